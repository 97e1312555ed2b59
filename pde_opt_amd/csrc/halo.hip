// Domain decomposition of one large field (BASELINE config 5): padded, non-periodic
// tile layout + halo pack / unpack kernels + the per-phase RK4 entry point.
//
// The reference has no counterpart (single device, jnp.roll wraps the whole array).  A rank's tile
// is stored with a 4-cell halo on every side; the stencil kernels then read neighbours straight
// from memory (Geo.periodic == 0).  Before each RK phase the field that phase differentiates is
// exchanged: every rank packs 8 pieces of its INTERIOR (4 edges + 4 corners) into one contiguous
// strip, the strips are all-gathered (RCCL over xGMI via torch.distributed on the GPU, or the
// loop-back buffer for a single rank), and every rank unpacks the 8 pieces it needs from its 8
// Cartesian neighbours.  One collective per phase: 2 per substep with the fused stage pairs
// (halo 4 covers two stages), 4 with per-stage kernels.
#include <algorithm>

#include "common.hpp"

namespace pdeopt {

namespace {

struct HaloGeo {
  int nx, ny, h;
  int64_t ld, bstride, off;  // padded row pitch / env stride / offset of interior (0,0)
  int64_t per_env;           // strip elements per environment
};

HaloGeo halo_geo(const pdeopt_ctx* ctx) {
  HaloGeo g;
  g.nx = ctx->prob.nx;
  g.ny = ctx->prob.ny;
  g.h = ctx->halo;
  g.ld = pad_ld(g.ny, g.h);
  g.bstride = (int64_t)pad_rows(g.nx, g.h) * g.ld;
  g.off = (int64_t)g.h * g.ld + g.h;
  g.per_env = 2LL * g.h * g.ny + 2LL * g.nx * g.h + 4LL * g.h * g.h;
  return g;
}

// strip element -> (piece, local i, local j) ; piece order: top, bottom, left, right, TL, TR, BL, BR
__device__ __forceinline__ void decode(const HaloGeo& g, int64_t e, int* piece, int* li, int* lj) {
  const int64_t rows = (int64_t)g.h * g.ny, cols = (int64_t)g.nx * g.h, cor = (int64_t)g.h * g.h;
  if (e < rows) { *piece = 0; *li = (int)(e / g.ny); *lj = (int)(e % g.ny); return; }
  e -= rows;
  if (e < rows) { *piece = 1; *li = (int)(e / g.ny); *lj = (int)(e % g.ny); return; }
  e -= rows;
  if (e < cols) { *piece = 2; *li = (int)(e / g.h); *lj = (int)(e % g.h); return; }
  e -= cols;
  if (e < cols) { *piece = 3; *li = (int)(e / g.h); *lj = (int)(e % g.h); return; }
  e -= cols;
  *piece = 4 + (int)(e / cor);
  e %= cor;
  *li = (int)(e / g.h);
  *lj = (int)(e % g.h);
}

// interior coordinates of the SOURCE cell of a piece element (what a rank sends)
__device__ __forceinline__ void src_cell(const HaloGeo& g, int piece, int li, int lj, int* i, int* j) {
  const int lastr = g.nx - g.h, lastc = g.ny - g.h;
  switch (piece) {
    case 0: *i = li; *j = lj; break;                 // top rows
    case 1: *i = lastr + li; *j = lj; break;         // bottom rows
    case 2: *i = li; *j = lj; break;                 // left columns
    case 3: *i = li; *j = lastc + lj; break;         // right columns
    case 4: *i = li; *j = lj; break;                 // TL corner
    case 5: *i = li; *j = lastc + lj; break;         // TR
    case 6: *i = lastr + li; *j = lj; break;         // BL
    default: *i = lastr + li; *j = lastc + lj; break;  // BR
  }
}

template <typename T>
__global__ void pack_kernel(const T* __restrict__ field, T* __restrict__ send, HaloGeo g) {
  const int b = blockIdx.y;
  const T* f = field + (int64_t)b * g.bstride + g.off;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < g.per_env;
       e += (int64_t)gridDim.x * blockDim.x) {
    int piece, li, lj, i, j;
    decode(g, e, &piece, &li, &lj);
    src_cell(g, piece, li, lj, &i, &j);
    send[(int64_t)b * g.per_env + e] = f[(int64_t)i * g.ld + j];
  }
}

struct Neighbours {
  int r[8];  // up, down, left, right, up-left, up-right, down-left, down-right
};

// my halo piece q is filled from neighbour nbr[q]'s piece `from[q]`:
//   top halo <- up's bottom rows, bottom <- down's top rows, left <- left's right columns, ...
template <typename T>
__global__ void unpack_kernel(T* __restrict__ field, const T* __restrict__ recv, HaloGeo g,
                              Neighbours nb, int64_t strip_elems) {
  const int b = blockIdx.y;
  T* f = field + (int64_t)b * g.bstride + g.off;
  const int from[8] = {1, 0, 3, 2, 7, 6, 5, 4};
  const int64_t rows = (int64_t)g.h * g.ny, cols = (int64_t)g.nx * g.h, cor = (int64_t)g.h * g.h;
  const int64_t piece_off[8] = {0, rows, 2 * rows, 2 * rows + cols, 2 * rows + 2 * cols,
                                2 * rows + 2 * cols + cor, 2 * rows + 2 * cols + 2 * cor,
                                2 * rows + 2 * cols + 3 * cor};
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < g.per_env;
       e += (int64_t)gridDim.x * blockDim.x) {
    int q, li, lj;
    decode(g, e, &q, &li, &lj);  // (q, li, lj) addresses MY halo piece q
    int i, j;
    switch (q) {
      case 0: i = -g.h + li; j = lj; break;
      case 1: i = g.nx + li; j = lj; break;
      case 2: i = li; j = -g.h + lj; break;
      case 3: i = li; j = g.ny + lj; break;
      case 4: i = -g.h + li; j = -g.h + lj; break;
      case 5: i = -g.h + li; j = g.ny + lj; break;
      case 6: i = g.nx + li; j = -g.h + lj; break;
      default: i = g.nx + li; j = g.ny + lj; break;
    }
    // same (li, lj) inside the neighbour's piece from[q] (pieces q and from[q] have equal shape)
    const int64_t within = e - piece_off[q];
    const T v = recv[(int64_t)nb.r[q] * strip_elems + (int64_t)b * g.per_env + piece_off[from[q]] + within];
    f[(int64_t)i * g.ld + j] = v;
  }
}

int ensure_scratch(pdeopt_ctx* ctx, size_t bytes) {
  if (ctx->halo_scratch_bytes >= bytes) return PDEOPT_OK;
  if (ctx->halo_scratch) (void)hipFree(ctx->halo_scratch);
  if (ctx->halo_scratch2) (void)hipFree(ctx->halo_scratch2);  // sized like the first (stencil.hip: loop-back, halo 8)
  ctx->halo_scratch = ctx->halo_scratch2 = nullptr;
  ctx->halo_scratch_bytes = ctx->halo_scratch2_bytes = 0;
  int rc = ensure_buffer(ctx, &ctx->halo_scratch, bytes);
  if (!rc) ctx->halo_scratch_bytes = bytes;
  return rc;
}

}  // namespace

size_t halo_strip_elems(const pdeopt_ctx* ctx) {
  return (size_t)halo_geo(ctx).per_env * (size_t)ctx->prob.batch;
}

void* field_ptr(pdeopt_ctx* ctx, int field) {
  switch (field) {
    case 0: return ctx->Y;
    case 1: return ctx->TA;
    case 2: return ctx->TB;
    case 3: return ctx->ACC;
    default: return nullptr;
  }
}

int halo_pack(pdeopt_ctx* ctx, int field, void* dev_send) {
  void* f = field_ptr(ctx, field);
  if (!f) return fail(ctx, PDEOPT_EINVAL, "field %d does not exist / is not allocated yet", field);
  const HaloGeo g = halo_geo(ctx);
  const size_t bytes = halo_strip_elems(ctx) * ctx->esize;
  if (!dev_send) {
    int rc = ensure_scratch(ctx, bytes);
    if (rc) return rc;
    dev_send = ctx->halo_scratch;
  }
  const int blocks = (int)std::min<int64_t>((g.per_env + 255) / 256, 1024);
  dim3 grid(blocks, ctx->prob.batch);
  if (ctx->prob.dtype == PDEOPT_F32)
    hipLaunchKernelGGL(pack_kernel<float>, grid, dim3(256), 0, ctx->stream, (const float*)f, (float*)dev_send, g);
  else
    hipLaunchKernelGGL(pack_kernel<double>, grid, dim3(256), 0, ctx->stream, (const double*)f, (double*)dev_send, g);
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  return PDEOPT_OK;
}

int halo_unpack(pdeopt_ctx* ctx, int field, const void* dev_recv, const int* nbr) {
  void* f = field_ptr(ctx, field);
  if (!f) return fail(ctx, PDEOPT_EINVAL, "field %d does not exist / is not allocated yet", field);
  const HaloGeo g = halo_geo(ctx);
  if (!dev_recv) {
    if (!ctx->halo_scratch) return fail(ctx, PDEOPT_ESTATE, "loop-back unpack without a preceding pack");
    dev_recv = ctx->halo_scratch;
    for (int q = 0; q < 8; ++q)
      if (nbr[q] != 0) return fail(ctx, PDEOPT_EINVAL, "loop-back exchange: every neighbour must be rank 0");
  }
  Neighbours nb;
  for (int q = 0; q < 8; ++q) {
    if (nbr[q] < 0) return fail(ctx, PDEOPT_EINVAL, "negative neighbour rank");
    nb.r[q] = nbr[q];
  }
  const int64_t strip = (int64_t)halo_strip_elems(ctx);
  const int blocks = (int)std::min<int64_t>((g.per_env + 255) / 256, 1024);
  dim3 grid(blocks, ctx->prob.batch);
  if (ctx->prob.dtype == PDEOPT_F32)
    hipLaunchKernelGGL(unpack_kernel<float>, grid, dim3(256), 0, ctx->stream, (float*)f, (const float*)dev_recv, g, nb, strip);
  else
    hipLaunchKernelGGL(unpack_kernel<double>, grid, dim3(256), 0, ctx->stream, (double*)f, (const double*)dev_recv, g, nb, strip);
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  return PDEOPT_OK;
}

}  // namespace pdeopt
