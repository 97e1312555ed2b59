// RCCL communicator of the domain-decomposed driver and the substep loop that uses it (BASELINE config 5:
// "RCCL allgather of halos over xGMI only for the domain-decomposed case").
//
// One process per GPU; the host language only carries the 128-byte ncclUniqueId from rank 0 to the others
// (any transport: torch.distributed, MPI, a file).  Everything per substep runs here, with no host round trip:
//
//     for phase in plan:                         2 phases with fused stage pairs
//       pack(field)                                                    stream S
//       ncclAllGather(send, recv)                                      stream C   (waits for the pack)
//       phase(interior tiles)      -- overlaps the collective --       stream S
//       unpack(field); phase(edge tiles)                               stream S   (waits for the collective)
//
// RCCL is resolved at run time from the library the process already has (dlopen "librccl.so": torch ships
// one, /opt/rocm/lib has one), so libpdeopt_hip.so carries no link-time dependency on it and a single-GPU
// user never loads it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "common.hpp"

namespace pdeopt {

struct CommState {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  ncclComm_t comm = nullptr;
  int world = 0, rank = 0;
  hipStream_t cstream = nullptr;          // collectives
  hipEvent_t packed = nullptr, gathered = nullptr;
  void* send = nullptr;                   // strip of this rank
  void* recv = nullptr;                   // strips of all ranks, rank-major
  size_t strip_bytes = 0;
};

namespace {

int load_rccl(pdeopt_ctx* ctx, CommState& c) {
  if (c.lib) return PDEOPT_OK;
  const char* names[] = {"librccl.so", "librccl.so.1"};
  // the copy the process already uses first (torch bundles its own), then the system one
  for (const char* n : names)
    if (!c.lib) c.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
  for (const char* n : names)
    if (!c.lib) c.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!c.lib) return fail(ctx, PDEOPT_EINVAL, "cannot load RCCL (librccl.so): %s", dlerror());
  c.get_unique_id = reinterpret_cast<decltype(c.get_unique_id)>(dlsym(c.lib, "ncclGetUniqueId"));
  c.comm_init_rank = reinterpret_cast<decltype(c.comm_init_rank)>(dlsym(c.lib, "ncclCommInitRank"));
  c.all_gather = reinterpret_cast<decltype(c.all_gather)>(dlsym(c.lib, "ncclAllGather"));
  c.comm_destroy = reinterpret_cast<decltype(c.comm_destroy)>(dlsym(c.lib, "ncclCommDestroy"));
  c.error_string = reinterpret_cast<decltype(c.error_string)>(dlsym(c.lib, "ncclGetErrorString"));
  if (!c.get_unique_id || !c.comm_init_rank || !c.all_gather || !c.comm_destroy || !c.error_string)
    return fail(ctx, PDEOPT_EINVAL, "librccl.so lacks a required symbol");
  return PDEOPT_OK;
}

#define PDEOPT_NCCL_CHECK(ctx, c, expr)                                                                    \
  do {                                                                                                      \
    ncclResult_t r_ = (expr);                                                                               \
    if (r_ != ncclSuccess)                                                                                  \
      return fail((ctx), PDEOPT_EHIP, "%s failed: %s (%s:%d)", #expr, (c).error_string(r_), __FILE__, __LINE__); \
  } while (0)

CommState g_probe;  // symbols for pdeopt_comm_unique_id (no ctx state needed)

}  // namespace

int comm_unique_id(pdeopt_ctx* ctx, char* out128) {
  int rc = load_rccl(ctx, g_probe);
  if (rc) return rc;
  ncclUniqueId id;
  PDEOPT_NCCL_CHECK(ctx, g_probe, g_probe.get_unique_id(&id));
  memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return PDEOPT_OK;
}

int comm_init(pdeopt_ctx* ctx, int world, int rank, const char* id128) {
  if (!ctx->comm) ctx->comm = new CommState();
  CommState& c = *ctx->comm;
  int rc = load_rccl(ctx, c);
  if (rc) return rc;
  if (c.comm) return fail(ctx, PDEOPT_ESTATE, "this ctx already has a communicator");
  ncclUniqueId id;
  memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  PDEOPT_NCCL_CHECK(ctx, c, c.comm_init_rank(&c.comm, world, id, rank));
  c.world = world;
  c.rank = rank;
  PDEOPT_HIP_CHECK(ctx, hipStreamCreateWithFlags(&c.cstream, hipStreamNonBlocking));
  PDEOPT_HIP_CHECK(ctx, hipEventCreateWithFlags(&c.packed, hipEventDisableTiming));
  PDEOPT_HIP_CHECK(ctx, hipEventCreateWithFlags(&c.gathered, hipEventDisableTiming));
  return PDEOPT_OK;
}

void comm_destroy(pdeopt_ctx* ctx) {
  CommState* c = ctx->comm;
  if (!c) return;
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (c->cstream) (void)hipStreamSynchronize(c->cstream);
  if (c->comm) (void)c->comm_destroy(c->comm);
  if (c->send) (void)hipFree(c->send);
  if (c->recv) (void)hipFree(c->recv);
  if (c->packed) (void)hipEventDestroy(c->packed);
  if (c->gathered) (void)hipEventDestroy(c->gathered);
  if (c->cstream) (void)hipStreamDestroy(c->cstream);
  delete c;
  ctx->comm = nullptr;
}

// n RK4 substeps of this rank's tile; nbr[8] = ranks of {up, down, left, right, UL, UR, DL, DR}
int rk4_decomposed_advance(pdeopt_ctx* ctx, double dt, int64_t n, const int* nbr, int overlap) {
  CommState* cp = ctx->comm;
  if (!cp || !cp->comm) return fail(ctx, PDEOPT_ESTATE, "pdeopt_comm_init has not been called");
  CommState& c = *cp;
  for (int q = 0; q < 8; ++q)
    if (nbr[q] < 0 || nbr[q] >= c.world) return fail(ctx, PDEOPT_EINVAL, "neighbour rank %d outside 0..%d", nbr[q], c.world - 1);
  const size_t strip_elems = halo_strip_elems(ctx);
  const size_t strip_bytes = strip_elems * ctx->esize;
  int rc;
  if (c.strip_bytes != strip_bytes) {
    PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (c.send) (void)hipFree(c.send);
    if (c.recv) (void)hipFree(c.recv);
    c.send = c.recv = nullptr;
    if ((rc = ensure_buffer(ctx, &c.send, strip_bytes))) return rc;
    if ((rc = ensure_buffer(ctx, &c.recv, strip_bytes * (size_t)c.world))) return rc;
    c.strip_bytes = strip_bytes;
  }
  int fields[4], nph = 0;
  rk4_phase_plan(ctx, fields, &nph);
  const bool split = overlap && nph == 2;  // interior / edge launches exist for the fused stage pairs
  const ncclDataType_t dtype = ctx->prob.dtype == PDEOPT_F32 ? ncclFloat32 : ncclFloat64;
  for (int64_t s = 0; s < n; ++s) {
    for (int ph = 0; ph < nph; ++ph) {
      if ((rc = halo_pack(ctx, fields[ph], c.send))) return rc;
      if (split) {
        PDEOPT_HIP_CHECK(ctx, hipEventRecord(c.packed, ctx->stream));
        PDEOPT_HIP_CHECK(ctx, hipStreamWaitEvent(c.cstream, c.packed, 0));
        PDEOPT_NCCL_CHECK(ctx, c, c.all_gather(c.send, c.recv, strip_elems, dtype, c.comm, c.cstream));
        PDEOPT_HIP_CHECK(ctx, hipEventRecord(c.gathered, c.cstream));
        if ((rc = rk4_phase(ctx, ph, dt, 1))) return rc;  // interior tiles: no halo reads
        PDEOPT_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, c.gathered, 0));
        if ((rc = halo_unpack(ctx, fields[ph], c.recv, nbr))) return rc;
        if ((rc = rk4_phase(ctx, ph, dt, 2))) return rc;  // edge tiles
      } else {
        PDEOPT_NCCL_CHECK(ctx, c, c.all_gather(c.send, c.recv, strip_elems, dtype, c.comm, ctx->stream));
        if ((rc = halo_unpack(ctx, fields[ph], c.recv, nbr))) return rc;
        if ((rc = rk4_phase(ctx, ph, dt, 0))) return rc;
      }
    }
  }
  return PDEOPT_OK;
}

}  // namespace pdeopt
