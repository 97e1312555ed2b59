// LDS-tiled fused RHS + integrator-stage kernel for Cahn-Hilliard / Allen-Cahn (gfx950).
//
// One 256-thread workgroup owns a TX x TY tile (32 rows x 32 16-byte vectors: 128 fp32 or 64 fp64
// columns).  Per stage and tile:
//   phase 1  stage-input tile + halo (2 rows / one aligned vector of columns each side) is
//            streamed HBM -> registers -> LDS with 16-byte loads only (wrap by index, no padded
//            copies of the field);  the pointwise operands of the stage update (y, acc) are
//            prefetched to registers before the first barrier so their latency hides under it.
//   phase 2  (CH) mu = mu_h(u) - kappa lap u is formed ONCE per point on the tile + 1 ring and
//            kept in a second LDS array (the closure -- log/exp -- is the expensive part).
//   phase 3  each thread walks a 4-row x 1-vector micro-tile, forms face mobilities, face
//            gradients and their divergence from LDS, and applies the Runge-Kutta stage update
//            straight to HBM with 16-byte stores.
// Algorithmic HBM traffic per stage is therefore exactly the compulsory words of SURVEY 8(d):
// the halo re-reads are served by the XCD's L2 because the workgroup -> tile map hands every XCD
// whole environments (blocks b and b+8 share an XCD; see remap below).
//
// Arithmetic: SURVEY Appendix A index form of pde_opt/numerics/equations/cahn_hilliard.py:89-109,
// allen_cahn.py:81-84 and pde_opt/numerics/utils/derivatives.py:8-61.
#pragma once

#include "stencil_generic.hpp"

namespace pdeopt {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <typename T>
struct VecOf;
template <>
struct VecOf<float> {
  using type = f32x4;
  static constexpr int V = 4;
};
template <>
struct VecOf<double> {
  using type = f64x2;
  static constexpr int V = 2;
};

// Whole-wave DPP shifts (gfx9: wave_shr:1 / wave_shl:1; checked on gfx950 with tools/dpp_test.hip):
// lane i receives the value of lane i-1 (from_prev) / lane i+1 (from_next); lane 0 / lane 63 keep
// `old`.  Neighbour exchange along a tile row in registers instead of a scalar LDS read whose
// 16-byte lane stride is a 4-way bank conflict.
__device__ __forceinline__ float lane_from_prev(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                                __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_from_next(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                                __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ double lane_from_prev(double old, double v) {
  const long long o = __builtin_bit_cast(long long, old), x = __builtin_bit_cast(long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)x, 0x138, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(o >> 32), (int)(x >> 32), 0x138, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
}
__device__ __forceinline__ double lane_from_next(double old, double v) {
  const long long o = __builtin_bit_cast(long long, old), x = __builtin_bit_cast(long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)x, 0x130, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(o >> 32), (int)(x >> 32), 0x130, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
}

// periodic index of a tile cell: one conditional add/subtract when tiles divide the grid, a true
// modulo for ragged tiles (a tile may then cover the grid more than once)
__device__ __forceinline__ int tile_wrap(int i, int n, bool ragged) {
  if (ragged) {
    i %= n;
    return i < 0 ? i + n : i;
  }
  if (i < 0) i += n;
  if (i >= n) i -= n;
  return i;
}

constexpr int kLanesPerRow = 32;   // vectors per tile row
constexpr int kPV = kLanesPerRow + 2;  // vectors per LDS row (one halo vector each side)
// RPT selects the tile height TX = 8 * RPT (2: 16 rows / 256 threads, 4: 32 rows / 512 threads)

template <typename T, int EQ, int RPT>
constexpr size_t tiled_lds_bytes() {
  constexpr int V = VecOf<T>::V;
  constexpr int HR = (EQ == PDEOPT_EQ_CAHN_HILLIARD) ? 2 : 1;
  size_t su = (size_t)(8 * RPT + 2 * HR) * kPV * V + 2 * V;
  size_t smu = (EQ == PDEOPT_EQ_CAHN_HILLIARD) ? (size_t)(8 * RPT + 2) * kPV * V + 2 * V : 0;
  return (su + smu) * sizeof(T);
}

template <typename T, int EQ, int CL, int OUT_MODE, int ACC_MODE, bool Y_FROM_TILE, int RPT>
__global__ __launch_bounds__(RPT == 4 ? 512 : 256) void stage_tiled_kernel(const StageArgs<T> a, const int tiles_i,
                                                          const int tiles_j, const int nblk,
                                                          const int xcd_remap) {
  using Vec = typename VecOf<T>::type;
  constexpr int V = VecOf<T>::V;
  constexpr int HR = (EQ == PDEOPT_EQ_CAHN_HILLIARD) ? 2 : 1;
  // RPT selects the tile height: 2 -> 16 rows x 256 threads, 4 -> 32 rows x 512 threads; a thread always
  // owns 2 rows (taller tiles by more threads, not more registers)
  constexpr int NT = RPT == 4 ? 512 : 256;
  constexpr int TX = 8 * RPT;
  constexpr int kRowsPerThread = 2;
  constexpr int PV = kPV;
  constexpr int P = PV * V;  // LDS row pitch in elements
  constexpr bool kIsCH = (EQ == PDEOPT_EQ_CAHN_HILLIARD);
  constexpr bool kMobInPlace = kIsCH && (CL == CL_GENERIC);
  constexpr bool kNeedY = (OUT_MODE == OUT_Y_PLUS_AK) || (ACC_MODE == ACC_INIT) || (OUT_MODE == OUT_K_LC);
  constexpr bool kNeedAcc = (ACC_MODE == ACC_ADD) || (OUT_MODE == OUT_ACC_PLUS_BK);

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* const su = reinterpret_cast<T*>(smem_raw) + V;              // one guard vector in front
  T* const smu = su + (TX + 2 * HR) * P + V;                    // (CH only) guard + mu array

  // ---- workgroup -> tile.  Blocks are dealt round-robin over the 8 XCDs (b and b+8 share one);
  // give each XCD a contiguous run of tiles (= whole environments) so halo re-reads hit its L2.
  int ti, tj, b;
  decode_tile(blockIdx.x, tiles_i, tiles_j, nblk, xcd_remap, &ti, &tj, &b);
  const int i0 = ti * TX;
  const int j0 = tj * (kLanesPerRow * V);

  const Geo& g = a.g;
  const int64_t ld = g.ld;
  const int64_t base = (int64_t)b * g.bstride + g.off;
  const EnvParams<T>& p = a.ep[b];
  const T* __restrict__ in = a.in + base;
  const bool ragged = (g.nx % TX != 0) || (g.ny % (kLanesPerRow * V) != 0);

  const int tid = threadIdx.x;
  const int lx = tid & 31;
  const int ly = tid >> 5;
  const int r0 = ly * kRowsPerThread;  // first tile row of this thread's micro-tile

  // Ragged tiles (grid extents that are not multiples of the tile): the tile is filled modulo the
  // grid, every lane computes, and only cells inside the grid are loaded pointwise / stored.
  const bool col_ok = (j0 + lx * V) < g.ny;
  auto cell_ok = [&](int r) { return col_ok && (i0 + r0 + r) < g.nx; };

  // ---- prefetch the pointwise operands of the stage update
  Vec yv[kRowsPerThread], av[kRowsPerThread];
  const int64_t pidx0 = base + (int64_t)(i0 + r0) * ld + (j0 + lx * V);
#pragma unroll
  for (int r = 0; r < kRowsPerThread; ++r) {
    yv[r] = Vec{};
    av[r] = Vec{};
    if (cell_ok(r)) {
      if constexpr (kNeedY && !Y_FROM_TILE) yv[r] = *reinterpret_cast<const Vec*>(a.y + pidx0 + r * ld);
      if constexpr (kNeedAcc) av[r] = *reinterpret_cast<const Vec*>(a.acc + pidx0 + r * ld);
    }
  }

  // ---- phase 1: tile + halo -> LDS
  load_rows_per_wave<T, V, PV, NT, TX + 2 * HR, Vec>(
      su, P, in, ld, i0 - HR, j0 - V, [&](int gi) { return g.periodic ? tile_wrap(gi, g.nx, ragged) : gi; },
      [&](int gj) { return g.periodic ? tile_wrap(gj, g.ny, ragged) : gj; }, tid);
  __syncthreads();

  const T kap = p.kappa;

  if (kIsCH && !(a.dbg & 1)) {
    // ---- phase 2: mu on the tile + 1 ring (rows -1..TX), one closure evaluation per point
    constexpr int kMuVecs = (TX + 2) * PV;
#pragma unroll 1
    for (int idx = tid; idx < kMuVecs; idx += NT) {
      const int r = idx / PV;  // mu row r <-> tile row r-1 <-> su row r+1
      const int cv = idx - r * PV;
      const T* c_ = su + (r + 1) * P + cv * V;
      const Vec c = *reinterpret_cast<const Vec*>(c_);
      const Vec xp = *reinterpret_cast<const Vec*>(c_ + P);
      const Vec xm = *reinterpret_cast<const Vec*>(c_ - P);
      const T left = c_[-1], right = c_[V];
      Vec m;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const T ym = (e == 0) ? left : c[e - 1];
        const T yp = (e == V - 1) ? right : c[e + 1];
        m[e] = eval_mu<T, CL>(a.mu, p.mu, c[e]) -
               kap * lap_at<T>(c[e], xp[e], xm[e], yp, ym, a.rhx2, a.rhy2);
      }
      *reinterpret_cast<Vec*>(smu + r * P + cv * V) = m;
    }
    __syncthreads();
    if constexpr (kMobInPlace) {
      // expensive mobility closures: evaluate once per point, in place over u (rows -1..TX)
#pragma unroll 1
      for (int idx = tid; idx < kMuVecs; idx += NT) {
        const int r = idx / PV;
        const int cv = idx - r * PV;
        T* c_ = su + (r + 1) * P + cv * V;
        Vec c = *reinterpret_cast<const Vec*>(c_);
#pragma unroll
        for (int e = 0; e < V; ++e) c[e] = eval_mob<T, CL>(a.mob, p.mob, c[e]);
        *reinterpret_cast<Vec*>(c_) = c;
      }
      __syncthreads();
    }
  }

  // ---- phase 3: fluxes, divergence and the stage update on a 4-row x 1-vector micro-tile
  const int cofs = (lx + 1) * V;  // array column of this thread's first cell
  Vec kout[kRowsPerThread];

  if (a.dbg & 2) {
#pragma unroll
    for (int r = 0; r < kRowsPerThread; ++r)
      kout[r] = *reinterpret_cast<const Vec*>(su + (r0 + r + HR) * P + cofs);
  } else if constexpr (kIsCH) {
    auto mob_vec = [&](const T* ptr) -> Vec {
      Vec d = *reinterpret_cast<const Vec*>(ptr);
      if constexpr (!kMobInPlace) {
#pragma unroll
        for (int e = 0; e < V; ++e) d[e] = eval_mob<T, CL>(a.mob, p.mob, d[e]);
      }
      return d;
    };
    auto mob_scalar = [&](T v) -> T {
      if constexpr (!kMobInPlace) return eval_mob<T, CL>(a.mob, p.mob, v);
      return v;
    };
    Vec m_prev, d_prev, fx_prev, divy_prev;
#pragma unroll
    for (int rr = -1; rr <= kRowsPerThread; ++rr) {
      const int r = r0 + rr;  // tile row
      const T* mp = smu + (r + 1) * P + cofs;
      const T* up = su + (r + HR) * P + cofs;
      const Vec m = *reinterpret_cast<const Vec*>(mp);
      const Vec d = mob_vec(up);
      Vec fx;
      if (rr >= 0) {
#pragma unroll
        for (int e = 0; e < V; ++e)
          fx[e] = (T(0.5) * (d_prev[e] + d[e])) * ((m[e] - m_prev[e]) * a.rhx);
      }
      if (rr >= 1) {
#pragma unroll
        for (int e = 0; e < V; ++e) kout[rr - 1][e] = (fx[e] - fx_prev[e]) * a.rhx + divy_prev[e];
      }
      if (rr >= 0 && rr < kRowsPerThread) {
        const T ml = mp[-1], mr = mp[V];
        const T dl = mob_scalar(up[-1]), dr = mob_scalar(up[V]);
        T fy[V + 1];
        fy[0] = (T(0.5) * (dl + d[0])) * ((m[0] - ml) * a.rhy);
#pragma unroll
        for (int e = 1; e < V; ++e)
          fy[e] = (T(0.5) * (d[e - 1] + d[e])) * ((m[e] - m[e - 1]) * a.rhy);
        fy[V] = (T(0.5) * (d[V - 1] + dr)) * ((mr - m[V - 1]) * a.rhy);
#pragma unroll
        for (int e = 0; e < V; ++e) divy_prev[e] = (fy[e + 1] - fy[e]) * a.rhy;
      }
      m_prev = m;
      d_prev = d;
      fx_prev = fx;
    }
  } else {
    // Allen-Cahn: k = -R(u) (mu_h(u) - kappa lap u), allen_cahn.py:81-84
    Vec u_prev = *reinterpret_cast<const Vec*>(su + (r0 - 1 + HR) * P + cofs);
    Vec u_cur = *reinterpret_cast<const Vec*>(su + (r0 + HR) * P + cofs);
#pragma unroll
    for (int rr = 0; rr < kRowsPerThread; ++rr) {
      const T* up = su + (r0 + rr + HR) * P + cofs;
      const Vec u_next = *reinterpret_cast<const Vec*>(up + P);
      const T left = up[-1], right = up[V];
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const T ym = (e == 0) ? left : u_cur[e - 1];
        const T yp = (e == V - 1) ? right : u_cur[e + 1];
        const T mu = eval_mu<T, CL>(a.mu, p.mu, u_cur[e]) -
                     kap * lap_at<T>(u_cur[e], u_next[e], u_prev[e], yp, ym, a.rhx2, a.rhy2);
        kout[rr][e] = -eval_mob<T, CL>(a.mob, p.mob, u_cur[e]) * mu;
      }
      u_prev = u_cur;
      u_cur = u_next;
    }
  }

  // ---- stage update, 16-byte stores
#pragma unroll
  for (int r = 0; r < kRowsPerThread; ++r) {
    if (!cell_ok(r)) continue;
    const int64_t idx = pidx0 + r * ld;
    Vec k = kout[r];
    if (a.scaled) k = k * p.kscale;  // per-environment step size: the slope carries dt_b / dt_ref
    if constexpr (kNeedY && Y_FROM_TILE) {
      // the stage input IS y (stage 1 / Euler): take it from the LDS tile.  Not valid when the
      // mobility pass overwrote the tile in place.
      yv[r] = *reinterpret_cast<const Vec*>(su + (r0 + r + HR) * P + cofs);
    }
    if constexpr (ACC_MODE == ACC_INIT) *reinterpret_cast<Vec*>(a.acc + idx) = yv[r] + a.b * k;
    if constexpr (OUT_MODE == OUT_K) *reinterpret_cast<Vec*>(a.out + idx) = k;
    if constexpr (OUT_MODE == OUT_K_LC) {
      *reinterpret_cast<Vec*>(a.out + idx) = k;
      Vec nxt = yv[r];
      for (int j = 0; j < a.lc.n; ++j) nxt += a.lc.c[j] * *reinterpret_cast<const Vec*>(a.lc.k[j] + idx);
      *reinterpret_cast<Vec*>(a.lc.next + idx) = nxt + a.lc.c[a.lc.n] * k;
    }
    if constexpr (OUT_MODE == OUT_Y_PLUS_AK) *reinterpret_cast<Vec*>(a.out + idx) = yv[r] + a.a * k;
    if constexpr (OUT_MODE == OUT_ACC_PLUS_BK) *reinterpret_cast<Vec*>(a.out + idx) = av[r] + a.b * k;
    if constexpr (ACC_MODE == ACC_ADD) *reinterpret_cast<Vec*>(a.acc + idx) = av[r] + a.b * k;
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------

// rows per thread for a problem: 4 (32-row tiles) unless the grid only divides by 16
inline int tiled_rpt(const pdeopt_ctx* ctx) {
  return ctx->opt_tile_rows == 32 ? 4 : 2;  // 16-row tiles measured fastest
}

template <typename T>
bool tiled_supported(const pdeopt_ctx* ctx) {
  constexpr int V = VecOf<T>::V;
  const pdeopt_problem& p = ctx->prob;
  if (p.equation != PDEOPT_EQ_CAHN_HILLIARD && p.equation != PDEOPT_EQ_ALLEN_CAHN) return false;
  if (p.mu.kind == PDEOPT_CL_JIT || p.mob.kind == PDEOPT_CL_JIT) return false;  // run-time-compiled closures: the generic kernel only (jit.hip)
  // 16-byte vectors need ny % V == 0.  Grids that the tiles do not divide run ragged tiles (periodic
  // layout only); degenerate extents (a single row / column, the 256 x 1 "1-D" runs) stay generic.
  if (p.ny % V != 0 || p.nx < 8 || p.ny < 4 * V) return false;
  const bool divides = p.nx % (8 * tiled_rpt(ctx)) == 0 && p.ny % (kLanesPerRow * V) == 0;
  if (!divides && ctx->halo) return false;
  return true;
}

template <typename T, int EQ, int CL, int OUT_MODE, int ACC_MODE, bool Y_FROM_TILE, int RPT>
int launch_tiled_rpt(pdeopt_ctx* ctx, const StageArgs<T>& s) {
  constexpr int V = VecOf<T>::V;
  constexpr int kTileRows = 8 * RPT;
  const pdeopt_problem& p = ctx->prob;
  const int tiles_i = (p.nx + kTileRows - 1) / kTileRows;
  const int tiles_j = (p.ny + kLanesPerRow * V - 1) / (kLanesPerRow * V);
  const int64_t nblk64 = (int64_t)tiles_i * tiles_j * ctx->win_n;
  if (nblk64 > 0x7fffffffLL) return fail(ctx, PDEOPT_EINVAL, "too many tiles");
  const int nblk = (int)nblk64;
  const size_t lds = tiled_lds_bytes<T, EQ, RPT>();
  hipLaunchKernelGGL((stage_tiled_kernel<T, EQ, CL, OUT_MODE, ACC_MODE, Y_FROM_TILE, RPT>), dim3(nblk),
                     dim3(RPT == 4 ? 512 : 256), lds, ctx->stream, s, tiles_i, tiles_j, nblk,
                     tile_flags(nblk, tiles_i, tiles_j));
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  return PDEOPT_OK;
}

template <typename T, int EQ, int CL, int OUT_MODE, int ACC_MODE, bool Y_FROM_TILE>
int launch_tiled_inst(pdeopt_ctx* ctx, const StageArgs<T>& s) {
  if (tiled_rpt(ctx) == 2)
    return launch_tiled_rpt<T, EQ, CL, OUT_MODE, ACC_MODE, Y_FROM_TILE, 2>(ctx, s);
  return launch_tiled_rpt<T, EQ, CL, OUT_MODE, ACC_MODE, Y_FROM_TILE, 4>(ctx, s);
}

template <typename T, int EQ, int CL>
int launch_tiled_modes(pdeopt_ctx* ctx, const StageArgs<T>& s) {
  const bool in_is_y = (s.in == s.y);
  // the in-place mobility pass (CH + generic closures) destroys u in the tile
  constexpr bool kCanReuseTile = !(EQ == PDEOPT_EQ_CAHN_HILLIARD && CL == CL_GENERIC);
  const int om = s.out_mode, am = s.acc_mode;
  if (om == OUT_K && am == ACC_NONE)
    return launch_tiled_inst<T, EQ, CL, OUT_K, ACC_NONE, false>(ctx, s);
  if (om == OUT_K_LC && am == ACC_NONE)
    return launch_tiled_inst<T, EQ, CL, OUT_K_LC, ACC_NONE, false>(ctx, s);
  if (om == OUT_Y_PLUS_AK && am == ACC_NONE) {
    if constexpr (kCanReuseTile)
      if (in_is_y) return launch_tiled_inst<T, EQ, CL, OUT_Y_PLUS_AK, ACC_NONE, true>(ctx, s);
    return launch_tiled_inst<T, EQ, CL, OUT_Y_PLUS_AK, ACC_NONE, false>(ctx, s);
  }
  if (om == OUT_Y_PLUS_AK && am == ACC_INIT) {
    if constexpr (kCanReuseTile)
      if (in_is_y) return launch_tiled_inst<T, EQ, CL, OUT_Y_PLUS_AK, ACC_INIT, true>(ctx, s);
    return launch_tiled_inst<T, EQ, CL, OUT_Y_PLUS_AK, ACC_INIT, false>(ctx, s);
  }
  if (om == OUT_Y_PLUS_AK && am == ACC_ADD)
    return launch_tiled_inst<T, EQ, CL, OUT_Y_PLUS_AK, ACC_ADD, false>(ctx, s);
  if (om == OUT_ACC_PLUS_BK && am == ACC_NONE)
    return launch_tiled_inst<T, EQ, CL, OUT_ACC_PLUS_BK, ACC_NONE, false>(ctx, s);
  return fail(ctx, PDEOPT_EINVAL, "stage mode (%d,%d) has no tiled kernel", om, am);
}

template <typename T, int EQ>
int launch_tiled_cl(pdeopt_ctx* ctx, const StageArgs<T>& s) {
  switch (classify_closures(ctx->prob.mu, ctx->prob.mob)) {
    case CL_POLY:
      return launch_tiled_modes<T, EQ, CL_POLY>(ctx, s);
    case CL_LOGIT:
      return launch_tiled_modes<T, EQ, CL_LOGIT>(ctx, s);
    default:
      return launch_tiled_modes<T, EQ, CL_GENERIC>(ctx, s);
  }
}

template <typename T>
int launch_tiled(pdeopt_ctx* ctx, const StageArgs<T>& s) {
  const int cl = classify_closures(ctx->prob.mu, ctx->prob.mob);
  static const char* kClName[] = {"generic", "poly", "logit"};
  char name[96];
  snprintf(name, sizeof(name), "stage_tiled<%s,%s,%s,rows%d>", sizeof(T) == 4 ? "f32" : "f64",
           ctx->prob.equation == PDEOPT_EQ_CAHN_HILLIARD ? "CH" : "AC", kClName[cl],
           8 * tiled_rpt(ctx));
  ctx->last_kernel = name;
  if (ctx->prob.equation == PDEOPT_EQ_CAHN_HILLIARD)
    return launch_tiled_cl<T, PDEOPT_EQ_CAHN_HILLIARD>(ctx, s);
  return launch_tiled_cl<T, PDEOPT_EQ_ALLEN_CAHN>(ctx, s);
}

}  // namespace pdeopt
