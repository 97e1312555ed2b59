// LDS-tiled fused RHS + integrator-stage kernel for the smoothed-boundary equations:
//   AllenCahn2DSmoothedBoundary   allen_cahn.py:139-156    k = -R(u) inner
//   CahnHilliard2DSmoothedBoundary cahn_hilliard.py:257-289 k = div(psi D(u) grad inner) / psi + |grad psi|/psi flux(t)
//   inner = mu_h(u) - kappa/psi div(psi grad u) - sqrt(kappa) |grad psi|/psi (cos theta(t) on / off the mask) sqrt(2 f(u))
//
// The one-thread-per-cell kernels (stencil_generic.hpp) fetch every neighbour of u, psi and -- for Cahn-Hilliard, in
// two launches through a work field -- of `inner` from global memory: 24.8 us per right-hand side at 1024^2 fp32,
// L2-bound.  Here one workgroup owns a 16 x 32-vector tile:
//   P1  u and psi on tile + 2 (Allen-Cahn: + 1)     HBM -> LDS, 16-byte loads, periodic wrap by index
//   P2  (CH) inner on tile + 1                       LDS -> LDS; |grad psi|/psi and the mask are read pointwise (each
//                                                    value is needed once per evaluation)
//   P3  k on the own cells, stage update             16-byte stores
// Compulsory traffic per cell and stage: u, psi, |grad psi|/psi, mask (4 words: SURVEY f3's "+8 B/cell/stage" over
// the periodic kernels, plus the mask) + the stage update's operands.  Same expressions, term for term, as
// rhs_generic_point / sbm_inner_kernel / sbm_ch_stage_kernel -- except that fp32 takes 1 / psi and sqrt(2 f) from the
// hardware approximations (the kernel is VALU-bound on its transcendentals: a logit, two logs, a square root and two
// divisions per cell); closures through closure_generic (the regular-solution free energy needs PDEOPT_CL_MIX_ENTROPY).  The time-dependent scalars arrive as kernel arguments (StageArgs::tw_a,
// tw_b, tsrc), one set per right-hand-side evaluation.
#pragma once

#include "stencil_tiled.hpp"

namespace pdeopt {

template <typename T, int EQ>
constexpr size_t sbm_tiled_lds_bytes() {
  constexpr int V = VecOf<T>::V;
  constexpr int TX = 16;
  constexpr int HR = EQ == PDEOPT_EQ_CAHN_HILLIARD_SBM ? 2 : 1;
  size_t n = 2 * ((size_t)(TX + 2 * HR) * kPV * V + 2 * V);                         // u, psi
  if (EQ == PDEOPT_EQ_CAHN_HILLIARD_SBM) n += (size_t)(TX + 2) * kPV * V + 2 * V;  // inner
  return n * sizeof(T);
}

// The closures of the smoothed-boundary runs in their fixed forms (FAST): mu_h = cubic [+ logit], mobility = quadratic,
// f = cubic [+ c ln c + (1 - c) ln(1 - c)] -- the regular-solution model of notebooks/smooth_boundary.ipynb and every
// reference test; coefficients past n are stored as zeros, so the Horner chains have a fixed length and no loop.
// closure_generic walks a run-time loop and run-time flag branches per evaluation: 288 lane-instructions per cell and
// stage against ~140 (SQ_INSTS_VALU, profiles/pmc_r03.json), in a kernel the VALU bounds.
template <typename T, bool FAST>
__device__ __forceinline__ T sbm_mu(const StageArgs<T>& a, const EnvParams<T>& p, T c) {
  if constexpr (!FAST) return closure_generic<T>(a.mu, p.mu, c);
  T r = ((p.mu[3] * c + p.mu[2]) * c + p.mu[1]) * c + p.mu[0];
  if (a.mu.flags & PDEOPT_CL_LOGIT_PRIOR) r += t_logit<T>(c);
  return r;
}
template <typename T, bool FAST>
__device__ __forceinline__ T sbm_fe(const StageArgs<T>& a, const EnvParams<T>& p, T c) {
  if constexpr (!FAST) return closure_generic<T>(a.fe, p.fe, c);
  T r = ((p.fe[3] * c + p.fe[2]) * c + p.fe[1]) * c + p.fe[0];
  if (a.fe.flags & PDEOPT_CL_MIX_ENTROPY) r += c * t_log<T>(c) + (T(1) - c) * t_log<T>(T(1) - c);
  return r;
}
template <typename T, bool FAST>
__device__ __forceinline__ T sbm_mob(const StageArgs<T>& a, const EnvParams<T>& p, T c) {
  if constexpr (!FAST) return closure_generic<T>(a.mob, p.mob, c);
  return (p.mob[2] * c + p.mob[1]) * c + p.mob[0];
}
// host: do the three closures have the fixed forms?
inline bool sbm_fast_closures(const pdeopt_problem& p) {
  const bool mu = p.mu.kind == PDEOPT_CL_POLY && p.mu.n <= 4 && (p.mu.flags & ~PDEOPT_CL_LOGIT_PRIOR) == 0;
  const bool mob = p.mob.kind == PDEOPT_CL_POLY && p.mob.n <= 3 && p.mob.flags == 0;
  const bool fe = p.fe.kind == PDEOPT_CL_POLY && p.fe.n <= 4 && (p.fe.flags & ~PDEOPT_CL_MIX_ENTROPY) == 0;
  return mu && mob && fe;
}

template <typename T, int EQ, bool FAST>
__global__ __launch_bounds__(256) void sbm_tiled_kernel(const StageArgs<T> a, const int tiles_i, const int tiles_j,
                                                        const int nblk, const int xcd_remap) {
  using Vec = typename VecOf<T>::type;
  constexpr int V = VecOf<T>::V;
  constexpr bool kIsCH = EQ == PDEOPT_EQ_CAHN_HILLIARD_SBM;
  constexpr int HR = kIsCH ? 2 : 1;
  constexpr int NT = 256, TX = 16, RPT = 2, PV = kPV, P = PV * V;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* const su = reinterpret_cast<T*>(smem_raw) + V;  // rows: tile row + HR; one guard vector in front
  T* const sp = su + (TX + 2 * HR) * P + 2 * V;      // psi, same geometry
  T* const sin = sp + (TX + 2 * HR) * P + 2 * V;     // (CH) inner: rows = tile row + 1

  int ti, tj, b;
  decode_tile(blockIdx.x, tiles_i, tiles_j, nblk, xcd_remap, &ti, &tj, &b);
  const int i0 = ti * TX;
  const int j0 = tj * (kLanesPerRow * V);

  const Geo& g = a.g;
  const int nx = g.nx, ny = g.ny;
  const int64_t base = (int64_t)b * g.bstride;  // periodic layout: ld == ny, off == 0
  const EnvParams<T>& p = a.ep[b];
  const bool ragged = (nx % TX != 0) || (ny % (kLanesPerRow * V) != 0);
  auto wrap_row = [&](int gi) { return tile_wrap(gi, nx, ragged); };
  auto wrap_col = [&](int gj) { return tile_wrap(gj, ny, ragged); };

  const int tid = threadIdx.x;
  const int lx = tid & 31;
  const int ly = tid >> 5;
  const int r0 = ly * RPT;
  const bool col_ok = (j0 + lx * V) < ny;
  auto cell_ok = [&](int r) { return col_ok && (i0 + r0 + r) < nx; };

  // ---- P1: u and psi tiles
  load_rows_per_wave<T, V, PV, NT, TX + 2 * HR, Vec>(su, P, a.in + base, ny, i0 - HR, j0 - V, wrap_row, wrap_col, tid);
  load_rows_per_wave<T, V, PV, NT, TX + 2 * HR, Vec>(sp, P, a.psi, ny, i0 - HR, j0 - V, wrap_row, wrap_col, tid);
  __syncthreads();

  const T sqk = sqrt(p.kappa);
  // inner at one vector: LDS row r (of su / sp), vector column cv; wl = the wall weights of its 4 cells
  auto inner_vec = [&](const int r, const int cv, const Vec ngp, const Vec msk) -> Vec {
    const T* c_ = su + r * P + cv * V;
    const T* q_ = sp + r * P + cv * V;
    const Vec c = *reinterpret_cast<const Vec*>(c_), xp = *reinterpret_cast<const Vec*>(c_ + P), xm = *reinterpret_cast<const Vec*>(c_ - P);
    const Vec pc = *reinterpret_cast<const Vec*>(q_), pxp = *reinterpret_cast<const Vec*>(q_ + P), pxm = *reinterpret_cast<const Vec*>(q_ - P);
    const T cl = c_[-1], cr = c_[V], pl = q_[-1], pr = q_[V];
    Vec out;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const T ym = (e == 0) ? cl : c[e - 1], yp = (e == V - 1) ? cr : c[e + 1];
      const T pym = (e == 0) ? pl : pc[e - 1], pyp = (e == V - 1) ? pr : pc[e + 1];
      const T w = sqk * ngp[e] * (a.tw_a * msk[e] + a.tw_b * (T(1) - msk[e]));
      // the expression of INNER in rhs_generic_point, term for term
      const T dx_hi = (T(0.5) * (pc[e] + pxp[e])) * ((xp[e] - c[e]) * a.rhx), dx_lo = (T(0.5) * (pxm[e] + pc[e])) * ((c[e] - xm[e]) * a.rhx);
      const T dy_hi = (T(0.5) * (pc[e] + pyp)) * ((yp - c[e]) * a.rhy), dy_lo = (T(0.5) * (pym + pc[e])) * ((c[e] - ym) * a.rhy);
      const T lap = (dx_hi - dx_lo) * a.rhx + (dy_hi - dy_lo) * a.rhy;
      // (fp32: kappa / psi and the square root on the hardware approximations, closures.hpp: t_rcp / t_sqrt)
      T rr = sbm_mu<T, FAST>(a, p, c[e]) - (p.kappa * t_rcp<T>(pc[e])) * lap;
      rr -= w * t_sqrt<T>(T(2) * sbm_fe<T, FAST>(a, p, c[e]));
      out[e] = rr;
    }
    return out;
  };
  // |grad psi| / psi and the mask at the (wrapped) cells of a vector
  auto aux_at = [&](const int gi, const int gj, Vec* ngp, Vec* msk) {
    const int64_t o = (int64_t)wrap_row(gi) * ny + wrap_col(gj);
    *ngp = *reinterpret_cast<const Vec*>(a.ngp + o);
    *msk = *reinterpret_cast<const Vec*>(a.mask + o);
  };

  if constexpr (kIsCH) {
    // ---- P2: inner on tile + 1 (inner row r <-> tile row r - 1 <-> su row r + 1)
    constexpr int kVecs = (TX + 2) * PV;
#pragma unroll 1
    for (int idx = tid; idx < kVecs; idx += NT) {
      const int r = idx / PV;
      const int cv = idx - r * PV;
      Vec ngp, msk;
      aux_at(i0 - 1 + r, j0 - V + cv * V, &ngp, &msk);
      *reinterpret_cast<Vec*>(sin + r * P + cv * V) = inner_vec(r + 1, cv, ngp, msk);
    }
    __syncthreads();
  }

  // ---- P3: k on the own cells, stage update
  const int cv = lx + 1;
  const int64_t pidx0 = base + (int64_t)(i0 + r0) * ny + (j0 + lx * V);
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    if (!cell_ok(r)) continue;
    const int tr = r0 + r;  // tile row
    Vec ngp, msk;
    aux_at(i0 + tr, j0 + lx * V, &ngp, &msk);
    Vec k;
    const T* c_ = su + (tr + HR) * P + cv * V;
    const Vec c = *reinterpret_cast<const Vec*>(c_);
    if constexpr (!kIsCH) {
      const Vec in00 = inner_vec(tr + HR, cv, ngp, msk);
#pragma unroll
      for (int e = 0; e < V; ++e) k[e] = -sbm_mob<T, FAST>(a, p, c[e]) * in00[e];
    } else {
      const T* q_ = sp + (tr + HR) * P + cv * V;
      const T* n_ = sin + (tr + 1) * P + cv * V;
      const Vec uxp = *reinterpret_cast<const Vec*>(c_ + P), uxm = *reinterpret_cast<const Vec*>(c_ - P);
      const Vec p00 = *reinterpret_cast<const Vec*>(q_), pxp = *reinterpret_cast<const Vec*>(q_ + P), pxm = *reinterpret_cast<const Vec*>(q_ - P);
      const Vec in00 = *reinterpret_cast<const Vec*>(n_), inxp = *reinterpret_cast<const Vec*>(n_ + P), inxm = *reinterpret_cast<const Vec*>(n_ - P);
      const T ul = c_[-1], ur = c_[V], pl = q_[-1], pr = q_[V], il = n_[-1], ir = n_[V];
      Vec d;
#pragma unroll
      for (int e = 0; e < V; ++e) d[e] = sbm_mob<T, FAST>(a, p, c[e]);
      const T dl = sbm_mob<T, FAST>(a, p, ul), dr = sbm_mob<T, FAST>(a, p, ur);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const T dxp = sbm_mob<T, FAST>(a, p, uxp[e]), dxm = sbm_mob<T, FAST>(a, p, uxm[e]);
        const T dyp = (e == V - 1) ? dr : d[e + 1], dym = (e == 0) ? dl : d[e - 1];
        const T pyp = (e == V - 1) ? pr : p00[e + 1], pym = (e == 0) ? pl : p00[e - 1];
        const T inyp = (e == V - 1) ? ir : in00[e + 1], inym = (e == 0) ? il : in00[e - 1];
        // sbm_ch_stage_kernel / rhs_generic_point, term for term
        const T fx0 = (T(0.5) * (p00[e] + pxp[e])) * (T(0.5) * (d[e] + dxp)) * ((inxp[e] - in00[e]) * a.rhx);
        const T fxm = (T(0.5) * (pxm[e] + p00[e])) * (T(0.5) * (dxm + d[e])) * ((in00[e] - inxm[e]) * a.rhx);
        const T fy0 = (T(0.5) * (p00[e] + pyp)) * (T(0.5) * (d[e] + dyp)) * ((inyp - in00[e]) * a.rhy);
        const T fym = (T(0.5) * (pym + p00[e])) * (T(0.5) * (dym + d[e])) * ((in00[e] - inym) * a.rhy);
        k[e] = ((fx0 - fxm) * a.rhx + (fy0 - fym) * a.rhy) * t_rcp<T>(p00[e]) + ngp[e] * a.tsrc;
      }
    }
    if (a.scaled) k = k * p.kscale;
    // stage update (the modes of stage_update, 16-byte accesses; wave-uniform branches)
    const int64_t idx = pidx0 + (int64_t)r * ny;
    const int om = a.out_mode, am = a.acc_mode;
    Vec yv = Vec{}, av = Vec{};
    if (om == OUT_Y_PLUS_AK || am == ACC_INIT || om == OUT_K_LC) yv = *reinterpret_cast<const Vec*>(a.y + idx);
    if (am == ACC_ADD || om == OUT_ACC_PLUS_BK) av = *reinterpret_cast<const Vec*>(a.acc + idx);
    if (am == ACC_INIT) *reinterpret_cast<Vec*>(a.acc + idx) = yv + a.b * k;
    if (om == OUT_K) {
      *reinterpret_cast<Vec*>(a.out + idx) = k;
    } else if (om == OUT_Y_PLUS_AK) {
      *reinterpret_cast<Vec*>(a.out + idx) = yv + a.a * k;
    } else if (om == OUT_ACC_PLUS_BK) {
      *reinterpret_cast<Vec*>(a.out + idx) = av + a.b * k;
    } else if (om == OUT_K_LC) {
      *reinterpret_cast<Vec*>(a.out + idx) = k;
      Vec nxt = yv;
      for (int j = 0; j < a.lc.n; ++j) nxt += a.lc.c[j] * *reinterpret_cast<const Vec*>(a.lc.k[j] + idx);
      *reinterpret_cast<Vec*>(a.lc.next + idx) = nxt + a.lc.c[a.lc.n] * k;
    }
    if (am == ACC_ADD) *reinterpret_cast<Vec*>(a.acc + idx) = av + a.b * k;
  }
}

// ---------------------------------------------------------------------------------------------- host
template <typename T>
bool sbm_tiled_supported(const pdeopt_ctx* ctx) {
  constexpr int V = VecOf<T>::V;
  const pdeopt_problem& p = ctx->prob;
  if (p.equation != PDEOPT_EQ_ALLEN_CAHN_SBM && p.equation != PDEOPT_EQ_CAHN_HILLIARD_SBM) return false;
  if (ctx->halo || ctx->opt_kernel_path == 1) return false;
  return p.ny % V == 0 && p.nx >= 8 && p.ny >= 4 * V;
}

template <typename T, int EQ>
int launch_sbm_tiled_eq(pdeopt_ctx* ctx, const StageArgs<T>& s) {
  constexpr int V = VecOf<T>::V;
  const pdeopt_problem& p = ctx->prob;
  const int tiles_i = (p.nx + 15) / 16;
  const int tiles_j = (p.ny + kLanesPerRow * V - 1) / (kLanesPerRow * V);
  const int64_t nblk64 = (int64_t)tiles_i * tiles_j * ctx->win_n;
  if (nblk64 > 0x7fffffffLL) return fail(ctx, PDEOPT_EINVAL, "too many tiles");
  const int nblk = (int)nblk64;
  const size_t lds = sbm_tiled_lds_bytes<T, EQ>();
  const int flags = tile_flags(nblk, tiles_i, tiles_j);
  if (sbm_fast_closures(p))
    hipLaunchKernelGGL((sbm_tiled_kernel<T, EQ, true>), dim3(nblk), dim3(256), lds, ctx->stream, s, tiles_i, tiles_j, nblk, flags);
  else
    hipLaunchKernelGGL((sbm_tiled_kernel<T, EQ, false>), dim3(nblk), dim3(256), lds, ctx->stream, s, tiles_i, tiles_j, nblk, flags);
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  return PDEOPT_OK;
}

template <typename T>
int launch_sbm_tiled(pdeopt_ctx* ctx, const StageArgs<T>& s) {
  if (ctx->prob.equation == PDEOPT_EQ_ALLEN_CAHN_SBM) {
    ctx->last_kernel = sizeof(T) == 4 ? "sbm_tiled<f32,AC-SBM>" : "sbm_tiled<f64,AC-SBM>";
    return launch_sbm_tiled_eq<T, PDEOPT_EQ_ALLEN_CAHN_SBM>(ctx, s);
  }
  ctx->last_kernel = sizeof(T) == 4 ? "sbm_tiled<f32,CH-SBM>" : "sbm_tiled<f64,CH-SBM>";
  return launch_sbm_tiled_eq<T, PDEOPT_EQ_CAHN_HILLIARD_SBM>(ctx, s);
}

}  // namespace pdeopt
