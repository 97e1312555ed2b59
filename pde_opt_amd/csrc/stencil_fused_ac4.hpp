// A whole classical RK4 substep of Allen-Cahn in ONE pass over HBM (fp32): the four radius-1 stages
// are chained inside a workgroup, so a substep reads y once (tile + 4 halo) and writes y' once --
// 2 words per cell instead of the 7 of the stage-pair kernels (stencil_fused_ac.hpp), which run at
// the fabric's bandwidth on BASELINE config 2 (512^2 x 64, cache-resident).
//
//   k1 = f(y)            on tile+3      w1 = y + dt/2 k1     (LDS, tile+3)
//   k2 = f(w1)           on tile+2      w2 = y + dt/2 k2     (LDS, in place, tile+2)
//   k3 = f(w2)           on tile+1      w3 = y + dt   k3     (LDS, in place, tile+1)
//   k4 = f(w3)           on tile        y' = y + dt/6 (k1 + 2 k2 + 2 k3 + k4)
//   f(u) = -R(u) (mu_h(u) - kappa lap u)                     allen_cahn.py:81-84, derivatives.py:8-12
//
// Every stage evaluates k on the thread's own micro-tile (2 rows x 1 vector, accumulated in registers)
// and on ONE vector of the ring the later stages still need (rings of 236 / 168 / 100 vectors <= 256
// threads), keeps the results in registers across a barrier and then overwrites the stage input in
// place.  y stays in a second LDS array as the base of every w.  Redundant ring work x1.46 / 1.33 / 1.2 / 1.
// Columns: the 4-cell halo is one 16-byte vector per side; validity shrinks by one column per stage
// from its outer edge, exactly as far as the next stage reads.
#pragma once

#include "stencil_fused_ac.hpp"

namespace pdeopt {

template <typename T>
struct QuadArgs {
  const T* y;  // state (read: tile + 4)
  T* out;      // y' (a different buffer: neighbouring tiles still read y)
  T dt;
  T rhx2, rhy2;
  Geo g;
  const EnvParams<T>* ep;
  ClosureSpec mu, mob;
};

#ifndef PDEOPT_AC4_FOLD
#define PDEOPT_AC4_FOLD 1
#endif
#ifndef PDEOPT_AC4_THREADS
#define PDEOPT_AC4_THREADS 512
#endif
struct Ac4Geom {
  static constexpr int NT = PDEOPT_AC4_THREADS;  // 256 -> 16-row tiles, 512 -> 32-row tiles
  static constexpr int V = 4, RPT = 2, TX = (NT / kLanesPerRow) * RPT, PV = kLanesPerRow + 2, P = PV * V,
                       TY = kLanesPerRow * V;
  static constexpr int kRows = TX + 8;  // LDS rows: tile row + 4
  static constexpr size_t lds_bytes() { return (size_t)(2 * kRows * P + 3 * V) * sizeof(float); }
  // ring of stage with halo h (the region tile+h minus the tile): 2h full rows + 2 side vectors per tile row
  static constexpr int ring(int h) { return 2 * h * PV + 2 * TX; }
};

template <int CL, bool RAGGED>
__global__ __launch_bounds__(Ac4Geom::NT) void ac_rk4_quad_kernel(const QuadArgs<float> a, const int tiles_i, const int tiles_j,
                                                          const int nblk, const int xcd_remap) {
  using T = float;
  using Vec = typename VecOf<T>::type;
  using G = Ac4Geom;
  constexpr int V = G::V, RPT = G::RPT, TX = G::TX, PV = G::PV, P = G::P, TY = G::TY;
  constexpr int NT = G::NT;
  static_assert(G::ring(3) <= NT, "largest ring must fit one pass");

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* const sY = reinterpret_cast<T*>(smem_raw) + V;  // rows: tile row + 4, cols: tile col + V
  T* const sW = sY + G::kRows * P + V;

  int ti, tj, b;
  decode_tile(blockIdx.x, tiles_i, tiles_j, nblk, xcd_remap, &ti, &tj, &b);
  const int i0 = ti * TX;
  const int j0 = tj * TY;

  const Geo& g = a.g;
  const int64_t ld = g.ld;
  const int64_t base = (int64_t)b * g.bstride + g.off;
  const EnvParams<T>& p = a.ep[b];
  const T kap = p.kappa;
  const T* __restrict__ in = a.y + base;

  const int tid = threadIdx.x;
  const int lx = tid & 31;
  const int ly = tid >> 5;
  const int r0 = ly * RPT;
  const int cvo = lx + 1;

  constexpr bool ragged = RAGGED;
  auto wrap_row = [&](int gi) { return g.periodic ? tile_wrap(gi, g.nx, ragged) : gi; };
  auto wrap_col = [&](int gj) { return g.periodic ? tile_wrap(gj, g.ny, ragged) : gj; };
  const bool col_ok = !RAGGED || (j0 + lx * V) < g.ny;
  auto cell_ok = [&](int r) { return !RAGGED || (col_ok && (i0 + r0 + r) < g.nx); };

  // ---- load y on tile + 4 into both arrays' source (sY); stage 1 reads sY directly
  load_rows_per_wave<T, V, PV, NT, G::kRows, Vec>(sY, P, in, ld, i0 - 4, j0 - V, wrap_row, wrap_col, tid);
  __syncthreads();

  // Constant mobility (CL_POLY_M0): k = -R (mu_h(u) - kappa lap u) is ONE cubic in u plus two weighted
  // neighbour sums,  k = q(u) + A (u_x+ + u_x-) + B (u_y+ + u_y-),  A = R kappa / hx^2, B = R kappa / hy^2,
  // q = -R mu_h - 2 (A + B) u  with the coefficients folded per environment: 7 instructions per cell
  // instead of 11 in this VALU-bound kernel.  Algebraically the same expression, re-associated: the
  // rounding differs from the literal form at the ulp level of the state per substep.
  constexpr bool FOLD = PDEOPT_AC4_FOLD && CL == CL_POLY_M0;
  T fA = T(0), fB = T(0), q0 = T(0), q1 = T(0), q2 = T(0), q3 = T(0);
  if constexpr (FOLD) {
    const T R = p.mob[0];
    fA = R * kap * a.rhx2;
    fB = R * kap * a.rhy2;
    q0 = -R * p.mu[0];
    q1 = -R * p.mu[1] - T(2) * (fA + fB);
    q2 = -R * p.mu[2];
    q3 = -R * p.mu[3];
  }
  // k at one vector (tile row r, LDS vector column cv) of the field held in `src`
  auto k_at = [&](const T* src, const int r, const int cv) -> Vec {
    const T* up = src + (r + 4) * P + cv * V;
    const Vec c = *reinterpret_cast<const Vec*>(up);
    const Vec xp = *reinterpret_cast<const Vec*>(up + P);
    const Vec xm = *reinterpret_cast<const Vec*>(up - P);
    const T left = up[-1], right = up[V];
    Vec k;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const T ym = (e == 0) ? left : c[e - 1];
      const T yp = (e == V - 1) ? right : c[e + 1];
      if constexpr (FOLD) {
        const T q = ((q3 * c[e] + q2) * c[e] + q1) * c[e] + q0;
        k[e] = fA * (xp[e] + xm[e]) + (fB * (yp + ym) + q);
        continue;
      }
      const T mu = eval_mu<T, CL>(a.mu, p.mu, c[e]) - kap * lap_at<T>(c[e], xp[e], xm[e], yp, ym, a.rhx2, a.rhy2);
      k[e] = -eval_mob<T, CL>(a.mob, p.mob, c[e]) * mu;
    }
    return k;
  };
  // ring vector `idx` of the region tile + h: h rows above, h rows below (all PV vectors), two side vectors per tile row
  auto ring_coord = [&](const int h, const int idx, int* r, int* cv) {
    const int top = 2 * h * PV;
    if (idx < top) {
      const int q = idx / PV;
      *r = (q < h) ? (q - h) : (TX + q - h);
      *cv = idx - q * PV;
    } else {
      const int t2 = idx - top;
      *r = t2 >> 1;
      *cv = (t2 & 1) ? (PV - 1) : 0;
    }
  };

  Vec acc[RPT];  // sum_i b_i k_i on the own cells, b = (1, 2, 2, 1) / 6
#pragma unroll
  for (int r = 0; r < RPT; ++r) acc[r] = Vec{};
  const T half = T(0.5) * a.dt;

  // one of the stages 1..3: k on own cells + the ring of tile+H, then w = y + c k in place into sW
  auto stage = [&](const T* src, const int H, const T cw, const T bw) {
    Vec w_own[RPT], w_ring;
    int rr = 0, rc = 0;
    const bool has_ring = tid < G::ring(H);
    if (has_ring) {
      ring_coord(H, tid, &rr, &rc);
      const Vec k = k_at(src, rr, rc);
      w_ring = *reinterpret_cast<const Vec*>(sY + (rr + 4) * P + rc * V) + cw * k;
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const Vec k = k_at(src, r0 + r, cvo);
      acc[r] += bw * k;
      w_own[r] = *reinterpret_cast<const Vec*>(sY + (r0 + r + 4) * P + cvo * V) + cw * k;
    }
    __syncthreads();  // every read of the stage input is done
#pragma unroll
    for (int r = 0; r < RPT; ++r) *reinterpret_cast<Vec*>(sW + (r0 + r + 4) * P + cvo * V) = w_own[r];
    if (has_ring) *reinterpret_cast<Vec*>(sW + (rr + 4) * P + rc * V) = w_ring;
    __syncthreads();
  };

  stage(sY, 3, half, T(1));  // k1 on tile+3, w1 = y + dt/2 k1
  stage(sW, 2, half, T(2));  // k2 on tile+2, w2 = y + dt/2 k2
  stage(sW, 1, a.dt, T(2));  // k3 on tile+1, w3 = y + dt k3

  // ---- stage 4 on the tile, combine, store
  const int64_t pidx0 = base + (int64_t)(i0 + r0) * ld + (j0 + lx * V);
  const T sixth = a.dt * T(1.0 / 6.0);
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const Vec k4 = k_at(sW, r0 + r, cvo);
    if (!cell_ok(r)) continue;
    const Vec y = *reinterpret_cast<const Vec*>(sY + (r0 + r + 4) * P + cvo * V);
    *reinterpret_cast<Vec*>(a.out + pidx0 + r * ld) = y + sixth * (acc[r] + k4);
  }
}

// whether the single-pass RK4 kernel covers the configured problem
inline bool ac_quad_supported(const pdeopt_ctx* ctx) {
  const pdeopt_problem& p = ctx->prob;
  if (p.equation != PDEOPT_EQ_ALLEN_CAHN || p.dtype != PDEOPT_F32 || p.derivs != PDEOPT_DERIVS_FD) return false;
  if (ctx->halo || ctx->opt_kernel_path == 1 || ctx->opt_fuse_stages < 0 || ctx->opt_fuse_stages == 1) return false;
  if (!tiled_supported<float>(ctx)) return false;
  return classify_closures(p.mu, p.mob) == CL_POLY;
}

inline int launch_ac_quad(pdeopt_ctx* ctx, const void* y, void* out, double dt) {
  using G = Ac4Geom;
  const pdeopt_problem& p = ctx->prob;
  QuadArgs<float> s{};
  s.g = make_geo(ctx);
  const int64_t woff = (int64_t)ctx->win_lo * s.g.bstride;
  s.y = static_cast<const float*>(y) + woff;
  s.out = static_cast<float*>(out) + woff;
  s.dt = (float)dt;
  s.rhx2 = (float)(1.0 / (p.hx * p.hx));
  s.rhy2 = (float)(1.0 / (p.hy * p.hy));
  s.ep = static_cast<const EnvParams<float>*>(ctx->env_params_dev) + ctx->win_lo;
  s.mu = ClosureSpec{p.mu.kind, p.mu.flags, p.mu.n};
  s.mob = ClosureSpec{p.mob.kind, p.mob.flags, p.mob.n};
  const int tiles_i = (p.nx + G::TX - 1) / G::TX;
  const int tiles_j = (p.ny + G::TY - 1) / G::TY;
  const int64_t nblk64 = (int64_t)tiles_i * tiles_j * ctx->win_n;
  if (nblk64 > 0x7fffffffLL) return fail(ctx, PDEOPT_EINVAL, "too many tiles");
  const int nblk = (int)nblk64;
  const bool ragged = p.nx % G::TX != 0 || p.ny % G::TY != 0;
  ctx->n_stage_launches++;
  ctx->last_kernel = G::TX == 32 ? "rk4_quad<f32,AC,poly,rows32>" : "rk4_quad<f32,AC,poly,rows16>";
  const int remap = tile_flags(nblk, tiles_i, tiles_j);
#ifndef PDEOPT_AC4_M0
#define PDEOPT_AC4_M0 1
#endif
  const bool m0 = PDEOPT_AC4_M0 && p.mob.n <= 1;  // constant mobility (coefficients past n are stored as zeros)
#define PDEOPT_QUAD_LAUNCH(CLV, RG) \
  hipLaunchKernelGGL((ac_rk4_quad_kernel<CLV, RG>), dim3(nblk), dim3(G::NT), G::lds_bytes(), ctx->stream, s, tiles_i, \
                     tiles_j, nblk, remap)
  if (m0) {
    if (ragged) PDEOPT_QUAD_LAUNCH(CL_POLY_M0, true); else PDEOPT_QUAD_LAUNCH(CL_POLY_M0, false);
  } else {
    if (ragged) PDEOPT_QUAD_LAUNCH(CL_POLY, true); else PDEOPT_QUAD_LAUNCH(CL_POLY, false);
  }
#undef PDEOPT_QUAD_LAUNCH
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  return PDEOPT_OK;
}

}  // namespace pdeopt
