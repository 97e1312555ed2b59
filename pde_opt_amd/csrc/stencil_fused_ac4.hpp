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
// threads) and writes w = y + c k into the LDS array the stage did not read: the two arrays alternate
// as input and output (y -> w1 -> w2 -> w3), one barrier per stage.  y, the base of every w, is held in
// registers for exactly the cells a thread updates (own cells + its ring vector of each stage), so the
// array y was loaded into is free from stage 2 on.  Redundant ring work x1.46 / 1.33 / 1.2 / 1.
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
#ifndef PDEOPT_AC4_DPP
#define PDEOPT_AC4_DPP 0  // measured: 17.9 k env-steps/s with the DPP shifts against 18.6 k with the conflicted LDS reads (512^2 x 64)
#endif
#ifndef PDEOPT_AC4_THREADS
#define PDEOPT_AC4_THREADS 512
#endif
#ifndef PDEOPT_AC4_HELPERS
#define PDEOPT_AC4_HELPERS 0  // measured 17.9 k against 19.7 k env-steps/s without (512^2 x 64): this kernel lives on its three workgroups per CU
#endif
struct Ac4Geom {
  // NOWN threads own the tile's cells (256 -> 16-row tiles, 512 -> 32-row tiles).  PDEOPT_AC4_HELPERS doubles the
  // workgroup: the second half owns nothing, shares the tile load and takes the ring vectors WHILE the owners
  // evaluate their cells (stencil_fused_ch4.hpp's scheme, where it pays) -- two 1024-thread workgroups per CU
  // (8 waves per SIMD) instead of three of 512 (6).  Slower here: the memory share of this kernel wants the third
  // workgroup more than its stages want the shorter path.
  static constexpr int NOWN = PDEOPT_AC4_THREADS;
  static constexpr int NT = PDEOPT_AC4_HELPERS ? 2 * NOWN : NOWN;
  static constexpr int V = 4, RPT = 2, TX = (NOWN / kLanesPerRow) * RPT, PV = kLanesPerRow + 2, P = PV * V,
                       TY = kLanesPerRow * V;
  static constexpr int kRows = TX + 8;  // LDS rows: tile row + 4
  static constexpr size_t lds_bytes() { return (size_t)(2 * kRows * P + 3 * V) * sizeof(float); }
  // ring of stage with halo h (the region tile+h minus the tile): 2h full rows + 2 side vectors per tile row
  static constexpr int ring(int h) { return 2 * h * PV + 2 * TX; }
};

// Where a launch's time goes, measured on 512^2 x 64 with the global load / the store / both compiled out
// (tools/mkvariant.sh -DPDEOPT_AC4_ABLATE=1|2|3): 34 us whole, 28 without the load, 26 without the store, 23 with
// neither.  Tried against that in round 3, same box, interleaved runs (profiles/r03_ac4_experiments.txt):
//   * y of the updated cells in registers + the two LDS arrays alternating as stage input / output (one barrier per
//     stage instead of two): +6.5 % -- this kernel;
//   * left / right neighbours by DPP wavefront shifts instead of the bank-conflicted scalar LDS reads: -3.6 %;
//   * persistent workgroups (3 per CU) walking tiles with the next tile's loads in flight in registers during the
//     stages and the stores draining behind: -4 % (after removing a vmcnt(0) the environment-parameter loads forced
//     and a spill of the prefetch registers; 80 VGPRs).  Hiding a workgroup's own load / store latency buys nothing:
//     the 11 us are the memory system's time for 134 MB with 768 workgroups' requests in flight, not exposed latency.
template <int CL, bool RAGGED>
__global__ __launch_bounds__(Ac4Geom::NT) void ac_rk4_quad_kernel(const QuadArgs<float> a, const int tiles_i, const int tiles_j,
                                                          const int nblk, const int xcd_remap) {
  using T = float;
  using Vec = typename VecOf<T>::type;
  using G = Ac4Geom;
  constexpr int V = G::V, RPT = G::RPT, TX = G::TX, PV = G::PV, P = G::P, TY = G::TY;
  constexpr int NT = G::NT;
  constexpr bool HELPERS = NT > G::NOWN;
  static_assert(G::ring(3) <= (HELPERS ? NT - G::NOWN : NT), "largest ring must fit one pass");

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
  const int lane = tid & 63;
  const bool own_l = lx == 0, own_r = lx == kLanesPerRow - 1;
  const bool owner = tid < G::NOWN;                 // wave-uniform
  const int rw = HELPERS ? tid - G::NOWN : tid;     // ring worker index (negative: none)

  constexpr bool ragged = RAGGED;
  auto wrap_row = [&](int gi) { return g.periodic ? tile_wrap(gi, g.nx, ragged) : gi; };
  auto wrap_col = [&](int gj) { return g.periodic ? tile_wrap(gj, g.ny, ragged) : gj; };
  const bool col_ok = !RAGGED || (j0 + lx * V) < g.ny;
  auto cell_ok = [&](int r) { return !RAGGED || (col_ok && (i0 + r0 + r) < g.nx); };

  // ---- load y on tile + 4
#if defined(PDEOPT_AC4_ABLATE) && (PDEOPT_AC4_ABLATE & 1)  // TIMING ONLY (tools/mkvariant.sh): no global load
  for (int i = tid; i < G::kRows * PV; i += NT) *reinterpret_cast<Vec*>(sY + (i / PV) * P + (i % PV) * V) = Vec{0.1f, 0.2f, 0.1f, 0.2f} * a.dt;
#else
  load_rows_per_wave<T, V, PV, NT, G::kRows, Vec>(sY, P, in, ld, i0 - 4, j0 - V, wrap_row, wrap_col, tid);
#endif
  __syncthreads();

  // Constant mobility (CL_POLY_M0): k = -R (mu_h(u) - kappa lap u) is ONE cubic in u plus two weighted
  // neighbour sums,  k = q(u) + A (u_x+ + u_x-) + B (u_y+ + u_y-),  A = R kappa / hx^2, B = R kappa / hy^2,
  // q = -R mu_h - 2 (A + B) u  with the coefficients folded per environment: 7 instructions per cell
  // instead of 11.  Algebraically the same expression, re-associated: the rounding differs from the literal
  // form at the ulp level of the state per substep.
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
  // k at one vector (tile row r, LDS vector column cv) of the field held in `src`.
  // The scalar neighbours left / right of the vector are read from LDS (ds_read_b32 with a 16-byte lane stride: a
  // 4-way bank conflict).  PDEOPT_AC4_DPP takes them from the adjacent lanes' registers instead (adjacent lanes hold
  // adjacent vectors; `lds_l` / `lds_r`: lanes at a row's or a wave's end and the ring's side vectors still read
  // LDS): measured slower (above) -- the LDS has room (SQ_LDS_IDX_ACTIVE 0.52 of the launch) and the shifts add VALU
  // work on the dependent path.  Off by default.
  auto k_at = [&](const T* src, const int r, const int cv, const bool lds_l, const bool lds_r) -> Vec {
    const T* up = src + (r + 4) * P + cv * V;
    const Vec c = *reinterpret_cast<const Vec*>(up);
    const Vec xp = *reinterpret_cast<const Vec*>(up + P);
    const Vec xm = *reinterpret_cast<const Vec*>(up - P);
#if PDEOPT_AC4_DPP
    T left = lane_from_prev(T(0), c[V - 1]), right = lane_from_next(T(0), c[0]);
    if (lds_l) left = up[-1];
    if (lds_r) right = up[V];
#else
    const T left = up[-1], right = up[V];
    (void)lds_l; (void)lds_r;
#endif
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

  Vec acc[RPT];   // sum_i b_i k_i on the own cells, b = (1, 2, 2, 1) / 6
  Vec yown[RPT];  // y on the own cells: the base of every w and of y', read from LDS once instead of once per stage
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    acc[r] = Vec{};
    yown[r] = owner ? *reinterpret_cast<const Vec*>(sY + (r0 + r + 4) * P + cvo * V) : Vec{};
  }
  // y at the ring vector this thread computes in a stage (the rings of tile+3, +2, +1 are different cells) is read
  // from the y array where the stage needs it: stages 1 and 2 find it intact -- stage 2 overwrites that array with
  // w2, but a cell is written by the one thread that also reads its y, in that order -- and stage 3's is fetched
  // up front, before stage 2 runs over it.  With yown this is every use of y after the load.
  int ring_r[3], ring_c[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    ring_r[q] = ring_c[q] = 0;
    if (rw >= 0 && rw < G::ring(3 - q)) ring_coord(3 - q, rw, &ring_r[q], &ring_c[q]);
  }
  Vec yring3 = Vec{};
  if (rw >= 0 && rw < G::ring(1)) yring3 = *reinterpret_cast<const Vec*>(sY + (ring_r[2] + 4) * P + ring_c[2] * V);
  const T half = T(0.5) * a.dt;

  // One of the stages 1..3: k on own cells + the ring of tile+H, then w = y + c k into the OTHER array -- the two
  // arrays alternate as stage input and output (y -> w1 -> w2 -> w3), so a stage ends in ONE barrier: nobody reads
  // `dst` during this stage (its last readers passed the previous stage's barrier), nobody writes `src`.
  // Round 2 wrote w in place and needed a second barrier per stage between the reads and the writes.
  auto stage = [&](const T* src, T* dst, const int q, const T cw, const T bw) {
    const int H = 3 - q;
    if (rw >= 0 && rw < G::ring(H)) {
      const int rr = ring_r[q], rc = ring_c[q];
      const bool side = rw >= 2 * H * PV;  // the two vectors beside each tile row: lanes alternate left / right
      const Vec k = k_at(src, rr, rc, side || rc == 0 || lane == 0, side || rc == PV - 1 || lane == 63);
      const Vec yr = q == 2 ? yring3 : *reinterpret_cast<const Vec*>(sY + (rr + 4) * P + rc * V);
      *reinterpret_cast<Vec*>(dst + (rr + 4) * P + rc * V) = yr + cw * k;
    }
    if (owner) {
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const Vec k = k_at(src, r0 + r, cvo, own_l, own_r);
        acc[r] += bw * k;
        *reinterpret_cast<Vec*>(dst + (r0 + r + 4) * P + cvo * V) = yown[r] + cw * k;
      }
    }
    __syncthreads();
  };

  stage(sY, sW, 0, half, T(1));  // k1 on tile+3, w1 = y + dt/2 k1
  stage(sW, sY, 1, half, T(2));  // k2 on tile+2, w2 = y + dt/2 k2
  stage(sY, sW, 2, a.dt, T(2));  // k3 on tile+1, w3 = y + dt   k3

  // ---- stage 4 on the tile, combine, store
  const int64_t pidx0 = base + (int64_t)(i0 + r0) * ld + (j0 + lx * V);
  const T sixth = a.dt * T(1.0 / 6.0);
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    if (!owner) break;
    const Vec k4 = k_at(sW, r0 + r, cvo, own_l, own_r);
    if (!cell_ok(r)) continue;
#if defined(PDEOPT_AC4_ABLATE) && (PDEOPT_AC4_ABLATE & 2)  // TIMING ONLY: no global store (the condition never holds)
    if (k4[0] != 1.2345e-30f) continue;
#endif
    *reinterpret_cast<Vec*>(a.out + pidx0 + r * ld) = yown[r] + sixth * (acc[r] + k4);
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
