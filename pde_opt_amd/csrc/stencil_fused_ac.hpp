// Temporally fused Runge-Kutta stage pairs for Allen-Cahn (radius-1 stencil per stage):
// the PAIR_12 / PAIR_34 scheme of stencil_fused.hpp with a 2-cell input halo and a 1-cell ring.
//   k = -R(u) (mu_h(u) - kappa lap u)                      allen_cahn.py:81-84, derivatives.py:8-12
// Phases: P1 stage-A input on tile+2 -> LDS;  P3 k_A on the own micro-tile + one ring vector per
// thread -> registers;  P4 w = base + a_A k_A on tile+1 -> LDS in place;  P6 k_B, RK updates, stores.
#pragma once

#include "stencil_fused.hpp"

namespace pdeopt {

template <typename T, int RPT>
constexpr size_t fused_ac_lds_bytes() {
  constexpr int V = VecOf<T>::V;
  return ((size_t)(8 * RPT + 4) * (kLanesPerRow + 2) * V + 2 * V) * sizeof(T);
}

template <typename T, int CL, int PAIR, int RPT, bool RAGGED>
__global__ __launch_bounds__(256) void stage_pair_ac_kernel(const PairArgs<T> a, const int tiles_i,
                                                            const int tiles_j, const int nblk,
                                                            const int xcd_remap) {
  using Vec = typename VecOf<T>::type;
  constexpr int V = VecOf<T>::V;
  constexpr int TX = 8 * RPT;
  constexpr int PV = kLanesPerRow + 2;  // one halo vector per side (>= 2 columns for both dtypes)
  constexpr int P = PV * V;
  constexpr int TY = kLanesPerRow * V;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* const sU = reinterpret_cast<T*>(smem_raw) + V;  // rows: tile row + 2, cols: tile col + V

  int ti, tj, b;
  decode_tile(blockIdx.x, tiles_i, tiles_j, nblk, xcd_remap, &ti, &tj, &b);
  if (tile_skipped(a.part, ti, tj, tiles_i, tiles_j)) return;
  const int i0 = ti * TX;
  const int j0 = tj * TY;

  const Geo& g = a.g;
  const int64_t ld = g.ld;
  const int64_t base = (int64_t)b * g.bstride + g.off;
  const EnvParams<T>& p = a.ep[b];
  const T* __restrict__ in = a.in + base;
  const T kap = p.kappa;

  const int tid = threadIdx.x;
  const int lx = tid & 31;
  const int ly = tid >> 5;
  const int r0 = ly * RPT;
  const int cvo = lx + 1;

  // tile + 1 ring minus the tile: rows -1 and TX (PV vectors each) + 2 side vectors per tile row
  constexpr int kRingTop = 2 * PV;
  constexpr int kRing = kRingTop + 2 * TX;
  static_assert(kRing <= 256, "ring must fit one pass");
  int ring_r = 0, ring_cv = 0;
  const bool has_ring = tid < kRing;
  if (tid < kRingTop) {
    const int q = tid / PV;
    ring_r = q ? TX : -1;
    ring_cv = tid - q * PV;
  } else if (has_ring) {
    const int t2 = tid - kRingTop;
    ring_r = t2 >> 1;
    ring_cv = (t2 & 1) ? (PV - 1) : 0;
  }

  constexpr bool ragged = RAGGED;  // compile-time: the divisible case pays nothing for the masks
  auto wrap_row = [&](int gi) { return g.periodic ? tile_wrap(gi, g.nx, ragged) : gi; };
  auto wrap_col = [&](int gj) { return g.periodic ? tile_wrap(gj, g.ny, ragged) : gj; };
  // ragged tiles: every lane computes, only cells inside the grid are loaded pointwise / stored
  const bool col_ok = !RAGGED || (j0 + lx * V) < g.ny;
  auto cell_ok = [&](int r) { return !RAGGED || (col_ok && (i0 + r0 + r) < g.nx); };

  const int64_t pidx0 = base + (int64_t)(i0 + r0) * ld + (j0 + lx * V);
  Vec ybase[RPT], accp[RPT], yring;
  if constexpr (PAIR == PAIR_34) {
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      accp[r] = Vec{};
      if constexpr (RAGGED) {
        // cells of the tile that lie beyond the grid are periodic images: stage B of the cells next
        // to the grid edge reads w there, so their base value is needed (wrapped), not masked
        const int gi = wrap_row(i0 + r0 + r), gj = wrap_col(j0 + lx * V);
        ybase[r] = *reinterpret_cast<const Vec*>(a.y + base + (int64_t)gi * ld + gj);
        if (cell_ok(r)) accp[r] = *reinterpret_cast<const Vec*>(a.acc + pidx0 + r * ld);
      } else {
        ybase[r] = *reinterpret_cast<const Vec*>(a.y + pidx0 + r * ld);
        accp[r] = *reinterpret_cast<const Vec*>(a.acc + pidx0 + r * ld);
      }
    }
    if (has_ring) {
      const int gi = wrap_row(i0 + ring_r);
      const int gj = wrap_col(j0 + (ring_cv - 1) * V);
      yring = *reinterpret_cast<const Vec*>(a.y + base + (int64_t)gi * ld + gj);
    }
  }

  // ---- P1: stage-A input, tile + 2 rows / one halo vector of columns
  load_rows_per_wave<T, V, PV, 256, TX + 4, Vec>(sU, P, in, ld, i0 - 2, j0 - V, wrap_row, wrap_col, tid);
  __syncthreads();

  // k at one vector: tile row r, LDS vector column cv
  auto k_at = [&](const int r, const int cv, Vec* centre) -> Vec {
    const T* up = sU + (r + 2) * P + cv * V;
    const Vec c = *reinterpret_cast<const Vec*>(up);
    const Vec xp = *reinterpret_cast<const Vec*>(up + P);
    const Vec xm = *reinterpret_cast<const Vec*>(up - P);
    const T left = up[-1], right = up[V];
    if (centre) *centre = c;
    Vec k;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const T ym = (e == 0) ? left : c[e - 1];
      const T yp = (e == V - 1) ? right : c[e + 1];
      const T mu = eval_mu<T, CL>(a.mu, p.mu, c[e]) - kap * lap_at<T>(c[e], xp[e], xm[e], yp, ym, a.rhx2, a.rhy2);
      k[e] = -eval_mob<T, CL>(a.mob, p.mob, c[e]) * mu;
    }
    return k;
  };

  // ---- P3: k_A on the own micro-tile and one ring vector
  Vec w_own[RPT], yown[RPT], w_ring;
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    Vec uc;
    const Vec kA = k_at(r0 + r, cvo, &uc);
    if constexpr (PAIR == PAIR_12) {
      yown[r] = uc;
      w_own[r] = uc + a.aA * kA;
      accp[r] = uc + a.bA * kA;
    } else {
      w_own[r] = ybase[r] + a.aA * kA;
      accp[r] = accp[r] + a.bA * kA;
    }
  }
  if (has_ring) {
    Vec uc;
    const Vec kA = k_at(ring_r, ring_cv, &uc);
    w_ring = (PAIR == PAIR_12 ? uc : yring) + a.aA * kA;
  }
  __syncthreads();

  // ---- P4: w -> sU in place (tile + 1)
#pragma unroll
  for (int r = 0; r < RPT; ++r) *reinterpret_cast<Vec*>(sU + (r0 + r + 2) * P + cvo * V) = w_own[r];
  if (has_ring) *reinterpret_cast<Vec*>(sU + (ring_r + 2) * P + ring_cv * V) = w_ring;
  __syncthreads();

  // ---- P6: k_B, stage updates, stores
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const Vec kB = k_at(r0 + r, cvo, nullptr);
    if (!cell_ok(r)) continue;
    const int64_t idx = pidx0 + r * ld;
    if constexpr (PAIR == PAIR_12) {
      *reinterpret_cast<Vec*>(a.out + idx) = yown[r] + a.aB * kB;
      *reinterpret_cast<Vec*>(a.acc_out + idx) = accp[r] + a.bB * kB;
    } else {
      *reinterpret_cast<Vec*>(a.out + idx) = accp[r] + a.bB * kB;
    }
  }
}

template <typename T, int CL, int PAIR, int RPT>
int launch_pair_ac_inst(pdeopt_ctx* ctx, const PairArgs<T>& s) {
  constexpr int V = VecOf<T>::V;
  const pdeopt_problem& p = ctx->prob;
  const int tiles_i = (p.nx + 8 * RPT - 1) / (8 * RPT);
  const int tiles_j = (p.ny + kLanesPerRow * V - 1) / (kLanesPerRow * V);
  const int64_t nblk64 = (int64_t)tiles_i * tiles_j * ctx->win_n;
  if (nblk64 > 0x7fffffffLL) return fail(ctx, PDEOPT_EINVAL, "too many tiles");
  const int nblk = (int)nblk64;
  const size_t lds = fused_ac_lds_bytes<T, RPT>();
  const bool ragged = p.nx % (8 * RPT) != 0 || p.ny % (kLanesPerRow * V) != 0;
  if (ragged)
    hipLaunchKernelGGL((stage_pair_ac_kernel<T, CL, PAIR, RPT, true>), dim3(nblk), dim3(256), lds, ctx->stream, s, tiles_i,
                       tiles_j, nblk, tile_flags(nblk, tiles_i, tiles_j));
  else
    hipLaunchKernelGGL((stage_pair_ac_kernel<T, CL, PAIR, RPT, false>), dim3(nblk), dim3(256), lds, ctx->stream, s, tiles_i,
                       tiles_j, nblk, tile_flags(nblk, tiles_i, tiles_j));
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  return PDEOPT_OK;
}

}  // namespace pdeopt
