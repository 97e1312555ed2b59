// Wave-local variant of the fused Cahn-Hilliard stage-pair kernel: every WAVE owns a 16 x 64-cell
// (fp64: 16 x 32) tile with its own LDS slice and runs all six phases of stencil_fused.hpp on it by
// itself, so there is no s_barrier in the kernel -- phases are ordered by program order alone (the
// LDS unit serves one wave's instructions in order).
//
// Why: the phase ablation of stage_pair_kernel shows LDS work, arithmetic and global
// memory being used almost serially -- the four waves of a workgroup move through the barrier-separated
// phases in lock step and the 4-5 co-resident workgroups overlap them poorly.  Independent waves drift
// apart, so one wave's tile load, another's mu pass (VALU) and a third's flux march (LDS + VALU) run
// at the same time on a SIMD.  The register FFT passes made the same move (fft_reg.hpp).
//
// Costs: the tile of a wave is half the workgroup tile, so stage A's redundant ring is relatively
// larger (mu_A x1.55, k_A x1.41 against x1.46, x1.33) and 13.2 KB of LDS per wave leave 12 waves per CU;
// each lane marches 4 rows (RPT 4) instead of 2.
// The tile load is an LDS-DMA (global_load_lds_dwordx4) of the wave's own 24 x 18 vectors.
// Arithmetic and flux routine are those of stage_pair_kernel (same results up to FMA contraction).
#pragma once

#include "stencil_fused_pipe.hpp"

namespace pdeopt {

__device__ __forceinline__ void wave_order() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ void wave_lds_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <typename T>
struct WaveGeom {
  static constexpr int V = VecOf<T>::V;
  static constexpr int HV = 4 / V;
#ifndef PDEOPT_WAVE_LPR
#define PDEOPT_WAVE_LPR 16
#endif
  static constexpr int LPR = PDEOPT_WAVE_LPR;  // lanes (vectors) per tile row: 16 (64-cell rows) or 8
  static constexpr int RPT = LPR / 4;          // rows per lane, so that the tile has 16 rows
  static constexpr int TX = (64 / LPR) * RPT;  // 16 rows
  static constexpr int PV = LPR + 2 * HV;
  static constexpr int P = PV * V;
  static constexpr int TY = LPR * V;
  static constexpr int kLoadVecs = (TX + 8) * PV;
  static constexpr int kLoadVecsPad = (kLoadVecs + 63) / 64 * 64;
  static constexpr int kSU = kLoadVecsPad * V;
  static constexpr int kSMu = (TX + 6) * P;
  static constexpr int kWaveElems = 3 * V + kSU + kSMu;  // [pad V][sU][pad V][sMu][pad V]
  static constexpr size_t lds_bytes() { return (size_t)4 * kWaveElems * sizeof(T); }
};

template <typename T, int CL, int PAIR, bool RAGGED>
__global__ __launch_bounds__(256) void stage_pair_wave_kernel(const PairArgs<T> a, const int tiles_i,
                                                              const int tiles_j, const int ntiles,
                                                              const int nblk, const int xcd_remap) {
  using Vec = typename VecOf<T>::type;
  using G = WaveGeom<T>;
  constexpr int V = G::V, HV = G::HV, LPR = G::LPR, RPT = G::RPT, TX = G::TX, PV = G::PV, P = G::P, TY = G::TY;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  T* const sU = reinterpret_cast<T*>(smem_raw) + wave * G::kWaveElems + V;
  T* const sMu = sU + G::kSU + V;

  int blk = blockIdx.x;
  if (xcd_remap) blk = (blk & 7) * (nblk >> 3) + (blk >> 3);
  const int t = uniform_i(blk * 4 + wave);
  if (t >= ntiles) return;  // whole wave; there is no barrier to miss
  const int tj = t % tiles_j;
  const int ti = (t / tiles_j) % tiles_i;
  const int tb = t / (tiles_j * tiles_i);

  const Geo& g = a.g;
  const int64_t ld = g.ld;
  const int lx = lane % LPR;
  const int ly = lane / LPR;
  const int r0 = ly * RPT;
  const int cvo = lx + HV;
  const int i0 = ti * TX;
  const int j0 = tj * TY;
  const int64_t base = (int64_t)tb * g.bstride + g.off;

  constexpr bool ragged = RAGGED;
  auto wrap_row = [&](int gi) { return g.periodic ? tile_wrap(gi, g.nx, ragged) : gi; };
  auto wrap_col = [&](int gj) { return g.periodic ? tile_wrap(gj, g.ny, ragged) : gj; };
  const bool col_ok = !RAGGED || (j0 + lx * V) < g.ny;
  auto cell_ok = [&](int r) { return !RAGGED || (col_ok && (i0 + r0 + r) < g.nx); };

  // ring of the tile (tile + 2 minus the tile): 4 full rows + 2 side vectors per row, two per lane
  constexpr int kRingRowVecs = LPR + 2;
  constexpr int kRingTop = 4 * kRingRowVecs;
  constexpr int kRing = kRingTop + 2 * TX;
  constexpr int kRingTrips = (kRing + 63) / 64;
  int ring_r[kRingTrips], ring_cv[kRingTrips];
  bool has_ring[kRingTrips];
#pragma unroll
  for (int q2 = 0; q2 < kRingTrips; ++q2) {
    const int rid = lane + 64 * q2;
    has_ring[q2] = rid < kRing;
    ring_r[q2] = 0;
    ring_cv[q2] = 0;
    if (rid < kRingTop) {
      const int q = rid / kRingRowVecs;
      ring_r[q2] = (q < 2) ? (q - 2) : (TX + q - 2);
      ring_cv[q2] = HV - 1 + (rid - q * kRingRowVecs);
    } else if (rid < kRing) {
      const int t2 = rid - kRingTop;
      ring_r[q2] = t2 >> 1;
      ring_cv[q2] = (t2 & 1) ? (HV + LPR) : (HV - 1);
    }
  }

  // ---- P1: the wave's input image (tile + 4) by LDS-DMA
  {
    const T* __restrict__ src = a.in + base;
#pragma unroll
    for (int it = 0; it < (G::kLoadVecs + 63) / 64; ++it) {
      const int idx = lane + it * 64;
      if (idx < G::kLoadVecs) {
        const int row = idx / PV;
        const int cv = idx - row * PV;
        const int gi = wrap_row(i0 - 4 + row);
        const int gj = wrap_col(j0 - HV * V + cv * V);
        glds16(src + (int64_t)gi * ld + gj, sU + it * 64 * V);
      }
    }
  }

  // per-environment parameters into scalar registers
  struct {
    T mu[4], mob[3];
  } p;
  const EnvParams<T>& ep = a.ep[tb];
#pragma unroll
  for (int k = 0; k < 4; ++k) p.mu[k] = uniform_f(ep.mu[k]);
#pragma unroll
  for (int k = 0; k < 3; ++k) p.mob[k] = uniform_f(ep.mob[k]);
  const T kap = uniform_f(ep.kappa);

  // ---- pointwise operands (PAIR_34: y on own cells + ring, acc on own cells)
  const int64_t pidx0 = base + (int64_t)(i0 + r0) * ld + (j0 + lx * V);
  Vec ybase[RPT], accp[RPT], yring[kRingTrips];
  if constexpr (PAIR == PAIR_34) {
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      accp[r] = Vec{};
      if constexpr (RAGGED) {
        const int gi = wrap_row(i0 + r0 + r), gj = wrap_col(j0 + lx * V);
        ybase[r] = *reinterpret_cast<const Vec*>(a.y + base + (int64_t)gi * ld + gj);
        if (cell_ok(r)) accp[r] = *reinterpret_cast<const Vec*>(a.acc + pidx0 + r * ld);
      } else {
        ybase[r] = *reinterpret_cast<const Vec*>(a.y + pidx0 + r * ld);
        accp[r] = *reinterpret_cast<const Vec*>(a.acc + pidx0 + r * ld);
      }
    }
#pragma unroll
    for (int q2 = 0; q2 < kRingTrips; ++q2) {
      if (has_ring[q2]) {
        const int gi = wrap_row(i0 + ring_r[q2]);
        const int gj = wrap_col(j0 + (ring_cv[q2] - HV) * V);
        yring[q2] = *reinterpret_cast<const Vec*>(a.y + base + (int64_t)gi * ld + gj);
      }
    }
  }
  vm_wait_all();  // DMA landed (and the pointwise operands)
  wave_order();

  auto mu_pass = [&](const int rm0, const int nrows) {
    const int nvec = nrows * PV;
#pragma unroll 1
    for (int idx = lane; idx < nvec; idx += 64) {
      const int rr = idx / PV;
      const int cv = idx - rr * PV;
      const int rm = rm0 + rr;
      const T* c_ = sU + (rm + 1) * P + cv * V;
      const Vec c = *reinterpret_cast<const Vec*>(c_);
      const Vec xp = *reinterpret_cast<const Vec*>(c_ + P);
      const Vec xm = *reinterpret_cast<const Vec*>(c_ - P);
      const T left = c_[-1], right = c_[V];
      Vec m;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const T ym = (e == 0) ? left : c[e - 1];
        const T yp = (e == V - 1) ? right : c[e + 1];
        m[e] = eval_mu<T, CL>(a.mu, p.mu, c[e]) - kap * lap_at<T>(c[e], xp[e], xm[e], yp, ym, a.rhx2, a.rhy2);
      }
      *reinterpret_cast<Vec*>(sMu + rm * P + cv * V) = m;
    }
  };

  auto k_at = [&](const int r, const int cv, Vec* centre) -> Vec {
    const T* mp = sMu + (r + 3) * P + cv * V;
    const T* up = sU + (r + 4) * P + cv * V;
    const Vec u_c = *reinterpret_cast<const Vec*>(up);
    if (centre) *centre = u_c;
    return flux_divergence<T, CL, Vec, V>(
        a.mob, p.mob, *reinterpret_cast<const Vec*>(mp - P), *reinterpret_cast<const Vec*>(mp),
        *reinterpret_cast<const Vec*>(mp + P), *reinterpret_cast<const Vec*>(up - P), u_c,
        *reinterpret_cast<const Vec*>(up + P), mp[-1], mp[V], up[-1], up[V], a.rhx, a.rhy);
  };

  auto march = [&](Vec* kout, Vec* centre) {
    const T* mp = sMu + (r0 + 2) * P + cvo * V;  // row r0 - 1
    const T* up = sU + (r0 + 3) * P + cvo * V;
    Vec m_lo = *reinterpret_cast<const Vec*>(mp);
    Vec d_lo = mob_vec<T, CL, Vec, V>(a.mob, p.mob, *reinterpret_cast<const Vec*>(up));
    mp += P;
    up += P;
    Vec m_c = *reinterpret_cast<const Vec*>(mp);
    Vec u_c = *reinterpret_cast<const Vec*>(up);
    Vec d_c = mob_vec<T, CL, Vec, V>(a.mob, p.mob, u_c);
    Vec fx_lo;
#pragma unroll
    for (int e = 0; e < V; ++e) fx_lo[e] = face_flux<T>(d_lo[e], d_c[e], m_lo[e], m_c[e], a.rhx);
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const Vec m_hi = *reinterpret_cast<const Vec*>(mp + P);
      const Vec u_hi = *reinterpret_cast<const Vec*>(up + P);
      const Vec d_hi = mob_vec<T, CL, Vec, V>(a.mob, p.mob, u_hi);
      const T ml = mp[-1], mr = mp[V];
      const T dl = eval_mob<T, CL>(a.mob, p.mob, up[-1]), dr = eval_mob<T, CL>(a.mob, p.mob, up[V]);
      const Vec dy = div_y<T, Vec, V>(m_c, d_c, ml, mr, dl, dr, a.rhy);
      Vec fx_hi, k;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        fx_hi[e] = face_flux<T>(d_c[e], d_hi[e], m_c[e], m_hi[e], a.rhx);
        k[e] = (fx_hi[e] - fx_lo[e]) * a.rhx + dy[e];
      }
      kout[r] = k;
      if (centre) centre[r] = u_c;
      m_c = m_hi;
      u_c = u_hi;
      d_c = d_hi;
      fx_lo = fx_hi;
      mp += P;
      up += P;
    }
  };

  // ---- P2: mu_A on tile + 3
  mu_pass(0, TX + 6);
  wave_order();

  // ---- P3: k_A on the own micro-tile and on the ring
  Vec w_own[RPT], yown[RPT], w_ring[kRingTrips];
  {
    Vec kA[RPT];
    march(kA, yown);
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      if constexpr (PAIR == PAIR_12) {
        w_own[r] = yown[r] + a.aA * kA[r];
        accp[r] = yown[r] + a.bA * kA[r];
      } else {
        w_own[r] = ybase[r] + a.aA * kA[r];
        accp[r] = accp[r] + a.bA * kA[r];
      }
    }
  }
#pragma unroll
  for (int q2 = 0; q2 < kRingTrips; ++q2) {
    if (has_ring[q2]) {
      Vec uc;
      const Vec kA = k_at(ring_r[q2], ring_cv[q2], &uc);
      if constexpr (PAIR == PAIR_12)
        w_ring[q2] = uc + a.aA * kA;
      else
        w_ring[q2] = yring[q2] + a.aA * kA;
    }
  }
  // every read of the stage-A input has returned before it is overwritten in place
  wave_lds_done();

  // ---- P4: w -> sU in place (tile + 2)
#pragma unroll
  for (int r = 0; r < RPT; ++r) *reinterpret_cast<Vec*>(sU + (r0 + r + 4) * P + cvo * V) = w_own[r];
#pragma unroll
  for (int q2 = 0; q2 < kRingTrips; ++q2)
    if (has_ring[q2]) *reinterpret_cast<Vec*>(sU + (ring_r[q2] + 4) * P + ring_cv[q2] * V) = w_ring[q2];
  wave_order();

  // ---- P5: mu_B on tile + 1
  mu_pass(2, TX + 2);
  wave_order();

  // ---- P6: k_B, stage updates, stores
  Vec kB[RPT];
  march(kB, nullptr);
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    if (!cell_ok(r)) continue;
    const int64_t idx = pidx0 + r * ld;
    if constexpr (PAIR == PAIR_12) {
      *reinterpret_cast<Vec*>(a.out + idx) = yown[r] + a.aB * kB[r];
      *reinterpret_cast<Vec*>(a.acc_out + idx) = accp[r] + a.bB * kB[r];
    } else {
      *reinterpret_cast<Vec*>(a.out + idx) = accp[r] + a.bB * kB[r];
    }
  }
}

template <typename T, int CL, int PAIR>
int launch_pair_wave_inst(pdeopt_ctx* ctx, const PairArgs<T>& s) {
  using G = WaveGeom<T>;
  const pdeopt_problem& p = ctx->prob;
  const int tiles_i = (p.nx + G::TX - 1) / G::TX;
  const int tiles_j = (p.ny + G::TY - 1) / G::TY;
  const int64_t ntiles64 = (int64_t)tiles_i * tiles_j * ctx->win_n;
  if (ntiles64 > 0x7fffffffLL) return fail(ctx, PDEOPT_EINVAL, "too many tiles");
  const int ntiles = (int)ntiles64;
  const int nblk = (ntiles + 3) / 4;
  const bool ragged = p.nx % G::TX != 0 || p.ny % G::TY != 0;
  const int remap = (nblk % 8 == 0) ? 1 : 0;
  const size_t lds = G::lds_bytes();
  if (ragged) {
    auto kern = stage_pair_wave_kernel<T, CL, PAIR, true>;
    if (lds > 48 * 1024)
      PDEOPT_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, ctx->stream, s, tiles_i, tiles_j, ntiles, nblk, remap);
  } else {
    auto kern = stage_pair_wave_kernel<T, CL, PAIR, false>;
    if (lds > 48 * 1024)
      PDEOPT_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, ctx->stream, s, tiles_i, tiles_j, ntiles, nblk, remap);
  }
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  return PDEOPT_OK;
}

}  // namespace pdeopt
