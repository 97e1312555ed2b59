// Software-pipelined variant of the fused Cahn-Hilliard stage-pair kernel (stencil_fused.hpp).
//
// Why: phase ablation of stage_pair_kernel (tools/ablate_pair.sh) shows its cost is nearly the SUM of
// its parts -- LDS/barrier skeleton 9.3, arithmetic 11.6, global loads+stores 8.6 of 25.2 ms per
// environment step -- i.e. a workgroup's tile load, its six barrier-separated phases and its stores
// run back to back and the 4-5 co-resident workgroups of a CU overlap them poorly.  Nothing is
// saturated (VALU ~50 %, HBM ~60 %): the kernel is bound by the latency of one tile's critical path.
//
// Here a workgroup is persistent over a contiguous run of tiles and the NEXT tile's input (tile + 4
// halo) is fetched by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write pass) into a second
// LDS buffer while the current tile is in phases P4-P6, so the tile load leaves the critical path:
//
//   prologue   DMA tile 0 -> sU[0]; vmcnt(0); barrier
//   tile t     P2 mu_A(sU[cur]) | P3 k_A -> regs | DMA tile t+1 -> sU[cur^1] | P4 w -> sU[cur]
//              P5 mu_B | P6 k_B, vmcnt(0), stores | barrier ; cur ^= 1
//
// Ordering rules used (cdna_hip_programming.md section 5, "Pipelining across barriers" / "Read a staged
// buffer one phase AFTER the wait that retires it"):
//   * while a DMA is in flight every barrier is a raw s_barrier behind "s_waitcnt lgkmcnt(0)" (inline
//     asm with a memory clobber): __syncthreads() would drain vmcnt(0) at once;
//   * every wave waits vmcnt(0) for its own DMA pieces BEFORE the tile's closing barrier, the next
//     tile reads the buffer after that barrier (RAW);
//   * a buffer is re-staged two barriers after its last ds_read (WAR);
//   * no ordinary global-load result is consumed while a DMA is in flight (PAIR_34's pointwise
//     operands are consumed in P3, the DMA is issued after P3's barrier), so hipcc's own waitcnt
//     bookkeeping never drains it early.
// Arithmetic, flux routine and tile geometry are those of stage_pair_kernel: results are bitwise
// identical to it (tests/test_gpu_parity.py::test_pipelined_pair_equals_classic).
#pragma once

#include "stencil_fused.hpp"

namespace pdeopt {

__device__ __forceinline__ void lds_barrier_raw() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ void vm_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// materialise a value here (the compiler may not sink the arithmetic producing it past this point).
// In a __device__ helper: register constraints inside a __global__ template body break the HOST-side
// instantiation of the kernel stub ("v" is not an x86 constraint; the stub is silently not emitted).
template <typename X>
__device__ __forceinline__ void pin_value(X& x) {
  asm volatile("" : "+v"(x));
}

// 16 bytes per lane, global -> LDS without a VGPR stop; lds_wave_base is wave-uniform, lane l lands at
// lds_wave_base + 16 l (in a __device__ helper for the same host-side reason: the LDS address-space cast)
__device__ __forceinline__ void glds16(const void* src, void* lds_wave_base) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#endif
}

template <typename T>
struct PipeGeom {
  static constexpr int V = VecOf<T>::V;
  static constexpr int HV = 4 / V;
  static constexpr int RPT = 2;
  static constexpr int NT = 256;
  static constexpr int TX = (NT / kLanesPerRow) * RPT;  // 16 rows
  static constexpr int PV = kLanesPerRow + 2 * HV;
  static constexpr int P = PV * V;
  static constexpr int TY = kLanesPerRow * V;
  static constexpr int kLoadVecs = (TX + 8) * PV;
  static constexpr int kLoadVecsPad = (kLoadVecs + 63) / 64 * 64;  // whole 1-KiB DMA pieces
  static constexpr int kSU = kLoadVecsPad * V;                       // elements per input buffer
  static constexpr int kSMu = (TX + 6) * P;
  // [pad V][sU0][pad V][sU1][pad V][sMu][pad V]
  static constexpr size_t lds_bytes() { return (size_t)(4 * V + 2 * kSU + kSMu) * sizeof(T); }
};

// P1 as LDS-DMA: this wave's 1-KiB pieces of the (TX+8) x PV input image of tile (b_, i_, j_).
// (A __device__ function, not a lambda of the kernel: the host pass must be able to instantiate the
// kernel's stub without resolving device-only calls.)
template <typename T, bool RAGGED>
__device__ __forceinline__ void issue_tile_dma(const T* __restrict__ in, const Geo& g, int b_, int i_, int j_, T* dst,
                                               int tid) {
  using G = PipeGeom<T>;
  constexpr int V = G::V, HV = G::HV, NT = G::NT, TX = G::TX, PV = G::PV, TY = G::TY;
  const T* __restrict__ src = in + (int64_t)b_ * g.bstride + g.off;
  const int64_t ld = g.ld;
  const int i0n = i_ * TX, j0n = j_ * TY;
  const int wave_base = tid & ~63;
#pragma unroll
  for (int it = 0; it < (G::kLoadVecs + NT - 1) / NT; ++it) {
    const int idx = tid + it * NT;
    if (idx < G::kLoadVecs) {
      const int row = idx / PV;
      const int cv = idx - row * PV;
      const int gi = g.periodic ? tile_wrap(i0n - 4 + row, g.nx, RAGGED) : i0n - 4 + row;
      const int gj = g.periodic ? tile_wrap(j0n - HV * V + cv * V, g.ny, RAGGED) : j0n - HV * V + cv * V;
      glds16(src + (int64_t)gi * ld + gj, dst + (it * NT + wave_base) * V);
    }
  }
}

template <typename T, int CL, int PAIR, bool RAGGED>
__global__ __launch_bounds__(256) void stage_pair_pipe_kernel(const PairArgs<T> a, const int tiles_i,
                                                              const int tiles_j, const int ntiles,
                                                              const int tiles_per_block, const int nchunks,
                                                              const int xcd_remap) {
  using Vec = typename VecOf<T>::type;
  using G = PipeGeom<T>;
  constexpr int V = G::V, HV = G::HV, RPT = G::RPT, NT = G::NT, TX = G::TX, PV = G::PV, P = G::P, TY = G::TY;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* const sU0 = reinterpret_cast<T*>(smem_raw) + V;
  T* const sU1 = sU0 + G::kSU + V;
  T* const sMu = sU1 + G::kSU + V;

  int chunk = blockIdx.x;
  if (xcd_remap) chunk = (chunk & 7) * (nchunks >> 3) + (chunk >> 3);
  int t = chunk * tiles_per_block;
  const int t_end = min(t + tiles_per_block, ntiles);
  if (t >= t_end) return;  // block-uniform

  const Geo& g = a.g;
  const int64_t ld = g.ld;
  const int tid = threadIdx.x;
  const int lx = tid & 31;
  const int ly = tid >> 5;
  const int r0 = ly * RPT;
  const int cvo = lx + HV;

  constexpr int kRingRowVecs = kLanesPerRow + 2;
  constexpr int kRingTop = 4 * kRingRowVecs;
  constexpr int kRing = kRingTop + 2 * TX;
  static_assert(kRing <= NT, "ring must fit one pass");
  int ring_r = 0, ring_cv = 0;
  const bool has_ring = tid < kRing;
  if (tid < kRingTop) {
    const int q = tid / kRingRowVecs;
    ring_r = (q < 2) ? (q - 2) : (TX + q - 2);
    ring_cv = HV - 1 + (tid - q * kRingRowVecs);
  } else if (has_ring) {
    const int t2 = tid - kRingTop;
    ring_r = t2 >> 1;
    ring_cv = (t2 & 1) ? (HV + kLanesPerRow) : (HV - 1);
  }

  constexpr bool ragged = RAGGED;
  auto wrap_row = [&](int gi) { return g.periodic ? tile_wrap(gi, g.nx, ragged) : gi; };
  auto wrap_col = [&](int gj) { return g.periodic ? tile_wrap(gj, g.ny, ragged) : gj; };

  // tile coordinates of t, advanced incrementally; (nb, ni, nj) run one tile ahead for the DMA
  int tj = uniform_i(t % tiles_j);
  int ti = uniform_i((t / tiles_j) % tiles_i);
  int tb = uniform_i(t / (tiles_j * tiles_i));
  auto advance = [&](int& b_, int& i_, int& j_) {
    if (++j_ == tiles_j) {
      j_ = 0;
      if (++i_ == tiles_i) {
        i_ = 0;
        ++b_;
      }
    }
  };

  issue_tile_dma<T, RAGGED>(a.in, g, tb, ti, tj, sU0, tid);
  vm_wait_all();
  lds_barrier_raw();

  int nb = tb, ni = ti, nj = tj;
  advance(nb, ni, nj);
  int cur = 0;
  struct {
    T mu[4], mob[3];
  } p;
  T kap = T(0);
  int p_env = -1;

#pragma unroll 1
  for (; t < t_end; ++t) {
    T* const sU = cur ? sU1 : sU0;
    T* const sUnext = cur ? sU0 : sU1;
    const int i0 = ti * TX;
    const int j0 = tj * TY;
    const int64_t base = (int64_t)tb * g.bstride + g.off;
    // per-environment parameters live in scalar registers and are re-read only when the run crosses
    // into another environment: a plain global load consumed here would make hipcc wait vmcnt(0),
    // which also waits for the previous tile's stores (one in-order counter on gfx9)
    if (tb != p_env) {
      const EnvParams<T>& ep = a.ep[tb];
#pragma unroll
      for (int k = 0; k < 4; ++k) p.mu[k] = uniform_f(ep.mu[k]);
#pragma unroll
      for (int k = 0; k < 3; ++k) p.mob[k] = uniform_f(ep.mob[k]);
      kap = uniform_f(ep.kappa);
      p_env = tb;
    }
    const bool col_ok = !RAGGED || (j0 + lx * V) < g.ny;
    auto cell_ok = [&](int r) { return !RAGGED || (col_ok && (i0 + r0 + r) < g.nx); };

    // ---- pointwise operands (PAIR_34: y on own cells + ring, acc on own cells), consumed in P3
    const int64_t pidx0 = base + (int64_t)(i0 + r0) * ld + (j0 + lx * V);
    Vec ybase[RPT], accp[RPT], yring;
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
      if (has_ring) {
        const int gi = wrap_row(i0 + ring_r);
        const int gj = wrap_col(j0 + (ring_cv - HV) * V);
        yring = *reinterpret_cast<const Vec*>(a.y + base + (int64_t)gi * ld + gj);
      }
    }

    auto mu_pass = [&](const int rm0, const int nrows) {
      const int nvec = nrows * PV;
#pragma unroll 1
      for (int idx = tid; idx < nvec; idx += NT) {
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
    lds_barrier_raw();

    // ---- P3: k_A on the own micro-tile and on one ring vector
    Vec w_own[RPT], yown[RPT], w_ring;
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
    if (has_ring) {
      Vec uc;
      const Vec kA = k_at(ring_r, ring_cv, &uc);
      if constexpr (PAIR == PAIR_12)
        w_ring = uc + a.aA * kA;
      else
        w_ring = yring + a.aA * kA;
      pin_value(w_ring);
    }
    // pin every value derived from the pointwise loads HERE: hipcc would otherwise sink the arithmetic
    // (and its vmcnt(0)) below the next barriers, into the span where the DMA is in flight
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      pin_value(w_own[r]);
      pin_value(accp[r]);
    }
    lds_barrier_raw();

    // ---- next tile's input: in flight during P4-P6 (its buffer was last read two barriers ago)
    if (t + 1 < t_end) issue_tile_dma<T, RAGGED>(a.in, g, nb, ni, nj, sUnext, tid);

    // ---- P4: w -> sU in place (tile + 2)
#pragma unroll
    for (int r = 0; r < RPT; ++r) *reinterpret_cast<Vec*>(sU + (r0 + r + 4) * P + cvo * V) = w_own[r];
    if (has_ring) *reinterpret_cast<Vec*>(sU + (ring_r + 4) * P + ring_cv * V) = w_ring;
    lds_barrier_raw();

    // ---- P5: mu_B on tile + 1
    mu_pass(2, TX + 2);
    lds_barrier_raw();

    // ---- P6: k_B, stage updates, stores
    Vec kB[RPT];
    march(kB, nullptr);
    Vec o0[RPT], o1[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      if constexpr (PAIR == PAIR_12) {
        o0[r] = yown[r] + a.aB * kB[r];
        o1[r] = accp[r] + a.bB * kB[r];
      } else {
        o0[r] = accp[r] + a.bB * kB[r];
      }
    }
    // this wave's pieces of the next tile have landed (ordered for the other waves by the barrier below)
    vm_wait_all();
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      if (!cell_ok(r)) continue;
      const int64_t idx = pidx0 + r * ld;
      *reinterpret_cast<Vec*>(a.out + idx) = o0[r];
      if constexpr (PAIR == PAIR_12) *reinterpret_cast<Vec*>(a.acc_out + idx) = o1[r];
    }
    lds_barrier_raw();

    cur ^= 1;
    tb = nb;
    ti = ni;
    tj = nj;
    advance(nb, ni, nj);
  }
}

template <typename T, int CL, int PAIR>
int launch_pair_pipe_inst(pdeopt_ctx* ctx, const PairArgs<T>& s) {
  using G = PipeGeom<T>;
  const pdeopt_problem& p = ctx->prob;
  const int tiles_i = (p.nx + G::TX - 1) / G::TX;
  const int tiles_j = (p.ny + G::TY - 1) / G::TY;
  const int64_t ntiles64 = (int64_t)tiles_i * tiles_j * ctx->win_n;
  if (ntiles64 > 0x7fffffffLL) return fail(ctx, PDEOPT_EINVAL, "too many tiles");
  const int ntiles = (int)ntiles64;
  // 4 workgroups per CU are co-resident (38.6 KB LDS each): size the runs so that one wave of
  // persistent workgroups covers the launch
  const int slots = ctx->num_cus * 4;
  const int tiles_per_block = std::max(1, (ntiles + slots - 1) / slots);
  const int nchunks = (ntiles + tiles_per_block - 1) / tiles_per_block;
  const bool ragged = p.nx % G::TX != 0 || p.ny % G::TY != 0;
  const int remap = (nchunks % 8 == 0) ? 1 : 0;
  if (ragged)
    hipLaunchKernelGGL((stage_pair_pipe_kernel<T, CL, PAIR, true>), dim3(nchunks), dim3(256), G::lds_bytes(), ctx->stream,
                       s, tiles_i, tiles_j, ntiles, tiles_per_block, nchunks, remap);
  else
    hipLaunchKernelGGL((stage_pair_pipe_kernel<T, CL, PAIR, false>), dim3(nchunks), dim3(256), G::lds_bytes(), ctx->stream,
                       s, tiles_i, tiles_j, ntiles, tiles_per_block, nchunks, remap);
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  return PDEOPT_OK;
}

}  // namespace pdeopt
