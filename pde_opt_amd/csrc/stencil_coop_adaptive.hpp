// Adaptive Tsit5 + PID controller inside ONE launch for grids beyond one compute unit: SEVERAL workgroups (= compute
// units) per environment, one exchange + one environment-wide barrier per trial step (round 4; SURVEY section 8 row f1,
// f3, a15).
//
// What the reference runs through diffrax.diffeqsolve(..., Tsit5(), PIDController(rtol, atol)) and where:
//   notebooks/smooth_boundary.ipynb:228,397   CahnHilliard2DSmoothedBoundary 100^2, 117 890 and 352 104 steps
//                                             (equation: cahn_hilliard.py:204-289, shapes.py:67-77)
//   notebooks/run_advection_diffusion.ipynb:84 advection-diffusion 64^2, 8 950 steps
//   tests/test_solvers.py:81,263              Allen-Cahn / Cahn-Hilliard 256 x 1, 32^2 ... (the single-workgroup
//                                             kernel of stencil_small_adaptive.hpp takes those)
// Host-driven (pdeopt_tsit5_trial / commit) a trial step is 8 launches + a reduction + a device->host read: 66-88 us
// whatever the grid.  The single-workgroup kernel stops at 2048 vectors (one CU's registers), and one CU's VALU needs
// ~30 us for the seven smoothed-boundary right-hand sides of a 100^2 grid anyway.
//
// Here an environment is cut into px x py sub-tiles, one workgroup each, and a trial step needs NO exchange between its
// seven stages: a workgroup holds its tile + H halo cells (H = 6 x the stencil radius of one right-hand side: 12 for the
// Cahn-Hilliard forms, 6 for Allen-Cahn / advection-diffusion) and evaluates stage s on the tile + r (7 - s) ring --
// the communication-avoiding form of the decomposed RK4 driver (comm.hip), here inside one CU's LDS:
//
//     k1 (FSAL) on T + 6r                  from the exchange of the previous accepted step
//     w2 = y + h a21 k1 on T + 6r
//     k_s = f(w_s) on T + r (7 - s), w_{s+1} = y + h sum_j a_{s+1,j} k_j there        s = 2 .. 6
//     k7 = f(y1) on T, scaled error norm of the tile; y1 and k7 of the tile -> global exchange buffers (speculative)
//     -- ONE barrier over the environment's workgroups (partial norms, fixed summation order: every workgroup holds the
//        same double and runs the same controller arithmetic: stencil_small_adaptive.hpp) --
//     accept: dense output of the tile at the save times passed; halo ring of y and k1 <- the neighbours' tiles
//
// Everything of a step lives in LDS (y, k1..k6, two stage-input buffers, mu / inner, the static fields psi,
// |grad psi|/psi, mask or the face velocities): 9 - 13 arrays of (tile + 2H)^2; the host picks the tile grid that fits
// 160 KB.  Redundant halo work (x2 at 25^2 tiles with H = 12) buys the absence of 6 cross-CU barriers per step (~1.5 us
// each against ~0.5 us of arithmetic per stage and CU).  Cells in overlapping regions are computed by several
// workgroups from the same inputs with the same code: same bits, so the copies never diverge.
//
// Cross-workgroup traffic goes through global memory with agent-scope release / acquire around the barrier (the
// workgroups of one environment are mapped to ONE XCD -- block index mod 8 -- wherever they fit its 32 compute units, so
// the data stays in that XCD's L2; small fp64 tiles may need two).
// Exit conditions every wave reaches: t >= t1, max_steps, a stalled step, and an abort flag that any workgroup raises
// when it has waited > 2 s at a barrier (a partner that never arrives): no wave can spin forever.
#pragma once

#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "stencil_small_adaptive.hpp"

namespace pdeopt {

template <typename T>
struct CoopArgs {
  T* y;  // [nenv][nx * ny] dense periodic states of this launch's environments (read at the start, written at the end)
  int nx, ny, px, py, nenv;
  int64_t bstride;
  T rhx, rhy, rhx2, rhy2;  // 1 / hx, 1 / hy, 1 / hx^2, 1 / hy^2
  const EnvParams<T>* ep;
  ClosureSpec mu, mob, fe;
  const T* s0;  // static fields [nx][ny]: smoothed boundary: psi, |grad psi| / psi, mask; advection-diffusion: vx, vy faces
  const T* s1;
  const T* s2;
  int64_t sstride;  // elements between environments of the static fields (0: shared)
  // time-dependent scalars of the smoothed-boundary forms: tmode 0: constants tw[3] = {cos theta on the mask, off it, flux};
  // 1: theta(t), flux(t) polynomials in t, evaluated at every stage time in the kernel
  int tmode;
  double theta[4], flux[4], tw[3];
  double t0, t1, dt0;
  PidConsts pid;
  int64_t max_steps;
  int n_save;
  const double* save_ts;
  T* save_out;  // [n_save][batch][nx * ny], pre-offset to this launch's first environment
  int64_t save_stride;
  pdeopt_tsit5_stats* stats;  // [nenv]
  T* xy[2];                   // exchange buffers [nenv][nx * ny]: candidate state / its slope, double-buffered
  T* xk[2];
  double* part;        // [2][nenv][px * py][2] partial error sums: slots of the per-step barrier (two tagged words each)
  unsigned* bar;       // [nenv][2]: arrivals, generation
  unsigned* abort_flag;
  int xs, wpx;         // XCDs an environment's workgroups are spread over (1, 2, 4, 8) and workgroups per XCD
  int rows, pitch;     // LDS array geometry: (largest tile + 2 H) rows x pitch
  int red_off;         // byte offset of the reduction / control scratch in LDS
  // fixed-step mode (MODE 1 of the kernel: n_sub substeps of explicit Euler / classical RK4 with step dt from t0)
  int fixed_rk4;
  int64_t n_sub;
  double dt;
  unsigned* tags;      // [nenv][px * py] exchange rounds a workgroup has published
};

// the fixed closure forms of stencil_sbm_tiled.hpp (FAST) or the run-time walk (closure_generic)
template <typename T, bool FAST>
__device__ __forceinline__ T coop_mu(const ClosureSpec& s, const T* __restrict__ cf, T c) {
  if constexpr (!FAST) return closure_generic<T>(s, cf, c);
  T r = ((cf[3] * c + cf[2]) * c + cf[1]) * c + cf[0];
  if (s.flags & PDEOPT_CL_LOGIT_PRIOR) r += t_logit<T>(c);
  return r;
}
template <typename T, bool FAST>
__device__ __forceinline__ T coop_fe(const ClosureSpec& s, const T* __restrict__ cf, T c) {
  if constexpr (!FAST) return closure_generic<T>(s, cf, c);
  T r = ((cf[3] * c + cf[2]) * c + cf[1]) * c + cf[0];
  if (s.flags & PDEOPT_CL_MIX_ENTROPY) r += c * t_log<T>(c) + (T(1) - c) * t_log<T>(T(1) - c);
  return r;
}
template <typename T, bool FAST>
__device__ __forceinline__ T coop_mob(const ClosureSpec& s, const T* __restrict__ cf, T c) {
  if constexpr (!FAST) return closure_generic<T>(s, cf, c);
  return (cf[2] * c + cf[1]) * c + cf[0];
}
inline bool coop_fast_closures(const pdeopt_problem& p, bool with_fe) {
  const bool mu = p.mu.kind == PDEOPT_CL_POLY && p.mu.n <= 4 && (p.mu.flags & ~PDEOPT_CL_LOGIT_PRIOR) == 0;
  const bool mob = p.mob.kind == PDEOPT_CL_POLY && p.mob.n <= 3 && p.mob.flags == 0;
  const bool fe = !with_fe || (p.fe.kind == PDEOPT_CL_POLY && p.fe.n <= 4 && (p.fe.flags & ~PDEOPT_CL_MIX_ENTROPY) == 0);
  return mu && mob && fe;
}

__device__ __forceinline__ int wrap1(int i, int n) {  // |offset| <= n: one conditional add / subtract
  if (i < 0) i += n;
  if (i >= n) i -= n;
  return i;
}

constexpr unsigned long long kCoopTimeoutTicks = 200000000ull;  // 2 s of s_memrealtime (100 MHz)

// Every store this wave has issued is acknowledged at its scope once this returns.  __syncthreads() alone does NOT give
// that: its workgroup-scope release needs no vmcnt wait when the workgroup's waves share one CU (the compiler emits
// `s_waitcnt lgkmcnt(0); s_barrier`), so a slot word stored by one wave behind the barrier could become visible to a
// partner before another wave's tile data -- seen once in ~10 runs of the fixed-step mode as a stale ring (an error of
// one round's increment in a few cells), never in the adaptive step, whose reduction sits between data and slot stores.
__device__ __forceinline__ void coop_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Counter barrier over the nwg workgroups of one environment (the solve's prologue; the step loop's barrier is the slot
// exchange in the kernel).  false: the solve was aborted (a partner did not arrive).  one_xcd = no fences: the exchange
// data is written and read with agent-scope accesses (xstore / xload below), which need no cache maintenance -- see the
// kernel; otherwise (-DPDEOPT_COOP_XCD_FENCES=1) agent-scope release / acquire fences write the L2 back and invalidate it.
__device__ __forceinline__ bool coop_env_barrier(unsigned* bar, unsigned* abort_flag, int nwg, unsigned* gen, bool one_xcd) {
  coop_stores_done();
  __syncthreads();  // every wave's stores have been acknowledged
  if (threadIdx.x == 0) {
    if (!one_xcd) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned g = *gen;
    if (__hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nwg - 1)) {
      __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&bar[1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const unsigned long long t_in = __builtin_amdgcn_s_memrealtime();
      while (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) {
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        if (__builtin_amdgcn_s_memrealtime() - t_in > kCoopTimeoutTicks) {
          __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
  }
  __syncthreads();
  if (!one_xcd) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // every wave: the partners' writes, not this CU's stale lines
  *gen += 1u;
  return __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
}
// exchange data: agent-scope accesses (sc1: past the per-CU vector cache, to the L2 the partners share)
template <typename T>
__device__ __forceinline__ T xload(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ void xstore(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// cells per trip of the stage loops (region_u).  Measured on the 100^2 smoothed-boundary solve (profiles/r04_adaptive_variants_ab.txt):
// 1, 2 and 4 cells a trip, 512 or 768 threads, hoisted or fresh loop geometry all land within 16.4 - 18.4 us per trial step,
// inside the run-to-run spread of one build (~1 us): the stage loops are bound neither by VALU issue (27 % of the step's
// cycles, SQ_INSTS_VALU) nor by the LDS pipe (33 %, SQ_LDS_IDX_ACTIVE) but by the ~14 workgroup barriers of a step and the
// dependent LDS -> VALU chains between them.  One cell a trip is the smallest code.
#ifndef PDEOPT_COOP_UNROLL
#define PDEOPT_COOP_UNROLL 1
#endif
// The cell loops' geometry (a thread's first row / column / LDS offset, the strides) is the same in every trial step, so
// the compiler hoists all of it -- for each of the ~13 loops of a step -- out of the step loop and keeps it in registers
// across the whole step: the smoothed-boundary forms then sit at exactly 256 registers, and anything added spills.  The
// empty asm statements make the thread index and the tile width opaque at a loop's entry: ~10 instructions per loop
// instead of ~5 live registers per loop.  Applied to the loops OUTSIDE the stages (load, stage-2 input, stage 7, dense
// output, accept, store: `region`); the five stage loops (`region_u`) keep their hoisted geometry unless
// -DPDEOPT_COOP_FRESH (measured: neither faster nor slower beyond the run-to-run spread).
#define PDEOPT_COOP_OPAQUE_GEOMETRY                         \
  int tid_ = tid, tw_ = __builtin_amdgcn_readfirstlane(tw); \
  asm volatile("" : "+v"(tid_));                            \
  asm volatile("" : "+s"(tw_))
#ifndef PDEOPT_COOP_RING_REGS
#define PDEOPT_COOP_RING_REGS 6
#endif
#ifndef PDEOPT_COOP_THREADS
#define PDEOPT_COOP_THREADS 512  // 1024 threads cap a thread at 128 registers: the step loop's uniform doubles then spill (48-140 B fp32)
#endif
// the fixed-step mode holds no slopes and no controller: 51 - 95 registers in fp32 -> 1024 threads, 4 waves per SIMD
#ifndef PDEOPT_COOP_FIXED_NEIGHBOURS
#define PDEOPT_COOP_FIXED_NEIGHBOURS 1  // the fixed-step mode's exchange waits for the 8 neighbouring workgroups only
#endif
#ifndef PDEOPT_COOP_FIXED_THREADS_F32
#define PDEOPT_COOP_FIXED_THREADS_F32 1024
#endif
// (fp64: the fixed-closure periodic forms and advection-diffusion need 60 - 84 registers -> 1024 threads as well; the
// smoothed-boundary forms (128 - 130) and the run-time closure walk stay at 512)
template <typename T, int MODE, int EQ, bool FAST>
constexpr int coop_threads() {
  constexpr bool sbm = EQ == PDEOPT_EQ_ALLEN_CAHN_SBM || EQ == PDEOPT_EQ_CAHN_HILLIARD_SBM;
  return MODE == 1 && (sizeof(T) == 4 || (FAST && !sbm)) ? PDEOPT_COOP_FIXED_THREADS_F32 : PDEOPT_COOP_THREADS;
}
// MODE 0: the adaptive Tsit5 solve.  MODE 1: n_sub substeps of explicit Euler / classical RK4 on the same tiling (one
// environment on several compute units; see the fixed-step loop below).
template <typename T, int EQ, bool FAST, int MODE = 0>
__global__ __launch_bounds__((coop_threads<T, MODE, EQ, FAST>())) void tsit5_coop_kernel(const CoopArgs<T> a) {
  constexpr int NT = coop_threads<T, MODE, EQ, FAST>();
  constexpr bool kTwoPass = EQ == PDEOPT_EQ_CAHN_HILLIARD || EQ == PDEOPT_EQ_CAHN_HILLIARD_SBM;  // mu / inner array
  constexpr bool kSBM = EQ == PDEOPT_EQ_ALLEN_CAHN_SBM || EQ == PDEOPT_EQ_CAHN_HILLIARD_SBM;
  constexpr bool kAD = EQ == PDEOPT_EQ_ADVECTION_DIFFUSION;
  constexpr int R = kTwoPass ? 2 : 1;  // stencil radius of one right-hand side
  constexpr int H = MODE == 1 ? 8 : 6 * R;  // fixed-step: 8 = one RK4 substep of the radius-2 forms, two of the radius-1 forms
  constexpr int NK = MODE == 1 ? 1 : 6;      // slope arrays (fixed-step: the Runge-Kutta accumulator)

  // block -> (environment of this launch, tile).  Blocks are dealt round-robin over the 8 XCDs (block index mod 8): an
  // environment's workgroups sit on a.xs neighbouring XCDs (1 wherever they fit one XCD's compute units: the exchange
  // then stays in that XCD's L2), a.wpx of them per XCD
  const int nwg = a.px * a.py;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int row = slot / a.wpx;
  const int be = row * (8 / a.xs) + xcd / a.xs;
  const int w = (slot - row * a.wpx) * a.xs + (xcd % a.xs);
  if (be >= a.nenv || w >= nwg) return;  // the whole workgroup
  const int wi = w / a.py, wj = w - wi * a.py;
  const int nx = a.nx, ny = a.ny;
  const int i0 = (int)((int64_t)wi * nx / a.px), i1 = (int)((int64_t)(wi + 1) * nx / a.px);
  const int j0 = (int)((int64_t)wj * ny / a.py), j1 = (int)((int64_t)(wj + 1) * ny / a.py);
  const int th = i1 - i0, tw = j1 - j0;
  const int tid = threadIdx.x;
  const int P = a.pitch;
  const int FS = a.rows * P;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* const sY = reinterpret_cast<T*>(smem_raw);
  T* const sK = sY + FS;          // k1 .. k6: sK + j FS
  T* sW = sK + NK * FS;           // current stage input
  T* sV = sW + FS;                // next stage input / k7
  T* const sM = sV + FS;          // mu / inner (two-pass forms)
  T* const sS = sM + (kTwoPass ? FS : 0);  // static fields: sS + j FS
  double* const red = reinterpret_cast<double*>(smem_raw + a.red_off);  // [0..15] wave partials, [19..22] controller / abort broadcast, [24..] partners' sums, [280..] the step's time terms

  // this environment's closure coefficients.  Fixed forms: copied into registers once -- read through the pointer inside
  // the cell loops the compiler re-loaded the coefficient arrays from memory in every trip (two global_load_dwordx4 per
  // cell).  Run-time closure walk: through the pointer (its loops index the arrays dynamically).
  struct CoefRegs {
    T mu[4], mob[4], fe[4];
  };
  struct CoefPtrs {
    const T *mu, *mob, *fe;
  };
  std::conditional_t<FAST, CoefRegs, CoefPtrs> p;
  {
    const EnvParams<T>& pe = a.ep[be];
    if constexpr (FAST) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        p.mu[i] = pe.mu[i];
        p.mob[i] = pe.mob[i];
        p.fe[i] = pe.fe[i];
      }
    } else {
      p.mu = pe.mu;
      p.mob = pe.mob;
      p.fe = pe.fe;
    }
  }
  const T kap = a.ep[be].kappa;
  T* const yg = a.y + (int64_t)be * a.bstride;
  const int64_t xoff = (int64_t)be * nx * ny;
  unsigned* const bar = a.bar + 2 * be;
  unsigned gen = 0;
  // Every cross-workgroup access of the step loop is an agent-scope access (xload / xstore, the slot words): its scope bits
  // take it past the caches that are not coherent at that scope -- the per-CU vector cache, and the per-XCD L2 when the
  // partner sits on another XCD -- so an environment spread over several XCDs needs no release / acquire fences (an L2
  // write-back + invalidate per barrier: 6.5 us on one XCD, 15 us across two, measured in this kernel's first version)
  // either.  Ordering: a workgroup's data stores are acknowledged (every wave: s_waitcnt vmcnt(0), then the s_barrier) before its
  // slot word is stored; a reader issues its data loads after it has seen the slot words.  -DPDEOPT_COOP_XCD_FENCES=1
  // keeps the fenced counter barrier for environments on several XCDs (384^2 CH: 61 us per trial step against 45).
#ifndef PDEOPT_COOP_XCD_FENCES
#define PDEOPT_COOP_XCD_FENCES 0
#endif
  const bool one_xcd = a.xs == 1 || !PDEOPT_COOP_XCD_FENCES;

  // cells of the region T + e, local coordinates (r, c) relative to the tile origin: f(LDS offset, r, c)
  auto region = [&](const int e, auto f) {
    // (row, column) of a thread's cells are carried from one trip to the next: idx += NT is rr += NT / wd, cc += NT % wd
    // with one carry -- a division per trip was ~12 of the ~110 instructions a cell costs
    PDEOPT_COOP_OPAQUE_GEOMETRY;
    const int wd = tw_ + 2 * e, total = (th + 2 * e) * wd;
    const float inv = 1.0f / (float)wd;
    int rr = (int)(((float)tid_ + 0.5f) * inv);
    int cc = tid_ - rr * wd;
    if (cc < 0) { cc += wd; --rr; }
    if (cc >= wd) { cc -= wd; ++rr; }
    const int dr = (int)(((float)NT + 0.5f) * inv), dc = NT - dr * wd;  // NT = dr wd + dc, 0 <= dc < wd (wd <= NT + ...: see below)
    int o = (rr - e + H) * P + (cc - e + H);
    const int dO = dr * P + dc;
    for (int idx = tid_; idx < total; idx += NT) {
      f(o, rr - e, cc - e);
      rr += dr;
      cc += dc;
      o += dO;
      if (cc >= wd) {
        cc -= wd;
        ++rr;
        o += P - wd;
      }
    }
  };
  // The same cells, PDEOPT_COOP_UNROLL of them per trip: compute(o, r, c) -> value for each, THEN commit(o, r, c, value)
  // for each.  A workgroup has 2 waves per SIMD: the LDS latency and the quarter-rate log / rcp / sqrt chains of one cell
  // are not hidden by other waves, and inside `region` the compiler may not move the next cell's loads above this
  // cell's LDS store (same address space).  With the loads of U cells issued before any store their chains overlap.
  // Trips where every lane of the wave has all U cells (a scalar condition) run unrolled, the remainder one cell a trip.
  auto region_n = [&](auto u_c, auto opaque_c, const int e, auto compute, auto commit) {
    constexpr int U = decltype(u_c)::value;
    int tid_ = tid, tw_ = tw;
    if constexpr (decltype(opaque_c)::value) {  // (PDEOPT_COOP_OPAQUE_GEOMETRY)
      tw_ = __builtin_amdgcn_readfirstlane(tw);
      asm volatile("" : "+v"(tid_));
      asm volatile("" : "+s"(tw_));
    }
    const int wd = tw_ + 2 * e, total = (th + 2 * e) * wd;
    const float inv = 1.0f / (float)wd;
    int rr = (int)(((float)tid_ + 0.5f) * inv);
    int cc = tid_ - rr * wd;
    if (cc < 0) { cc += wd; --rr; }
    if (cc >= wd) { cc -= wd; ++rr; }
    const int dr = (int)(((float)NT + 0.5f) * inv), dc = NT - dr * wd;
    int o = (rr - e + H) * P + (cc - e + H);
    const int dO = dr * P + dc;
    auto advance = [&]() {
      rr += dr;
      cc += dc;
      o += dO;
      if (cc >= wd) {
        cc -= wd;
        ++rr;
        o += P - wd;
      }
    };
    int idx = tid_;
    if constexpr (U > 1) {
      int wave_first = __builtin_amdgcn_readfirstlane(idx);  // lane 0 holds the wave's smallest index
      while (wave_first + 63 + (U - 1) * NT < total) {
        int ou[U], ru[U], cu[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          ou[u] = o; ru[u] = rr - e; cu[u] = cc - e;
          advance();
        }
        decltype(compute(0, 0, 0)) v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = compute(ou[u], ru[u], cu[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) commit(ou[u], ru[u], cu[u], v[u]);
        idx += U * NT;
        wave_first += U * NT;
      }
    }
    for (; idx < total; idx += NT) {
      const auto v = compute(o, rr - e, cc - e);
      commit(o, rr - e, cc - e, v);
      advance();
    }
  };
  // the five stage loops: PDEOPT_COOP_UNROLL cells a trip (fp64 smoothed-boundary cells -- three software logarithms each
  // -- fill the registers on their own: one a trip), geometry hoisted unless -DPDEOPT_COOP_FRESH
  auto region_u = [&](const int e, auto compute, auto commit) {
    constexpr int U = (sizeof(T) == 8 && kSBM) ? 1 : PDEOPT_COOP_UNROLL;
#ifdef PDEOPT_COOP_FRESH
    region_n(std::integral_constant<int, U>{}, std::true_type{}, e, compute, commit);
#else
    region_n(std::integral_constant<int, U>{}, std::false_type{}, e, compute, commit);
#endif
  };
  // the short pointwise passes (stage-2 input, accept): a trip is two LDS reads, one multiply-add and a write, all
  // latency -- four cells' loads issued before the first store instead of four dependent round trips (a trip cost
  // ~500 ticks: the stage-2 input pass alone was 1.9 k of a step's 39 k)
  auto region_4 = [&](const int e, auto compute, auto commit) {
    region_n(std::integral_constant<int, 4>{}, std::true_type{}, e, compute, commit);
  };
  auto gidx = [&](int r, int c) -> int64_t { return (int64_t)wrap1(i0 + r, nx) * ny + wrap1(j0 + c, ny); };

  // ---- the state and the static fields on T + H
  {
    const T* const s0 = a.s0 ? a.s0 + be * a.sstride : nullptr;
    const T* const s1 = a.s1 ? a.s1 + be * a.sstride : nullptr;
    const T* const s2 = a.s2 ? a.s2 + be * a.sstride : nullptr;
    region(H, [&](int o, int r, int c) {
      const int64_t g = gidx(r, c);
      sY[o] = yg[g];
      if constexpr (kSBM || kAD) {
        sS[o] = s0[g];
        sS[FS + o] = s1[g];
      }
      if constexpr (kSBM) sS[2 * FS + o] = s2[g];
    });
  }
  const T sqk = kSBM ? T(sqrt((double)kap)) : T(0);

  // Time terms of the right-hand-side evaluations of a trial step -- cos theta(ts) on / off the mask, flux(ts) at the six
  // stage times ts = t + c_i h -- in a table red[kTT + 3 i ..]: six LANES fill it side by side at the start of the step
  // (one lane evaluating the double-precision cosines stage by stage while every other wave waited cost ~2 us per step
  // of the notebook's theta(t) solve), the stages read their entry after the barrier that follows.
  constexpr int kTT = 24 + 256;
  auto put_time_terms = [&](const int slot, const double ts) {  // (called by the lane that owns the slot)
    if constexpr (kSBM) {
      double* const tt = red + kTT + 3 * slot;
      if (a.tmode == 0) {
        tt[0] = a.tw[0]; tt[1] = a.tw[1]; tt[2] = a.tw[2];
      } else {
        const double thv = ((a.theta[3] * ts + a.theta[2]) * ts + a.theta[1]) * ts + a.theta[0];
        tt[0] = cos(thv);
        // Cahn-Hilliard: cos(pi - theta) off the mask (cahn_hilliard.py:271-272); Allen-Cahn: nothing there (allen_cahn.py:150)
        tt[1] = EQ == PDEOPT_EQ_ALLEN_CAHN_SBM ? 0.0 : cos(3.14159265358979323846 - thv);
        tt[2] = ((a.flux[3] * ts + a.flux[2]) * ts + a.flux[1]) * ts + a.flux[0];
      }
    }
  };

  // ---- right-hand side on a region.  pass1 (two-pass forms): mu / inner of `src` on T + e + 1 -> sM;
  //      kcell: k at one cell of T + e from `src` (and sM).  Expressions: rhs_generic_point / sbm_tiled_kernel, term for term.
  auto inner_at = [&](const T* src, int o, T twa, T twb) -> T {  // smoothed boundary: cahn_hilliard.py:262-279, allen_cahn.py:141-155
    const T* q = sS;  // psi
    const T c = src[o], xp = src[o + P], xm = src[o - P], yp = src[o + 1], ym = src[o - 1];
    const T pc = q[o], pxp = q[o + P], pxm = q[o - P], pyp = q[o + 1], pym = q[o - 1];
    const T m = sS[2 * FS + o];
    const T wl = sqk * sS[FS + o] * (twa * m + twb * (T(1) - m));
    const T dx_hi = (T(0.5) * (pc + pxp)) * ((xp - c) * a.rhx), dx_lo = (T(0.5) * (pxm + pc)) * ((c - xm) * a.rhx);
    const T dy_hi = (T(0.5) * (pc + pyp)) * ((yp - c) * a.rhy), dy_lo = (T(0.5) * (pym + pc)) * ((c - ym) * a.rhy);
    const T lap = (dx_hi - dx_lo) * a.rhx + (dy_hi - dy_lo) * a.rhy;
    T rr = coop_mu<T, FAST>(a.mu, p.mu, c) - (kap * t_rcp<T>(pc)) * lap;
    rr -= wl * t_sqrt<T>(T(2) * coop_fe<T, FAST>(a.fe, p.fe, c));
    return rr;
  };
  auto pass1 = [&](const T* src, int e, T twa, T twb) {
    if constexpr (kTwoPass) {
      region_u(e + 1, [&](int o, int, int) -> T {
        if constexpr (EQ == PDEOPT_EQ_CAHN_HILLIARD) {
          const T c = src[o];
          return coop_mu<T, FAST>(a.mu, p.mu, c) - kap * lap_at<T>(c, src[o + P], src[o - P], src[o + 1], src[o - 1], a.rhx2, a.rhy2);
        } else {
          return inner_at(src, o, twa, twb);
        }
      }, [&](int o, int, int, T v) { sM[o] = v; });
      __syncthreads();
    }
  };
  auto kcell = [&](const T* src, int o, T twa, T twb, T tsrc) -> T {
    if constexpr (EQ == PDEOPT_EQ_ALLEN_CAHN) {
      const T c = src[o];
      const T m = coop_mu<T, FAST>(a.mu, p.mu, c) - kap * lap_at<T>(c, src[o + P], src[o - P], src[o + 1], src[o - 1], a.rhx2, a.rhy2);
      return -coop_mob<T, FAST>(a.mob, p.mob, c) * m;
    } else if constexpr (EQ == PDEOPT_EQ_ALLEN_CAHN_SBM) {
      return -coop_mob<T, FAST>(a.mob, p.mob, src[o]) * inner_at(src, o, twa, twb);
    } else if constexpr (kAD) {
      // -div(v u) + D lap u with face-averaged u and face velocities (stencil_generic.hpp)
      const T* vx = sS;
      const T* vy = sS + FS;
      const T u00 = src[o], uxp = src[o + P], uxm = src[o - P], uyp = src[o + 1], uym = src[o - 1];
      const T fx0 = vx[o] * (T(0.5) * (u00 + uxp)), fxm = vx[o - P] * (T(0.5) * (uxm + u00));
      const T fy0 = vy[o] * (T(0.5) * (u00 + uyp)), fym = vy[o - 1] * (T(0.5) * (uym + u00));
      return -((fx0 - fxm) * a.rhx + (fy0 - fym) * a.rhy) + kap * lap_at<T>(u00, uxp, uxm, uyp, uym, a.rhx2, a.rhy2);
    } else {
      const T m00 = sM[o], mxp = sM[o + P], mxm = sM[o - P], myp = sM[o + 1], mym = sM[o - 1];
      const T d00 = coop_mob<T, FAST>(a.mob, p.mob, src[o]);
      const T dxp = coop_mob<T, FAST>(a.mob, p.mob, src[o + P]), dxm = coop_mob<T, FAST>(a.mob, p.mob, src[o - P]);
      const T dyp = coop_mob<T, FAST>(a.mob, p.mob, src[o + 1]), dym = coop_mob<T, FAST>(a.mob, p.mob, src[o - 1]);
      if constexpr (EQ == PDEOPT_EQ_CAHN_HILLIARD) {
        const T fx0 = (T(0.5) * (d00 + dxp)) * ((mxp - m00) * a.rhx), fxm = (T(0.5) * (dxm + d00)) * ((m00 - mxm) * a.rhx);
        const T fy0 = (T(0.5) * (d00 + dyp)) * ((myp - m00) * a.rhy), fym = (T(0.5) * (dym + d00)) * ((m00 - mym) * a.rhy);
        return (fx0 - fxm) * a.rhx + (fy0 - fym) * a.rhy;
      } else {  // smoothed-boundary Cahn-Hilliard: cahn_hilliard.py:280-289
        const T* q = sS;
        const T p00 = q[o], pxp = q[o + P], pxm = q[o - P], pyp = q[o + 1], pym = q[o - 1];
        const T fx0 = (T(0.5) * (p00 + pxp)) * (T(0.5) * (d00 + dxp)) * ((mxp - m00) * a.rhx);
        const T fxm = (T(0.5) * (pxm + p00)) * (T(0.5) * (dxm + d00)) * ((m00 - mxm) * a.rhx);
        const T fy0 = (T(0.5) * (p00 + pyp)) * (T(0.5) * (d00 + dyp)) * ((myp - m00) * a.rhy);
        const T fym = (T(0.5) * (pym + p00)) * (T(0.5) * (dym + d00)) * ((m00 - mym) * a.rhy);
        return ((fx0 - fxm) * a.rhx + (fy0 - fym) * a.rhy) * t_rcp<T>(p00) + sS[FS + o] * tsrc;
      }
    }
  };
  auto time_terms = [&](const int slot, T* twa, T* twb, T* tsrc) {
    *twa = kSBM ? T(red[kTT + 3 * slot]) : T(0);
    *twb = kSBM ? T(red[kTT + 3 * slot + 1]) : T(0);
    *tsrc = kSBM ? T(red[kTT + 3 * slot + 2]) : T(0);
  };
  // the halo ring (T + H minus T) of y and k1 <- the exchange buffers: ring cells only (2 H full rows above and below,
  // H columns left and right of every tile row), both fields' loads of a cell issued together
  auto load_rings = [&](const T* srcy, const T* srck) {
    const int wd = tw + 2 * H, top = 2 * H * wd, total = top + 2 * H * th;
    const float inv = 1.0f / (float)wd;
    for (int idx = tid; idx < total; idx += NT) {
      int r, c;
      if (idx < top) {
        int rr = (int)(((float)idx + 0.5f) * inv);
        int cc = idx - rr * wd;
        if (cc < 0) { cc += wd; --rr; }
        if (cc >= wd) { cc -= wd; ++rr; }
        r = rr < H ? rr - H : th + rr - H;
        c = cc - H;
      } else {
        const int i2 = idx - top;
        const int rr = i2 / (2 * H), s2 = i2 - rr * (2 * H);
        r = rr;
        c = s2 < H ? s2 - H : tw + s2 - H;
      }
      const int64_t g = gidx(r, c);
      const T vy = srcy ? xload(&srcy[g]) : T(0), vk = xload(&srck[g]);
      const int o = (r + H) * P + (c + H);
      if (srcy) sY[o] = vy;
      sK[o] = vk;
    }
  };

  // The same ring through REGISTERS on an accepted step: load_rings' loop waits for a trip's loads before it stores them,
  // so a thread's 3 trips are 3 global-memory latencies in a row (~4 k ticks of the step's ~39 k); ring_fetch issues
  // all of a thread's loads, ring_commit stores them.  (-DPDEOPT_COOP_RING_EARLY fetches before the controller -- the
  // candidate y1 / k7 are in the exchange buffers once the step's barrier has passed, whatever the decision -- to hide
  // the loads under the controller's pow: measured slower, see below.)
  // ring cells per thread held in registers (larger rings: load_rings after the decision); fp64 -- two registers a value, smaller tiles -- half
  constexpr int kRingRegs = sizeof(T) == 8 ? PDEOPT_COOP_RING_REGS / 2 : PDEOPT_COOP_RING_REGS;
  const int ring_total = 2 * H * (tw + 2 * H) + 2 * H * th;
  const bool ring_in_regs = kRingRegs > 0 && ring_total <= kRingRegs * NT;
  T ring_y[kRingRegs > 0 ? kRingRegs : 1], ring_k[kRingRegs > 0 ? kRingRegs : 1];
  auto ring_cell = [&](int idx, int* r, int* c) {
    const int wd = tw + 2 * H, top = 2 * H * wd;
    if (idx < top) {
      int rr = (int)(((float)idx + 0.5f) * (1.0f / (float)wd));
      int cc = idx - rr * wd;
      if (cc < 0) { cc += wd; --rr; }
      if (cc >= wd) { cc -= wd; ++rr; }
      *r = rr < H ? rr - H : th + rr - H;
      *c = cc - H;
    } else {
      const int i2 = idx - top;
      const int rr = i2 / (2 * H), s2 = i2 - rr * (2 * H);
      *r = rr;
      *c = s2 < H ? s2 - H : tw + s2 - H;
    }
  };
  // (the cells' coordinates are the same in every step: the opaque copy of the thread index keeps the compiler from
  // carrying all of them in registers across the whole step loop, where the stage loops then spill)
  auto ring_fetch = [&](const T* srcy, const T* srck) {
    int tid_ = tid;
    asm volatile("" : "+v"(tid_));
#pragma unroll
    for (int j = 0; j < kRingRegs; ++j) {
      const int idx = tid_ + j * NT;
      if (idx < ring_total) {
        int r, c;
        ring_cell(idx, &r, &c);
        const int64_t g = gidx(r, c);
        ring_y[j] = xload(&srcy[g]);
        ring_k[j] = xload(&srck[g]);
      }
    }
  };
  auto ring_commit = [&]() {
    int tid_ = tid;
    asm volatile("" : "+v"(tid_));
#pragma unroll
    for (int j = 0; j < kRingRegs; ++j) {
      const int idx = tid_ + j * NT;
      if (idx < ring_total) {
        int r, c;
        ring_cell(idx, &r, &c);
        const int o = (r + H) * P + (c + H);
        sY[o] = ring_y[j];
        sK[o] = ring_k[j];
      }
    }
  };

  if constexpr (MODE == 1) {
    // ------------------------------------------------------------------------------------------------ fixed step
    // n_sub substeps of explicit Euler / classical RK4 for ONE environment on several compute units (VERDICT r3 Weak #6:
    // a single 96^2 - 256^2 environment is 2 dependent launches of ~5.8 us per RK4 substep on the tiled path whatever
    // its size).  A round = as many substeps as the 8-cell halo pays for (a substep consumes stages x radius of it: RK4
    // Cahn-Hilliard 1 per round, RK4 Allen-Cahn / advection-diffusion 2, Euler 4 / 8), every stage on the region the later
    // stages still need, nothing exchanged inside a round; then the tile of y goes to the exchange buffer of the round's
    // parity, the workgroup publishes the round number, waits for its partners' and reloads its halo ring.  Update
    // formulas in the association of the stage-pair / whole-step kernels (stencil_small.hpp):
    //   w2 = y + dt/2 k1, acc = y + dt/6 k1;  w3 = y + dt/2 k2, acc += dt/3 k2;  w4 = y + dt k3, acc += dt/3 k3;  y' = acc + dt/6 k4
    const bool rk4 = a.fixed_rk4 != 0;
    const int stages = rk4 ? 4 : 1;
    const int S = H / (stages * R);  // substeps per round
    const T dtT = T(a.dt), h2 = T(a.dt / 2), h3 = T(a.dt / 3), h6 = T(a.dt / 6);
    T* y = sY;
    T* acc = sK;
    unsigned* const tags = a.tags + (size_t)be * nwg;
    const int ring_n = 2 * H * (tw + 2 * H) + 2 * H * th;
    constexpr int kRing = sizeof(T) == 8 ? 3 : 6;
    __syncthreads();  // y and the static fields are in place
    // one right-hand side on T + e_out from `src`, its update applied per cell
    auto rhs_update = [&](const T* src, const int e_out, const int slot, auto update) {
      T twa, twb, tsrc;
      time_terms(slot, &twa, &twb, &tsrc);
      pass1(src, e_out, twa, twb);
      region_u(e_out, [&](int o, int, int) -> T { return kcell(src, o, twa, twb, tsrc); }, update);
      __syncthreads();
    };
    unsigned round = 0;
    for (int64_t done = 0; done < a.n_sub; ++round) {
      const int m = (int)((a.n_sub - done) < (int64_t)S ? (a.n_sub - done) : (int64_t)S);
      if constexpr (kSBM) {  // theta(t), flux(t) at the round's stage times: one lane each
        if (tid < m * stages) {
          const int sub = tid / stages, st = tid - sub * stages;
          const double t_sub = a.t0 + (double)(done + sub) * a.dt;
          put_time_terms(tid, !rk4 || st == 0 ? t_sub : (st == 3 ? t_sub + a.dt : t_sub + 0.5 * a.dt));
        }
        __syncthreads();
      }
      int e = H;
      for (int sub = 0; sub < m; ++sub) {
        if (rk4) {
          rhs_update(y, e - R, 4 * sub, [&](int o, int, int, T k) {
            const T y0 = y[o];
            sW[o] = y0 + h2 * k;
            acc[o] = y0 + h6 * k;
          });
          rhs_update(sW, e - 2 * R, 4 * sub + 1, [&](int o, int, int, T k) {
            sV[o] = y[o] + h2 * k;
            acc[o] = acc[o] + h3 * k;
          });
          rhs_update(sV, e - 3 * R, 4 * sub + 2, [&](int o, int, int, T k) {
            sW[o] = y[o] + dtT * k;
            acc[o] = acc[o] + h3 * k;
          });
          rhs_update(sW, e - 4 * R, 4 * sub + 3, [&](int o, int, int, T k) { y[o] = acc[o] + h6 * k; });
          e -= 4 * R;
        } else {
          T* const nxt = y == sY ? sV : sY;
          rhs_update(y, e - R, sub, [&](int o, int, int, T k) { nxt[o] = y[o] + dtT * k; });
          y = nxt;
          e -= R;
        }
      }
      done += m;
      if (done >= a.n_sub) break;
      // ---- exchange: tile -> buffer of this round's parity; publish; wait for every partner; ring <- their tiles.
      // A workgroup reaches its next write of this buffer (two rounds on) only after every partner has published the round
      // between, i.e. has finished reading this one.
      T* const xb = a.xy[round & 1u] + xoff;
      region(0, [&](int o, int r, int c) { xstore(&xb[(int64_t)(i0 + r) * ny + (j0 + c)], y[o]); });
      coop_stores_done();
      __syncthreads();  // every wave's exchange stores have been acknowledged
      const unsigned tag = round + 1u;
      if (tid == 0) xstore(&tags[w], tag);
      if (tid < 64) {
        const unsigned long long t_in = __builtin_amdgcn_s_memrealtime();
        bool gave_up = false;
        // Whom to wait for: the workgroups whose tiles this one's ring is read from, and who read this one's tile -- the 8
        // neighbours when every tile is at least H cells wide and high (lanes 0 .. 7, one each; workgroups further away may
        // be a round ahead or behind: no environment-wide rendezvous), else everybody.
        const bool near_only = PDEOPT_COOP_FIXED_NEIGHBOURS && nx / a.px >= H && ny / a.py >= H;
        const int n_wait = near_only ? 8 : nwg;
        for (int q = tid; q < n_wait; q += 64) {
          int i = q;
          if (near_only) {
            const int di = q < 3 ? -1 : (q < 5 ? 0 : 1);                       // rows of the 3 x 3 block without its centre
            const int dj = q < 3 ? q - 1 : (q < 5 ? (q == 3 ? -1 : 1) : q - 6);
            i = wrap1(wi + di, a.px) * a.py + wrap1(wj + dj, a.py);
          }
          for (;;) {
            if (xload(&tags[i]) >= tag) break;
            if (xload(a.abort_flag) != 0u) {
              gave_up = true;
              break;
            }
            if (__builtin_amdgcn_s_memrealtime() - t_in > kCoopTimeoutTicks) {
              xstore(a.abort_flag, 1u);
              gave_up = true;
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        const bool any = __any(gave_up);
        if (tid == 0) red[22] = any ? 1.0 : 0.0;
      }
      __syncthreads();
      if (red[22] != 0.0) return;  // (the host reports the abort; the state is then garbage)
      {
        T ring[kRing];
        const int trips = (ring_n + NT - 1) / NT;
        for (int j0r = 0; j0r < trips; j0r += kRing) {  // (one pass for tiles up to ~40^2)
#pragma unroll
          for (int j = 0; j < kRing; ++j) {
            const int idx = tid + (j0r + j) * NT;
            if (idx < ring_n) {
              int r, c;
              ring_cell(idx, &r, &c);
              ring[j] = xload(&xb[gidx(r, c)]);
            }
          }
#pragma unroll
          for (int j = 0; j < kRing; ++j) {
            const int idx = tid + (j0r + j) * NT;
            if (idx < ring_n) {
              int r, c;
              ring_cell(idx, &r, &c);
              y[(r + H) * P + (c + H)] = ring[j];
            }
          }
        }
      }
      __syncthreads();
    }
    region(0, [&](int o, int r, int c) { yg[(int64_t)(i0 + r) * ny + (j0 + c)] = y[o]; });
    return;
  } else {
  // ---- k1 = f(t0, y0) on the tile; its halo through the exchange
  if (tid == 0) put_time_terms(0, a.t0);
  __syncthreads();  // sY, the static fields and the time terms are in place
  {
    T twa, twb, tsrc;
    time_terms(0, &twa, &twb, &tsrc);
    pass1(sY, 0, twa, twb);
    T* const xk0 = a.xk[0] + xoff;
    region(0, [&](int o, int r, int c) {
      const T k = kcell(sY, o, twa, twb, tsrc);
      sK[o] = k;
      xstore(&xk0[(int64_t)(i0 + r) * ny + (j0 + c)], k);
    });
  }
  if (!coop_env_barrier(bar, a.abort_flag, nwg, &gen, one_xcd)) return;
  load_rings(nullptr, a.xk[0] + xoff);
  int cur = 0;  // exchange buffer parity of the accepted state
  __syncthreads();

#ifdef PDEOPT_COOP_PROF  // TIMING ONLY (tools/mkvariant.sh): shader-clock ticks per phase, printed by the first workgroup at exit
  unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt = __builtin_amdgcn_s_memtime();
#define PDEOPT_COOP_TICK(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); prof[i] += n_ - pt; pt = n_; } while (0)
#else
#define PDEOPT_COOP_TICK(i) do { } while (0)
#endif
  const T rtol = T(a.pid.rtol), atol = T(a.pid.atol);
  const double inv_cells = 1.0 / ((double)nx * (double)ny);
  double t = a.t0, dt = a.dt0, prev_inv = 1.0, prev_prev_inv = 1.0;
  int64_t accepted = 0, rejected = 0;
  int qi = 0, status = PDEOPT_TSIT5_DONE;
  double next_tq = uniform_f(a.n_save > 0 ? a.save_ts[0] : __builtin_inf());
  unsigned step = 0;
  while (t < a.t1) {
    if (accepted + rejected >= a.max_steps) {
      status = PDEOPT_TSIT5_MAX_STEPS;
      break;
    }
    // (the controller's state is the same in every lane: kept in scalar registers -- readfirstlane -- so that the step
    // loop's doubles do not occupy vector registers across the stages)
    const double h = uniform_f(fmin(dt, a.t1 - t));
    if (!(h > 0.0) || t + h == t) {
      status = PDEOPT_TSIT5_STALLED;
      break;
    }
    // stage 2 input on T + 6 R (where k1 lives)
    {
      const T c0 = T(h * kTsA[0][0]);
      region_4(H, [&](int o, int, int) -> T { return sY[o] + c0 * sK[o]; }, [&](int o, int, int, T v) { sW[o] = v; });
    }
    if (tid < 6) put_time_terms(tid, t + kTsC[tid] * h);  // (the previous step's last read lies behind its barriers)
    __syncthreads();
    PDEOPT_COOP_TICK(0);
    // stages 2 .. 6 (slope index s = 1 .. 5): k on T + R (6 - s), with it the next stage's input there.  One instantiation
    // per stage: the weights stay in registers (a run-time stage index would index them dynamically -> scratch)
    auto stage = [&](auto s_c) {
      constexpr int s = decltype(s_c)::value;
      constexpr int e = R * (6 - s);
      T cs[s + 1];
#pragma unroll
      for (int i = 0; i <= s; ++i) cs[i] = T(h * kTsA[s][i]);
      T twa, twb, tsrc;
      time_terms(s - 1, &twa, &twb, &tsrc);
      pass1(sW, e, twa, twb);
      T* const ks = sK + s * FS;
      struct KW {
        T k, w;
      };
      region_u(e, [&](int o, int, int) -> KW {
        const T k = kcell(sW, o, twa, twb, tsrc);
        T r = sY[o];
#pragma unroll
        for (int i = 0; i < s; ++i) r = r + cs[i] * sK[i * FS + o];
        return KW{k, r + cs[s] * k};
      }, [&](int o, int, int, const KW& v) {
        ks[o] = v.k;
        sV[o] = v.w;
      });
      __syncthreads();
      T* const tmp = sW;
      sW = sV;
      sV = tmp;
    };
    stage(std::integral_constant<int, 1>{});
    stage(std::integral_constant<int, 2>{});
    stage(std::integral_constant<int, 3>{});
    stage(std::integral_constant<int, 4>{});
    stage(std::integral_constant<int, 5>{});
    PDEOPT_COOP_TICK(1);
    // stage 7 on the tile: k7 = f(y1), the scaled error, the speculative exchange write
    double part = 0.0;
    {
      T ce[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) ce[i] = T(h * kTsE[i]);
      T twa, twb, tsrc;
      time_terms(5, &twa, &twb, &tsrc);
      pass1(sW, 0, twa, twb);
      T* const xyn = a.xy[cur ^ 1] + xoff;
      T* const xkn = a.xk[cur ^ 1] + xoff;
      region(0, [&](int o, int r, int c) {
        const T k7 = kcell(sW, o, twa, twb, tsrc);
        sV[o] = k7;
        T ev = T(0);
#pragma unroll
        for (int i = 0; i < 6; ++i) ev = ev + ce[i] * sK[i * FS + o];
        ev = ev + ce[6] * k7;
        const T y0 = sY[o], y1 = sW[o];
        const T sc = atol + rtol * fmax(fabs(y0), fabs(y1));
        const double qv = (double)(ev / sc);
        part += qv * qv;
        const int64_t g = (int64_t)(i0 + r) * ny + (j0 + c);
        xstore(&xyn[g], y1);
        xstore(&xkn[g], k7);
      });
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) part += __shfl_down(part, sft, 64);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    coop_stores_done();  // (this wave's y1 / k7 stores; the wait overlaps the other waves' reductions)
    __syncthreads();     // red[] is complete and every wave's exchange stores have been acknowledged
    PDEOPT_COOP_TICK(2);
    // The per-step barrier IS the exchange of the partial norms: a workgroup posts its sum as two 64-bit words, each
    // carrying 32 bits of the double and the step's tag, into its slot of the step's parity; wave 0 of every workgroup
    // polls all slots (one lane each) until both tags match.  One store + one poll round per workgroup instead of an
    // atomic counter round trip, a generation word and a second read of the sums (one XCD: ~1 us less per step).  Slots
    // alternate by parity: a workgroup reaches its next write of a slot only after every partner has posted the step
    // between, i.e. has finished reading this one.  (-DPDEOPT_COOP_XCD_FENCES=1: across XCDs the fenced counter barrier instead.)
    // The controller: ONE lane per workgroup (the same arithmetic on the same partial sums in the same order in every
    // workgroup: one decision for the environment), broadcast through LDS -- all waves running the double-precision
    // pow / sqrt / divisions redundantly cost 6 us per step (14 k ticks), a lone lane ~1
    auto controller = [&]() {
      double sum = 0.0;
      for (int i = 0; i < nwg; ++i) sum += red[24 + i];  // one order in every workgroup: the same double everywhere
      const double err = sqrt(sum * inv_cells);  // diffrax rms_norm
      const bool keep_ = err < 1.0;              // a NaN norm rejects
      const double inv_ = (err > 0.0 && err < __builtin_inf()) ? 1.0 / err : (err == 0.0 ? __builtin_inf() : 0.0);
      double f_ = pid_term(inv_, a.pid.k1, a.pid) * pid_term(prev_inv, a.pid.k2, a.pid) * pid_term(prev_prev_inv, a.pid.k3, a.pid);
      f_ = fmin(a.pid.factormax, fmax(a.pid.factormin, a.pid.safety * f_));
      if (!keep_) f_ = fmin(1.0, f_);
      red[19] = keep_ ? 1.0 : 0.0;
      red[20] = inv_;
      red[21] = f_;
    };
    double* const parts = a.part + ((size_t)(step & 1u) * a.nenv + be) * nwg * 2;  // two words per workgroup
    const unsigned tag = step + 1u;
    bool aborted = false;
    if (one_xcd) {
      if (tid == 0) {
        double sum = 0.0;
        for (int i = 0; i < NT / 64; ++i) sum += red[i];
        const unsigned long long bits = (unsigned long long)__double_as_longlong(sum);
        unsigned long long* const slot = reinterpret_cast<unsigned long long*>(parts) + 2 * w;
        xstore(&slot[0], (bits & 0xffffffff00000000ull) | tag);
        xstore(&slot[1], (bits << 32) | tag);
      }
      if (tid < 64) {
        const unsigned long long t_in = __builtin_amdgcn_s_memrealtime();
        bool gave_up = false;
        for (int i = tid; i < nwg; i += 64) {
          const unsigned long long* const slot = reinterpret_cast<const unsigned long long*>(parts) + 2 * i;
          unsigned long long w0, w1;
          for (;;) {
            w0 = xload(&slot[0]);
            w1 = xload(&slot[1]);
            if ((unsigned)w0 == tag && (unsigned)w1 == tag) break;
            if (xload(a.abort_flag) != 0u) {
              gave_up = true;
              break;
            }
            if (__builtin_amdgcn_s_memrealtime() - t_in > kCoopTimeoutTicks) {
              xstore(a.abort_flag, 1u);
              gave_up = true;
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
          red[24 + i] = __longlong_as_double((long long)((w0 & 0xffffffff00000000ull) | (w1 >> 32)));
        }
        // The abort flag is only ever looked at while a slot is missing: a step whose slots all arrived needs no
        // global-memory round trip to learn that nobody gave up (an abort raised elsewhere after this workgroup passed
        // is met at the next step's poll, where the partner's slot stays missing)
        const bool any = __any(gave_up);
        // the polling wave's lane 0 runs the controller straight away: the sums it needs were staged by its own wave
        // (LDS operations of one wave complete in order), so the step needs one workgroup barrier here, not two
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (tid == 0) {
          red[22] = any ? 1.0 : 0.0;
          if (!any) controller();
        }
      }
      __syncthreads();
      aborted = red[22] != 0.0;
    } else {
      if (tid == 0) {
        double sum = 0.0;
        for (int i = 0; i < NT / 64; ++i) sum += red[i];
        xstore(&parts[w], sum);
      }
      if (!coop_env_barrier(bar, a.abort_flag, nwg, &gen, one_xcd)) return;
      if (tid < 64) {  // wave 0: the partners' partial sums, one lane each (the loads overlap), staged in LDS
        for (int i = tid; i < nwg; i += 64) red[24 + i] = xload(&parts[i]);
      }
    }
    if (aborted) return;
    PDEOPT_COOP_TICK(3);
    ++step;
#ifdef PDEOPT_COOP_RING_EARLY  // (measured slower: the fetched values live across the dense-output code and spill)
    if (ring_in_regs) ring_fetch(a.xy[cur ^ 1] + xoff, a.xk[cur ^ 1] + xoff);  // in flight under the controller
#endif
    if (!one_xcd) {  // (the fenced path: sums staged behind its own barrier)
      __syncthreads();
      if (tid == 0) controller();
      __syncthreads();
    }
    const bool keep = red[19] != 0.0;
    const double inv = red[20];
    const double f = red[21];
    PDEOPT_COOP_TICK(4);
    if (keep) {
      ++accepted;
      const double t_new = t + h;
      while (qi < a.n_save) {
        const double tq = next_tq;  // (a.save_ts[qi], loaded when qi last moved: not a global-memory round trip per step)
        if (!(tq <= t_new + 1e-14 * fmax(1.0, fabs(t_new)))) break;
        double bw[7];
        tsit5_dense_weights(fmin(1.0, fmax(0.0, (tq - t) / h)), bw);
        T cw[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) cw[i] = T(h * bw[i]);
        T* const out = a.save_out + (int64_t)qi * a.save_stride + xoff;
        region(0, [&](int o, int r, int c) {
          T rv = sY[o];
#pragma unroll
          for (int i = 0; i < 6; ++i) rv = rv + cw[i] * sK[i * FS + o];
          rv = rv + cw[6] * sV[o];
          out[(int64_t)(i0 + r) * ny + (j0 + c)] = rv;
        });
        ++qi;
        next_tq = uniform_f(qi < a.n_save ? a.save_ts[qi] : __builtin_inf());
      }
      // y <- y1, k1 <- k7 (FSAL): the tile from LDS, the halo ring from the neighbours' tiles
      struct YK {
        T y, k;
      };
      region_4(0, [&](int o, int, int) -> YK { return YK{sW[o], sV[o]}; }, [&](int o, int, int, const YK& v) {
        sY[o] = v.y;
        sK[o] = v.k;
      });
      cur ^= 1;
      if (ring_in_regs) {
#ifndef PDEOPT_COOP_RING_EARLY
        ring_fetch(a.xy[cur] + xoff, a.xk[cur] + xoff);  // every load of the thread's ring cells in flight at once
#endif
        ring_commit();
      } else {
        load_rings(a.xy[cur] + xoff, a.xk[cur] + xoff);
      }
      t = uniform_f(t_new < a.t1 - 1e-14 * fmax(1.0, fabs(a.t1)) ? t_new : a.t1);
      prev_prev_inv = prev_inv;
      prev_inv = uniform_f(inv);
    } else {
      ++rejected;
    }
    dt = uniform_f(fmin(a.pid.dtmax, fmax(a.pid.dtmin, h * f)));
    __syncthreads();  // sY / sK complete before the next step's stage-2 input reads them
    PDEOPT_COOP_TICK(5);
  }
#ifdef PDEOPT_COOP_PROF
  if (blockIdx.x == 0 && tid == 0)
    printf("coop prof (ticks/step over %lld steps): w2 %llu | stages2-6 %llu | stage7+reduce %llu | env barrier %llu | controller %llu | accept+ring %llu\n",
           (long long)(accepted + rejected), prof[0] / (accepted + rejected), prof[1] / (accepted + rejected), prof[2] / (accepted + rejected),
           prof[3] / (accepted + rejected), prof[4] / (accepted + rejected), prof[5] / (accepted + rejected));
#endif
  region(0, [&](int o, int r, int c) { yg[(int64_t)(i0 + r) * ny + (j0 + c)] = sY[o]; });
  if (w == 0 && tid == 0) {
    pdeopt_tsit5_stats st;
    st.t = t;
    st.dt = dt;
    st.accepted = accepted;
    st.rejected = rejected;
    st.status = status;
    st.saved = qi;
    a.stats[be] = st;
  }
  }  // MODE 0
}

// --------------------------------------------------------------------------------------------- host

constexpr size_t kCoopLdsMax = 160 * 1024;

// launches whose workgroups wait for each other: one at a time per process (coop_tsit5_solve)
inline std::mutex& coop_launch_mutex() {
  static std::mutex m;
  return m;
}

struct CoopPlan {
  int px = 0, py = 0, rows = 0, pitch = 0, nfields = 0, halo = 0;
  size_t lds = 0;
  int red_off = 0;
};

inline int coop_radius(int equation) {
  return (equation == PDEOPT_EQ_CAHN_HILLIARD || equation == PDEOPT_EQ_CAHN_HILLIARD_SBM) ? 2 : 1;
}

// The tile grid.  Among the px x py splits whose arrays fit the LDS, the one with the least arithmetic per workgroup (the
// sum over the stages of the largest tile's region sizes) -- first among those with at most one XCD's worth of
// workgroups (the exchange then stays in one L2 and the barrier needs no cache maintenance: 2 us against 15 us per step,
// profiles/r04_coop_adaptive.txt), only then among larger grids (fp64 with many static fields: small tiles).
// Measured on the 100^2 smoothed-boundary solve: 4 x 4 tiles 23 us per trial step, 5 x 5 19.5, 7 x 7 (two XCDs) 39.
// environments of one launch for a plan: every workgroup of a launch must be resident at once
struct CoopLaunchShape {
  int xs, wpx, envs_per_launch;
};
inline bool coop_launch_shape(int nwg, int num_cus, CoopLaunchShape* o) {
  const int cus_per_xcd = std::max(1, num_cus / 8);
  int xs = 1;
  while (xs < 8 && (nwg + xs - 1) / xs > cus_per_xcd) xs *= 2;
  const int wpx = (nwg + xs - 1) / xs;
  if (wpx > cus_per_xcd) return false;
  o->xs = xs;
  o->wpx = wpx;
  o->envs_per_launch = (cus_per_xcd / wpx) * (8 / xs);
  return true;
}

// fixed: 0 = the adaptive solve (H = 6 R, six slope arrays), 1 / 2 = explicit Euler / RK4 substeps (H = 8, one accumulator)
template <typename T>
bool coop_plan(const pdeopt_problem& p, int num_cus, CoopPlan* out, int fixed = 0) {
  const int eq = p.equation;
  const int R = coop_radius(eq), H = fixed ? 8 : 6 * R;
  const bool sbm = eq == PDEOPT_EQ_ALLEN_CAHN_SBM || eq == PDEOPT_EQ_CAHN_HILLIARD_SBM;
  const int nstatic = sbm ? 3 : (eq == PDEOPT_EQ_ADVECTION_DIFFUSION ? 2 : 0);
  const int nf = 1 + (fixed ? 1 : 6) + 2 + (R == 2 ? 1 : 0) + nstatic;
  if (p.nx < H || p.ny < H) return false;
  auto fits = [&](int px, int py, CoopPlan* pl) {
    const int tx = (p.nx + px - 1) / px, ty = (p.ny + py - 1) / py;
    const int rows = tx + 2 * H;
    int pitch = ty + 2 * H;
    if (pitch % 2 == 0) ++pitch;  // odd pitch: vertically adjacent cells on different banks
    const size_t field = (size_t)rows * pitch * sizeof(T);
    const size_t arrays = (nf * field + 15) / 16 * 16;
    const size_t lds = arrays + (24 + 256 + 24) * sizeof(double);  // wave partials, controller broadcast, partial-sum staging, the step's time-term table
    if (lds > kCoopLdsMax) return false;
    pl->px = px; pl->py = py; pl->rows = rows; pl->pitch = pitch; pl->nfields = nf; pl->halo = H; pl->lds = lds; pl->red_off = (int)arrays;
    return true;
  };
  auto work = [&](int px, int py) {
    const int tx = (p.nx + px - 1) / px, ty = (p.ny + py - 1) / py;
    int64_t wk = 0;
    if (fixed) {  // one round: the right-hand sides of 8 / R halo cells' worth of stages, on their shrinking regions
      for (int e = H - R; e >= 0; e -= R) wk += (int64_t)(tx + 2 * e + (R == 2 ? 1 : 0)) * (ty + 2 * e + (R == 2 ? 1 : 0));
      // (tools/coop_fixed_tile_sweep.py, 100 substeps: 128^2 6 x 8 tiles 0.744 ms, 8 x 8 0.703, 8 x 10 0.767; 96^2 6 x 6 0.605, 8 x 8 0.673;
      //  192^2 8 x 8 ... 12 x 12 0.84-0.87: flat)
      return wk + 200 * (int64_t)px * py / 8;
    }
    for (int s = 1; s <= 6; ++s) wk += (int64_t)(tx + 2 * R * (6 - s) + (R == 2 ? 1 : 0)) * (ty + 2 * R * (6 - s) + (R == 2 ? 1 : 0));
    return wk + 150 * (int64_t)px * py / 8;  // + a little for every partner the barrier waits for
  };
  int forced = 0;
  if (const char* ev = getenv("PDEOPT_COOP_TILE")) {  // tuning knob (tools/adaptive_coop_bench.py): tile edge in cells
    const int v = atoi(ev);
    if (v >= 4 && v <= 128) forced = v;
  }
  if (forced) return fits(std::max(1, (p.nx + forced - 1) / forced), std::max(1, (p.ny + forced - 1) / forced), out) && out->px * out->py <= num_cus;
  const int per_xcd = std::max(1, num_cus / 8);
  // The split with the least modelled TIME for the whole batch: launches x work per step, where a launch holds as many
  // environments as its workgroups leave room for (every workgroup of a launch is resident at once).  One environment: the
  // least work per step wherever it lands -- the per-step exchange needs no cache maintenance across XCDs (see the kernel;
  // the notebook's 100^2 solve: 5 x 6 tiles 16.3 us per trial step, 7 x 7 14.6, 8 x 8 14.4, 10 x 10 14.3); a large batch:
  // an environment on one XCD's compute units, 8 environments per launch.  At most 7 of the 8 XCDs' compute units per
  // environment: a launch that needs EVERY compute unit free waits for any other kernel.
  int64_t best = -1;
  for (const int cap : {num_cus - per_xcd, num_cus}) {
    for (int px = 1; px <= std::min(p.nx / 4, cap); ++px)
      for (int py = 1; py <= std::min(p.ny / 4, cap / px); ++py) {
        CoopPlan pl;
        CoopLaunchShape sh;
        if (!fits(px, py, &pl) || !coop_launch_shape(px * py, num_cus, &sh)) continue;
        const int64_t launches = (std::max(1, p.batch) + sh.envs_per_launch - 1) / sh.envs_per_launch;
        const int64_t wk = launches * work(px, py);
        if (best < 0 || wk < best) {
          best = wk;
          *out = pl;
        }
      }
    if (best >= 0) return true;
  }
  return false;
}

// the part of "does the multi-workgroup kernel cover this problem" that does not depend on the tile plan
inline bool coop_problem_supported(const pdeopt_ctx* ctx) {
  const pdeopt_problem& p = ctx->prob;
  if (ctx->opt_small_persist < 0 || ctx->opt_kernel_path == 1 || ctx->opt_debug_ablate) return false;
  if (ctx->halo || p.nz > 1) return false;
  if (p.mu.kind == PDEOPT_CL_JIT || p.mob.kind == PDEOPT_CL_JIT) return false;  // run-time-compiled closures live in the generic stage kernel
  const int eq = p.equation;
  const bool sbm = eq == PDEOPT_EQ_ALLEN_CAHN_SBM || eq == PDEOPT_EQ_CAHN_HILLIARD_SBM;
  if (eq != PDEOPT_EQ_CAHN_HILLIARD && eq != PDEOPT_EQ_ALLEN_CAHN && !sbm && eq != PDEOPT_EQ_ADVECTION_DIFFUSION) return false;
  if (eq != PDEOPT_EQ_ADVECTION_DIFFUSION && p.derivs != PDEOPT_DERIVS_FD) return false;
  if (sbm) {
    // the kernel evaluates theta(t) / flux(t) itself: constants or polynomials (pdeopt_set_time_terms_poly); a host
    // callback per stage time cannot be asked from inside a launch
    if (ctx->time_fn && !ctx->time_poly_valid) return false;
    if (!ctx->aux[PDEOPT_AUX_SBM_PSI].dev || !ctx->aux[PDEOPT_AUX_SBM_NORM_GRAD].dev || !ctx->aux[PDEOPT_AUX_SBM_MASK].dev) return false;
  }
  if (eq == PDEOPT_EQ_ADVECTION_DIFFUSION) {
    if (has_time_aux(ctx, PDEOPT_AUX_VX_FACE) || has_time_aux(ctx, PDEOPT_AUX_VY_FACE)) return false;  // velocity_fn(t, .) varies
    if (!ctx->aux[PDEOPT_AUX_VX_FACE].dev || !ctx->aux[PDEOPT_AUX_VY_FACE].dev) return false;
  }
  return true;
}
template <typename T>
bool coop_tsit5_supported(const pdeopt_ctx* ctx) {
  CoopPlan pl;
  return coop_problem_supported(ctx) && coop_plan<T>(ctx->prob, ctx->num_cus, &pl);
}

template <typename T>
int coop_tsit5_solve(pdeopt_ctx* ctx, double t0, double t1, double dt0, const pdeopt_pid* pid, int64_t max_steps, int n_save,
                     const double* save_ts, void* save_host, pdeopt_tsit5_stats* stats_host) {
  const pdeopt_problem& p = ctx->prob;
  const int batch = p.batch;
  const int64_t cells = (int64_t)p.nx * p.ny;
  CoopPlan pl;
  if (!coop_plan<T>(p, ctx->num_cus, &pl)) return fail(ctx, PDEOPT_EINVAL, "the multi-workgroup adaptive kernel does not cover this problem");
  const int nwg = pl.px * pl.py;
  const int eq = p.equation;
  const bool sbm = eq == PDEOPT_EQ_ALLEN_CAHN_SBM || eq == PDEOPT_EQ_CAHN_HILLIARD_SBM;
  int rc;
  // exchange buffers: the integrators' work fields
  if ((rc = ensure_buffer(ctx, &ctx->TA, ctx->total_bytes))) return rc;
  if ((rc = ensure_buffer(ctx, &ctx->TB, ctx->total_bytes))) return rc;
  if ((rc = ensure_buffer(ctx, &ctx->ACC, ctx->total_bytes))) return rc;
  if ((rc = ensure_buffer(ctx, &ctx->KS, ctx->total_bytes))) return rc;

  CoopArgs<T> s{};
  s.nx = p.nx; s.ny = p.ny; s.px = pl.px; s.py = pl.py;
  s.bstride = make_geo(ctx).bstride;
  s.rhx = T(1.0 / p.hx); s.rhy = T(1.0 / p.hy);
  s.rhx2 = T(1.0 / (p.hx * p.hx)); s.rhy2 = T(1.0 / (p.hy * p.hy));
  s.mu = ClosureSpec{p.mu.kind, p.mu.flags, p.mu.n};
  s.mob = ClosureSpec{p.mob.kind, p.mob.flags, p.mob.n};
  s.fe = ClosureSpec{p.fe.kind, p.fe.flags, p.fe.n};
  s.t0 = t0; s.t1 = t1; s.dt0 = dt0;
  const double order = 5.0;
  s.pid.rtol = pid->rtol; s.pid.atol = pid->atol;
  s.pid.k1 = (pid->icoeff + pid->pcoeff + pid->dcoeff) / order;
  s.pid.k2 = -(pid->pcoeff + 2 * pid->dcoeff) / order;
  s.pid.k3 = pid->dcoeff / order;
  s.pid.factormin = pid->factormin; s.pid.factormax = pid->factormax; s.pid.safety = pid->safety;
  s.pid.dtmin = pid->dtmin; s.pid.dtmax = pid->dtmax;
  s.max_steps = max_steps;
  s.n_save = n_save;
  s.save_stride = (int64_t)batch * cells;
  s.rows = pl.rows; s.pitch = pl.pitch; s.red_off = pl.red_off;
  if (sbm) {
    const bool poly_const = ctx->time_poly_valid && ctx->time_theta[1] == 0.0 && ctx->time_theta[2] == 0.0 && ctx->time_theta[3] == 0.0 &&
                            ctx->time_flux[1] == 0.0 && ctx->time_flux[2] == 0.0 && ctx->time_flux[3] == 0.0;
    if (poly_const) {  // theta, flux constant (the notebook's first solve): the cosines once, here
      s.tmode = 0;
      s.tw[0] = cos(ctx->time_theta[0]);
      s.tw[1] = eq == PDEOPT_EQ_ALLEN_CAHN_SBM ? 0.0 : cos(3.14159265358979323846 - ctx->time_theta[0]);
      s.tw[2] = ctx->time_flux[0];
    } else if (ctx->time_poly_valid) {
      s.tmode = 1;
      for (int i = 0; i < 4; ++i) { s.theta[i] = ctx->time_theta[i]; s.flux[i] = ctx->time_flux[i]; }
    } else {
      s.tmode = 0;
      for (int i = 0; i < 3; ++i) s.tw[i] = ctx->time_const[i];
    }
  }

  // one device block: save times, statistics, barrier words + abort flag, partial sums, save slots
  const size_t ts_bytes = ((size_t)n_save * sizeof(double) + 255) / 256 * 256;
  const size_t st_bytes = ((size_t)batch * sizeof(pdeopt_tsit5_stats) + 255) / 256 * 256;
  const size_t bar_bytes = ((size_t)(2 * batch + 1) * sizeof(unsigned) + 255) / 256 * 256;
  const size_t part_bytes = ((size_t)2 * batch * nwg * 2 * sizeof(double) + 255) / 256 * 256;  // [parity][env][workgroup][2 words]
  const size_t out_bytes = (size_t)n_save * batch * cells * sizeof(T);
  const size_t need = ts_bytes + st_bytes + bar_bytes + part_bytes + out_bytes + 256;
  if (ctx->adaptive_cap < need) {
    if (ctx->adaptive_blk) {
      PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
      (void)hipFree(ctx->adaptive_blk);
      ctx->adaptive_blk = nullptr;
      ctx->adaptive_cap = 0;
    }
    PDEOPT_HIP_CHECK(ctx, hipMalloc(&ctx->adaptive_blk, need));
    ctx->adaptive_cap = need;
  }
  char* const blk = static_cast<char*>(ctx->adaptive_blk);
  s.save_ts = reinterpret_cast<const double*>(blk);
  pdeopt_tsit5_stats* const stats_dev = reinterpret_cast<pdeopt_tsit5_stats*>(blk + ts_bytes);
  unsigned* const bar_dev = reinterpret_cast<unsigned*>(blk + ts_bytes + st_bytes);
  double* const part_dev = reinterpret_cast<double*>(blk + ts_bytes + st_bytes + bar_bytes);
  T* const out_dev = reinterpret_cast<T*>(blk + ts_bytes + st_bytes + bar_bytes + part_bytes);
  if (n_save) {
    PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(blk, save_ts, (size_t)n_save * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PDEOPT_HIP_CHECK(ctx, hipMemsetAsync(out_dev, 0xFF, out_bytes, ctx->stream));  // NaN fill
  }
  PDEOPT_HIP_CHECK(ctx, hipMemsetAsync(stats_dev, 0, st_bytes + bar_bytes + part_bytes, ctx->stream));

  char name[112];
  const char* eqn = eq == PDEOPT_EQ_CAHN_HILLIARD ? "CH" : eq == PDEOPT_EQ_ALLEN_CAHN ? "AC" : eq == PDEOPT_EQ_CAHN_HILLIARD_SBM ? "CH-SBM"
                    : eq == PDEOPT_EQ_ALLEN_CAHN_SBM ? "AC-SBM" : "AD";
  const bool fast = eq == PDEOPT_EQ_ADVECTION_DIFFUSION || coop_fast_closures(p, sbm);
  snprintf(name, sizeof(name), "tsit5_coop<%s,%s,%s,%dx%d workgroups>", sizeof(T) == 4 ? "f32" : "f64", eqn, fast ? "fixed closures" : "generic closures",
           pl.px, pl.py);
  ctx->last_kernel = name;
  ctx->tsit5_pending = false;
  ctx->tsit5_fsal_valid = false;

  auto kern = [&]() -> const void* {
#define PDEOPT_COOP_K(EQV) (fast ? reinterpret_cast<const void*>(tsit5_coop_kernel<T, EQV, true>) : reinterpret_cast<const void*>(tsit5_coop_kernel<T, EQV, false>))
    switch (eq) {
      case PDEOPT_EQ_CAHN_HILLIARD: return PDEOPT_COOP_K(PDEOPT_EQ_CAHN_HILLIARD);
      case PDEOPT_EQ_ALLEN_CAHN: return PDEOPT_COOP_K(PDEOPT_EQ_ALLEN_CAHN);
      case PDEOPT_EQ_CAHN_HILLIARD_SBM: return PDEOPT_COOP_K(PDEOPT_EQ_CAHN_HILLIARD_SBM);
      case PDEOPT_EQ_ALLEN_CAHN_SBM: return PDEOPT_COOP_K(PDEOPT_EQ_ALLEN_CAHN_SBM);
      default: return reinterpret_cast<const void*>(tsit5_coop_kernel<T, PDEOPT_EQ_ADVECTION_DIFFUSION, true>);
    }
#undef PDEOPT_COOP_K
  }();
  PDEOPT_HIP_CHECK(ctx, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds));
  // Every workgroup of a launch must be resident at once (they wait for each other): one workgroup per compute unit,
  // an environment's workgroups on one XCD (blocks are dealt round-robin over the 8 XCDs of 32 CUs each)
  const int cus_per_xcd = std::max(1, ctx->num_cus / 8);
  int xs = 1;
  while (xs < 8 && (nwg + xs - 1) / xs > cus_per_xcd) xs *= 2;  // fp64 tiles are small: an environment may need two XCDs
  const int wpx = (nwg + xs - 1) / xs;
  if (wpx > cus_per_xcd) return fail(ctx, PDEOPT_EINVAL, "%d workgroups per environment exceed the device's %d compute units", nwg, ctx->num_cus);
  const int rows_per_launch = cus_per_xcd / wpx;
  const int envs_per_launch = rows_per_launch * (8 / xs);
  s.xs = xs;
  s.wpx = wpx;

  const T* s0 = nullptr; const T* s1 = nullptr; const T* s2 = nullptr;
  int64_t sstride = 0;
  if (sbm) {
    s0 = static_cast<const T*>(ctx->aux[PDEOPT_AUX_SBM_PSI].dev);
    s1 = static_cast<const T*>(ctx->aux[PDEOPT_AUX_SBM_NORM_GRAD].dev);
    s2 = static_cast<const T*>(ctx->aux[PDEOPT_AUX_SBM_MASK].dev);
  } else if (eq == PDEOPT_EQ_ADVECTION_DIFFUSION) {
    s0 = static_cast<const T*>(ctx->aux[PDEOPT_AUX_VX_FACE].dev);
    s1 = static_cast<const T*>(ctx->aux[PDEOPT_AUX_VY_FACE].dev);
    sstride = ctx->aux[PDEOPT_AUX_VX_FACE].per_env ? cells : 0;
  }
  // The workgroups of a launch wait for each other: two such launches of one process in flight at once (two engines on
  // two host threads) could each hold half of the chip and starve the other until the 2 s abort.  One at a time per
  // process (the call is synchronous anyway); other PROCESSES on the same GPU are the caller's to keep apart.
  std::lock_guard<std::mutex> coop_lock(coop_launch_mutex());
  for (int e0 = 0; e0 < batch; e0 += envs_per_launch) {
    const int ne = std::min(envs_per_launch, batch - e0);
    CoopArgs<T> c = s;
    c.nenv = ne;
    c.y = static_cast<T*>(ctx->Y) + (int64_t)e0 * s.bstride;
    c.ep = static_cast<const EnvParams<T>*>(ctx->env_params_dev) + e0;
    c.s0 = s0 ? s0 + (int64_t)e0 * sstride : nullptr;
    c.s1 = s1 ? s1 + (int64_t)e0 * sstride : nullptr;
    c.s2 = s2;
    c.sstride = sstride;
    c.save_out = out_dev + (int64_t)e0 * cells;
    c.stats = stats_dev + e0;
    c.xy[0] = static_cast<T*>(ctx->TA) + (int64_t)e0 * cells; c.xy[1] = static_cast<T*>(ctx->TB) + (int64_t)e0 * cells;
    c.xk[0] = static_cast<T*>(ctx->ACC) + (int64_t)e0 * cells; c.xk[1] = static_cast<T*>(ctx->KS) + (int64_t)e0 * cells;
    c.part = part_dev + (size_t)2 * e0 * nwg * 2;
    c.bar = bar_dev + 2 * e0;
    c.abort_flag = bar_dev + 2 * batch;
    const int rows_used = (ne + 8 / xs - 1) / (8 / xs);
    void* params[] = {&c};
    PDEOPT_HIP_CHECK(ctx, hipLaunchKernel(kern, dim3(8 * rows_used * wpx), dim3(PDEOPT_COOP_THREADS), params, pl.lds, ctx->stream));
    ctx->n_stage_launches++;
  }
  unsigned aborted = 0;
  PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(stats_host, stats_dev, (size_t)batch * sizeof(pdeopt_tsit5_stats), hipMemcpyDeviceToHost, ctx->stream));
  PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(&aborted, bar_dev + 2 * batch, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
  if (n_save) PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(save_host, out_dev, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
  PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (aborted)
    return fail(ctx, PDEOPT_ESTATE, "the multi-workgroup adaptive solve was aborted: a workgroup waited more than 2 s for its partners "
                                    "(the launch's workgroups were not all resident, or the device was shared)");
  return PDEOPT_OK;
}


// ------------------------------------------------------------------------------------- fixed step on several CUs

// Explicit Euler / RK4 with the MODE 1 kernel: when is it the better path?  One launch must hold every environment (the
// tiled kernels sweep a whole batch per launch; this kernel spends 16 - 100 compute units on ONE environment), and the
// grid must be beyond the one-CU whole-step kernel's range -- that kernel's 4.6 us per 64^2 substep is below what an
// exchange per substep allows here.  PDEOPT_OPT_SMALL_PERSIST = 2 takes it wherever it can run, -1 never.
template <typename T>
bool coop_fixed_chosen(const pdeopt_ctx* ctx, int integrator, int64_t n) {
  if (integrator != PDEOPT_INT_EULER && integrator != PDEOPT_INT_RK4) return false;
  if (!coop_problem_supported(ctx)) return false;
  const pdeopt_problem& p = ctx->prob;
  CoopPlan pl;
  if (!coop_plan<T>(p, ctx->num_cus, &pl, integrator == PDEOPT_INT_RK4 ? 2 : 1)) return false;
  CoopLaunchShape sh;
  if (!coop_launch_shape(pl.px * pl.py, ctx->num_cus, &sh)) return false;
  if (ctx->opt_small_persist == 2) return true;
  if (ctx->opt_small_persist != 0) return false;
  // a caller who turned one of the tiled path's knobs is asking for that path
  if (ctx->opt_fuse_stages != 0 || ctx->opt_kernel_path != 0 || ctx->opt_graph != 0 || ctx->opt_group_envs != 0 || ctx->opt_tile_rows != 0)
    return false;
  // measured, 100 RK4 substeps of one CH environment (tools/small_grid_bench.py, profiles/r04_small_grid_ch.txt), this kernel
  // / tiled / one CU: fp32 96^2 0.61 / 1.30 / 1.63 ms, 128^2 0.69 / 0.84 / 1.65, 256^2 0.92 / 0.88 / -; 64^2 0.54 / 0.82 /
  // 0.47; fp64 64^2 0.70 / 1.35 / 1.14, 32^2 0.56 / 1.42 / 0.37
  const int64_t cells = (int64_t)p.nx * p.ny;
  const int64_t lo = sizeof(T) == 8 ? kSmallAutoCells - 1 : kSmallAutoCells;  // (fp64 64^2: 0.70 ms, 16 of them 0.79, against the one-CU kernel's 1.15)
  // (up to 16 environments on >= 16 compute units each -- 16 x 96^2: 0.73 ms against 1.29 tiled, 16 x 128^2: 0.87 / 0.85;
  // beyond that the tiled kernels' one launch per batch is the better use of the chip)
  return n >= 8 && p.batch <= 16 && p.batch <= sh.envs_per_launch && pl.px * pl.py >= 16 && cells > lo && cells <= (sizeof(T) == 8 ? 256 * 256 : 192 * 192);  // (fp64 256^2: 1.08 ms against the stage pairs' 1.44)
}

template <typename T>
int coop_fixed_advance(pdeopt_ctx* ctx, int integrator, double t0, double dt, int64_t n) {
  const pdeopt_problem& p = ctx->prob;
  const int batch = p.batch;
  const int64_t cells = (int64_t)p.nx * p.ny;
  const bool rk4 = integrator == PDEOPT_INT_RK4;
  CoopPlan pl;
  CoopLaunchShape sh;
  if (!coop_plan<T>(p, ctx->num_cus, &pl, rk4 ? 2 : 1) || !coop_launch_shape(pl.px * pl.py, ctx->num_cus, &sh))
    return fail(ctx, PDEOPT_EINVAL, "the multi-workgroup fixed-step kernel does not cover this problem");
  const int nwg = pl.px * pl.py;
  const int eq = p.equation;
  const bool sbm = eq == PDEOPT_EQ_ALLEN_CAHN_SBM || eq == PDEOPT_EQ_CAHN_HILLIARD_SBM;
  int rc;
  if ((rc = ensure_buffer(ctx, &ctx->TA, ctx->total_bytes))) return rc;
  if ((rc = ensure_buffer(ctx, &ctx->TB, ctx->total_bytes))) return rc;
  CoopArgs<T> s{};
  s.nx = p.nx; s.ny = p.ny; s.px = pl.px; s.py = pl.py;
  s.bstride = make_geo(ctx).bstride;
  s.rhx = T(1.0 / p.hx); s.rhy = T(1.0 / p.hy);
  s.rhx2 = T(1.0 / (p.hx * p.hx)); s.rhy2 = T(1.0 / (p.hy * p.hy));
  s.mu = ClosureSpec{p.mu.kind, p.mu.flags, p.mu.n};
  s.mob = ClosureSpec{p.mob.kind, p.mob.flags, p.mob.n};
  s.fe = ClosureSpec{p.fe.kind, p.fe.flags, p.fe.n};
  s.t0 = t0; s.dt = dt; s.n_sub = n; s.fixed_rk4 = rk4 ? 1 : 0;
  s.rows = pl.rows; s.pitch = pl.pitch; s.red_off = pl.red_off;
  s.xs = sh.xs; s.wpx = sh.wpx;
  if (sbm) {
    if (ctx->time_poly_valid) {
      s.tmode = 1;
      for (int i = 0; i < 4; ++i) { s.theta[i] = ctx->time_theta[i]; s.flux[i] = ctx->time_flux[i]; }
    } else {
      s.tmode = 0;
      for (int i = 0; i < 3; ++i) s.tw[i] = ctx->time_const[i];
    }
  }
  // device block: the published round numbers of every workgroup + the abort flag
  const size_t need = ((size_t)batch * nwg + 1) * sizeof(unsigned) + 256;
  if (ctx->adaptive_cap < need) {
    if (ctx->adaptive_blk) {
      PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
      (void)hipFree(ctx->adaptive_blk);
      ctx->adaptive_blk = nullptr;
      ctx->adaptive_cap = 0;
    }
    PDEOPT_HIP_CHECK(ctx, hipMalloc(&ctx->adaptive_blk, need));
    ctx->adaptive_cap = need;
  }
  unsigned* const tags_dev = static_cast<unsigned*>(ctx->adaptive_blk);
  unsigned* const abort_dev = tags_dev + (size_t)batch * nwg;
  PDEOPT_HIP_CHECK(ctx, hipMemsetAsync(tags_dev, 0, ((size_t)batch * nwg + 1) * sizeof(unsigned), ctx->stream));

  const char* eqn = eq == PDEOPT_EQ_CAHN_HILLIARD ? "CH" : eq == PDEOPT_EQ_ALLEN_CAHN ? "AC" : eq == PDEOPT_EQ_CAHN_HILLIARD_SBM ? "CH-SBM"
                    : eq == PDEOPT_EQ_ALLEN_CAHN_SBM ? "AC-SBM" : "AD";
  const bool fast = eq == PDEOPT_EQ_ADVECTION_DIFFUSION || coop_fast_closures(p, sbm);
  char name[112];
  snprintf(name, sizeof(name), "%s_coop<%s,%s,%s,%dx%d workgroups>", rk4 ? "rk4" : "euler", sizeof(T) == 4 ? "f32" : "f64", eqn,
           fast ? "fixed closures" : "generic closures", pl.px, pl.py);
  ctx->last_kernel = name;
  auto kern = [&]() -> const void* {
#define PDEOPT_COOP_KF(EQV) (fast ? reinterpret_cast<const void*>(tsit5_coop_kernel<T, EQV, true, 1>) : reinterpret_cast<const void*>(tsit5_coop_kernel<T, EQV, false, 1>))
    switch (eq) {
      case PDEOPT_EQ_CAHN_HILLIARD: return PDEOPT_COOP_KF(PDEOPT_EQ_CAHN_HILLIARD);
      case PDEOPT_EQ_ALLEN_CAHN: return PDEOPT_COOP_KF(PDEOPT_EQ_ALLEN_CAHN);
      case PDEOPT_EQ_CAHN_HILLIARD_SBM: return PDEOPT_COOP_KF(PDEOPT_EQ_CAHN_HILLIARD_SBM);
      case PDEOPT_EQ_ALLEN_CAHN_SBM: return PDEOPT_COOP_KF(PDEOPT_EQ_ALLEN_CAHN_SBM);
      default: return reinterpret_cast<const void*>(tsit5_coop_kernel<T, PDEOPT_EQ_ADVECTION_DIFFUSION, true, 1>);
    }
#undef PDEOPT_COOP_KF
  }();
  PDEOPT_HIP_CHECK(ctx, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds));
  // (coop_threads<T, 1, EQ, FAST>() of the instantiation picked above)
  const int nthreads = sizeof(T) == 4 || (fast && !sbm) ? PDEOPT_COOP_FIXED_THREADS_F32 : PDEOPT_COOP_THREADS;
  const T* s0 = nullptr; const T* s1 = nullptr; const T* s2 = nullptr;
  int64_t sstride = 0;
  if (sbm) {
    s0 = static_cast<const T*>(ctx->aux[PDEOPT_AUX_SBM_PSI].dev);
    s1 = static_cast<const T*>(ctx->aux[PDEOPT_AUX_SBM_NORM_GRAD].dev);
    s2 = static_cast<const T*>(ctx->aux[PDEOPT_AUX_SBM_MASK].dev);
  } else if (eq == PDEOPT_EQ_ADVECTION_DIFFUSION) {
    s0 = static_cast<const T*>(ctx->aux[PDEOPT_AUX_VX_FACE].dev);
    s1 = static_cast<const T*>(ctx->aux[PDEOPT_AUX_VY_FACE].dev);
    sstride = ctx->aux[PDEOPT_AUX_VX_FACE].per_env ? cells : 0;
  }
  std::lock_guard<std::mutex> coop_lock(coop_launch_mutex());  // (see coop_tsit5_solve)
  for (int e0 = 0; e0 < batch; e0 += sh.envs_per_launch) {
    const int ne = std::min(sh.envs_per_launch, batch - e0);
    CoopArgs<T> c = s;
    c.nenv = ne;
    c.y = static_cast<T*>(ctx->Y) + (int64_t)e0 * s.bstride;
    c.ep = static_cast<const EnvParams<T>*>(ctx->env_params_dev) + e0;
    c.s0 = s0 ? s0 + (int64_t)e0 * sstride : nullptr;
    c.s1 = s1 ? s1 + (int64_t)e0 * sstride : nullptr;
    c.s2 = s2;
    c.sstride = sstride;
    c.xy[0] = static_cast<T*>(ctx->TA) + (int64_t)e0 * cells; c.xy[1] = static_cast<T*>(ctx->TB) + (int64_t)e0 * cells;
    c.tags = tags_dev + (size_t)e0 * nwg;
    c.abort_flag = abort_dev;
    const int rows_used = (ne + 8 / sh.xs - 1) / (8 / sh.xs);
    void* params[] = {&c};
    PDEOPT_HIP_CHECK(ctx, hipLaunchKernel(kern, dim3(8 * rows_used * sh.wpx), dim3(nthreads), params, pl.lds, ctx->stream));
    ctx->n_stage_launches++;
  }
  unsigned aborted = 0;
  PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(&aborted, abort_dev, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
  PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (aborted)
    return fail(ctx, PDEOPT_ESTATE, "the multi-workgroup fixed-step advance was aborted: a workgroup waited more than 2 s for its partners "
                                    "(the launch's workgroups were not all resident, or the device was shared)");
  return PDEOPT_OK;
}

}  // namespace pdeopt
