// Whole-environment-step kernel for grids that fit one compute unit's LDS (gfx950: 160 KiB).
//
// The reference's own tests and notebooks integrate 32^2 ... 128^2 grids (tests/test_solvers.py:25,68,145;
// the diffeqsolve loop at pde_opt/pde_env.py:293-303).  At those sizes the tiled kernels are latency-bound: one RK4
// substep is two dependent launches of ~5.8 us each whatever the grid, 1.15 ms per 100
// substeps from 32^2 to 256^2.  Here ONE launch runs all n substeps of Euler / RK4:
//
//   * one workgroup per environment (batch on blockIdx.x: environments are independent);
//   * the stage input w and the chemical potential mu live in LDS for the whole launch, the state y and the RK
//     accumulator stay in the owning threads' registers; global memory is touched twice -- y in at the start, y out
//     at the end;
//   * a thread owns up to KMAX 16-byte vectors of cells; periodic neighbours are LDS offsets derived from the centre's;
//   * up to 2 vectors per thread the workgroup has 1024 threads (128 registers each); beyond (grids past 64 x 128
//     fp32) the state of a thread -- y, the accumulator and the next stage input of its vectors -- no longer fits
//     beside the ~70 transient registers of a flux evaluation, so the workgroup has 512 threads with 256 registers
//     each and up to 8 vectors per thread (128^2 fp32, 64 x 128 fp64);
//   * per stage:  mu pass -> barrier -> flux divergence + Runge-Kutta update in registers -> barrier -> the next
//     stage input into LDS -> barrier   (Allen-Cahn: no mu array, two barriers).
//
// Arithmetic: the SAME expressions as the fused stage-pair kernels (stencil_fused.hpp: the mu form, face_flux /
// flux_divergence, the update formulas in the same association), so a state advanced here equals the tiled path's
// to the last bit wherever the compiler contracts alike (checked to <= 1 ulp, tests/test_gpu_small.py).
// Reference arithmetic: cahn_hilliard.py:89-109, allen_cahn.py:81-84, derivatives.py:8-61.
//
// Bound: VALU issue of ONE compute unit -- 4 SIMDs -- plus 12 workgroup barriers per RK4 substep; the chip runs up to
// 256 such environments at once (512 at <= 64^2 fp32, two workgroups per CU).  bench.py --workload ch_rk4_64_f32_small.
#pragma once

#include "stencil_fused.hpp"

namespace pdeopt {

template <typename T>
struct SmallArgs {
  T* y;             // [batch][nx][ny] state, advanced in place
  int nx, ny;
  int64_t bstride;  // elements between environments
  int64_t n;        // substeps
  int rk4;          // 1: classical RK4, 0: explicit Euler
  T dt, h2, h3, h6;  // dt, dt/2, dt/3, dt/6 (formed in double on the host, as the stage-pair launches do)
  T rhx, rhy;        // 0.5 / hx^2, 0.5 / hy^2 (folded flux constants, stencil_fused.hpp: face_flux)
  T rhx2, rhy2;      // 1 / hx^2, 1 / hy^2
  const EnvParams<T>* ep;
  ClosureSpec mu, mob;
};

// What a workgroup keeps for its environment: the LDS arrays, the LDS offsets of the thread's vectors and the folded
// constants; `stage` = one right-hand-side evaluation on the owned vectors + whatever the integrator does with it.
// Shared by the fixed-step kernel below and the adaptive Tsit5 kernel (stencil_small_adaptive.hpp).
template <typename T, int EQ, int CL, int KMAX>
struct SmallTile {
  using Vec = typename VecOf<T>::type;
  static constexpr int V = VecOf<T>::V;
  static constexpr bool kIsCH = EQ == PDEOPT_EQ_CAHN_HILLIARD;
  static constexpr bool FOLD_MU = PDEOPT_PAIR_FOLD_MU && CL == CL_LOGIT1;

  T* sU;
  T* sMu;  // CH only
  int ny, cells;
  // owned vectors: the LDS offset of the centre; the rows above / below and the scalar neighbours left / right
  // (periodic) are re-derived from it where they are used -- two instructions each instead of four more registers
  // per vector (a thread keeps y and the integrator's slopes of up to KMAX vectors in registers)
  int oc[KMAX];
  unsigned cflags;  // bit 3k: owned, bit 3k + 1: first vector of its row, bit 3k + 2: last vector of its row
  const EnvParams<T>* p;
  ClosureSpec mu, mob;
  T kap, rhx, rhy, rhx2, rhy2, fA, fB, q1;

  // map the thread's vectors, copy the environment's state y0 (global) into sU and into yout[]
  __device__ __forceinline__ void init(char* smem, int nx, int ny_, const EnvParams<T>* ep, ClosureSpec mu_, ClosureSpec mob_, T rhx_,
                                       T rhy_, T rhx2_, T rhy2_, const T* yg, Vec* yout) {
    ny = ny_;
    cells = nx * ny_;
    sU = reinterpret_cast<T*>(smem);
    sMu = sU + cells;
    p = ep;
    mu = mu_; mob = mob_;
    kap = ep->kappa; rhx = rhx_; rhy = rhy_; rhx2 = rhx2_; rhy2 = rhy2_;
    fA = fB = q1 = T(0);
    // mu = mu_h(c) - kappa lap c: the linear-logit class in the folded form of stencil_fused.hpp (FOLD_MU)
    if constexpr (FOLD_MU) {
      fA = -kap * rhx2;
      fB = -kap * rhy2;
      q1 = ep->mu[1] - T(2) * (fA + fB);
    }
    const int tid = threadIdx.x, NT = blockDim.x;
    const int nvr = ny / V;  // vectors per row
    const int nvec = nx * nvr;
    cflags = 0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      // A thread whose k-th vector index runs past the environment works on the LAST vector instead: the same inputs,
      // the same arithmetic, the same values stored to the same LDS words as that vector's real owner -- straight-line
      // stage code for every thread (with `if (owned)` around each vector the register allocator kept both arms'
      // values alive and spilled: 1.2 KB of scratch at 8 vectors per thread); only stores to global memory are the
      // owner's alone.
      const int vid = tid + k * NT;
      const bool mine = vid < nvec;
      const int v = mine ? vid : nvec - 1;
      const int row = v / nvr, cv = v - row * nvr;
      oc[k] = row * ny + cv * V;
      cflags |= (mine ? 1u : 0u) << (3 * k) | (cv == 0 ? 1u : 0u) << (3 * k + 1) | (cv == nvr - 1 ? 1u : 0u) << (3 * k + 2);
      const Vec y0 = *reinterpret_cast<const Vec*>(yg + oc[k]);
      *reinterpret_cast<Vec*>(sU + oc[k]) = y0;
      yout[k] = y0;
    }
    __syncthreads();
  }
  __device__ __forceinline__ bool own(int k) const { return (cflags >> (3 * k)) & 1u; }
  __device__ __forceinline__ int up_of(int o) const { const int r = o + ny; return r >= cells ? r - cells : r; }  // row + 1
  __device__ __forceinline__ int dn_of(int o) const { const int r = o - ny; return r < 0 ? r + cells : r; }       // row - 1
  __device__ __forceinline__ int left_of(int o, int k) const { return o - 1 + (((cflags >> (3 * k + 1)) & 1u) ? ny : 0); }
  __device__ __forceinline__ int right_of(int o, int k) const { return o + V - (((cflags >> (3 * k + 2)) & 1u) ? ny : 0); }

  // One Runge-Kutta stage: k = f(w) on the owned vectors (w in sU), `update(j, k_j)` applied to each while it is in
  // registers, then the next stage input `next` published.  The vectors of a thread are processed ONE AFTER THE
  // OTHER (scheduling barriers): interleaved, their ~70 transient registers each would not fit beside the state.
  template <typename U>
  __device__ __forceinline__ void stage(U update, const Vec* next) {
    rhs(update);
    publish(next);
  }
  // the next stage input into sU, once every read of the current one is done
  __device__ __forceinline__ void publish(const Vec* next) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KMAX; ++k) *reinterpret_cast<Vec*>(sU + oc[k]) = next[k];
    __syncthreads();
  }
  template <typename U>
  __device__ __forceinline__ void rhs(U update) {
    if constexpr (kIsCH) {
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        {
          const int o = oc[k];
          const Vec c = *reinterpret_cast<const Vec*>(sU + o);
          const Vec xp = *reinterpret_cast<const Vec*>(sU + up_of(o));
          const Vec xm = *reinterpret_cast<const Vec*>(sU + dn_of(o));
          const T left = sU[left_of(o, k)], right = sU[right_of(o, k)];
          Vec m;
#pragma unroll
          for (int e = 0; e < V; ++e) {
            const T ym = (e == 0) ? left : c[e - 1];
            const T yp = (e == V - 1) ? right : c[e + 1];
            if constexpr (FOLD_MU)
              m[e] = fA * (xp[e] + xm[e]) + (fB * (yp + ym) + (q1 * c[e] + p->mu[0] + t_logit<T>(c[e])));
            else
              m[e] = eval_mu<T, CL>(mu, p->mu, c[e]) - kap * lap_at<T>(c[e], xp[e], xm[e], yp, ym, rhx2, rhy2);
          }
          *reinterpret_cast<Vec*>(sMu + o) = m;
        }
        if constexpr (KMAX > 2) __builtin_amdgcn_sched_barrier(0);  // one vector at a time
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        {
          const int o = oc[k], ou = up_of(o), od = dn_of(o), ol = left_of(o, k), orr = right_of(o, k);
          const Vec kv = flux_divergence<T, CL, Vec, V>(
              mob, p->mob, *reinterpret_cast<const Vec*>(sMu + od), *reinterpret_cast<const Vec*>(sMu + o),
              *reinterpret_cast<const Vec*>(sMu + ou), *reinterpret_cast<const Vec*>(sU + od),
              *reinterpret_cast<const Vec*>(sU + o), *reinterpret_cast<const Vec*>(sU + ou), sMu[ol], sMu[orr], sU[ol], sU[orr], rhx,
              rhy);
          update(k, kv);
        }
        if constexpr (KMAX > 2) __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // Allen-Cahn (stencil_fused_ac.hpp: k_at): k = -R(c) (mu_h(c) - kappa lap c)
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        {
          const int o = oc[k];
          const Vec c = *reinterpret_cast<const Vec*>(sU + o);
          const Vec xp = *reinterpret_cast<const Vec*>(sU + up_of(o));
          const Vec xm = *reinterpret_cast<const Vec*>(sU + dn_of(o));
          const T left = sU[left_of(o, k)], right = sU[right_of(o, k)];
          Vec r;
#pragma unroll
          for (int e = 0; e < V; ++e) {
            const T ym = (e == 0) ? left : c[e - 1];
            const T yp = (e == V - 1) ? right : c[e + 1];
            const T m = eval_mu<T, CL>(mu, p->mu, c[e]) - kap * lap_at<T>(c[e], xp[e], xm[e], yp, ym, rhx2, rhy2);
            r[e] = -eval_mob<T, CL>(mob, p->mob, c[e]) * m;
          }
          update(k, r);
        }
        if constexpr (KMAX > 2) __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
};

// DB: the stage input is double-buffered in LDS ([input A][mu][input B]: three arrays, where they fit) -- a stage's update
// writes the next input straight into the idle buffer while neighbours may still read the current one, so a stage costs
// two barriers (mu pass | flux + update) instead of three (... | publish), Allen-Cahn one instead of two, and the next
// input needs no registers (stencil_small_adaptive.hpp's scheme).  Same formulas, same bits.
template <typename T, int EQ, int CL, int KMAX, int NTMAX, bool DB = false>
__global__ __launch_bounds__(NTMAX) void small_persist_kernel(const SmallArgs<T> a) {
  using Tile = SmallTile<T, EQ, CL, KMAX>;
  using Vec = typename Tile::Vec;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int b = blockIdx.x;
  T* const yg = a.y + (int64_t)b * a.bstride;
  Tile tile;
  if constexpr (DB) {
    Vec y[KMAX], acc[KMAX];
    tile.init(smem_raw, a.nx, a.ny, a.ep + b, a.mu, a.mob, a.rhx, a.rhy, a.rhx2, a.rhy2, yg, y);
    T* sNext = tile.sU + 2 * tile.cells;
    auto put = [&](int k, const Vec v) { *reinterpret_cast<Vec*>(sNext + tile.oc[k]) = v; };
    auto flip = [&]() {  // the next input is complete once every thread is here; the current one is then free
      __syncthreads();
      T* const tmp = tile.sU;
      tile.sU = sNext;
      sNext = tmp;
    };
#pragma unroll
    for (int k = 0; k < KMAX; ++k) acc[k] = Vec{};
    for (int64_t s = 0; s < a.n; ++s) {
      if (a.rk4) {
        tile.rhs([&](int k, const Vec kv) { put(k, y[k] + a.h2 * kv); acc[k] = y[k] + a.h6 * kv; });
        flip();
        tile.rhs([&](int k, const Vec kv) { put(k, y[k] + a.h2 * kv); acc[k] = acc[k] + a.h3 * kv; });
        flip();
        tile.rhs([&](int k, const Vec kv) { put(k, y[k] + a.dt * kv); acc[k] = acc[k] + a.h3 * kv; });
        flip();
        tile.rhs([&](int k, const Vec kv) { y[k] = acc[k] + a.h6 * kv; put(k, y[k]); });
        flip();
      } else {
        tile.rhs([&](int k, const Vec kv) { y[k] = y[k] + a.dt * kv; put(k, y[k]); });
        flip();
      }
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (tile.own(k)) *reinterpret_cast<Vec*>(yg + tile.oc[k]) = y[k];
    return;
  }
  Vec y[KMAX], acc[KMAX], w[KMAX];
  tile.init(smem_raw, a.nx, a.ny, a.ep + b, a.mu, a.mob, a.rhx, a.rhy, a.rhx2, a.rhy2, yg, y);
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    acc[k] = Vec{};
    w[k] = Vec{};
  }
  for (int64_t s = 0; s < a.n; ++s) {
    if (a.rk4) {
      // the update formulas of the stage-pair kernels, in their association (stencil_fused.hpp: PAIR_12 / PAIR_34):
      //   w2 = y + dt/2 k1,  acc = y + dt/6 k1;   w3 = y + dt/2 k2,  acc += dt/3 k2;
      //   w4 = y + dt k3,    acc += dt/3 k3;      y' = acc + dt/6 k4
      tile.stage([&](int k, const Vec kv) { w[k] = y[k] + a.h2 * kv; acc[k] = y[k] + a.h6 * kv; }, w);
      tile.stage([&](int k, const Vec kv) { w[k] = y[k] + a.h2 * kv; acc[k] = acc[k] + a.h3 * kv; }, w);
      tile.stage([&](int k, const Vec kv) { w[k] = y[k] + a.dt * kv; acc[k] = acc[k] + a.h3 * kv; }, w);
      tile.stage([&](int k, const Vec kv) { y[k] = acc[k] + a.h6 * kv; }, y);
    } else {
      tile.stage([&](int k, const Vec kv) { y[k] = y[k] + a.dt * kv; }, y);
    }
  }
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    if (tile.own(k)) *reinterpret_cast<Vec*>(yg + tile.oc[k]) = y[k];
}

// --------------------------------------------------------------------------------------------- host

template <typename T>
inline size_t small_lds_bytes(const pdeopt_ctx* ctx) {
  const size_t cells = (size_t)ctx->prob.nx * ctx->prob.ny;
  return cells * sizeof(T) * (ctx->prob.equation == PDEOPT_EQ_CAHN_HILLIARD ? 2 : 1);
}

constexpr size_t kSmallLdsMax = 160 * 1024;
#ifndef PDEOPT_SMALL_DOUBLE_BUFFER
#define PDEOPT_SMALL_DOUBLE_BUFFER 1
#endif
constexpr int kSmallMaxVec = 8 * 512;  // vectors of one environment: 8 per thread of the 512-thread form (10 spill)
// auto policy (stencil.hip: small_chosen)
// measured, 100 RK4 substeps, Cahn-Hilliard fp32, whole-step vs tiled (tools/small_grid_bench.py, profiles/r03_small_grid.txt):
//   32^2: 0.24 vs 1.15 ms   64^2: 0.46 vs 1.23 ms (one environment; the same up to 256)   128^2: 1.92 vs 1.15 ms for 1..64
//   environments, 2.0 vs 3.2 ms for 256 -- one CU per environment is slower than eight until the chip is full
constexpr int64_t kSmallAutoCells = 4096;  // up to 64^2: always
constexpr int kSmallAutoBatch = 192;       // larger LDS-resident grids: from this many environments on

// The grid as the kernel walks it.  An (nx, 1) column -- the reference's 1-D runs (tests/test_solvers.py:25,68: 256 x 1)
// -- is the same memory as a (1, nx) row, and the kernel's vectors run along rows: walk it as the row.  The direction
// with one point contributes differences of a value with itself; its reciprocal spacing is set to 0 so that the zero
// stays a zero whatever the domain's extent there.
struct SmallDims {
  int nx, ny;
  double rx2, ry2;  // 1 / hx^2, 1 / hy^2
};
inline SmallDims small_dims(const pdeopt_problem& p) {
  if (p.ny == 1 && p.nx > 1) return {1, p.nx, 0.0, 1.0 / (p.hx * p.hx)};
  return {p.nx, p.ny, p.nx > 1 ? 1.0 / (p.hx * p.hx) : 0.0, p.ny > 1 ? 1.0 / (p.hy * p.hy) : 0.0};
}

// can the whole-step kernel run this problem at all?
template <typename T>
bool small_supported(const pdeopt_ctx* ctx) {
  constexpr int V = VecOf<T>::V;
  const pdeopt_problem& p = ctx->prob;
  if (p.equation != PDEOPT_EQ_CAHN_HILLIARD && p.equation != PDEOPT_EQ_ALLEN_CAHN) return false;
  if (p.derivs != PDEOPT_DERIVS_FD || ctx->halo || p.nz > 1) return false;
  const SmallDims d = small_dims(p);
  if (d.ny % V != 0 || d.nx < 1) return false;
  if (classify_closures(p.mu, p.mob) == CL_GENERIC) return false;
  if (small_lds_bytes<T>(ctx) > kSmallLdsMax) return false;
  const int64_t nvec = (int64_t)d.nx * (d.ny / V);
  return nvec <= (int64_t)kSmallMaxVec;
}

// threads per workgroup and vectors per thread for nvec vectors per environment
inline void small_shape(int64_t nvec, int* nt, int* kmax) {
  if (nvec <= 1024) { *nt = (int)((nvec + 63) / 64 * 64); *kmax = 1; }
  else if (nvec <= 2048) { *nt = 1024; *kmax = 2; }
  else { *nt = 512; *kmax = (int)((nvec + 511) / 512); }
}

template <typename T, int EQ, int CL>
int launch_small_k(pdeopt_ctx* ctx, const SmallArgs<T>& s, int nt, int kmax, size_t lds) {
  auto go = [&](auto kern) -> int {
    if (lds > 48 * 1024)
      PDEOPT_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(ctx->win_n), dim3(nt), lds, ctx->stream, s);
    PDEOPT_HIP_CHECK(ctx, hipGetLastError());
    return PDEOPT_OK;
  };
  // room for the double-buffered stage input (launch_small decides; grids of that size run 1 or 2 vectors per thread)
  if (lds >= (size_t)3 * s.nx * s.ny * sizeof(T) && (nt > 512 || kmax <= 2))
    return kmax <= 1 ? go(small_persist_kernel<T, EQ, CL, 1, 1024, true>) : go(small_persist_kernel<T, EQ, CL, 2, 1024, true>);
  if (nt > 512 || kmax <= 2) return kmax <= 1 ? go(small_persist_kernel<T, EQ, CL, 1, 1024>) : go(small_persist_kernel<T, EQ, CL, 2, 1024>);
  return go(small_persist_kernel<T, EQ, CL, 8, 512>);
}

// n substeps of Euler / RK4 for the environments of the current window, one launch
template <typename T>
int launch_small(pdeopt_ctx* ctx, int integrator, double dt, int64_t n) {
  constexpr int V = VecOf<T>::V;
  const pdeopt_problem& p = ctx->prob;
  SmallArgs<T> s{};
  const Geo g = make_geo(ctx);
  s.y = static_cast<T*>(ctx->Y) + (int64_t)ctx->win_lo * g.bstride;
  const SmallDims d = small_dims(p);
  s.nx = d.nx;
  s.ny = d.ny;
  s.bstride = g.bstride;
  s.n = n;
  s.rk4 = integrator == PDEOPT_INT_RK4 ? 1 : 0;
  s.dt = T(dt); s.h2 = T(dt / 2); s.h3 = T(dt / 3); s.h6 = T(dt / 6);
  s.rhx = T(0.5 * d.rx2); s.rhy = T(0.5 * d.ry2);
  s.rhx2 = T(d.rx2); s.rhy2 = T(d.ry2);
  s.ep = static_cast<const EnvParams<T>*>(ctx->env_params_dev) + ctx->win_lo;
  s.mu = ClosureSpec{p.mu.kind, p.mu.flags, p.mu.n};
  s.mob = ClosureSpec{p.mob.kind, p.mob.flags, p.mob.n};
  const int64_t nvec = (int64_t)d.nx * (d.ny / V);
  int nt, kmax;
  small_shape(nvec, &nt, &kmax);
  // the stage input double-buffered where three arrays leave room for a second workgroup on the CU (<= 80 KB: up to 64^2
  // fp32 -- the grids that run two environments per CU in large batches)
  const size_t lds3 = (size_t)3 * d.nx * d.ny * sizeof(T);
  const bool db = PDEOPT_SMALL_DOUBLE_BUFFER && lds3 <= kSmallLdsMax / 2;
  const size_t lds = db ? lds3 : small_lds_bytes<T>(ctx);
  const int cl = classify_closures(p.mu, p.mob);
  char name[112];
  snprintf(name, sizeof(name), "small_persist<%s,%s,%s,%dx%d threads,%d vec/thread%s>", sizeof(T) == 4 ? "f32" : "f64",
           p.equation == PDEOPT_EQ_ALLEN_CAHN ? "AC" : "CH", cl == CL_LOGIT ? "logit" : "poly", 1, nt, kmax, db ? ",2 inputs" : "");
  ctx->last_kernel = name;
  ctx->n_stage_launches++;
  if (p.equation == PDEOPT_EQ_CAHN_HILLIARD) {
    if (cl == CL_LOGIT && p.mu.n <= 2) return launch_small_k<T, PDEOPT_EQ_CAHN_HILLIARD, CL_LOGIT1>(ctx, s, nt, kmax, lds);
    if (cl == CL_LOGIT) return launch_small_k<T, PDEOPT_EQ_CAHN_HILLIARD, CL_LOGIT>(ctx, s, nt, kmax, lds);
    return launch_small_k<T, PDEOPT_EQ_CAHN_HILLIARD, CL_POLY>(ctx, s, nt, kmax, lds);
  }
  if (cl == CL_LOGIT) return launch_small_k<T, PDEOPT_EQ_ALLEN_CAHN, CL_LOGIT>(ctx, s, nt, kmax, lds);
  return launch_small_k<T, PDEOPT_EQ_ALLEN_CAHN, CL_POLY>(ctx, s, nt, kmax, lds);
}

}  // namespace pdeopt
