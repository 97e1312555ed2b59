// Strang split step with the FFTs held in LDS and the pointwise operators fused into the passes.
//
// Reference step (pde_opt/numerics/solvers.py:99-122), tau = dt * time_scale, E = exp(A_term tau / 2):
//   psi1 = ifft2(fft2(psi0) E);  b = -i (V + k |psi0|^2)   (gross_pitaevskii.py:67-75, PRE-half-step state)
//   psi2 = psi1 exp(b tau);      psi3 = psi2 / sqrt(sum |psi2|^2 dx^2);      psi4 = ifft2(fft2(psi3) E)
//
// Passes over HBM per step (each reads and writes the field once, in place; 16 B/cell at c64):
//   row  FIRST/JOIN :            [IFFT_y of the previous step] -> |psi0|^2 -> FFT_y
//   col             :  FFT_x -> * E/(nx ny)            -> IFFT_x
//   row  MID        :  IFFT_y -> * exp(b tau), partial sums of |psi2|^2 -> FFT_y
//   col  (scaled)   :  FFT_x -> * E/(nx ny) * scale_b  -> IFFT_x        (scale_b from the partial sums)
//   row  LAST       :  IFFT_y                           (only after the final step)
// = 4 passes per step in steady state, against 8 rocFFT passes + 4 pointwise kernels on the library
// path (csrc/spectral.hip, which stays the path for sizes that are not powers of two in 64..1024).
// The normalisation scalar is folded into the second spectral multiply (the FFT is linear), the
// reduction is deterministic (fixed partition, fp64 partials summed in a fixed order).
#include <cmath>
#include <complex>

#include "common.hpp"
#include "fft_lds.hpp"
#include "fft_reg.hpp"

namespace pdeopt {

struct StrangFused {
  void* tw_x = nullptr;  // twiddle tables exp(-2 pi i n / N) in the problem dtype
  void* tw_y = nullptr;
  void* mult = nullptr;  // E / (nx ny), complex [nx][ny]
  void* dens = nullptr;  // |psi0|^2, real [batch][nx][ny]
  double* partial = nullptr;
  double key_dt = NAN, key_tr = NAN, key_ti = NAN;
  bool mult_valid = false;
  // IMEX on the same transforms
  void* cwork = nullptr;      // complex work field [batch][nx][ny]
  void* imex_mult = nullptr;  // 1 / ((1 + A dt symbol) nx ny)
  double imex_dt = NAN, imex_A = NAN;
  bool imex_valid = false;
};

namespace {

enum { ROW_FIRST = 0, ROW_MID = 1, ROW_JOIN = 2, ROW_LAST = 3 };

template <typename T>
__device__ __forceinline__ void sincos_t(T x, T* s, T* c);
// fp32: hardware v_sin_f32 / v_cos_f32 (arguments in revolutions) behind a two-term reduction of
// x / 2 pi, instead of ocml's sincosf (~45 VALU instructions per call plus a private-memory slow
// path): the row pass evaluates one per cell and was VALU-bound on it.  Absolute error ~2e-7 on a
// unit-modulus factor, below the fp32 rounding of the transforms around it; fp64 keeps sincos().
template <>
__device__ __forceinline__ void sincos_t<float>(float x, float* s, float* c) {
  const float c1 = 0.15915494f;        // fl(1 / 2 pi)
  const float c2 = 6.4206383e-09f;     // 1 / 2 pi - c1
  const float hi = x * c1;
  const float lo = __builtin_fmaf(x, c1, -hi) + x * c2;
  const float r = __builtin_amdgcn_fractf(hi) + lo;  // revolutions, |lo| tiny: sin / cos are 1-periodic in r
  *s = __builtin_amdgcn_sinf(r);
  *c = __builtin_amdgcn_cosf(r);
}
template <>
__device__ __forceinline__ void sincos_t<double>(double x, double* s, double* c) { sincos(x, s, c); }
template <typename T>
__device__ __forceinline__ T exp_t(T x);
template <>
__device__ __forceinline__ float exp_t<float>(float x) { return expf(x); }
template <>
__device__ __forceinline__ double exp_t<double>(double x) { return exp(x); }

// F consecutive rows (FFT along the contiguous axis) per 256-thread workgroup
template <typename T, int N, int F, int MODE>
__global__ __launch_bounds__(256) void strang_row_kernel(Cx<T>* __restrict__ psi, T* __restrict__ dens,
                                                         const T* __restrict__ pot, int64_t pot_env_stride,
                                                         const EnvParams<T>* __restrict__ ep,
                                                         const Cx<T>* __restrict__ tw_g, T tr, T ti, int nx,
                                                         double* __restrict__ partial) {
  constexpr int NP = fft_lds_pitch<N>();
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  Cx<T>* const s = reinterpret_cast<Cx<T>*>(smem_raw);
  Cx<T>* const tw = s + F * NP;
  const int tid = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * F;
  Cx<T>* const g = psi + row0 * N;
  for (int n = tid; n < N; n += 256) tw[n] = tw_g[n];
  for (int idx = tid; idx < F * N; idx += 256) {
    const int f = idx / N, k = idx - f * N;
    const Cx<T> v = g[idx];
    if constexpr (MODE == ROW_FIRST) {
      s[f * NP + fft_lds_addr(k)] = v;
      dens[row0 * N + idx] = v.re * v.re + v.im * v.im;
    } else {
      s[f * NP + fft_lds_addr(fft_pos_of<N>(k))] = v;  // natural-order spectrum -> the layout dit() consumes
    }
  }
  __syncthreads();
  if constexpr (MODE != ROW_FIRST) {
    fft_dit<T, N, F, NP, +1>(s, tw, tid);  // unnormalised inverse: 1/(nx ny) sits in the column multiplier
    if constexpr (MODE == ROW_LAST) {
      for (int idx = tid; idx < F * N; idx += 256) {
        const int f = idx / N, k = idx - f * N;
        g[idx] = s[f * NP + fft_lds_addr(k)];
      }
      return;
    }
    const int env = (int)(row0 / nx);
    if constexpr (MODE == ROW_MID) {
      const T kk = ep[env].gpe_k;
      const T* vrow = pot ? pot + (int64_t)env * pot_env_stride + (row0 - (int64_t)env * nx) * N : nullptr;
      double acc = 0.0;
      for (int idx = tid; idx < F * N; idx += 256) {
        const int f = idx / N, k = idx - f * N;
        const T w = (vrow ? vrow[idx] : T(0)) + kk * dens[row0 * N + idx];
        // exp(-i w (tr + i ti)) = exp(w ti) (cos(w tr) - i sin(w tr))
        T sn, cs;
        sincos_t<T>(w * tr, &sn, &cs);
        const T mag = (ti == T(0)) ? T(1) : exp_t<T>(w * ti);
        const Cx<T> e{mag * cs, -mag * sn};
        const Cx<T> r = cmul(s[f * NP + fft_lds_addr(k)], e);
        s[f * NP + fft_lds_addr(k)] = r;
        acc += (double)r.re * (double)r.re + (double)r.im * (double)r.im;
      }
      __shared__ double red[4];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
      if ((tid & 63) == 0) red[tid >> 6] = acc;
      __syncthreads();
      if (tid == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    } else {  // ROW_JOIN: the real-space field is psi0 of the next step
      for (int idx = tid; idx < F * N; idx += 256) {
        const int f = idx / N, k = idx - f * N;
        const Cx<T> v = s[f * NP + fft_lds_addr(k)];
        dens[row0 * N + idx] = v.re * v.re + v.im * v.im;
      }
    }
    __syncthreads();
  }
  fft_dif<T, N, F, NP, -1>(s, tw, tid);
  for (int idx = tid; idx < F * N; idx += 256) {
    const int f = idx / N, k = idx - f * N;
    g[idx] = s[f * NP + fft_lds_addr(fft_pos_of<N>(k))];
  }
}

// The row pass with the transforms in registers (fft_reg.hpp): N/8 threads per row, 256/(N/8) rows per
// workgroup; for N <= 512 a row lives in one wave and the pass has no s_barrier except the one of the
// norm reduction.  Spectrum side: thread j holds the frequencies j + m N/8; real-space side: the cells
// j + m N/8 -- both coalesced.  Same modes and arithmetic as strang_row_kernel.
template <typename T, int N, int MODE>
__global__ __launch_bounds__(256) void strang_row_reg_kernel(Cx<T>* __restrict__ psi, T* __restrict__ dens,
                                                             const T* __restrict__ pot, int64_t pot_env_stride,
                                                             const EnvParams<T>* __restrict__ ep,
                                                             const Cx<T>* __restrict__ tw, T tr, T ti, int nx,
                                                             double* __restrict__ partial) {
  constexpr int TT = N / 8, F = 256 / TT, NP = fft_lds_pitch<N>();
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x;
  const int f = tid / TT, j = tid - f * TT;
  Cx<T>* const seq = reinterpret_cast<Cx<T>*>(smem_raw) + f * NP;
  const int64_t row = (int64_t)blockIdx.x * F + f;
  Cx<T>* const g = psi + row * N;
  Cx<T> v[8];
  if constexpr (MODE == ROW_FIRST) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      v[m] = g[j + m * TT];
      dens[row * N + j + m * TT] = v[m].re * v[m].re + v[m].im * v[m].im;
    }
  } else {
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) v[sl] = g[reg_freq<N>(j, sl)];
    reg_fft_dit<T, N, +1>(v, seq, tw, j);  // unnormalised inverse: 1/(nx ny) sits in the column multiplier
    if constexpr (MODE == ROW_LAST) {
#pragma unroll
      for (int m = 0; m < 8; ++m) g[j + m * TT] = v[m];
      return;
    }
    const int env = (int)(row / nx);
    if constexpr (MODE == ROW_MID) {
      const T kk = ep[env].gpe_k;
      const T* vrow = pot ? pot + (int64_t)env * pot_env_stride + (row - (int64_t)env * nx) * N : nullptr;
      double acc = 0.0;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int n = j + m * TT;
        const T w = (vrow ? vrow[n] : T(0)) + kk * dens[row * N + n];
        // exp(-i w (tr + i ti)) = exp(w ti) (cos(w tr) - i sin(w tr))
        T sn, cs;
        sincos_t<T>(w * tr, &sn, &cs);
        const T mag = (ti == T(0)) ? T(1) : exp_t<T>(w * ti);
        v[m] = cmul(v[m], Cx<T>{mag * cs, -mag * sn});
        acc += (double)v[m].re * (double)v[m].re + (double)v[m].im * (double)v[m].im;
      }
      __shared__ double red[4];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
      if ((tid & 63) == 0) red[tid >> 6] = acc;
      __syncthreads();
      if (tid == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    } else {  // ROW_JOIN: the real-space field is psi0 of the next step
#pragma unroll
      for (int m = 0; m < 8; ++m) dens[row * N + j + m * TT] = v[m].re * v[m].re + v[m].im * v[m].im;
    }
  }
  reg_fft_dif<T, N, -1>(v, seq, tw, j);
#pragma unroll
  for (int sl = 0; sl < 8; ++sl) g[reg_freq<N>(j, sl)] = v[sl];
}

// F consecutive columns (FFT along the strided axis) per workgroup: FFT_x -> * mult (* scale) -> IFFT_x
template <typename T, int N, int F, bool SCALED>
__global__ __launch_bounds__(1024) void strang_col_kernel(Cx<T>* __restrict__ psi, const Cx<T>* __restrict__ mult,
                                                         const Cx<T>* __restrict__ tw_g, int ny,
                                                         const double* __restrict__ partial, int blocks_per_env,
                                                         double dx2) {
  constexpr int NP = fft_lds_pitch<N>();
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  Cx<T>* const s = reinterpret_cast<Cx<T>*>(smem_raw);
  Cx<T>* const tw = s + F * NP;
  const int tid = threadIdx.x;
  const int env = blockIdx.y;
  const int col0 = blockIdx.x * F;
  Cx<T>* const g = psi + (int64_t)env * N * ny + col0;
  for (int n = tid; n < N; n += (int)blockDim.x) tw[n] = tw_g[n];
  for (int idx = tid; idx < F * N; idx += (int)blockDim.x) {
    const int i = idx / F, c = idx - i * F;
    s[c * NP + fft_lds_addr(i)] = g[(int64_t)i * ny + c];
  }
  __syncthreads();
  fft_dif<T, N, F, NP, -1>(s, tw, tid);
  T scale = T(1);
  if constexpr (SCALED) {
    // sum of the row pass's partial norms: first wave, fixed lane assignment + fixed shuffle tree
    // (the same order in every workgroup of the environment -> one scale per environment, bitwise)
    __shared__ double scale_sh;
    if (tid < 64) {
      double sum = 0.0;
      for (int q = tid; q < blocks_per_env; q += 64) sum += partial[(int64_t)env * blocks_per_env + q];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
      if (tid == 0) scale_sh = 1.0 / sqrt(sum * dx2);
    }
    __syncthreads();
    scale = (T)scale_sh;
  }
  for (int idx = tid; idx < F * N; idx += (int)blockDim.x) {
    const int p = idx / F, c = idx - p * F;
    const int k = fft_rev<N>(p);
    Cx<T> m = mult[(int64_t)k * ny + col0 + c];
    m.re *= scale;
    m.im *= scale;
    s[c * NP + fft_lds_addr(p)] = cmul(s[c * NP + fft_lds_addr(p)], m);
  }
  __syncthreads();
  fft_dit<T, N, F, NP, +1>(s, tw, tid);
  for (int idx = tid; idx < F * N; idx += (int)blockDim.x) {
    const int i = idx / F, c = idx - i * F;
    g[(int64_t)i * ny + c] = s[c * NP + fft_lds_addr(i)];
  }
}

// The column pass with the transforms in registers: C adjacent columns x N/8 threads per workgroup,
// the column index fastest across lanes (C x 8 bytes contiguous per row: 128-byte segments at C = 16),
// so the threads of one column sit in different waves and the exchanges use workgroup barriers.
template <typename T, int N, int C, bool SCALED>
__global__ __launch_bounds__(C* N / 8) void strang_col_reg_kernel(Cx<T>* __restrict__ psi,
                                                                  const Cx<T>* __restrict__ mult,
                                                                  const Cx<T>* __restrict__ tw, int ny,
                                                                  const double* __restrict__ partial,
                                                                  int blocks_per_env, double dx2) {
  constexpr int TT = N / 8, NP = fft_lds_pitch<N>();
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x;
  const int j = tid / C, c = tid - j * C;
  Cx<T>* const seq = reinterpret_cast<Cx<T>*>(smem_raw) + c * NP;
  const int env = blockIdx.y;
  const int col = blockIdx.x * C + c;
  Cx<T>* const g = psi + (int64_t)env * N * ny + col;
  Cx<T> v[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) v[m] = g[(int64_t)(j + m * TT) * ny];
  reg_fft_dif<T, N, -1, false>(v, seq, tw, j);
  T scale = T(1);
  if constexpr (SCALED) {
    __shared__ double scale_sh;
    if (tid < 64) {
      double sum = 0.0;
      for (int q = tid; q < blocks_per_env; q += 64) sum += partial[(int64_t)env * blocks_per_env + q];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
      if (tid == 0) scale_sh = 1.0 / sqrt(sum * dx2);
    }
    __syncthreads();
    scale = (T)scale_sh;
  }
#pragma unroll
  for (int sl = 0; sl < 8; ++sl) {
    Cx<T> m = mult[(int64_t)reg_freq<N>(j, sl) * ny + col];
    m.re *= scale;
    m.im *= scale;
    v[sl] = cmul(v[sl], m);
  }
  reg_fft_dit<T, N, +1, false>(v, seq, tw, j);
#pragma unroll
  for (int m = 0; m < 8; ++m) g[(int64_t)(j + m * TT) * ny] = v[m];
}

// sequences per 256-thread workgroup.  Measured on 128 x 512^2 c64: 16 -> 1036, 8 -> 1210, 4 -> 1312
// env-steps/s: the pass is latency-bound (load -> 3 barrier-separated stages -> store), so more,
// smaller workgroups per CU win over wider HBM segments on the column pass (neighbouring column
// blocks run back to back and share their 128-byte lines through L2).
#ifndef PDEOPT_FFT_ROWS
#define PDEOPT_FFT_ROWS 4
#endif
template <typename T>
constexpr int rows_per_block() { return PDEOPT_FFT_ROWS; }
// the column pass wants 128-byte segments (16 fp32 / 8 fp64 complex columns) and hides its strided
// loads with many waves per workgroup instead of many workgroups
#ifndef PDEOPT_FFT_COL_THREADS
#define PDEOPT_FFT_COL_THREADS 1024
#endif
#ifndef PDEOPT_FFT_COLS
#define PDEOPT_FFT_COLS 16
#endif
template <typename T>
constexpr int cols_per_block() { return sizeof(T) == 4 ? PDEOPT_FFT_COLS : PDEOPT_FFT_COLS / 2; }
template <typename T, int N>
size_t col_lds_bytes() { return ((size_t)cols_per_block<T>() * fft_lds_pitch<N>() + N) * sizeof(Cx<T>); }

template <typename T, int N>
size_t lds_bytes() { return ((size_t)rows_per_block<T>() * fft_lds_pitch<N>() + N) * sizeof(Cx<T>); }

template <typename K>
int allow_lds(pdeopt_ctx* ctx, K kernel, size_t bytes) {
  if (bytes > 48 * 1024)
    PDEOPT_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return PDEOPT_OK;
}

// rows per workgroup of the row pass that is in use for this N (the column pass sums that many partials)
template <typename T, int N>
constexpr bool row_in_registers() { return N <= 512; }
template <typename T, int N>
constexpr int row_pass_rows() { return row_in_registers<T, N>() ? 256 / (N / 8) : rows_per_block<T>(); }

template <typename T>
int row_pass_rows_rt(int ny) { return ny <= 512 ? 256 / (ny / 8) : rows_per_block<T>(); }

template <typename T, int N, int MODE>
int launch_row(pdeopt_ctx* ctx, StrangFused& sf, double tr, double ti) {
  const pdeopt_problem& p = ctx->prob;
  const AuxField& pot = ctx->aux[PDEOPT_AUX_GPE_POTENTIAL];
  if constexpr (row_in_registers<T, N>()) {
    constexpr int F = row_pass_rows<T, N>();
    const size_t lds = (size_t)F * fft_lds_pitch<N>() * sizeof(Cx<T>);
    auto kern = strang_row_reg_kernel<T, N, MODE>;
    int rc = allow_lds(ctx, kern, lds);
    if (rc) return rc;
    const int blocks = (int)((int64_t)p.batch * p.nx / F);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, ctx->stream, (Cx<T>*)ctx->Y, (T*)sf.dens,
                       (const T*)pot.dev, pot.per_env ? (int64_t)p.nx * p.ny : (int64_t)0,
                       (const EnvParams<T>*)ctx->env_params_dev, (const Cx<T>*)sf.tw_y, (T)tr, (T)ti, p.nx,
                       sf.partial);
    PDEOPT_HIP_CHECK(ctx, hipGetLastError());
    return PDEOPT_OK;
  }
  constexpr int F = rows_per_block<T>();
  const size_t lds = lds_bytes<T, N>();
  auto kern = strang_row_kernel<T, N, F, MODE>;
  int rc = allow_lds(ctx, kern, lds);
  if (rc) return rc;
  const int blocks = (int)((int64_t)p.batch * p.nx / F);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, ctx->stream, (Cx<T>*)ctx->Y, (T*)sf.dens,
                     (const T*)pot.dev, pot.per_env ? (int64_t)p.nx * p.ny : (int64_t)0,
                     (const EnvParams<T>*)ctx->env_params_dev, (const Cx<T>*)sf.tw_y, (T)tr, (T)ti, p.nx,
                     sf.partial);
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  return PDEOPT_OK;
}

template <typename T, int N, bool SCALED>
int launch_col(pdeopt_ctx* ctx, StrangFused& sf) {
  const pdeopt_problem& p = ctx->prob;
  if constexpr (N <= 512) {
    constexpr int C = cols_per_block<T>();
    const size_t lds = (size_t)C * fft_lds_pitch<N>() * sizeof(Cx<T>);
    auto kern = strang_col_reg_kernel<T, N, C, SCALED>;
    int rc = allow_lds(ctx, kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(p.ny / C, p.batch), dim3(C * N / 8), lds, ctx->stream, (Cx<T>*)ctx->Y,
                       (const Cx<T>*)sf.mult, (const Cx<T>*)sf.tw_x, p.ny, (const double*)sf.partial,
                       p.nx / row_pass_rows_rt<T>(p.ny), ctx->strang_dx * ctx->strang_dx);
    PDEOPT_HIP_CHECK(ctx, hipGetLastError());
    return PDEOPT_OK;
  }
  constexpr int F = cols_per_block<T>();
  const size_t lds = col_lds_bytes<T, N>();
  auto kern = strang_col_kernel<T, N, F, SCALED>;
  int rc = allow_lds(ctx, kern, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3(p.ny / F, p.batch), dim3(PDEOPT_FFT_COL_THREADS), lds, ctx->stream, (Cx<T>*)ctx->Y,
                     (const Cx<T>*)sf.mult, (const Cx<T>*)sf.tw_x, p.ny, (const double*)sf.partial, p.nx / row_pass_rows_rt<T>(p.ny),
                     ctx->strang_dx * ctx->strang_dx);
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  return PDEOPT_OK;
}

// run-time size -> template instantiation
#define PDEOPT_FFT_SIZES(X) X(64) X(128) X(256) X(512) X(1024)

template <typename T, int MODE>
int row_dispatch(pdeopt_ctx* ctx, StrangFused& sf, double tr, double ti) {
  switch (ctx->prob.ny) {
#define X(NN) case NN: return launch_row<T, NN, MODE>(ctx, sf, tr, ti);
    PDEOPT_FFT_SIZES(X)
#undef X
    default: return fail(ctx, PDEOPT_EINVAL, "fused Strang: ny=%d is not covered", ctx->prob.ny);
  }
}
template <typename T, bool SCALED>
int col_dispatch(pdeopt_ctx* ctx, StrangFused& sf) {
  switch (ctx->prob.nx) {
#define X(NN) case NN: return launch_col<T, NN, SCALED>(ctx, sf);
    PDEOPT_FFT_SIZES(X)
#undef X
    default: return fail(ctx, PDEOPT_EINVAL, "fused Strang: nx=%d is not covered", ctx->prob.nx);
  }
}

bool size_ok(int n, bool f64) { return n == 64 || n == 128 || n == 256 || n == 512 || (n == 1024 && !f64); }

template <typename T>
int upload_table(pdeopt_ctx* ctx, void** dev, int n) {
  std::vector<Cx<T>> h((size_t)n);
  for (int k = 0; k < n; ++k) {
    const double a = -2.0 * M_PI * (double)k / (double)n;
    h[k] = Cx<T>{(T)std::cos(a), (T)std::sin(a)};
  }
  int rc = ensure_buffer(ctx, dev, h.size() * sizeof(Cx<T>));
  if (rc) return rc;
  PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(*dev, h.data(), h.size() * sizeof(Cx<T>), hipMemcpyHostToDevice, ctx->stream));
  PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return PDEOPT_OK;
}

template <typename T>
int strang_fused_t(pdeopt_ctx* ctx, double dt, int64_t n) {
  if (!ctx->strang_fused) ctx->strang_fused = new StrangFused();
  StrangFused& sf = *ctx->strang_fused;
  const pdeopt_problem& p = ctx->prob;
  const int64_t cells = (int64_t)p.nx * p.ny;
  int rc;
  if (!sf.tw_x) {
    if ((rc = upload_table<T>(ctx, &sf.tw_x, p.nx))) return rc;
    if ((rc = upload_table<T>(ctx, &sf.tw_y, p.ny))) return rc;
    if ((rc = ensure_buffer(ctx, &sf.dens, (size_t)cells * p.batch * sizeof(T)))) return rc;
    if ((rc = ensure_buffer(ctx, (void**)&sf.partial, sizeof(double) * (size_t)p.batch * p.nx))) return rc;
  }
  const std::complex<double> tau = dt * std::complex<double>(ctx->ts_re, ctx->ts_im);
  if (!sf.mult_valid || sf.key_dt != dt || sf.key_tr != ctx->ts_re || sf.key_ti != ctx->ts_im) {
    // E / (nx ny), E = exp(A_term tau / 2)  (solvers.py:105); A_term is caller data (aux field)
    const AuxField& a = ctx->aux[PDEOPT_AUX_GPE_A_TERM];
    std::vector<std::complex<double>> at((size_t)cells);
    if (p.dtype == PDEOPT_F32) {
      std::vector<float> h((size_t)cells * 2);
      PDEOPT_HIP_CHECK(ctx, hipMemcpy(h.data(), a.dev, (size_t)cells * 8, hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < cells; ++i) at[i] = {h[2 * i], h[2 * i + 1]};
    } else {
      PDEOPT_HIP_CHECK(ctx, hipMemcpy(at.data(), a.dev, (size_t)cells * 16, hipMemcpyDeviceToHost));
    }
    std::vector<Cx<T>> m((size_t)cells);
    const double inv_n = 1.0 / (double)cells;
    for (int64_t i = 0; i < cells; ++i) {
      const std::complex<double> e = std::exp(at[i] * 0.5 * tau) * inv_n;
      m[i] = Cx<T>{(T)e.real(), (T)e.imag()};
    }
    if ((rc = ensure_buffer(ctx, &sf.mult, (size_t)cells * sizeof(Cx<T>)))) return rc;
    PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(sf.mult, m.data(), (size_t)cells * sizeof(Cx<T>), hipMemcpyHostToDevice, ctx->stream));
    PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    sf.mult_valid = true;
    sf.key_dt = dt;
    sf.key_tr = ctx->ts_re;
    sf.key_ti = ctx->ts_im;
  }
  const double tr = tau.real(), ti = tau.imag();
  if ((rc = row_dispatch<T, ROW_FIRST>(ctx, sf, tr, ti))) return rc;
  for (int64_t s = 0; s < n; ++s) {
    if ((rc = col_dispatch<T, false>(ctx, sf))) return rc;
    if ((rc = row_dispatch<T, ROW_MID>(ctx, sf, tr, ti))) return rc;
    if ((rc = col_dispatch<T, true>(ctx, sf))) return rc;
    if (s + 1 < n)
      rc = row_dispatch<T, ROW_JOIN>(ctx, sf, tr, ti);
    else
      rc = row_dispatch<T, ROW_LAST>(ctx, sf, tr, ti);
    if (rc) return rc;
  }
  ctx->last_kernel = "strang_fused_lds_fft";
  return PDEOPT_OK;
}

// ---------------------------------------------------------------------------------------------
// IMEX (SemiImplicitFourierSpectral.step, solvers.py:56-63) on the same LDS transforms:
//   k = rhs(y)  (stencil kernel)  ->  row pass: FFT_y of the real k  ->  column pass:
//   FFT_x -> * 1/((1 + A dt symbol) nx ny) -> IFFT_x  ->  row pass: IFFT_y, y += dt Re(.)
// Full complex transforms of the real field like the reference (cahn_hilliard.py:72-73), so any
// complex fourier_symbol is honoured as is; 4 kernels per substep.
// ---------------------------------------------------------------------------------------------

template <typename T, int N, int F>
__global__ __launch_bounds__(256) void imex_row_fwd_kernel(const T* __restrict__ k, Cx<T>* __restrict__ c,
                                                           const Cx<T>* __restrict__ tw_g) {
  constexpr int NP = fft_lds_pitch<N>();
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  Cx<T>* const s = reinterpret_cast<Cx<T>*>(smem_raw);
  Cx<T>* const tw = s + F * NP;
  const int tid = threadIdx.x;
  const int64_t o = (int64_t)blockIdx.x * F * N;
  for (int n = tid; n < N; n += 256) tw[n] = tw_g[n];
  for (int idx = tid; idx < F * N; idx += 256) {
    const int f = idx / N, j = idx - f * N;
    s[f * NP + fft_lds_addr(j)] = Cx<T>{k[o + idx], T(0)};
  }
  __syncthreads();
  fft_dif<T, N, F, NP, -1>(s, tw, tid);
  for (int idx = tid; idx < F * N; idx += 256) {
    const int f = idx / N, j = idx - f * N;
    c[o + idx] = s[f * NP + fft_lds_addr(fft_pos_of<N>(j))];
  }
}

template <typename T, int N, int F>
__global__ __launch_bounds__(256) void imex_row_inv_kernel(const Cx<T>* __restrict__ c, T* __restrict__ y,
                                                           const Cx<T>* __restrict__ tw_g, T dt) {
  constexpr int NP = fft_lds_pitch<N>();
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  Cx<T>* const s = reinterpret_cast<Cx<T>*>(smem_raw);
  Cx<T>* const tw = s + F * NP;
  const int tid = threadIdx.x;
  const int64_t o = (int64_t)blockIdx.x * F * N;
  for (int n = tid; n < N; n += 256) tw[n] = tw_g[n];
  for (int idx = tid; idx < F * N; idx += 256) {
    const int f = idx / N, j = idx - f * N;
    s[f * NP + fft_lds_addr(fft_pos_of<N>(j))] = c[o + idx];
  }
  __syncthreads();
  fft_dit<T, N, F, NP, +1>(s, tw, tid);
  for (int idx = tid; idx < F * N; idx += 256) {
    const int f = idx / N, j = idx - f * N;
    y[o + idx] += dt * s[f * NP + fft_lds_addr(j)].re;  // y1 = y0 + dt Re ifft(...)   solvers.py:63
  }
}

template <typename T, int N>
int imex_rows(pdeopt_ctx* ctx, StrangFused& sf, bool forward, double dt) {
  constexpr int F = rows_per_block<T>();
  const pdeopt_problem& p = ctx->prob;
  const size_t lds = lds_bytes<T, N>();
  const int blocks = (int)((int64_t)p.batch * p.nx / F);
  if (forward) {
    auto kern = imex_row_fwd_kernel<T, N, F>;
    int rc = allow_lds(ctx, kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, ctx->stream, (const T*)ctx->TA, (Cx<T>*)sf.cwork,
                       (const Cx<T>*)sf.tw_y);
  } else {
    auto kern = imex_row_inv_kernel<T, N, F>;
    int rc = allow_lds(ctx, kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, ctx->stream, (const Cx<T>*)sf.cwork, (T*)ctx->Y,
                       (const Cx<T>*)sf.tw_y, (T)dt);
  }
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  return PDEOPT_OK;
}

template <typename T, int N>
int imex_cols(pdeopt_ctx* ctx, StrangFused& sf) {
  constexpr int F = cols_per_block<T>();
  const pdeopt_problem& p = ctx->prob;
  const size_t lds = col_lds_bytes<T, N>();
  auto kern = strang_col_kernel<T, N, F, false>;  // FFT_x -> * multiplier -> IFFT_x, in place
  int rc = allow_lds(ctx, kern, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3(p.ny / F, p.batch), dim3(PDEOPT_FFT_COL_THREADS), lds, ctx->stream,
                     (Cx<T>*)sf.cwork, (const Cx<T>*)sf.imex_mult, (const Cx<T>*)sf.tw_x, p.ny,
                     (const double*)nullptr, 0, 1.0);
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  return PDEOPT_OK;
}

template <typename T>
int imex_fused_t(pdeopt_ctx* ctx, double dt, int64_t n) {
  if (!ctx->strang_fused) ctx->strang_fused = new StrangFused();
  StrangFused& sf = *ctx->strang_fused;
  const pdeopt_problem& p = ctx->prob;
  const int64_t cells = (int64_t)p.nx * p.ny;
  int rc;
  if (!sf.tw_x) {
    if ((rc = upload_table<T>(ctx, &sf.tw_x, p.nx))) return rc;
    if ((rc = upload_table<T>(ctx, &sf.tw_y, p.ny))) return rc;
  }
  if ((rc = ensure_buffer(ctx, &ctx->TA, ctx->total_bytes))) return rc;
  if ((rc = ensure_buffer(ctx, &sf.cwork, (size_t)cells * p.batch * sizeof(Cx<T>)))) return rc;
  if (!sf.imex_valid || sf.imex_dt != dt || sf.imex_A != ctx->imex_A) {
    // 1 / ((1 + A dt fourier_symbol) nx ny): solvers.py:62-63 with the 1/N of the inverse folded in
    const AuxField& a = ctx->aux[PDEOPT_AUX_IMEX_SYMBOL];
    std::vector<std::complex<double>> sym((size_t)cells);
    if (p.dtype == PDEOPT_F32) {
      std::vector<float> h((size_t)cells * 2);
      PDEOPT_HIP_CHECK(ctx, hipMemcpy(h.data(), a.dev, (size_t)cells * 8, hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < cells; ++i) sym[i] = {h[2 * i], h[2 * i + 1]};
    } else {
      PDEOPT_HIP_CHECK(ctx, hipMemcpy(sym.data(), a.dev, (size_t)cells * 16, hipMemcpyDeviceToHost));
    }
    std::vector<Cx<T>> m((size_t)cells);
    const double inv_n = 1.0 / (double)cells;
    for (int64_t i = 0; i < cells; ++i) {
      const std::complex<double> v = inv_n / (1.0 + ctx->imex_A * dt * sym[i]);
      m[i] = Cx<T>{(T)v.real(), (T)v.imag()};
    }
    if ((rc = ensure_buffer(ctx, &sf.imex_mult, (size_t)cells * sizeof(Cx<T>)))) return rc;
    PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(sf.imex_mult, m.data(), (size_t)cells * sizeof(Cx<T>), hipMemcpyHostToDevice, ctx->stream));
    PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    sf.imex_valid = true;
    sf.imex_dt = dt;
    sf.imex_A = ctx->imex_A;
  }
  for (int64_t s = 0; s < n; ++s) {
    if ((rc = launch_rhs(ctx, ctx->Y, ctx->TA, 0.0))) return rc;
    switch (p.ny) {
#define X(NN) case NN: rc = imex_rows<T, NN>(ctx, sf, true, dt); break;
      PDEOPT_FFT_SIZES(X)
#undef X
    }
    if (rc) return rc;
    switch (p.nx) {
#define X(NN) case NN: rc = imex_cols<T, NN>(ctx, sf); break;
      PDEOPT_FFT_SIZES(X)
#undef X
    }
    if (rc) return rc;
    switch (p.ny) {
#define X(NN) case NN: rc = imex_rows<T, NN>(ctx, sf, false, dt); break;
      PDEOPT_FFT_SIZES(X)
#undef X
    }
    if (rc) return rc;
  }
  ctx->last_kernel += "+imex_fused_lds_fft";
  return PDEOPT_OK;
}

}  // namespace

bool imex_fused_supported(const pdeopt_ctx* ctx) {
  const pdeopt_problem& p = ctx->prob;
  const bool f64 = p.dtype == PDEOPT_F64;
  // Opt-in (PDEOPT_OPT_IMEX_LDS_FFT): measured on CH 1024^2 x 32 the full-complex LDS transforms run
  // 408 env-steps/s against 496 for rocFFT's real<->hermitian plans, which move half the bytes.
  if (ctx->opt_imex_lds_fft <= 0) return false;
  if (!size_ok(p.nx, f64) || !size_ok(p.ny, f64)) return false;
  return !ctx->aux[PDEOPT_AUX_IMEX_SYMBOL].per_env;
}

int advance_imex_fused(pdeopt_ctx* ctx, double dt, int64_t n) {
  return ctx->prob.dtype == PDEOPT_F32 ? imex_fused_t<float>(ctx, dt, n) : imex_fused_t<double>(ctx, dt, n);
}

bool strang_fused_supported(const pdeopt_ctx* ctx) {
  const pdeopt_problem& p = ctx->prob;
  const bool f64 = p.dtype == PDEOPT_F64;
  if (ctx->opt_kernel_path == 1) return false;  // "generic" path = the rocFFT pipeline
  if (!size_ok(p.nx, f64) || !size_ok(p.ny, f64)) return false;
  if (ctx->aux[PDEOPT_AUX_GPE_A_TERM].per_env) return false;
  return true;
}

int advance_strang_fused(pdeopt_ctx* ctx, double dt, int64_t n) {
  return ctx->prob.dtype == PDEOPT_F32 ? strang_fused_t<float>(ctx, dt, n) : strang_fused_t<double>(ctx, dt, n);
}

void strang_fused_invalidate(pdeopt_ctx* ctx) {
  if (ctx->strang_fused) {
    ctx->strang_fused->mult_valid = false;
    ctx->strang_fused->imex_valid = false;
  }
}

void strang_fused_destroy(pdeopt_ctx* ctx) {
  StrangFused* sf = ctx->strang_fused;
  if (!sf) return;
  void* bufs[] = {sf->tw_x, sf->tw_y, sf->mult, sf->dens, sf->partial, sf->cwork, sf->imex_mult};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  delete sf;
  ctx->strang_fused = nullptr;
}

}  // namespace pdeopt
