// Strang split step (and the IMEX step) with hand-written FFTs -- butterflies in registers, exchanges
// through LDS (fft_reg.hpp) -- and the pointwise operators fused into the passes.
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
#include <algorithm>
#include <cmath>
#include <complex>

#include "common.hpp"
#include "fft_lds.hpp"
#include "fft_reg.hpp"
#include "fft_reg32.hpp"

namespace pdeopt {

struct StrangFused {
  void* tw_x = nullptr;  // twiddle tables exp(-2 pi i n / N) in the problem dtype
  void* tw_y = nullptr;
  void* mult = nullptr;  // E / (nx ny), complex, laid out by col_mult_index()
  void* dens = nullptr;  // |psi0|^2, real [batch][nx][ny]
  double* partial = nullptr;
  double key_dt = NAN, key_tr = NAN, key_ti = NAN;
  bool mult_valid = false;
  // IMEX on the same transforms
  void* cwork = nullptr;      // complex work field [batch][nx][ny]
  size_t cwork_bytes = 0;
  bool imex_sym_real = true;  // the uploaded fourier_symbol has no imaginary part (per-environment scales need that)
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

// The row pass with the transforms in registers (fft_reg.hpp): N/8 threads per row, 256/(N/8) rows per
// workgroup; for N <= 512 a row lives in one wave and the pass has no s_barrier except the one of the
// norm reduction.  Spectrum side: thread j holds the frequencies j + m N/8; real-space side: the cells
// j + m N/8 -- both coalesced.
template <typename T, int N, int MODE>
__global__ __launch_bounds__(256) void strang_row_reg_kernel(Cx<T>* __restrict__ psi, T* __restrict__ dens,
                                                             const T* __restrict__ pot, int64_t pot_env_stride,
                                                             const EnvParams<T>* __restrict__ ep,
                                                             const Cx<T>* __restrict__ tw, T tr, T ti, int nx,
                                                             double* __restrict__ partial, const SpotArgs<T> spots) {
  using E = RegFft<T, N>;
  constexpr int PTS = reg_default_pts<N>(), TT = E::TT, F = 256 / TT, NP = fft_lds_pitch<N>();
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x;
  const int f = tid / TT, j = tid - f * TT;
  Cx<T>* const seq = reinterpret_cast<Cx<T>*>(smem_raw) + f * NP;
  const int64_t row = (int64_t)blockIdx.x * F + f;
  Cx<T>* const g = psi + row * N;
  Cx<T> v[PTS];
  if constexpr (MODE == ROW_FIRST) {
#pragma unroll
    for (int m = 0; m < PTS; ++m) {
      const int n = E::natural(j, m);
      v[m] = g[n];
      dens[row * N + n] = v[m].re * v[m].re + v[m].im * v[m].im;
    }
  } else {
#pragma unroll
    for (int sl = 0; sl < PTS; ++sl) v[sl] = g[E::freq(j, sl)];
    E::template dit<+1>(v, seq, tw, j);  // unnormalised inverse: 1/(nx ny) sits in the column multiplier
    if constexpr (MODE == ROW_LAST) {
#pragma unroll
      for (int m = 0; m < PTS; ++m) g[E::natural(j, m)] = v[m];
      return;
    }
    const int env = (int)(row / nx);
    if constexpr (MODE == ROW_MID) {
      const T kk = ep[env].gpe_k;
      const int ix = (int)(row - (int64_t)env * nx);
      const T* vrow = pot ? pot + (int64_t)env * pot_env_stride + (int64_t)ix * N : nullptr;
      const T xs = spots.x_first + T(ix) * spots.hx;
      // |psi|^2 of the thread's points in the working precision (the reference sums in it too,
      // solvers.py:111), the sums across threads / workgroups / the column pass in fp64
      T accp = T(0);
#pragma unroll
      for (int m = 0; m < PTS; ++m) {
        const int n = E::natural(j, m);
        T w = (vrow ? vrow[n] : T(0)) + kk * dens[row * N + n];
        if (spots.n) w += spots_value<T>(spots, env, xs, spots.y_first + T(n) * spots.hy);  // lights(t0, x, y)
        // exp(-i w (tr + i ti)) = exp(w ti) (cos(w tr) - i sin(w tr))
        T sn, cs;
        sincos_t<T>(w * tr, &sn, &cs);
        const T mag = (ti == T(0)) ? T(1) : exp_t<T>(w * ti);
        v[m] = cmul(v[m], Cx<T>{mag * cs, -mag * sn});
        accp += v[m].re * v[m].re + v[m].im * v[m].im;
      }
      double acc = (double)accp;
      __shared__ double red[4];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
      if ((tid & 63) == 0) red[tid >> 6] = acc;
      __syncthreads();
      if (tid == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    } else {  // ROW_JOIN: the real-space field is psi0 of the next step
#pragma unroll
      for (int m = 0; m < PTS; ++m) dens[row * N + E::natural(j, m)] = v[m].re * v[m].re + v[m].im * v[m].im;
    }
  }
  E::template dif<-1>(v, seq, tw, j);
#pragma unroll
  for (int sl = 0; sl < PTS; ++sl) g[E::freq(j, sl)] = v[sl];
}

// The column pass with the transforms in registers: C adjacent columns x N/8 threads per workgroup,
// the column index fastest across lanes on the global side (C x 8 bytes contiguous per row: 128-byte
// segments at C = 16), so there the threads of one column sit in different waves.
//
// Up to N = 512 only stage 0 of a transform runs in that layout; the exchange after it hands every
// column to ONE wave (RegFft::dif_split / dit_split) and the remaining stages need no barrier: 2 per
// pass instead of 8.  Same-box: the 512^2 pass drops from 77 to ~40 us, which is what the same kernel
// takes with the transforms removed (pure traffic, 6.7 TB/s out of the Infinity Cache); split step
// x 128 environments 2498 -> 2913 env-steps/s.  At N = 1024 (16 points per thread, one workgroup per CU)
// the wave-per-column layout measured 14 % SLOWER (50 vs 44 us: its last-stage LDS reads are 4-way
// instead of 2-way bank-conflicted and nothing overlaps them), so that size keeps the barrier form.
#ifndef PDEOPT_COL_WL
#define PDEOPT_COL_WL 1
#endif
#ifndef PDEOPT_COL_WL_MAX
#define PDEOPT_COL_WL_MAX 512
#endif
constexpr bool col_wave_local(int n) { return PDEOPT_COL_WL && n <= PDEOPT_COL_WL_MAX; }
template <typename T, int N, int C, int PTS, bool SCALED, bool PER_ENV = false>
__global__ __launch_bounds__(C* N / PTS) void strang_col_reg_kernel(Cx<T>* __restrict__ psi,
                                                                  const Cx<T>* __restrict__ mult,
                                                                  const Cx<T>* __restrict__ tw, int ny,
                                                                  const double* __restrict__ partial,
                                                                  int blocks_per_env, double dx2,
                                                                  const EnvParams<T>* __restrict__ sigma_ep = nullptr,
                                                                  T inv_n = T(0)) {
  using E = RegFft<T, N, PTS>;
  constexpr int NP = fft_lds_pitch<N>();
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x;
  const int j = tid / C, c = tid - j * C;
  Cx<T>* const seq = reinterpret_cast<Cx<T>*>(smem_raw) + c * NP;
  const int env = blockIdx.y;
#ifdef PDEOPT_COL_XCD_PAIR
  // workgroups are dealt to the 8 XCDs round-robin by linear id: hand each XCD PAIRS of adjacent column
  // blocks so that the two halves of a 128-byte line are fetched into the same L2 (narrow blocks)
  const int bx_ = blockIdx.x;
  const int cblk = (bx_ & ~15) + 2 * (bx_ & 7) + ((bx_ >> 3) & 1);
#else
  const int cblk = blockIdx.x;
#endif
  // uniform base pointers + 32-bit per-thread offsets: one VGPR per address instead of a 64-bit pair
  // (the 16-point threads of the N = 1024 pass run at the 128-VGPR limit of a 1024-thread workgroup)
  Cx<T>* const gb = psi + (int64_t)env * N * ny + cblk * C;
  Cx<T> v[PTS];
#pragma unroll
  for (int m = 0; m < PTS; ++m) v[m] = gb[E::natural(j, m) * ny + c];
  // 1 / sqrt(sum |psi|^2 dx^2) from the row pass's partial sums; published by the barriers of the transform
  __shared__ double scale_sh;
  if constexpr (SCALED) {
    if (tid < 64) {
      double sum = 0.0;
      for (int q = tid; q < blocks_per_env; q += 64) sum += partial[(int64_t)env * blocks_per_env + q];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
      if (tid == 0) scale_sh = 1.0 / sqrt(sum * dx2);
    }
  }
  constexpr bool WL = col_wave_local(N);
  // WL: stage 0 in the load layout, everything after it with a column per wave
  const int ji = tid % E::TT;
  Cx<T>* const seqi = reinterpret_cast<Cx<T>*>(smem_raw) + (tid / E::TT) * NP;
  const Cx<T>* const mb = WL ? mult + (int64_t)(cblk * C + tid / E::TT) * N  // transposed: [ny][nx]
                             : mult + cblk * C;
  if constexpr (WL)
    E::template dif_split<-1>(v, seq, j, seqi, ji, tw);
  else
    E::template dif<-1, false>(v, seq, tw, j);
  T scale = T(1);
  if constexpr (SCALED) scale = (T)scale_sh;
  // IMEX with a per-environment implicit operator (one environment per complex field): the stored multiplier is
  // M0 = 1 / (N (1 + A dt symbol)), real; this environment's is 1 / (N (1 + sigma (1 / (N M0) - 1)))
  // (a template parameter: as a run-time branch its divisions cost every instantiation 4-5 VGPRs, and the 512^2
  // Strang pass -- two 1024-thread workgroups per CU -- lives at the 64-VGPR limit: 54 -> 71 us)
  const T sigma = PER_ENV ? sigma_ep[env].imex_scale : T(1);
#pragma unroll
  for (int sl = 0; sl < PTS; ++sl) {
    Cx<T> m = WL ? mb[E::freq(ji, sl)] : mb[E::freq(j, sl) * ny + c];
    if constexpr (PER_ENV) {
      m.re = inv_n / (T(1) + sigma * (inv_n / m.re - T(1)));
      m.im = T(0);
    }
    m.re *= scale;
    m.im *= scale;
    v[sl] = cmul(v[sl], m);
    if constexpr (PTS > 8) {
      if (sl % 4 == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (WL)
    E::template dit_split<+1>(v, seqi, ji, seq, j, tw);
  else
    E::template dit<+1, false>(v, seq, tw, j);
#pragma unroll
  for (int m = 0; m < PTS; ++m) gb[E::natural(j, m) * ny + c] = v[m];
}

// N = 1024 fp32 column pass on the 32 x 32 plan (fft_reg32.hpp): thread (column c, j) owns rows j + 32 m of its
// column -- global loads / stores as above (column index fastest across lanes: 128-byte segments), but ONE
// exchange per transform instead of two, 512-thread workgroups with 68 KB of LDS (two per CU, so one's barriers
// overlap the other's arithmetic) instead of one of 1024 threads with 148 KB.
#ifndef PDEOPT_COL32
#define PDEOPT_COL32 1
#endif
template <typename T, int N>
constexpr bool col_use32() { return PDEOPT_COL32 && sizeof(T) == 4 && N == 1024; }
// LDS geometry of the column pass.  The LDS has 32 banks of 4 bytes and serves 32 lanes' dwords per clock; a
// half-wave here is 16 columns x 2 rows.  With 1060 (= 4 mod 32) floats between the columns' images -- the first
// cut, laid out for 64 banks -- columns c and c + 8 met on one bank: SQ_LDS_BANK_CONFLICT = 50 % of the LDS cycles
// (profiles/pmc_r02.json).  1058 = 2 mod 32 with the odd row pitch 33 puts the 32 accesses of a clock on 32 banks
// in both directions of the transpose (writes 2c + j, reads 2c + 33j): conflicts 0.000, LDS cycles halved,
// +1.0 % on ch_imex_1024_f32 (1438 -> 1452 env-steps/s same-box).
#ifndef PDEOPT_COL32_ROW_PITCH
#define PDEOPT_COL32_ROW_PITCH 33
#endif
#ifndef PDEOPT_COL32_SEQ_PITCH
#define PDEOPT_COL32_SEQ_PITCH 1058
#endif
constexpr int kCol32Pitch = PDEOPT_COL32_SEQ_PITCH;
static_assert(kCol32Pitch >= 32 * PDEOPT_COL32_ROW_PITCH, "a column's image must fit its slot");

// complex fp32 element through a buffer descriptor: uniform descriptor + SGPR row offset + ONE per-thread VGPR
// offset for all 32 rows (plain global loads keep a 64-bit address pair per row alive: 230 VGPRs)
typedef unsigned pdeopt_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ Cx<float> buf_load_cx(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  const pdeopt_v2u d = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
  const unsigned x = d.x, y = d.y;  // (__builtin_bit_cast of a vector ELEMENT reads element 0 on this compiler)
  return Cx<float>{__uint_as_float(x), __uint_as_float(y)};
}
__device__ __forceinline__ void buf_store_cx(Cx<float> v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  const pdeopt_v2u d = {__float_as_uint(v.re), __float_as_uint(v.im)};
  __builtin_amdgcn_raw_buffer_store_b64(d, r, voff, soff, 0);
}

template <typename T, int C, bool SCALED, bool PER_ENV>
__global__ __launch_bounds__(C * 32, 4) void strang_col32_kernel(Cx<T>* __restrict__ psi, const Cx<T>* __restrict__ mult,
                                                                 const Cx<T>* __restrict__ tw, int ny,
                                                                 const double* __restrict__ partial, int blocks_per_env,
                                                                 double dx2, const EnvParams<T>* __restrict__ sigma_ep,
                                                                 T inv_n) {
  static_assert(sizeof(T) == 4, "fp32 only");
  using E = RegFft32x32<T, PDEOPT_COL32_ROW_PITCH>;
  constexpr int N = 1024, PTS = 32;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x;
  const int j = tid / C, c = tid - j * C;
  T* const seq = reinterpret_cast<T*>(smem_raw) + c * kCol32Pitch;
  const int env = blockIdx.y;
  const int cblk = blockIdx.x;
  // descriptors over the rest of this environment's field / of the multiplier from the block's first column on
#ifdef PDEOPT_COL32_NOMEM  // timing-only build: the range check drops every load and store of the pass
  const int bytes = 0;
#else
  const int bytes = (N * ny - cblk * C) * (int)sizeof(Cx<T>);
#endif
  const __amdgpu_buffer_rsrc_t rpsi =
      __builtin_amdgcn_make_buffer_rsrc((void*)(psi + (int64_t)env * N * ny + cblk * C), 0, bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rmul = __builtin_amdgcn_make_buffer_rsrc((void*)(mult + cblk * C), 0, bytes, 0x00020000);
  // thread (c, j) owns rows j + 32 m (natural) / j + 32 sl (frequency): byte offset of row j + m row steps
  const int voff = (j * ny + c) * (int)sizeof(Cx<T>), rstep = 32 * ny * (int)sizeof(Cx<T>);
  Cx<T> v[PTS];
#pragma unroll
  for (int m = 0; m < PTS; ++m) v[m] = buf_load_cx(rpsi, voff, m * rstep);
  __shared__ double scale_sh;  // see strang_col_reg_kernel
  if constexpr (SCALED) {
    if (tid < 64) {
      double sum = 0.0;
      for (int q = tid; q < blocks_per_env; q += 64) sum += partial[(int64_t)env * blocks_per_env + q];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
      if (tid == 0) scale_sh = 1.0 / sqrt(sum * dx2);
    }
  }
#ifndef PDEOPT_COL32_NOFFT  // timing-only build: traffic without the transforms
  E::template transform<-1, false>(v, seq, tw, j);
#endif
  T scale = T(1);
  if constexpr (SCALED) scale = (T)scale_sh;
  const T sigma = PER_ENV ? sigma_ep[env].imex_scale : T(1);
#pragma unroll
  for (int sl = 0; sl < PTS; ++sl) {
    Cx<T> m = buf_load_cx(rmul, voff, sl * rstep);
    if constexpr (PER_ENV) {  // per-environment implicit operator, see strang_col_reg_kernel
      m.re = inv_n / (T(1) + sigma * (inv_n / m.re - T(1)));
      m.im = T(0);
    }
    m.re *= scale;
    m.im *= scale;
    v[sl] = cmul(v[sl], m);
  }
#ifndef PDEOPT_COL32_NOFFT
  E::template transform<+1, false>(v, seq, tw, j);
#endif
#pragma unroll
  for (int m = 0; m < PTS; ++m) buf_store_cx(v[m], rpsi, voff, m * rstep);
}

// Column pass geometry.  128-byte segments (16 fp32 / 8 fp64 complex columns) are needed -- 8 columns
// measured 6 % slower -- but nothing beyond: 32 columns x 16 points per thread (256-byte segments) and
// 16 x 16 (half-size workgroups) both measured within 0.5 % of 16 x 8 on 128 x 512^2 c64, and padding the
// row pitch of the IMEX work field by 128 / 576 bytes changed nothing (0 % / -4 %).  The pass runs at
// ~3 TB/s against ~5 TB/s for the row passes whatever its shape.
#ifndef PDEOPT_FFT_COLS
#define PDEOPT_FFT_COLS 16
#endif
template <typename T>
constexpr int cols_per_block() { return sizeof(T) == 4 ? PDEOPT_FFT_COLS : PDEOPT_FFT_COLS / 2; }
// points per thread of the column pass (0: 8 up to N = 512, 16 at N = 1024)
#ifndef PDEOPT_FFT_COL_PTS
#define PDEOPT_FFT_COL_PTS 0
#endif
template <typename T, int N>
constexpr int col_pts() { return PDEOPT_FFT_COL_PTS ? PDEOPT_FFT_COL_PTS : reg_default_pts<N>(); }

// where the column pass expects the multiplier of frequency (kx, ky): it reads along kx
inline size_t col_mult_index(int64_t kx, int64_t ky, int nx, int ny) {
  return col_wave_local(nx) ? (size_t)(ky * nx + kx) : (size_t)(kx * ny + ky);
}

template <typename K>
int allow_lds(pdeopt_ctx* ctx, K kernel, size_t bytes) {
  if (bytes > 48 * 1024)
    PDEOPT_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return PDEOPT_OK;
}

// rows per workgroup of the row pass (the column pass sums that many norm partials per environment)
template <typename T, int N>
constexpr int row_pass_rows() { return 256 / RegFft<T, N>::TT; }
template <typename T>
int row_pass_rows_rt(int ny) { return ny <= 512 ? 256 / (ny / 8) : 256 / (ny / 16); }

template <typename T, int N, int MODE>
int launch_row(pdeopt_ctx* ctx, StrangFused& sf, double tr, double ti, double t = 0.0) {
  const pdeopt_problem& p = ctx->prob;
  const AuxField& pot = ctx->aux[PDEOPT_AUX_GPE_POTENTIAL];
  constexpr int F = row_pass_rows<T, N>();
  const size_t lds = (size_t)F * fft_lds_pitch<N>() * sizeof(Cx<T>);
  auto kern = strang_row_reg_kernel<T, N, MODE>;
  int rc = allow_lds(ctx, kern, lds);
  if (rc) return rc;
  // environment window [win_lo, win_lo + win_n): all pointers pre-offset, the kernel sees win_n environments
  const int64_t cells = (int64_t)p.nx * p.ny, w0 = ctx->win_lo;
  const int blocks = (int)((int64_t)ctx->win_n * p.nx / F);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, ctx->stream, (Cx<T>*)ctx->Y + w0 * cells,
                     (T*)sf.dens + w0 * cells, pot.dev ? (const T*)pot.dev + (pot.per_env ? w0 * cells : 0) : nullptr,
                     pot.per_env ? cells : (int64_t)0, (const EnvParams<T>*)ctx->env_params_dev + w0,
                     (const Cx<T>*)sf.tw_y, (T)tr, (T)ti, p.nx, sf.partial + w0 * (p.nx / F),
                     MODE == ROW_MID ? make_spot_args<T>(ctx, t) : SpotArgs<T>{});
  ctx->n_stage_launches++;
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  return PDEOPT_OK;
}

template <typename T, int N, bool SCALED>
int launch_col(pdeopt_ctx* ctx, StrangFused& sf) {
  const pdeopt_problem& p = ctx->prob;
  constexpr int C = cols_per_block<T>();
  constexpr int PTS = col_pts<T, N>();
  const int64_t cells = (int64_t)p.nx * p.ny, w0 = ctx->win_lo;
  const int bpe = p.nx / row_pass_rows_rt<T>(p.ny);  // norm partials per environment
  if constexpr (col_use32<T, N>()) {
    const size_t lds = (size_t)C * kCol32Pitch * sizeof(T);
    auto kern = strang_col32_kernel<T, C, SCALED, false>;
    int rc = allow_lds(ctx, kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(p.ny / C, ctx->win_n), dim3(C * 32), lds, ctx->stream, (Cx<T>*)ctx->Y + w0 * cells,
                       (const Cx<T>*)sf.mult, (const Cx<T>*)sf.tw_x, p.ny, (const double*)sf.partial + w0 * bpe, bpe,
                       ctx->strang_dx * ctx->strang_dx, (const EnvParams<T>*)nullptr, T(0));
  } else {
    const size_t lds = (size_t)C * fft_lds_pitch<N>() * sizeof(Cx<T>);
    auto kern = strang_col_reg_kernel<T, N, C, PTS, SCALED>;
    int rc = allow_lds(ctx, kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(p.ny / C, ctx->win_n), dim3(C * N / PTS), lds, ctx->stream,
                       (Cx<T>*)ctx->Y + w0 * cells, (const Cx<T>*)sf.mult, (const Cx<T>*)sf.tw_x, p.ny,
                       (const double*)sf.partial + w0 * bpe, bpe, ctx->strang_dx * ctx->strang_dx,
                       (const EnvParams<T>*)nullptr, T(0));
  }
  ctx->n_stage_launches++;
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  return PDEOPT_OK;
}

// run-time size -> template instantiation
#define PDEOPT_FFT_SIZES(X) X(64) X(128) X(256) X(512) X(1024)

template <typename T, int MODE>
int row_dispatch(pdeopt_ctx* ctx, StrangFused& sf, double tr, double ti, double t = 0.0) {
  switch (ctx->prob.ny) {
#define X(NN) case NN: return launch_row<T, NN, MODE>(ctx, sf, tr, ti, t);
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

bool size_ok(int n, bool) { return n == 64 || n == 128 || n == 256 || n == 512 || n == 1024; }

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
int strang_fused_t(pdeopt_ctx* ctx, double t0, double dt, int64_t n) {
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
      m[col_mult_index(i / p.ny, i % p.ny, p.nx, p.ny)] = Cx<T>{(T)e.real(), (T)e.imag()};
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
  // Environments are independent: run the whole n-step pipeline group by group, a group's wavefunction +
  // density (12 B/cell at fp32) sized to stay in the 256 MiB Infinity Cache between its passes.
  // A time-dependent potential (lights(t, x, y), gross_pitaevskii.py:61) is brought to each substep's t0
  // before the pass that applies exp(b tau) (b = terms.vf(t0, y0), solvers.py:109): one sweep over the batch.
  const bool timed = has_time_aux(ctx, PDEOPT_AUX_GPE_POTENTIAL);
  int group = p.batch;
  if (timed) {
    group = p.batch;
  } else if (ctx->opt_group_envs > 0) {
    group = (int)std::min<int64_t>(ctx->opt_group_envs, p.batch);
  } else if (ctx->opt_group_envs == 0) {
    const size_t per_env = (size_t)cells * (sizeof(Cx<T>) + sizeof(T));
    const int64_t fit = std::max<int64_t>(1, (int64_t)((192ull << 20) / per_env));
    if (fit < p.batch && n > 1) {
      const int ngroups = (int)((p.batch + fit - 1) / fit);
      group = (p.batch + ngroups - 1) / ngroups;
    }
  }
  // two groups side by side on two streams, each half the size (PDEOPT_OPT_GROUP_STREAMS; stencil.hip:
  // advance_explicit has the reasoning and the measurement for the explicit integrators)
  bool side_by_side = false;
  if (!timed && ctx->opt_group_streams != 1 && n > 1 && group >= 2 &&
      (group < p.batch ? (ctx->opt_group_envs == 0 || ctx->opt_group_streams == 2)
                       : (ctx->opt_group_streams == 0 && ctx->opt_group_envs == 0 && cells * p.batch > (1 << 21)))) {
    if (ctx->opt_group_envs == 0) group = (group + 1) / 2;
    side_by_side = true;
  }
  ctx->last_groups = (p.batch + group - 1) / group;
  auto first = [&]() -> int { return row_dispatch<T, ROW_FIRST>(ctx, sf, tr, ti); };
  auto substep = [&](int64_t s) -> int {
    int r;
    if ((r = col_dispatch<T, false>(ctx, sf))) return r;
    if (timed && (r = refresh_time_aux(ctx, PDEOPT_AUX_GPE_POTENTIAL, t0 + (double)s * dt))) return r;
    if ((r = row_dispatch<T, ROW_MID>(ctx, sf, tr, ti, t0 + (double)s * dt))) return r;
    if ((r = col_dispatch<T, true>(ctx, sf))) return r;
    return s + 1 < n ? row_dispatch<T, ROW_JOIN>(ctx, sf, tr, ti) : row_dispatch<T, ROW_LAST>(ctx, sf, tr, ti);
  };
  if (side_by_side && ctx->last_groups >= 2) {
    ctx->last_group_streams = 2;
    if ((rc = ensure_stream2(ctx))) return rc;
    PDEOPT_HIP_CHECK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    PDEOPT_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
    auto on = [&](int lo, bool second, auto fn) -> int {
      ctx->win_lo = lo;
      ctx->win_n = std::min(group, p.batch - lo);
      if (second) std::swap(ctx->stream, ctx->stream2);  // the dispatch helpers take the ctx stream
      const int r = fn();
      if (second) std::swap(ctx->stream, ctx->stream2);
      return r;
    };
    for (int lo = 0; lo < p.batch && !rc; lo += 2 * group) {
      const bool two = lo + group < p.batch;
      rc = on(lo, false, first);
      if (two && !rc) rc = on(lo + group, true, first);
      for (int64_t s = 0; s < n && !rc; ++s) {
        rc = on(lo, false, [&] { return substep(s); });
        if (two && !rc) rc = on(lo + group, true, [&] { return substep(s); });
      }
    }
    const hipError_t e1 = hipEventRecord(ctx->ev_join, ctx->stream2);
    const hipError_t e2 = hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0);
    ctx->win_lo = 0;
    ctx->win_n = p.batch;
    if (rc) return rc;
    PDEOPT_HIP_CHECK(ctx, e1);
    PDEOPT_HIP_CHECK(ctx, e2);
    ctx->last_kernel = "strang_fused_lds_fft";
    return PDEOPT_OK;
  }
  for (int lo = 0; lo < p.batch && !rc; lo += group) {
    ctx->win_lo = lo;
    ctx->win_n = std::min(group, p.batch - lo);
    if ((rc = first())) break;
    for (int64_t s = 0; s < n && !rc; ++s) rc = substep(s);
  }
  ctx->win_lo = 0;
  ctx->win_n = p.batch;
  if (rc) return rc;
  ctx->last_kernel = "strang_fused_lds_fft";
  return PDEOPT_OK;
}

// ---------------------------------------------------------------------------------------------
// IMEX (SemiImplicitFourierSpectral.step, solvers.py:56-63) on the same LDS transforms:
//   k = rhs(y)  (stencil kernel)  ->  row pass: FFT_y of the real k  ->  column pass:
//   FFT_x -> * 1/((1 + A dt symbol) nx ny) -> IFFT_x  ->  row pass: IFFT_y, y += dt Re(.)
// Complex transforms carrying two real environments each (see imex_row_fwd_reg_kernel); the result is
// Re ifft2(M fft2 f) for ANY complex fourier_symbol, as the reference's `.real` (solvers.py:63); 4 kernels
// per substep.
// ---------------------------------------------------------------------------------------------

// Engine of the IMEX row passes: the 32 x 32 plan (one LDS exchange per transform, fft_reg32.hpp) at N = 1024,
// the generic register engine otherwise.  -DPDEOPT_FFT32=0 keeps the generic engine everywhere (A/B).
#ifndef PDEOPT_FFT32
#define PDEOPT_FFT32 1
#endif
template <typename T, int N>
struct ImexRowEngine {
  using type = RegFft<T, N>;
};
#if PDEOPT_FFT32
template <>
struct ImexRowEngine<float, 1024> {  // fp64 would hold 128 VGPRs of points alone: generic engine
  using type = RegFft32x32<float>;
};
#endif

// waves per SIMD the row kernels are compiled for: 4 for the 32-point engine (VGPR cap 128), no constraint else
#ifndef PDEOPT_IMEX_ROW_OCC
#define PDEOPT_IMEX_ROW_OCC 4
#endif
#define PDEOPT_IMEX_ROW_WAVES(T, N) ((PDEOPT_FFT32 && sizeof(T) == 4 && (N) == 1024) ? PDEOPT_IMEX_ROW_OCC : 1)

// IMEX row passes on the register engine, two ENVIRONMENTS per complex sequence: the operator
// f -> Re ifft2(M fft2 f) equals ifft2(M_h fft2 f) with the symmetrised multiplier
// M_h(k) = (M(k) + conj M(-k)) / 2, which maps real fields to real fields, so by linearity
//   z = f_a + i f_b   ->   ifft2(M_h fft2 z) = L f_a + i L f_b :
// plain complex transforms move the bytes of a real<->hermitian plan (8 per cell and pass instead of
// 16) with no hermitian packing.  An odd batch leaves the last sequence with a zero imaginary part.
template <typename T, int N>
__global__ __launch_bounds__(256, PDEOPT_IMEX_ROW_WAVES(T, N)) void imex_row_fwd_reg_kernel(const T* __restrict__ k, Cx<T>* __restrict__ c,
                                                               const Cx<T>* __restrict__ tw, int nx, int batch,
                                                               int pack /* environments per complex field: 2, or 1 */) {
  using E = typename ImexRowEngine<T, N>::type;
  constexpr int PTS = E::kPts, TT = E::TT, F = 256 / TT, NP = E::NP;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x;
  const int f = tid / TT, j = tid - f * TT;
  typename E::LdsT* const seq = reinterpret_cast<typename E::LdsT*>(smem_raw) + f * NP;
  const int64_t prow = (int64_t)blockIdx.x * F + f;  // row index in the pair-packed field
  const int pair = (int)(prow / nx);
  const int64_t r = prow - (int64_t)pair * nx;
  const T* const ka = k + ((int64_t)(pack * pair) * nx + r) * N;
  const bool has_b = pack == 2 && 2 * pair + 1 < batch;
  const T* const kb = ka + (int64_t)nx * N;
  Cx<T> v[PTS];
#pragma unroll
  for (int m = 0; m < PTS; ++m) {
    const int n = E::natural(j, m);
    v[m] = Cx<T>{ka[n], has_b ? kb[n] : T(0)};
  }
  E::template dif<-1>(v, seq, tw, j);
  Cx<T>* const g = c + prow * N;
#pragma unroll
  for (int sl = 0; sl < PTS; ++sl) g[E::freq(j, sl)] = v[sl];
}

template <typename T, int N>
__global__ __launch_bounds__(256, PDEOPT_IMEX_ROW_WAVES(T, N)) void imex_row_inv_reg_kernel(const Cx<T>* __restrict__ c, T* __restrict__ y,
                                                               const Cx<T>* __restrict__ tw, T dt, int nx,
                                                               int batch, int pack) {
  using E = typename ImexRowEngine<T, N>::type;
  constexpr int PTS = E::kPts, TT = E::TT, F = 256 / TT, NP = E::NP;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x;
  const int f = tid / TT, j = tid - f * TT;
  typename E::LdsT* const seq = reinterpret_cast<typename E::LdsT*>(smem_raw) + f * NP;
  const int64_t prow = (int64_t)blockIdx.x * F + f;
  const int pair = (int)(prow / nx);
  const int64_t r = prow - (int64_t)pair * nx;
  const Cx<T>* const g = c + prow * N;
  Cx<T> v[PTS];
#pragma unroll
  for (int sl = 0; sl < PTS; ++sl) v[sl] = g[E::freq(j, sl)];
  E::template dit<+1>(v, seq, tw, j);
  T* const ya = y + ((int64_t)(pack * pair) * nx + r) * N;
  const bool has_b = pack == 2 && 2 * pair + 1 < batch;
  T* const yb = ya + (int64_t)nx * N;
  // y1 = y0 + dt Re ifft(...)   solvers.py:63.  Groups of 8 points: keeps the y loads of the whole sequence from
  // being hoisted above the transform (VGPRs)
#pragma unroll
  for (int m = 0; m < PTS; ++m) {
    if (m % 8 == 0) asm volatile("" ::: "memory");
    ya[E::natural(j, m)] += dt * v[m].re;
  }
  if (has_b) {
#pragma unroll
    for (int m = 0; m < PTS; ++m) {
      if (m % 8 == 0) asm volatile("" ::: "memory");
      yb[E::natural(j, m)] += dt * v[m].im;
    }
  }
}

template <typename T, int N>
int imex_rows(pdeopt_ctx* ctx, StrangFused& sf, bool forward, double dt) {
  using E = typename ImexRowEngine<T, N>::type;
  constexpr int F = 256 / E::TT;
  const pdeopt_problem& p = ctx->prob;
  // environment window (win_lo even): pointers pre-offset, the kernels see win_n environments
  const int64_t cells = (int64_t)p.nx * p.ny, w0 = ctx->win_lo;
  const int pack = ctx->imex_per_env ? 1 : 2;  // environments per complex field
  const int npairs = (ctx->win_n + pack - 1) / pack;
  const size_t lds = (size_t)F * E::NP * sizeof(typename E::LdsT);
  const int blocks = (int)((int64_t)npairs * p.nx / F);
  Cx<T>* const cw = (Cx<T>*)sf.cwork + (w0 / pack) * cells;
  if (forward) {
    auto kern = imex_row_fwd_reg_kernel<T, N>;
    int rc = allow_lds(ctx, kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, ctx->stream, (const T*)ctx->TA + w0 * cells, cw,
                       (const Cx<T>*)sf.tw_y, p.nx, ctx->win_n, pack);
  } else {
    auto kern = imex_row_inv_reg_kernel<T, N>;
    int rc = allow_lds(ctx, kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, ctx->stream, (const Cx<T>*)cw, (T*)ctx->Y + w0 * cells,
                       (const Cx<T>*)sf.tw_y, (T)dt, p.nx, ctx->win_n, pack);
  }
  ctx->n_stage_launches++;
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  return PDEOPT_OK;
}

template <typename T, int N>
int imex_cols(pdeopt_ctx* ctx, StrangFused& sf) {
  constexpr int C = cols_per_block<T>();
  constexpr int PTS = col_pts<T, N>();
  const pdeopt_problem& p = ctx->prob;
  const int pack = ctx->imex_per_env ? 1 : 2;
  const int npairs = (ctx->win_n + pack - 1) / pack;
  Cx<T>* const cw = (Cx<T>*)sf.cwork + (int64_t)(ctx->win_lo / pack) * p.nx * p.ny;
  const EnvParams<T>* const ep = ctx->imex_per_env ? (const EnvParams<T>*)ctx->env_params_dev + ctx->win_lo : nullptr;
  const T inv_n = (T)(1.0 / ((double)p.nx * p.ny));
  // FFT_x -> * multiplier -> IFFT_x, in place
  if constexpr (col_use32<T, N>()) {
    const size_t lds = (size_t)C * kCol32Pitch * sizeof(T);
    auto kern = ep ? strang_col32_kernel<T, C, false, true> : strang_col32_kernel<T, C, false, false>;
    int rc = allow_lds(ctx, kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(p.ny / C, npairs), dim3(C * 32), lds, ctx->stream, cw, (const Cx<T>*)sf.imex_mult,
                       (const Cx<T>*)sf.tw_x, p.ny, (const double*)nullptr, 0, 1.0, ep, inv_n);
  } else {
    const size_t lds = (size_t)C * fft_lds_pitch<N>() * sizeof(Cx<T>);
    auto kern = ep ? strang_col_reg_kernel<T, N, C, PTS, false, true> : strang_col_reg_kernel<T, N, C, PTS, false, false>;
    int rc = allow_lds(ctx, kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(p.ny / C, npairs), dim3(C * N / PTS), lds, ctx->stream, cw,
                       (const Cx<T>*)sf.imex_mult, (const Cx<T>*)sf.tw_x, p.ny, (const double*)nullptr, 0, 1.0, ep,
                       inv_n);
  }
  ctx->n_stage_launches++;
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
  // one complex field per PAIR of environments, or per environment when their implicit operators differ
  const size_t cwork_need = (size_t)cells * (ctx->imex_per_env ? (size_t)p.batch : (size_t)((p.batch + 1) / 2)) * sizeof(Cx<T>);
  if (sf.cwork && sf.cwork_bytes < cwork_need) {
    PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(sf.cwork);
    sf.cwork = nullptr;
  }
  if ((rc = ensure_buffer(ctx, &sf.cwork, cwork_need))) return rc;
  sf.cwork_bytes = std::max(sf.cwork_bytes, cwork_need);
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
    // M(k) = 1 / ((1 + A dt symbol(k)) nx ny), then M_h(k) = (M(k) + conj M(-k)) / 2: the transforms carry
    // two real environments per complex field (imex_row_fwd_reg_kernel), which needs the real-to-real form
    std::vector<std::complex<double>> mfull((size_t)cells);
    const double inv_n = 1.0 / (double)cells;
    sf.imex_sym_real = true;
    for (int64_t i = 0; i < cells; ++i) {
      mfull[i] = inv_n / (1.0 + ctx->imex_A * dt * sym[i]);
      sf.imex_sym_real = sf.imex_sym_real && sym[i].imag() == 0.0;
    }
    std::vector<Cx<T>> m((size_t)cells);
    for (int kx = 0; kx < p.nx; ++kx) {
      const int mx = (p.nx - kx) % p.nx;
      for (int ky = 0; ky < p.ny; ++ky) {
        const int my = (p.ny - ky) % p.ny;
        const std::complex<double> v =
            0.5 * (mfull[(size_t)kx * p.ny + ky] + std::conj(mfull[(size_t)mx * p.ny + my]));
        m[col_mult_index(kx, ky, p.nx, p.ny)] = Cx<T>{(T)v.real(), (T)v.imag()};
      }
    }
    if ((rc = ensure_buffer(ctx, &sf.imex_mult, (size_t)cells * sizeof(Cx<T>)))) return rc;
    PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(sf.imex_mult, m.data(), (size_t)cells * sizeof(Cx<T>), hipMemcpyHostToDevice, ctx->stream));
    PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    sf.imex_valid = true;
    sf.imex_dt = dt;
    sf.imex_A = ctx->imex_A;
  }
  if (ctx->imex_per_env && !sf.imex_sym_real)
    return fail(ctx, PDEOPT_EINVAL, "per-environment IMEX scales need a real fourier_symbol");
  // group by group (an even number of environments each): state + slope + spectrum work field are 12 B/cell
  int group = p.batch;
  if (ctx->opt_group_envs > 0) {
    group = (int)std::min<int64_t>((ctx->opt_group_envs + 1) & ~1LL, p.batch);
  } else if (ctx->opt_group_envs == 0) {
    const size_t per_env = (size_t)cells * 3 * sizeof(T);
    const int64_t fit = std::max<int64_t>(2, (int64_t)((192ull << 20) / per_env) & ~1LL);
    if (fit < p.batch && n > 1) {
      const int ngroups = (int)((p.batch + fit - 1) / fit);
      group = ((p.batch + ngroups - 1) / ngroups + 1) & ~1;
    }
  }
  ctx->last_groups = (p.batch + group - 1) / group;
  // one substep of the window [win_lo, win_lo + win_n) on ctx->stream
  auto substep = [&]() -> int {
    int r;
    if ((r = launch_rhs_slope(ctx, ctx->Y, ctx->TA, 0.0))) return r;
    switch (p.ny) {
#define X(NN) case NN: r = imex_rows<T, NN>(ctx, sf, true, dt); break;
      PDEOPT_FFT_SIZES(X)
#undef X
    }
    if (r) return r;
    switch (p.nx) {
#define X(NN) case NN: r = imex_cols<T, NN>(ctx, sf); break;
      PDEOPT_FFT_SIZES(X)
#undef X
    }
    if (r) return r;
    switch (p.ny) {
#define X(NN) case NN: r = imex_rows<T, NN>(ctx, sf, false, dt); break;
      PDEOPT_FFT_SIZES(X)
#undef X
    }
    return r;
  };
  // Two groups in flight on two streams (PDEOPT_OPT_GROUP_STREAMS = 2 only).  Round 2 measured 1445 vs 1443
  // env-steps/s at 2 x 8 environments against 1 x 16, and slower at 2 x 16 / 2 x 4; round 3 re-measured after the
  // explicit and the Strang pipelines gained 6-7 % from it: see profiles/r03_group_streams_ab.txt.
  if (ctx->opt_group_streams == 2 && n > 1 && group >= 4 && ctx->last_groups >= 1 && p.batch >= 4) {
    if (ctx->opt_group_envs == 0) group = std::max(2, (group / 2 + 1) & ~1);
    ctx->last_groups = (p.batch + group - 1) / group;
  }
  if (ctx->opt_group_streams == 2 && ctx->last_groups >= 2 && n > 1) {
    ctx->last_group_streams = 2;
    if ((rc = ensure_stream2(ctx))) return rc;
    PDEOPT_HIP_CHECK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    PDEOPT_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
    for (int lo = 0; lo < p.batch && !rc; lo += 2 * group) {
      const bool two = lo + group < p.batch;
      for (int64_t s = 0; s < n && !rc; ++s) {
        ctx->win_lo = lo;
        ctx->win_n = std::min(group, p.batch - lo);
        rc = substep();
        if (two && !rc) {
          ctx->win_lo = lo + group;
          ctx->win_n = std::min(group, p.batch - lo - group);
          std::swap(ctx->stream, ctx->stream2);
          rc = substep();
          std::swap(ctx->stream, ctx->stream2);
        }
      }
    }
    const hipError_t e1 = hipEventRecord(ctx->ev_join, ctx->stream2);
    const hipError_t e2 = hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0);
    ctx->win_lo = 0;
    ctx->win_n = p.batch;
    if (rc) return rc;
    PDEOPT_HIP_CHECK(ctx, e1);
    PDEOPT_HIP_CHECK(ctx, e2);
    ctx->last_kernel += "+imex_fused_lds_fft";
    return PDEOPT_OK;
  }
  for (int lo = 0; lo < p.batch && !rc; lo += group) {
    ctx->win_lo = lo;
    ctx->win_n = std::min(group, p.batch - lo);
    for (int64_t s = 0; s < n && !rc; ++s) rc = substep();
  }
  ctx->win_lo = 0;
  ctx->win_n = p.batch;
  if (rc) return rc;
  ctx->last_kernel += "+imex_fused_lds_fft";
  return PDEOPT_OK;
}

}  // namespace

bool imex_fused_supported(const pdeopt_ctx* ctx) {
  const pdeopt_problem& p = ctx->prob;
  const bool f64 = p.dtype == PDEOPT_F64;
  // PDEOPT_OPT_IMEX_LDS_FFT: 0 auto (these transforms where the size is covered), -1 rocFFT real<->hermitian
  // plans (csrc/spectral.hip), 1 as 0.  The "generic" kernel path also means the library pipeline.
  if (ctx->opt_imex_lds_fft < 0 || ctx->opt_kernel_path == 1) return false;
  if (p.nz > 1) return false;  // 3-D fields go through rocFFT's 3-D real<->hermitian plans
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

int advance_strang_fused(pdeopt_ctx* ctx, double t0, double dt, int64_t n) {
  return ctx->prob.dtype == PDEOPT_F32 ? strang_fused_t<float>(ctx, t0, dt, n) : strang_fused_t<double>(ctx, t0, dt, n);
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
