// Spectral integrators on rocFFT: SemiImplicitFourierSpectral.step and StrangSplitting.step
// (pde_opt/numerics/solvers.py:56-70 and :99-122), batched over environments.
//
//   IMEX    y1 = y0 + dt Re ifftn( fftn(rhs(y0)) / (1 + A dt fourier_symbol) )        solvers.py:58-63
//   Strang  tau = dt * time_scale;  E = exp(A_term tau / 2)
//           psi1 = ifft(fft(psi0) E);  b = B(t0, psi0)  (PRE-half-step state, :109)
//           psi2 = psi1 exp(b tau);    psi3 = psi2 / sqrt(sum |psi2|^2 dx^2)  (:111, every step)
//           psi4 = ifft(fft(psi3) E)
//   with b = -i (V + k |psi0|^2)  (gross_pitaevskii.py:67-75; V = trap + lights, an aux field).
//
// rocFFT transforms are unnormalised; the 1/N of ifft is folded into the spectral multipliers,
// and the per-environment normalisation scalar of psi3 is folded into the second spectral
// multiply (the FFT is linear), which saves one pass over the field.
#include <cmath>
#include <complex>
#include <mutex>

#include "closures.hpp"
#include "common.hpp"

namespace pdeopt {

struct Spectral {
  bool setup = false;
  rocfft_plan fwd = nullptr, inv = nullptr;
  rocfft_execution_info info = nullptr;
  void* work = nullptr;
  size_t work_bytes = 0;
  void* cbuf = nullptr;    // complex work field (IMEX, rhs_fourier)
  void* cbuf2 = nullptr;   // rhs_fourier work fields
  void* cbuf3 = nullptr;
  void* mult = nullptr;    // complex spectral multiplier, shared [nx][ny]
  void* dens = nullptr;    // real |psi0|^2 (Strang)
  double* partial = nullptr;  // [batch][kNormBlocks]
  double* scale = nullptr;    // [batch]
  // real <-> hermitian transforms of the IMEX step (half the bytes of C2C)
  rocfft_plan r2c = nullptr, c2r = nullptr;
  rocfft_execution_info rinfo = nullptr;
  void* rwork = nullptr;
  void* hbuf = nullptr;    // hermitian half-spectrum work field
  void* hmult = nullptr;   // multiplier on the half-spectrum
  int64_t half_cells = 0;  // complex elements of one half-spectrum
  // cache key of `mult` / `hmult`
  int mult_kind = -1;
  double mult_dt = NAN, mult_A = NAN, mult_tr = NAN, mult_ti = NAN;
};

namespace {

constexpr int kNormBlocks = 128;

#define PDEOPT_FFT_CHECK(ctx, expr)                                                         \
  do {                                                                                      \
    rocfft_status s_ = (expr);                                                              \
    if (s_ != rocfft_status_success)                                                        \
      return fail((ctx), PDEOPT_EFFT, "%s failed with rocfft_status %d (%s:%d)", #expr,    \
                  (int)s_, __FILE__, __LINE__);                                            \
  } while (0)

std::once_flag g_rocfft_once;

int ensure_plans(pdeopt_ctx* ctx) {
  if (!ctx->spectral) ctx->spectral = new Spectral();
  Spectral& sp = *ctx->spectral;
  if (sp.setup) return PDEOPT_OK;
  std::call_once(g_rocfft_once, [] { rocfft_setup(); });
  const pdeopt_problem& p = ctx->prob;
  // rocFFT lengths are fastest-first; a dimension of length 1 is dropped (256 x 1 "1-D" runs
  // of tests/test_solvers.py:21-61)
  size_t lengths[3];
  size_t dims = 0;
  if (p.nz > 1) lengths[dims++] = (size_t)p.nz;  // 3-D equations: z is the contiguous axis
  if (p.ny > 1) lengths[dims++] = (size_t)p.ny;
  if (p.nx > 1) lengths[dims++] = (size_t)p.nx;
  if (dims == 0) lengths[dims++] = 1;
  const rocfft_precision prec =
      p.dtype == PDEOPT_F32 ? rocfft_precision_single : rocfft_precision_double;
  PDEOPT_FFT_CHECK(ctx, rocfft_plan_create(&sp.fwd, rocfft_placement_inplace,
                                           rocfft_transform_type_complex_forward, prec, dims,
                                           lengths, (size_t)p.batch, nullptr));
  PDEOPT_FFT_CHECK(ctx, rocfft_plan_create(&sp.inv, rocfft_placement_inplace,
                                           rocfft_transform_type_complex_inverse, prec, dims,
                                           lengths, (size_t)p.batch, nullptr));
  size_t w1 = 0, w2 = 0;
  PDEOPT_FFT_CHECK(ctx, rocfft_plan_get_work_buffer_size(sp.fwd, &w1));
  PDEOPT_FFT_CHECK(ctx, rocfft_plan_get_work_buffer_size(sp.inv, &w2));
  sp.work_bytes = w1 > w2 ? w1 : w2;
  PDEOPT_FFT_CHECK(ctx, rocfft_execution_info_create(&sp.info));
  if (sp.work_bytes) {
    int rc = ensure_buffer(ctx, &sp.work, sp.work_bytes);
    if (rc) return rc;
    PDEOPT_FFT_CHECK(ctx, rocfft_execution_info_set_work_buffer(sp.info, sp.work, sp.work_bytes));
  }
  PDEOPT_FFT_CHECK(ctx, rocfft_execution_info_set_stream(sp.info, ctx->stream));
  sp.setup = true;
  return PDEOPT_OK;
}

// real-to-hermitian plans.  The fastest non-trivial dimension is the one rocFFT halves.
int ensure_real_plans(pdeopt_ctx* ctx) {
  Spectral& sp = *ctx->spectral;
  if (sp.r2c) return PDEOPT_OK;
  const pdeopt_problem& p = ctx->prob;
  size_t lengths[3];
  size_t dims = 0;
  if (p.nz > 1) lengths[dims++] = (size_t)p.nz;  // 3-D equations: z is the contiguous axis
  if (p.ny > 1) lengths[dims++] = (size_t)p.ny;
  if (p.nx > 1) lengths[dims++] = (size_t)p.nx;
  if (dims == 0) lengths[dims++] = 1;
  sp.half_cells = (int64_t)(lengths[0] / 2 + 1);
  for (size_t d = 1; d < dims; ++d) sp.half_cells *= (int64_t)lengths[d];
  const rocfft_precision prec =
      p.dtype == PDEOPT_F32 ? rocfft_precision_single : rocfft_precision_double;
  PDEOPT_FFT_CHECK(ctx, rocfft_plan_create(&sp.r2c, rocfft_placement_notinplace,
                                           rocfft_transform_type_real_forward, prec, dims, lengths,
                                           (size_t)p.batch, nullptr));
  PDEOPT_FFT_CHECK(ctx, rocfft_plan_create(&sp.c2r, rocfft_placement_notinplace,
                                           rocfft_transform_type_real_inverse, prec, dims, lengths,
                                           (size_t)p.batch, nullptr));
  size_t w1 = 0, w2 = 0;
  PDEOPT_FFT_CHECK(ctx, rocfft_plan_get_work_buffer_size(sp.r2c, &w1));
  PDEOPT_FFT_CHECK(ctx, rocfft_plan_get_work_buffer_size(sp.c2r, &w2));
  const size_t wb = w1 > w2 ? w1 : w2;
  PDEOPT_FFT_CHECK(ctx, rocfft_execution_info_create(&sp.rinfo));
  if (wb) {
    int rc = ensure_buffer(ctx, &sp.rwork, wb);
    if (rc) return rc;
    PDEOPT_FFT_CHECK(ctx, rocfft_execution_info_set_work_buffer(sp.rinfo, sp.rwork, wb));
  }
  PDEOPT_FFT_CHECK(ctx, rocfft_execution_info_set_stream(sp.rinfo, ctx->stream));
  return PDEOPT_OK;
}

int real_fft_exec(pdeopt_ctx* ctx, bool forward, void* in, void* out) {
  Spectral& sp = *ctx->spectral;
  void* ib[1] = {in};
  void* ob[1] = {out};
  PDEOPT_FFT_CHECK(ctx, rocfft_execute(forward ? sp.r2c : sp.c2r, ib, ob, sp.rinfo));
  return PDEOPT_OK;
}

int fft_exec(pdeopt_ctx* ctx, bool forward, void* buf) {
  Spectral& sp = *ctx->spectral;
  void* in[1] = {buf};
  PDEOPT_FFT_CHECK(ctx, rocfft_execute(forward ? sp.fwd : sp.inv, in, nullptr, sp.info));
  return PDEOPT_OK;
}

// ---------------------------------------------------------------------------------- kernels

template <typename T>
struct C2 {
  T re, im;
};

template <typename T>
__device__ __forceinline__ void t_sincos(T x, T* s, T* c);
template <>
__device__ __forceinline__ void t_sincos<float>(float x, float* s, float* c) {
  sincosf(x, s, c);
}
template <>
__device__ __forceinline__ void t_sincos<double>(double x, double* s, double* c) {
  sincos(x, s, c);
}
template <typename T>
__device__ __forceinline__ T t_expr(T x);
template <>
__device__ __forceinline__ float t_expr<float>(float x) {
  return expf(x);
}
template <>
__device__ __forceinline__ double t_expr<double>(double x) {
  return exp(x);
}

// cbuf = (k, 0)
template <typename T>
__global__ void embed_real_kernel(const T* __restrict__ k, C2<T>* __restrict__ c, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t st = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) c[i] = C2<T>{k[i], T(0)};
}

// c[b][cell] *= m[cell] (* scale[b])
template <typename T, bool SCALED>
__global__ void spectral_mul_kernel(C2<T>* __restrict__ c, const C2<T>* __restrict__ m,
                                    const double* __restrict__ scale, int64_t cells) {
  const int b = blockIdx.y;
  C2<T>* cb = c + (int64_t)b * cells;
  const T s = SCALED ? (T)scale[b] : T(1);
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t st = (int64_t)gridDim.x * blockDim.x;
  for (; i < cells; i += st) {
    const C2<T> v = cb[i], w = m[i];
    C2<T> r;
    r.re = (v.re * w.re - v.im * w.im) * s;
    r.im = (v.re * w.im + v.im * w.re) * s;
    cb[i] = r;
  }
}

// y += dt * Re(c)
template <typename T>
__global__ void imex_update_kernel(T* __restrict__ y, const C2<T>* __restrict__ c, T dt, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t st = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) y[i] += dt * c[i].re;
}

// dens = |psi|^2
template <typename T>
__global__ void density_kernel(const C2<T>* __restrict__ psi, T* __restrict__ d, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t st = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) {
    const C2<T> v = psi[i];
    d[i] = v.re * v.re + v.im * v.im;
  }
}

// psi *= exp(b tau), b = -i (V + k dens);  partial[b][block] = sum |psi|^2   (deterministic)
template <typename T>
__global__ __launch_bounds__(256) void strang_b_kernel(C2<T>* __restrict__ psi,
                                                       const T* __restrict__ dens,
                                                       const T* __restrict__ pot, int64_t pot_stride,
                                                       const EnvParams<T>* __restrict__ ep, T tr,
                                                       T ti, int64_t cells,
                                                       double* __restrict__ partial, const SpotArgs<T> spots, int ny) {
  const int b = blockIdx.y;
  C2<T>* pb = psi + (int64_t)b * cells;
  const T* db = dens + (int64_t)b * cells;
  const T* vb = pot ? pot + (int64_t)b * pot_stride : nullptr;
  const T kk = ep[b].gpe_k;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cells; i += (int64_t)gridDim.x * 256) {
    T w = (vb ? vb[i] : T(0)) + kk * db[i];
    if (spots.n)  // lights(t0, x, y) as Gaussian spots (pdeopt_set_gpe_spots)
      w += spots_value<T>(spots, b, spots.x_first + T(i / ny) * spots.hx, spots.y_first + T(i % ny) * spots.hy);
    // exp(-i w (tr + i ti)) = exp(w ti) (cos(w tr) - i sin(w tr))
    T sn, cs;
    t_sincos<T>(w * tr, &sn, &cs);
    const T mag = (ti == T(0)) ? T(1) : t_expr<T>(w * ti);
    const T er = mag * cs, ei = -mag * sn;
    const C2<T> v = pb[i];
    C2<T> r;
    r.re = v.re * er - v.im * ei;
    r.im = v.re * ei + v.im * er;
    pb[i] = r;
    acc += (double)r.re * (double)r.re + (double)r.im * (double)r.im;
  }
  __shared__ double sh[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[(int64_t)b * gridDim.x + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// scale[b] = 1 / sqrt(sum_b dx^2)
__global__ void norm_finalize_kernel(const double* __restrict__ partial, int nblocks, double dx2,
                                     double* __restrict__ scale, int batch) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  double s = 0.0;
  for (int i = 0; i < nblocks; ++i) s += partial[(int64_t)b * nblocks + i];
  scale[b] = 1.0 / sqrt(s * dx2);
}

inline int grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

template <typename T>
int upload_mult(pdeopt_ctx* ctx, const std::vector<std::complex<double>>& m) {
  Spectral& sp = *ctx->spectral;
  const size_t cells = m.size();
  int rc = ensure_buffer(ctx, &sp.mult, cells * 2 * sizeof(T));
  if (rc) return rc;
  std::vector<C2<T>> h(cells);
  for (size_t i = 0; i < cells; ++i) h[i] = C2<T>{(T)m[i].real(), (T)m[i].imag()};
  PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(sp.mult, h.data(), cells * sizeof(C2<T>),
                                       hipMemcpyHostToDevice, ctx->stream));
  PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return PDEOPT_OK;
}

// host copy of a complex aux field as complex<double>
int fetch_complex_aux(pdeopt_ctx* ctx, int which, std::vector<std::complex<double>>& out) {
  const AuxField& a = ctx->aux[which];
  const size_t cells = (size_t)ctx->prob.nx * ctx->prob.ny * (ctx->prob.nz > 1 ? ctx->prob.nz : 1);
  out.resize(cells);
  if (a.per_env) return fail(ctx, PDEOPT_EINVAL, "spectral aux fields must be shared across the batch");
  if (ctx->prob.dtype == PDEOPT_F32) {
    std::vector<float> h(cells * 2);
    PDEOPT_HIP_CHECK(ctx, hipMemcpy(h.data(), a.dev, cells * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < cells; ++i) out[i] = {h[2 * i], h[2 * i + 1]};
  } else {
    PDEOPT_HIP_CHECK(ctx, hipMemcpy(out.data(), a.dev, cells * 16, hipMemcpyDeviceToHost));
  }
  return PDEOPT_OK;
}

// y += dt * r
template <typename T>
__global__ void axpy_real_kernel(T* __restrict__ y, const T* __restrict__ r, T dt, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t st = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) y[i] += dt * r[i];
}

// IMEX on real <-> hermitian transforms.  The reference runs full complex transforms on the real
// field (cahn_hilliard.py:72-73) and keeps `.real` of the result (solvers.py:63).  For a real input F
// is hermitian, so  Re ifft(F m) = ifft(F m_h)  with  m_h(k) = (m(k) + conj(m(-k))) / 2 : the
// symmetrised multiplier on the half-spectrum reproduces the reference for ANY (also non-even,
// complex) fourier_symbol while moving half the bytes through rocFFT.
template <typename T>
int imex_t(pdeopt_ctx* ctx, double dt, int64_t n) {
  Spectral& sp = *ctx->spectral;
  const pdeopt_problem& p = ctx->prob;
  const int nzz = p.nz > 1 ? p.nz : 1;
  const int64_t cells = (int64_t)p.nx * p.ny * nzz;
  const int64_t total = cells * p.batch;
  int rc;
  if ((rc = ensure_real_plans(ctx))) return rc;
  const int64_t hc = sp.half_cells;
  if ((rc = ensure_buffer(ctx, &ctx->TA, ctx->total_bytes))) return rc;
  if ((rc = ensure_buffer(ctx, &sp.hbuf, (size_t)hc * p.batch * 2 * sizeof(T)))) return rc;
  if (sp.mult_kind != 2 || sp.mult_dt != dt || sp.mult_A != ctx->imex_A) {
    std::vector<std::complex<double>> sym;
    if ((rc = fetch_complex_aux(ctx, PDEOPT_AUX_IMEX_SYMBOL, sym))) return rc;
    const double inv_n = 1.0 / (double)cells;
    const int n3[3] = {p.nx, p.ny, nzz};
    auto m = [&](int i, int j, int k) {
      return inv_n / (1.0 + ctx->imex_A * dt * sym[((size_t)i * p.ny + j) * nzz + k]);
    };
    std::vector<C2<T>> h((size_t)hc);
    // half-spectrum layout [slow ...][fast/2+1]: the halved axis is the fastest non-trivial one
    // (z for the 3-D equations, else y, else x for the ny == 1 "1-D" runs)
    const int fast = nzz > 1 ? 2 : (p.ny > 1 ? 1 : 0);
    const int nh = n3[fast] / 2 + 1;
    int64_t o = 0;
    for (int i = 0; i < (fast == 0 ? nh : p.nx); ++i)
      for (int j = 0; j < (fast == 1 ? nh : (fast == 0 ? 1 : p.ny)); ++j)
        for (int k = 0; k < (fast == 2 ? nh : 1); ++k) {
          const std::complex<double> v =
              0.5 * (m(i, j, k) + std::conj(m((p.nx - i) % p.nx, (p.ny - j) % p.ny, (nzz - k) % nzz)));
          h[(size_t)o++] = C2<T>{(T)v.real(), (T)v.imag()};
        }
    if ((rc = ensure_buffer(ctx, &sp.hmult, (size_t)hc * sizeof(C2<T>)))) return rc;
    PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(sp.hmult, h.data(), (size_t)hc * sizeof(C2<T>), hipMemcpyHostToDevice, ctx->stream));
    PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    sp.mult_kind = 2;
    sp.mult_dt = dt;
    sp.mult_A = ctx->imex_A;
  }
  for (int64_t s = 0; s < n; ++s) {
    if ((rc = launch_rhs(ctx, ctx->Y, ctx->TA, 0.0))) return rc;
    if ((rc = real_fft_exec(ctx, true, ctx->TA, sp.hbuf))) return rc;
    hipLaunchKernelGGL((spectral_mul_kernel<T, false>), dim3(grid_for(hc), p.batch), dim3(256), 0,
                       ctx->stream, (C2<T>*)sp.hbuf, (const C2<T>*)sp.hmult, nullptr, hc);
    if ((rc = real_fft_exec(ctx, false, sp.hbuf, ctx->TA))) return rc;
    hipLaunchKernelGGL(axpy_real_kernel<T>, dim3(grid_for(total)), dim3(256), 0, ctx->stream,
                       (T*)ctx->Y, (const T*)ctx->TA, (T)dt, total);
  }
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  ctx->last_kernel += "+imex_rocfft_r2c";
  return PDEOPT_OK;
}

template <typename T>
int strang_t(pdeopt_ctx* ctx, double t0, double dt, int64_t n) {
  Spectral& sp = *ctx->spectral;
  const pdeopt_problem& p = ctx->prob;
  const int64_t cells = (int64_t)p.nx * p.ny;
  const int64_t total = cells * p.batch;
  int rc;
  if ((rc = ensure_buffer(ctx, &sp.dens, (size_t)total * sizeof(T)))) return rc;
  if ((rc = ensure_buffer(ctx, (void**)&sp.partial, sizeof(double) * kNormBlocks * p.batch))) return rc;
  if ((rc = ensure_buffer(ctx, (void**)&sp.scale, sizeof(double) * p.batch))) return rc;
  const std::complex<double> tau = dt * std::complex<double>(ctx->ts_re, ctx->ts_im);
  if (sp.mult_kind != 1 || sp.mult_dt != dt || sp.mult_tr != ctx->ts_re || sp.mult_ti != ctx->ts_im) {
    // E / N,  E = exp(A_term tau / 2)   (solvers.py:105)
    std::vector<std::complex<double>> a;
    if ((rc = fetch_complex_aux(ctx, PDEOPT_AUX_GPE_A_TERM, a))) return rc;
    const double inv_n = 1.0 / (double)cells;
    for (auto& v : a) v = std::exp(v * 0.5 * tau) * inv_n;
    if ((rc = upload_mult<T>(ctx, a))) return rc;
    sp.mult_kind = 1;
    sp.mult_dt = dt;
    sp.mult_tr = ctx->ts_re;
    sp.mult_ti = ctx->ts_im;
  }
  const AuxField& pot = ctx->aux[PDEOPT_AUX_GPE_POTENTIAL];
  const int64_t pot_stride = pot.per_env ? cells : 0;
  const dim3 mgrid(grid_for(cells), p.batch);
  for (int64_t s = 0; s < n; ++s) {
    hipLaunchKernelGGL(density_kernel<T>, dim3(grid_for(total)), dim3(256), 0, ctx->stream,
                       (const C2<T>*)ctx->Y, (T*)sp.dens, total);
    if ((rc = fft_exec(ctx, true, ctx->Y))) return rc;
    hipLaunchKernelGGL((spectral_mul_kernel<T, false>), mgrid, dim3(256), 0, ctx->stream,
                       (C2<T>*)ctx->Y, (const C2<T>*)sp.mult, nullptr, cells);
    if ((rc = fft_exec(ctx, false, ctx->Y))) return rc;
    // b = terms.vf(t0, y0): a time-dependent potential is evaluated at this substep's start time (solvers.py:109)
    if ((rc = refresh_time_aux(ctx, PDEOPT_AUX_GPE_POTENTIAL, t0 + (double)s * dt))) return rc;
    hipLaunchKernelGGL(strang_b_kernel<T>, dim3(kNormBlocks, p.batch), dim3(256), 0, ctx->stream,
                       (C2<T>*)ctx->Y, (const T*)sp.dens, (const T*)pot.dev, pot_stride,
                       (const EnvParams<T>*)ctx->env_params_dev, (T)tau.real(), (T)tau.imag(), cells,
                       sp.partial, make_spot_args<T>(ctx, t0 + (double)s * dt), p.ny);
    hipLaunchKernelGGL(norm_finalize_kernel, dim3((p.batch + 63) / 64), dim3(64), 0, ctx->stream,
                       (const double*)sp.partial, kNormBlocks, ctx->strang_dx * ctx->strang_dx,
                       sp.scale, p.batch);
    if ((rc = fft_exec(ctx, true, ctx->Y))) return rc;
    hipLaunchKernelGGL((spectral_mul_kernel<T, true>), mgrid, dim3(256), 0, ctx->stream,
                       (C2<T>*)ctx->Y, (const C2<T>*)sp.mult, (const double*)sp.scale, cells);
    if ((rc = fft_exec(ctx, false, ctx->Y))) return rc;
  }
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  ctx->last_kernel = "strang_rocfft_c2c";
  return PDEOPT_OK;
}

// ---------------------------------------------------------------------------------------------
// pseudo-spectral right-hand sides (cahn_hilliard.py:82-87, allen_cahn.py:74-79)
//   t_hat = F[mu_h(u)] - kappa K2 F[u],   K2 = (2 pi i kx)^2 + (2 pi i ky)^2
//   CH:  rhs = Re F^-1[ ikx F[D(u) F^-1[ikx t_hat]] + iky F[D(u) F^-1[iky t_hat]] ]     (7 FFTs)
//   AC:  rhs = -R(u) Re F^-1[t_hat]                                                     (3 FFTs)
// wave numbers are formed in-kernel from the index (fftfreq(n, h), domains.py:44-47).
// ---------------------------------------------------------------------------------------------

template <typename T>
__device__ __forceinline__ T wavenumber_2pi(int idx, int n, T inv_len) {
  // 2 pi * fftfreq(n, h)[idx], inv_len = 1 / (n h)
  const int k = (idx < (n + 1) / 2) ? idx : idx - n;
  return T(6.283185307179586476925286766559) * T(k) * inv_len;
}

// a = (u, 0), b = (mu_h(u), 0)
template <typename T>
__global__ void fourier_embed_kernel(const T* __restrict__ u, C2<T>* __restrict__ a, C2<T>* __restrict__ b,
                                     const EnvParams<T>* __restrict__ ep, ClosureSpec mu, int64_t cells) {
  const int env = blockIdx.y;
  const EnvParams<T>& p = ep[env];
  const int64_t o = (int64_t)env * cells;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (int64_t)gridDim.x * blockDim.x) {
    const T c = u[o + i];
    a[o + i] = C2<T>{c, T(0)};
    b[o + i] = C2<T>{closure_generic<T>(mu, p.mu, c), T(0)};
  }
}

// t_hat = b - kappa K2 a ;  CH: a <- ikx t_hat, b <- iky t_hat ;  AC: b <- t_hat
template <typename T, bool CH>
__global__ void fourier_that_kernel(C2<T>* __restrict__ a, C2<T>* __restrict__ b,
                                    const EnvParams<T>* __restrict__ ep, int nx, int ny, T inv_lx, T inv_ly) {
  const int env = blockIdx.y;
  const T kap = ep[env].kappa;
  const int64_t cells = (int64_t)nx * ny, o = (int64_t)env * cells;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (int64_t)gridDim.x * blockDim.x) {
    const int ix = (int)(i / ny), iy = (int)(i % ny);
    const T kx = wavenumber_2pi<T>(ix, nx, inv_lx), ky = wavenumber_2pi<T>(iy, ny, inv_ly);
    const T k2 = -(kx * kx + ky * ky);  // (i kx)^2 + (i ky)^2
    const C2<T> ua = a[o + i], mb = b[o + i];
    const C2<T> t{mb.re - kap * k2 * ua.re, mb.im - kap * k2 * ua.im};
    if (CH) {
      a[o + i] = C2<T>{-kx * t.im, kx * t.re};  // i kx t
      b[o + i] = C2<T>{-ky * t.im, ky * t.re};
    } else {
      b[o + i] = t;
    }
  }
}

// a *= D(u)/N, b *= D(u)/N   (the 1/N of the preceding inverse transform folded in)
template <typename T>
__global__ void fourier_mob_kernel(const T* __restrict__ u, C2<T>* __restrict__ a, C2<T>* __restrict__ b,
                                   const EnvParams<T>* __restrict__ ep, ClosureSpec mob, int64_t cells, T inv_n) {
  const int env = blockIdx.y;
  const EnvParams<T>& p = ep[env];
  const int64_t o = (int64_t)env * cells;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (int64_t)gridDim.x * blockDim.x) {
    const T d = closure_generic<T>(mob, p.mob, u[o + i]) * inv_n;
    C2<T> va = a[o + i], vb = b[o + i];
    a[o + i] = C2<T>{va.re * d, va.im * d};
    b[o + i] = C2<T>{vb.re * d, vb.im * d};
  }
}

// a <- ikx a + iky b
template <typename T>
__global__ void fourier_div_kernel(C2<T>* __restrict__ a, const C2<T>* __restrict__ b, int nx, int ny,
                                   T inv_lx, T inv_ly) {
  const int env = blockIdx.y;
  const int64_t cells = (int64_t)nx * ny, o = (int64_t)env * cells;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (int64_t)gridDim.x * blockDim.x) {
    const int ix = (int)(i / ny), iy = (int)(i % ny);
    const T kx = wavenumber_2pi<T>(ix, nx, inv_lx), ky = wavenumber_2pi<T>(iy, ny, inv_ly);
    const C2<T> va = a[o + i], vb = b[o + i];
    a[o + i] = C2<T>{-kx * va.im - ky * vb.im, kx * va.re + ky * vb.re};
  }
}

// CH: out = Re(a)/N ;  AC: out = -R(u) Re(b)/N
template <typename T, bool CH>
__global__ void fourier_out_kernel(const T* __restrict__ u, const C2<T>* __restrict__ c, T* __restrict__ out,
                                   const EnvParams<T>* __restrict__ ep, ClosureSpec mob, int64_t cells, T inv_n) {
  const int env = blockIdx.y;
  const EnvParams<T>& p = ep[env];
  const int64_t o = (int64_t)env * cells;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (int64_t)gridDim.x * blockDim.x) {
    const T r = c[o + i].re * inv_n;
    out[o + i] = CH ? r : -closure_generic<T>(mob, p.mob, u[o + i]) * r;
  }
}

template <typename T>
int rhs_fourier_t(pdeopt_ctx* ctx, const void* in, void* out) {
  Spectral& sp = *ctx->spectral;
  const pdeopt_problem& p = ctx->prob;
  const int64_t cells = (int64_t)p.nx * p.ny, total = cells * p.batch;
  int rc;
  if ((rc = ensure_buffer(ctx, &sp.cbuf2, (size_t)total * 2 * sizeof(T)))) return rc;
  if ((rc = ensure_buffer(ctx, &sp.cbuf3, (size_t)total * 2 * sizeof(T)))) return rc;
  C2<T>* a = (C2<T>*)sp.cbuf2;
  C2<T>* b = (C2<T>*)sp.cbuf3;
  const T* u = (const T*)in;
  const auto* ep = (const EnvParams<T>*)ctx->env_params_dev;
  const ClosureSpec mu{p.mu.kind, p.mu.flags, p.mu.n}, mob{p.mob.kind, p.mob.flags, p.mob.n};
  const dim3 grid(grid_for(cells), p.batch), blk(256);
  const T inv_lx = T(1.0 / (p.nx * p.hx)), inv_ly = T(1.0 / (p.ny * p.hy)), inv_n = T(1.0 / (double)cells);
  const bool ch = p.equation == PDEOPT_EQ_CAHN_HILLIARD;
  hipLaunchKernelGGL(fourier_embed_kernel<T>, grid, blk, 0, ctx->stream, u, a, b, ep, mu, cells);
  if ((rc = fft_exec(ctx, true, a))) return rc;
  if ((rc = fft_exec(ctx, true, b))) return rc;
  if (ch) {
    hipLaunchKernelGGL((fourier_that_kernel<T, true>), grid, blk, 0, ctx->stream, a, b, ep, p.nx, p.ny, inv_lx, inv_ly);
    if ((rc = fft_exec(ctx, false, a))) return rc;
    if ((rc = fft_exec(ctx, false, b))) return rc;
    hipLaunchKernelGGL(fourier_mob_kernel<T>, grid, blk, 0, ctx->stream, u, a, b, ep, mob, cells, inv_n);
    if ((rc = fft_exec(ctx, true, a))) return rc;
    if ((rc = fft_exec(ctx, true, b))) return rc;
    hipLaunchKernelGGL(fourier_div_kernel<T>, grid, blk, 0, ctx->stream, a, (const C2<T>*)b, p.nx, p.ny, inv_lx, inv_ly);
    if ((rc = fft_exec(ctx, false, a))) return rc;
    hipLaunchKernelGGL((fourier_out_kernel<T, true>), grid, blk, 0, ctx->stream, u, (const C2<T>*)a, (T*)out, ep, mob, cells, inv_n);
  } else {
    hipLaunchKernelGGL((fourier_that_kernel<T, false>), grid, blk, 0, ctx->stream, a, b, ep, p.nx, p.ny, inv_lx, inv_ly);
    if ((rc = fft_exec(ctx, false, b))) return rc;
    hipLaunchKernelGGL((fourier_out_kernel<T, false>), grid, blk, 0, ctx->stream, u, (const C2<T>*)b, (T*)out, ep, mob, cells, inv_n);
  }
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  ctx->last_kernel = ch ? "rhs_fourier<CH>(rocfft x7)" : "rhs_fourier<AC>(rocfft x3)";
  return PDEOPT_OK;
}

// ---------------------------------------------------------------------------------------------
// CahnHilliard3DPeriodic.rhs_fourier (cahn_hilliard.py:167-175), fields [b][nx][ny][nz]:
//   t_hat = F[mu_h(u)] - kappa K2 F[u];   rhs = Re F^-1[ sum_d i k_d F[ D(u) F^-1[ i k_d t_hat ] ] ]   (9 FFTs)
// one axis at a time through a single work field, the divergence accumulated in spectral space.
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void wave3(int64_t i, int nx, int ny, int nz, T ilx, T ily, T ilz, T* k) {
  const int iz = (int)(i % nz), iy = (int)((i / nz) % ny), ix = (int)(i / ((int64_t)nz * ny));
  k[0] = wavenumber_2pi<T>(ix, nx, ilx);
  k[1] = wavenumber_2pi<T>(iy, ny, ily);
  k[2] = wavenumber_2pi<T>(iz, nz, ilz);
}

// b <- t_hat = b - kappa K2 a
template <typename T>
__global__ void fourier3_that_kernel(const C2<T>* __restrict__ a, C2<T>* __restrict__ b, const EnvParams<T>* __restrict__ ep,
                                     int nx, int ny, int nz, T ilx, T ily, T ilz) {
  const int env = blockIdx.y;
  const T kap = ep[env].kappa;
  const int64_t cells = (int64_t)nx * ny * nz, o = (int64_t)env * cells;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (int64_t)gridDim.x * blockDim.x) {
    T k[3];
    wave3<T>(i, nx, ny, nz, ilx, ily, ilz, k);
    const T k2 = -(k[0] * k[0] + k[1] * k[1] + k[2] * k[2]);
    const C2<T> ua = a[o + i], mb = b[o + i];
    b[o + i] = C2<T>{mb.re - kap * k2 * ua.re, mb.im - kap * k2 * ua.im};
  }
}

// w <- i k_axis t_hat
template <typename T>
__global__ void fourier3_grad_kernel(const C2<T>* __restrict__ that, C2<T>* __restrict__ w, int axis, int nx, int ny,
                                     int nz, T ilx, T ily, T ilz) {
  const int env = blockIdx.y;
  const int64_t cells = (int64_t)nx * ny * nz, o = (int64_t)env * cells;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (int64_t)gridDim.x * blockDim.x) {
    T k[3];
    wave3<T>(i, nx, ny, nz, ilx, ily, ilz, k);
    const C2<T> t = that[o + i];
    w[o + i] = C2<T>{-k[axis] * t.im, k[axis] * t.re};
  }
}

// w *= D(u) / N
template <typename T>
__global__ void fourier3_mob_kernel(const T* __restrict__ u, C2<T>* __restrict__ w, const EnvParams<T>* __restrict__ ep,
                                    ClosureSpec mob, int64_t cells, T inv_n) {
  const int env = blockIdx.y;
  const EnvParams<T>& p = ep[env];
  const int64_t o = (int64_t)env * cells;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (int64_t)gridDim.x * blockDim.x) {
    const T d = closure_generic<T>(mob, p.mob, u[o + i]) * inv_n;
    const C2<T> v = w[o + i];
    w[o + i] = C2<T>{v.re * d, v.im * d};
  }
}

// acc (+)= i k_axis w
template <typename T>
__global__ void fourier3_div_kernel(C2<T>* __restrict__ acc, const C2<T>* __restrict__ w, int axis, int first, int nx,
                                    int ny, int nz, T ilx, T ily, T ilz) {
  const int env = blockIdx.y;
  const int64_t cells = (int64_t)nx * ny * nz, o = (int64_t)env * cells;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (int64_t)gridDim.x * blockDim.x) {
    T k[3];
    wave3<T>(i, nx, ny, nz, ilx, ily, ilz, k);
    const C2<T> v = w[o + i];
    C2<T> r{-k[axis] * v.im, k[axis] * v.re};
    if (!first) {
      const C2<T> old = acc[o + i];
      r.re += old.re;
      r.im += old.im;
    }
    acc[o + i] = r;
  }
}

template <typename T>
int rhs_fourier3_t(pdeopt_ctx* ctx, const void* in, void* out) {
  Spectral& sp = *ctx->spectral;
  const pdeopt_problem& p = ctx->prob;
  const int64_t cells = (int64_t)p.nx * p.ny * p.nz, total = cells * p.batch;
  int rc;
  if ((rc = ensure_buffer(ctx, &sp.cbuf, (size_t)total * 2 * sizeof(T)))) return rc;
  if ((rc = ensure_buffer(ctx, &sp.cbuf2, (size_t)total * 2 * sizeof(T)))) return rc;
  if ((rc = ensure_buffer(ctx, &sp.cbuf3, (size_t)total * 2 * sizeof(T)))) return rc;
  C2<T>* a = (C2<T>*)sp.cbuf2;   // F[u], later the accumulated divergence
  C2<T>* b = (C2<T>*)sp.cbuf3;   // F[mu_h(u)], then t_hat
  C2<T>* w = (C2<T>*)sp.cbuf;    // one axis at a time
  const T* u = (const T*)in;
  const auto* ep = (const EnvParams<T>*)ctx->env_params_dev;
  const ClosureSpec mu{p.mu.kind, p.mu.flags, p.mu.n}, mob{p.mob.kind, p.mob.flags, p.mob.n};
  const dim3 grid(grid_for(cells), p.batch), blk(256);
  const T ilx = T(1.0 / (p.nx * p.hx)), ily = T(1.0 / (p.ny * p.hy)), ilz = T(1.0 / (p.nz * p.hz));
  const T inv_n = T(1.0 / (double)cells);
  hipLaunchKernelGGL(fourier_embed_kernel<T>, grid, blk, 0, ctx->stream, u, a, b, ep, mu, cells);
  if ((rc = fft_exec(ctx, true, a))) return rc;
  if ((rc = fft_exec(ctx, true, b))) return rc;
  hipLaunchKernelGGL(fourier3_that_kernel<T>, grid, blk, 0, ctx->stream, (const C2<T>*)a, b, ep, p.nx, p.ny, p.nz, ilx, ily, ilz);
  for (int axis = 0; axis < 3; ++axis) {
    hipLaunchKernelGGL(fourier3_grad_kernel<T>, grid, blk, 0, ctx->stream, (const C2<T>*)b, w, axis, p.nx, p.ny, p.nz, ilx, ily, ilz);
    if ((rc = fft_exec(ctx, false, w))) return rc;
    hipLaunchKernelGGL(fourier3_mob_kernel<T>, grid, blk, 0, ctx->stream, u, w, ep, mob, cells, inv_n);
    if ((rc = fft_exec(ctx, true, w))) return rc;
    hipLaunchKernelGGL(fourier3_div_kernel<T>, grid, blk, 0, ctx->stream, a, (const C2<T>*)w, axis, axis == 0 ? 1 : 0, p.nx,
                       p.ny, p.nz, ilx, ily, ilz);
  }
  if ((rc = fft_exec(ctx, false, a))) return rc;
  hipLaunchKernelGGL((fourier_out_kernel<T, true>), grid, blk, 0, ctx->stream, u, (const C2<T>*)a, (T*)out, ep, mob, cells, inv_n);
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  ctx->last_kernel = "rhs_fourier<CH-3D>(rocfft x9)";
  return PDEOPT_OK;
}

}  // namespace

int rhs_fourier(pdeopt_ctx* ctx, const void* in, void* out) {
  if (ctx->halo) return fail(ctx, PDEOPT_EINVAL, "the pseudo-spectral RHS needs the periodic layout");
  int rc = ensure_plans(ctx);
  if (rc) return rc;
  if (ctx->prob.equation == PDEOPT_EQ_CAHN_HILLIARD_3D)
    return ctx->prob.dtype == PDEOPT_F32 ? rhs_fourier3_t<float>(ctx, in, out) : rhs_fourier3_t<double>(ctx, in, out);
  return ctx->prob.dtype == PDEOPT_F32 ? rhs_fourier_t<float>(ctx, in, out) : rhs_fourier_t<double>(ctx, in, out);
}

int advance_imex(pdeopt_ctx* ctx, double, double dt, int64_t n) {
  if (!ctx->aux[PDEOPT_AUX_IMEX_SYMBOL].dev)
    return fail(ctx, PDEOPT_ESTATE, "IMEX needs the IMEX_SYMBOL aux field (fourier_symbol)");
  if (imex_fused_supported(ctx)) return advance_imex_fused(ctx, dt, n);  // FFTs in LDS, 4 kernels per substep
  if (ctx->imex_per_env)
    return fail(ctx, PDEOPT_EINVAL, "per-environment IMEX scales need the hand-written FFT passes (power-of-two grids 64..1024)");
  int rc = ensure_plans(ctx);
  if (rc) return rc;
  return ctx->prob.dtype == PDEOPT_F32 ? imex_t<float>(ctx, dt, n) : imex_t<double>(ctx, dt, n);
}

int advance_strang(pdeopt_ctx* ctx, double t0, double dt, int64_t n) {
  if (!ctx->aux[PDEOPT_AUX_GPE_A_TERM].dev)
    return fail(ctx, PDEOPT_ESTATE, "Strang splitting needs the GPE_A_TERM aux field");
  if (strang_fused_supported(ctx)) return advance_strang_fused(ctx, t0, dt, n);  // LDS FFTs, fused passes
  int rc = ensure_plans(ctx);
  if (rc) return rc;
  return ctx->prob.dtype == PDEOPT_F32 ? strang_t<float>(ctx, t0, dt, n) : strang_t<double>(ctx, t0, dt, n);
}

void spectral_invalidate(pdeopt_ctx* ctx) {
  if (ctx->spectral) ctx->spectral->mult_kind = -1;
  strang_fused_invalidate(ctx);
}

void spectral_destroy(pdeopt_ctx* ctx) {
  Spectral* sp = ctx->spectral;
  if (!sp) return;
  if (sp->r2c) rocfft_plan_destroy(sp->r2c);
  if (sp->c2r) rocfft_plan_destroy(sp->c2r);
  if (sp->rinfo) rocfft_execution_info_destroy(sp->rinfo);
  if (sp->fwd) rocfft_plan_destroy(sp->fwd);
  if (sp->inv) rocfft_plan_destroy(sp->inv);
  if (sp->info) rocfft_execution_info_destroy(sp->info);
  void* bufs[] = {sp->work, sp->rwork, sp->hbuf, sp->hmult, sp->cbuf, sp->cbuf2, sp->cbuf3, sp->mult, sp->dens, sp->partial, sp->scale};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  delete sp;
  ctx->spectral = nullptr;
}

}  // namespace pdeopt
