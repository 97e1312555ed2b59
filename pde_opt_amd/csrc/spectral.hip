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

#include "common.hpp"

namespace pdeopt {

struct Spectral {
  bool setup = false;
  rocfft_plan fwd = nullptr, inv = nullptr;
  rocfft_execution_info info = nullptr;
  void* work = nullptr;
  size_t work_bytes = 0;
  void* cbuf = nullptr;    // complex work field (IMEX)
  void* mult = nullptr;    // complex spectral multiplier, shared [nx][ny]
  void* dens = nullptr;    // real |psi0|^2 (Strang)
  double* partial = nullptr;  // [batch][kNormBlocks]
  double* scale = nullptr;    // [batch]
  // cache key of `mult`
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
  size_t lengths[2];
  size_t dims = 0;
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
                                                       double* __restrict__ partial) {
  const int b = blockIdx.y;
  C2<T>* pb = psi + (int64_t)b * cells;
  const T* db = dens + (int64_t)b * cells;
  const T* vb = pot ? pot + (int64_t)b * pot_stride : nullptr;
  const T kk = ep[b].gpe_k;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cells; i += (int64_t)gridDim.x * 256) {
    const T w = (vb ? vb[i] : T(0)) + kk * db[i];
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
  const size_t cells = (size_t)ctx->prob.nx * ctx->prob.ny;
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

template <typename T>
int imex_t(pdeopt_ctx* ctx, double dt, int64_t n) {
  Spectral& sp = *ctx->spectral;
  const pdeopt_problem& p = ctx->prob;
  const int64_t cells = (int64_t)p.nx * p.ny;
  const int64_t total = cells * p.batch;
  int rc;
  if ((rc = ensure_buffer(ctx, &ctx->TA, ctx->total_bytes))) return rc;
  if ((rc = ensure_buffer(ctx, &sp.cbuf, (size_t)total * 2 * sizeof(T)))) return rc;
  if (sp.mult_kind != 0 || sp.mult_dt != dt || sp.mult_A != ctx->imex_A) {
    // multiplier 1 / ((1 + A dt symbol) N): solvers.py:62-63 with ifft's 1/N folded in
    std::vector<std::complex<double>> sym;
    if ((rc = fetch_complex_aux(ctx, PDEOPT_AUX_IMEX_SYMBOL, sym))) return rc;
    const double inv_n = 1.0 / (double)cells;
    for (auto& s : sym) s = inv_n / (1.0 + ctx->imex_A * dt * s);
    if ((rc = upload_mult<T>(ctx, sym))) return rc;
    sp.mult_kind = 0;
    sp.mult_dt = dt;
    sp.mult_A = ctx->imex_A;
  }
  for (int64_t s = 0; s < n; ++s) {
    if ((rc = launch_rhs(ctx, ctx->Y, ctx->TA, 0.0))) return rc;
    hipLaunchKernelGGL(embed_real_kernel<T>, dim3(grid_for(total)), dim3(256), 0, ctx->stream,
                       (const T*)ctx->TA, (C2<T>*)sp.cbuf, total);
    if ((rc = fft_exec(ctx, true, sp.cbuf))) return rc;
    hipLaunchKernelGGL((spectral_mul_kernel<T, false>), dim3(grid_for(cells), p.batch), dim3(256), 0,
                       ctx->stream, (C2<T>*)sp.cbuf, (const C2<T>*)sp.mult, nullptr, cells);
    if ((rc = fft_exec(ctx, false, sp.cbuf))) return rc;
    hipLaunchKernelGGL(imex_update_kernel<T>, dim3(grid_for(total)), dim3(256), 0, ctx->stream,
                       (T*)ctx->Y, (const C2<T>*)sp.cbuf, (T)dt, total);
  }
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  ctx->last_kernel += "+imex_rocfft_c2c";
  return PDEOPT_OK;
}

template <typename T>
int strang_t(pdeopt_ctx* ctx, double dt, int64_t n) {
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
    hipLaunchKernelGGL(strang_b_kernel<T>, dim3(kNormBlocks, p.batch), dim3(256), 0, ctx->stream,
                       (C2<T>*)ctx->Y, (const T*)sp.dens, (const T*)pot.dev, pot_stride,
                       (const EnvParams<T>*)ctx->env_params_dev, (T)tau.real(), (T)tau.imag(), cells,
                       sp.partial);
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

}  // namespace

int advance_imex(pdeopt_ctx* ctx, double, double dt, int64_t n) {
  if (!ctx->aux[PDEOPT_AUX_IMEX_SYMBOL].dev)
    return fail(ctx, PDEOPT_ESTATE, "IMEX needs the IMEX_SYMBOL aux field (fourier_symbol)");
  int rc = ensure_plans(ctx);
  if (rc) return rc;
  return ctx->prob.dtype == PDEOPT_F32 ? imex_t<float>(ctx, dt, n) : imex_t<double>(ctx, dt, n);
}

int advance_strang(pdeopt_ctx* ctx, double, double dt, int64_t n) {
  if (!ctx->aux[PDEOPT_AUX_GPE_A_TERM].dev)
    return fail(ctx, PDEOPT_ESTATE, "Strang splitting needs the GPE_A_TERM aux field");
  int rc = ensure_plans(ctx);
  if (rc) return rc;
  return ctx->prob.dtype == PDEOPT_F32 ? strang_t<float>(ctx, dt, n) : strang_t<double>(ctx, dt, n);
}

void spectral_destroy(pdeopt_ctx* ctx) {
  Spectral* sp = ctx->spectral;
  if (!sp) return;
  if (sp->fwd) rocfft_plan_destroy(sp->fwd);
  if (sp->inv) rocfft_plan_destroy(sp->inv);
  if (sp->info) rocfft_execution_info_destroy(sp->info);
  void* bufs[] = {sp->work, sp->cbuf, sp->mult, sp->dens, sp->partial, sp->scale};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  delete sp;
  ctx->spectral = nullptr;
}

}  // namespace pdeopt
