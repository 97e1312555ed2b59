// Per-environment reductions of the state (reward helpers of PDEEnv.step, pde_env.py:309:
// e.g. reward_function = np.var, notebooks/test_pde_env.ipynb:57).  Deterministic: fixed
// partition into chunks, fp64 partials, combined on the host in a fixed order.
#include <algorithm>

#include "common.hpp"

namespace pdeopt {

namespace {

constexpr int kChunks = 64;  // partial sums per environment

struct Partial {
  double sum, sumsq, mn, mx, bad;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}

// grid (kChunks, batch), block 256.  shift[b] is subtracted before squaring (two-pass variance).
template <typename T>
__global__ __launch_bounds__(256) void reduce_kernel(const T* __restrict__ y, int64_t env_elems,
                                                     const double* __restrict__ shift,
                                                     Partial* __restrict__ out) {
  const int b = blockIdx.y;
  const T* p = y + (int64_t)b * env_elems;
  const double sh = shift ? shift[b] : 0.0;
  const int64_t per = (env_elems + kChunks - 1) / kChunks;
  const int64_t lo = (int64_t)blockIdx.x * per;
  const int64_t hi = lo + per < env_elems ? lo + per : env_elems;
  double s = 0, q = 0, mn = INFINITY, mx = -INFINITY, bad = 0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const double v = (double)p[i];
    if (isfinite(v)) {
      s += v;
      q += (v - sh) * (v - sh);
      mn = fmin(mn, v);
      mx = fmax(mx, v);
    } else {
      bad += 1.0;
    }
  }
  __shared__ double sh_s[4], sh_q[4], sh_mn[4], sh_mx[4], sh_bad[4];
  s = wave_sum(s);
  q = wave_sum(q);
  mn = wave_min(mn);
  mx = wave_max(mx);
  bad = wave_sum(bad);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sh_s[w] = s; sh_q[w] = q; sh_mn[w] = mn; sh_mx[w] = mx; sh_bad[w] = bad;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Partial r;
    r.sum = sh_s[0] + sh_s[1] + sh_s[2] + sh_s[3];
    r.sumsq = sh_q[0] + sh_q[1] + sh_q[2] + sh_q[3];
    r.mn = fmin(fmin(sh_mn[0], sh_mn[1]), fmin(sh_mn[2], sh_mn[3]));
    r.mx = fmax(fmax(sh_mx[0], sh_mx[1]), fmax(sh_mx[2], sh_mx[3]));
    r.bad = sh_bad[0] + sh_bad[1] + sh_bad[2] + sh_bad[3];
    out[(int64_t)b * kChunks + blockIdx.x] = r;
  }
}

int run_pass(pdeopt_ctx* ctx, const double* shift_dev, std::vector<Partial>& host) {
  const int batch = ctx->prob.batch;
  const size_t need = (size_t)batch * kChunks * sizeof(Partial);
  if (ctx->red_cap < need) {
    if (ctx->red_dev) (void)hipFree(ctx->red_dev);
    ctx->red_dev = nullptr;
    PDEOPT_HIP_CHECK(ctx, hipMalloc((void**)&ctx->red_dev, need));
    ctx->red_cap = need;
  }
  dim3 grid(kChunks, batch), block(256);
  if (ctx->prob.dtype == PDEOPT_F32) {
    hipLaunchKernelGGL(reduce_kernel<float>, grid, block, 0, ctx->stream, (const float*)ctx->Y,
                       (int64_t)ctx->env_elems, shift_dev, (Partial*)ctx->red_dev);
  } else {
    hipLaunchKernelGGL(reduce_kernel<double>, grid, block, 0, ctx->stream, (const double*)ctx->Y,
                       (int64_t)ctx->env_elems, shift_dev, (Partial*)ctx->red_dev);
  }
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  host.resize((size_t)batch * kChunks);
  PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(host.data(), ctx->red_dev, need, hipMemcpyDeviceToHost,
                                       ctx->stream));
  PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return PDEOPT_OK;
}

}  // namespace

// uint8 frames for image observations (the reference declares Box(0, 255, (1, *points), uint8),
// pde_env.py:118-126): q = rint(clip((x - lo) / (hi - lo), 0, 1) * 255), 4 cells per thread
template <typename T>
__global__ void observe_u8_kernel(const T* __restrict__ y, uint32_t* __restrict__ out, int64_t n4, T lo,
                                  T scale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t w = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      T v = (y[4 * i + e] - lo) * scale;
      v = v < T(0) ? T(0) : (v > T(255) ? T(255) : v);  // NaN -> 0 through the first comparison chain
      if (!(v == v)) v = T(0);
      w |= (uint32_t)__builtin_rint((double)v) << (8 * e);
    }
    out[i] = w;
  }
}

int observe_u8(pdeopt_ctx* ctx, double lo, double hi, int env_first, int env_count, void* host_out) {
  if (ctx->halo) return fail(ctx, PDEOPT_EINVAL, "observations are not available in the padded layout");
  if (!(hi > lo)) return fail(ctx, PDEOPT_EINVAL, "observe_u8 needs hi > lo");
  const int64_t n = (int64_t)ctx->env_elems * env_count;
  if (n % 4) return fail(ctx, PDEOPT_EINVAL, "observe_u8 needs a multiple of 4 cells");
  int rc = ensure_buffer(ctx, &ctx->obs_dev, ctx->env_elems * (size_t)ctx->prob.batch);
  if (rc) return rc;
  const int64_t n4 = n / 4;
  const int blocks = (int)std::min<int64_t>((n4 + 255) / 256, 4096);
  const size_t off = (size_t)env_first * ctx->env_elems;
  if (ctx->prob.dtype == PDEOPT_F32)
    hipLaunchKernelGGL(observe_u8_kernel<float>, dim3(blocks), dim3(256), 0, ctx->stream,
                       (const float*)ctx->Y + off, (uint32_t*)ctx->obs_dev, n4, (float)lo, (float)(255.0 / (hi - lo)));
  else
    hipLaunchKernelGGL(observe_u8_kernel<double>, dim3(blocks), dim3(256), 0, ctx->stream,
                       (const double*)ctx->Y + off, (uint32_t*)ctx->obs_dev, n4, lo, 255.0 / (hi - lo));
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  if (host_out)  // nullptr: the frames stay in ctx->obs_dev (pdeopt_observe_u8_device)
    PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(host_out, ctx->obs_dev, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return PDEOPT_OK;
}

// ------------------------------------------------------------------------------------------
// Quantised vortices of a GPE wavefunction: phase circulation around every grid plaquette
// (pde_opt/rl_utils.py:19-84 detect_vortices).  One thread per cell evaluates the four corner
// phases and the four wrapped edge differences of ITS plaquette in the reference's order
//   circulation(i,j) = dth_x(i,j) + dth_y(i,j+1) - dth_x(i+1,j) - dth_y(i,j)
// with dth_x the wrapped difference along axis 1 and dth_y along axis 0 (rl_utils.py:48-56), rounds
// it to an integer winding, applies the tol / density masks and adds to three per-environment
// counters {cells with winding != 0, sum winding, sum |winding|}.
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wrap_to_pi(T x) {
  // (x + pi) % (2 pi) - pi with numpy's sign-of-divisor modulo: result in [-pi, pi)  (rl_utils.py:14-16)
  const T two_pi = T(6.283185307179586476925286766559);
  const T pi = T(3.141592653589793238462643383279);
  T r = fmod(x + pi, two_pi);
  if (r < T(0)) r += two_pi;
  return r - pi;
}

template <typename T>
__global__ __launch_bounds__(256) void vortex_kernel(const T* __restrict__ psi, int32_t* __restrict__ winding,
                                                     long long* __restrict__ counts, int nx, int ny, T amp_thresh,
                                                     T tol) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
  int n_int = 0;
  if (i < nx && j < ny) {
    const T* p = psi + (int64_t)b * nx * ny * 2;
    const int i1 = (i + 1 == nx) ? 0 : i + 1, j1 = (j + 1 == ny) ? 0 : j + 1;
    auto at = [&](int ii, int jj, T& re, T& im) {
      re = p[((int64_t)ii * ny + jj) * 2];
      im = p[((int64_t)ii * ny + jj) * 2 + 1];
    };
    T r00, m00, r01, m01, r10, m10, r11, m11;
    at(i, j, r00, m00);
    at(i, j1, r01, m01);
    at(i1, j, r10, m10);
    at(i1, j1, r11, m11);
    const T t00 = atan2(m00, r00), t01 = atan2(m01, r01), t10 = atan2(m10, r10), t11 = atan2(m11, r11);
    const T dx_ij = wrap_to_pi<T>(t01 - t00);    // dth_x(i, j)
    const T dy_ij = wrap_to_pi<T>(t10 - t00);    // dth_y(i, j)
    const T dy_ij1 = wrap_to_pi<T>(t11 - t01);   // dth_y(i, j+1)
    const T dx_i1j = wrap_to_pi<T>(t11 - t10);   // dth_x(i+1, j)
    const T circ = dx_ij + dy_ij1 - dx_i1j - dy_ij;
    const T n_float = circ / T(6.283185307179586476925286766559);
    n_int = (int)rint(n_float);
    if (!(fabs(n_float) >= tol)) n_int = 0;
    if (amp_thresh > T(0)) {
      const T rho = T(0.25) * ((r00 * r00 + m00 * m00) + (r10 * r10 + m10 * m10) + (r01 * r01 + m01 * m01) +
                                (r11 * r11 + m11 * m11));
      if (!(rho >= amp_thresh)) n_int = 0;
    }
    if (winding) winding[((int64_t)b * nx + i) * ny + j] = n_int;
  }
  // block totals -> three atomics per block (vortex cells are rare: most blocks add nothing)
  int c0 = n_int != 0, c1 = n_int, c2 = n_int < 0 ? -n_int : n_int;
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) {
    c0 += __shfl_down(c0, s, 64);
    c1 += __shfl_down(c1, s, 64);
    c2 += __shfl_down(c2, s, 64);
  }
  if ((threadIdx.x & 63) == 0 && (c0 | c2)) {
    atomicAdd((unsigned long long*)&counts[b * 3 + 0], (unsigned long long)(long long)c0);
    atomicAdd((unsigned long long*)&counts[b * 3 + 1], (unsigned long long)(long long)c1);
    atomicAdd((unsigned long long*)&counts[b * 3 + 2], (unsigned long long)(long long)c2);
  }
}

int detect_vortices(pdeopt_ctx* ctx, double amp_thresh, double tol, int env_first, int env_count,
                    int32_t* host_winding, int64_t* host_counts) {
  const pdeopt_problem& p = ctx->prob;
  const size_t cells = (size_t)p.nx * p.ny;
  const size_t wbytes = host_winding ? cells * env_count * sizeof(int32_t) : 0;
  const size_t need = wbytes + (size_t)env_count * 3 * sizeof(long long);
  if (ctx->vort_dev && ctx->vort_cap < need) {
    PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(ctx->vort_dev);
    ctx->vort_dev = nullptr;
  }
  int rc = ensure_buffer(ctx, &ctx->vort_dev, need);
  if (rc) return rc;
  if (ctx->vort_cap < need) ctx->vort_cap = need;
  long long* counts = reinterpret_cast<long long*>(ctx->vort_dev);
  int32_t* wdev = host_winding ? reinterpret_cast<int32_t*>(counts + (size_t)env_count * 3) : nullptr;
  PDEOPT_HIP_CHECK(ctx, hipMemsetAsync(counts, 0, (size_t)env_count * 3 * sizeof(long long), ctx->stream));
  dim3 grid((p.ny + 63) / 64, (p.nx + 3) / 4, env_count), block(256);
  if (grid.y > 65535u || grid.z > 65535u) return fail(ctx, PDEOPT_EINVAL, "grid too large for the vortex kernel");
  const size_t off = (size_t)env_first * ctx->env_elems;
  if (p.dtype == PDEOPT_F32)
    hipLaunchKernelGGL(vortex_kernel<float>, grid, block, 0, ctx->stream, (const float*)ctx->Y + off, wdev, counts,
                       p.nx, p.ny, (float)amp_thresh, (float)tol);
  else
    hipLaunchKernelGGL(vortex_kernel<double>, grid, block, 0, ctx->stream, (const double*)ctx->Y + off, wdev, counts,
                       p.nx, p.ny, amp_thresh, tol);
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  static_assert(sizeof(long long) == sizeof(int64_t), "counter width");
  PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(host_counts, counts, (size_t)env_count * 3 * sizeof(int64_t),
                                       hipMemcpyDeviceToHost, ctx->stream));
  if (host_winding)
    PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(host_winding, wdev, wbytes, hipMemcpyDeviceToHost, ctx->stream));
  PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return PDEOPT_OK;
}

// Point probes: the state at a list of grid cells, for every environment of a range (observation / reward
// functions that read a few sensors: n_probes * comps doubles per environment cross PCIe instead of the field).
template <typename T>
__global__ void probe_kernel(const T* __restrict__ y, const int64_t* __restrict__ cell, int n_probes, int comps,
                             int64_t env_elems, int env_first, double* __restrict__ out, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % comps);
  const int q = (int)((i / comps) % n_probes);
  const int64_t e = i / ((int64_t)comps * n_probes);
  out[i] = (double)y[(env_first + e) * env_elems + cell[q] * comps + c];
}

int probe_state(pdeopt_ctx* ctx, const int32_t* cells, int n_probes, int env_first, int env_count, double* host_out) {
  const pdeopt_problem& p = ctx->prob;
  const int nz = p.nz > 1 ? p.nz : 1;
  const int nd = nz > 1 ? 3 : 2;
  std::vector<int64_t> flat((size_t)n_probes);
  for (int q = 0; q < n_probes; ++q) {
    const int i = cells[(size_t)q * nd], j = cells[(size_t)q * nd + 1], k = nd == 3 ? cells[(size_t)q * nd + 2] : 0;
    if (i < 0 || i >= p.nx || j < 0 || j >= p.ny || k < 0 || k >= nz)
      return fail(ctx, PDEOPT_EINVAL, "probe %d at (%d, %d, %d) outside the %d x %d x %d grid", q, i, j, k, p.nx, p.ny, nz);
    flat[q] = ((int64_t)i * p.ny + j) * nz + k;
  }
  const int64_t total = (int64_t)env_count * n_probes * ctx->comps;
  const size_t need = (size_t)n_probes * sizeof(int64_t) + (size_t)total * sizeof(double);
  if (ctx->red_cap < need) {
    if (ctx->red_dev) (void)hipFree(ctx->red_dev);
    ctx->red_dev = nullptr;
    ctx->red_cap = 0;
    PDEOPT_HIP_CHECK(ctx, hipMalloc((void**)&ctx->red_dev, need));
    ctx->red_cap = need;
  }
  double* out_dev = ctx->red_dev;
  int64_t* cell_dev = reinterpret_cast<int64_t*>(ctx->red_dev + total);
  PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(cell_dev, flat.data(), (size_t)n_probes * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
  const int blocks = (int)((total + 255) / 256);
  if (p.dtype == PDEOPT_F32)
    hipLaunchKernelGGL(probe_kernel<float>, dim3(blocks), dim3(256), 0, ctx->stream, (const float*)ctx->Y, cell_dev, n_probes,
                       ctx->comps, (int64_t)ctx->env_elems, env_first, out_dev, total);
  else
    hipLaunchKernelGGL(probe_kernel<double>, dim3(blocks), dim3(256), 0, ctx->stream, (const double*)ctx->Y, cell_dev, n_probes,
                       ctx->comps, (int64_t)ctx->env_elems, env_first, out_dev, total);
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(host_out, out_dev, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // flat[] and host_out are the caller's from here on
  return PDEOPT_OK;
}

int reduce_state(pdeopt_ctx* ctx, int op, double* out) {
  const int batch = ctx->prob.batch;
  const double n = (double)ctx->env_elems;
  std::vector<Partial> part;
  int rc = run_pass(ctx, nullptr, part);
  if (rc) return rc;
  std::vector<double> mean(batch);
  for (int b = 0; b < batch; ++b) {
    double s = 0, q = 0, mn = INFINITY, mx = -INFINITY, bad = 0;
    for (int c = 0; c < kChunks; ++c) {
      const Partial& p = part[(size_t)b * kChunks + c];
      s += p.sum; q += p.sumsq; bad += p.bad;
      mn = p.mn < mn ? p.mn : mn;
      mx = p.mx > mx ? p.mx : mx;
    }
    mean[b] = s / n;
    switch (op) {
      case PDEOPT_RED_MEAN: out[b] = bad > 0 ? NAN : s / n; break;
      case PDEOPT_RED_MIN: out[b] = bad > 0 ? NAN : mn; break;
      case PDEOPT_RED_MAX: out[b] = bad > 0 ? NAN : mx; break;
      case PDEOPT_RED_SUMSQ: out[b] = bad > 0 ? NAN : q; break;
      case PDEOPT_RED_NONFINITE: out[b] = bad; break;
      case PDEOPT_RED_VAR: break;
      default: return fail(ctx, PDEOPT_EINVAL, "unknown reduction %d", op);
    }
  }
  if (op == PDEOPT_RED_VAR) {
    // second pass around the mean, like np.var
    if (!ctx->red_mean_dev)
      PDEOPT_HIP_CHECK(ctx, hipMalloc((void**)&ctx->red_mean_dev, sizeof(double) * 65536));
    if (batch > 65536) return fail(ctx, PDEOPT_EINVAL, "batch too large for variance");
    PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(ctx->red_mean_dev, mean.data(), sizeof(double) * batch,
                                         hipMemcpyHostToDevice, ctx->stream));
    rc = run_pass(ctx, ctx->red_mean_dev, part);
    if (rc) return rc;
    for (int b = 0; b < batch; ++b) {
      double q = 0, bad = 0;
      for (int c = 0; c < kChunks; ++c) {
        q += part[(size_t)b * kChunks + c].sumsq;
        bad += part[(size_t)b * kChunks + c].bad;
      }
      out[b] = bad > 0 ? NAN : q / n;
    }
  }
  return PDEOPT_OK;
}

}  // namespace pdeopt
