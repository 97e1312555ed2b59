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
  PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(host_out, ctx->obs_dev, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
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
