// Calibration micro-benchmark: what HBM rate do plain float4 streaming kernels reach on this GPU
// for the read/write stream mixes of the RK4 stage kernels (1R2W, 3R2W, 2R1W) and for a copy?
// build: hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o gpurun_out/membench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NR, int NW, bool NT>
__global__ __launch_bounds__(256) void stream_kernel(const f32x4* __restrict__ r0,
                                                     const f32x4* __restrict__ r1,
                                                     const f32x4* __restrict__ r2, f32x4* __restrict__ w0,
                                                     f32x4* __restrict__ w1, size_t n, int per_thread) {
  size_t base = ((size_t)blockIdx.x * per_thread) * 256 + threadIdx.x;
  for (int k = 0; k < per_thread; ++k) {
    const size_t i = base + (size_t)k * 256;
    if (i >= n) return;
    f32x4 v = r0[i];
    if (NR > 1) v += r1[i];
    if (NR > 2) v += r2[i];
    if (NT) {
      __builtin_nontemporal_store(v, &w0[i]);
      if (NW > 1) __builtin_nontemporal_store(v * 2.0f, &w1[i]);
    } else {
      w0[i] = v;
      if (NW > 1) w1[i] = v * 2.0f;
    }
  }
}

#define CK(x)                                                            \
  do {                                                                   \
    hipError_t e = (x);                                                  \
    if (e != hipSuccess) {                                               \
      printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__);    \
      return 1;                                                          \
    }                                                                    \
  } while (0)

template <int NR, int NW, bool NT>
int run(const char* name, f32x4** buf, size_t n, int per_thread) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const size_t threads = (n + per_thread - 1) / per_thread;
  const int blocks = (int)((threads + 255) / 256);
  for (int it = 0; it < 3; ++it)
    hipLaunchKernelGGL((stream_kernel<NR, NW, NT>), dim3(blocks), dim3(256), 0, 0, buf[0], buf[1], buf[2],
                       buf[3], buf[4], n, per_thread);
  CK(hipEventRecord(a));
  const int reps = 20;
  for (int it = 0; it < reps; ++it)
    hipLaunchKernelGGL((stream_kernel<NR, NW, NT>), dim3(blocks), dim3(256), 0, 0, buf[0], buf[1], buf[2],
                       buf[3], buf[4], n, per_thread);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  const double bytes = (double)(NR + NW) * n * 16.0 * reps;
  printf("%-28s per_thread=%d blocks=%d  %8.1f us/launch  %7.1f GB/s\n", name, per_thread, blocks,
         1e3 * ms / reps, bytes / (ms * 1e-3) / 1e9);
  return 0;
}

int main(int argc, char** argv) {
  // MiB per array (default 128 = 32 envs x 1024^2 floats: HBM streams; 64 = BASELINE config 2's 64 x 512^2 state, whose
  // read + write arrays stay resident in the 256 MiB Infinity Cache: the ceiling of the whole-substep kernels)
  const size_t mib = argc > 1 ? (size_t)atoi(argv[1]) : 128;
  const size_t n = mib * 1024 * 1024 / 16;
  f32x4* buf[5];
  for (auto& p : buf) {
    CK(hipMalloc((void**)&p, n * 16));
    CK(hipMemset(p, 0, n * 16));
  }
  for (int pt : {1, 4, 16}) {
    run<1, 1, false>("copy 1R1W", buf, n, pt);
    run<1, 2, false>("stage1 1R2W", buf, n, pt);
    run<3, 2, false>("stage2/3 3R2W", buf, n, pt);
    run<2, 1, false>("stage4 2R1W", buf, n, pt);
    run<3, 2, true>("stage2/3 3R2W nt-store", buf, n, pt);
    run<1, 2, true>("stage1 1R2W nt-store", buf, n, pt);
  }
  return 0;
}
