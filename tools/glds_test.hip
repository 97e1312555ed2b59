// Semantics probe for global_load_lds_dwordx4 on gfx950: where does lane l's 16 bytes land, and what
// happens under a partial exec mask?   build: hipcc -O3 --offload-arch=gfx950 tools/glds_test.hip -o variants/glds_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void glds16(const void* src, void* lds_wave_base) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#endif
}

__global__ __launch_bounds__(256) void k(const float* in, float* out, int nactive) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x;
  for (int i = tid; i < 2048; i += 256) s[i] = -1.0f;
  __syncthreads();
  float* dst = s + 4;  // 16-byte offset like sU0
  const int wave_base = tid & ~63;
  if (tid < nactive) glds16(in + tid * 4, dst + wave_base * 4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  for (int i = tid; i < 2048; i += 256) out[i] = s[i];
}

int main() {
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  float *d_in, *d_out;
  hipMalloc(&d_in, 4096);
  hipMalloc(&d_out, 8192);
  hipMemcpy(d_in, h.data(), 4096, hipMemcpyHostToDevice);
  for (int nactive : {256, 200, 48}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 8192, 0, d_in, d_out, nactive);
    std::vector<float> o(2048);
    hipMemcpy(o.data(), d_out, 8192, hipMemcpyDeviceToHost);
    int bad = 0, first_bad = -1;
    for (int i = 0; i < 2048; ++i) {
      const int e = i - 4;  // element index relative to dst
      const float want = (e >= 0 && e < nactive * 4) ? (float)e : -1.0f;
      if (o[i] != want) {
        if (first_bad < 0) first_bad = i;
        ++bad;
      }
    }
    printf("nactive %d: %d mismatches (first at %d: got %.0f)\n", nactive, bad, first_bad, first_bad >= 0 ? o[first_bad] : 0.f);
    if (bad) {
      for (int i = 0; i < 24; ++i) printf("%.0f ", o[i]);
      printf("\n");
    }
  }
  return 0;
}
