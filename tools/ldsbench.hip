// LDS bank behaviour of gfx950, measured directly: clocks per wave64 ds_read of 4 / 8 / 16 bytes per lane as a
// function of the lane stride (in dwords).  One wave per CU (no contention from other waves), a dependent chain
// of reads is NOT used: 16 independent reads per iteration so that the LDS pipe, not the latency, is timed.
// Reading the table: a stride whose cost equals stride 1's is conflict-free; cost x k = k-way conflict.
// build: hipcc -O3 --offload-arch=gfx950 tools/ldsbench.hip -o tools/ldsbench.bin
#include <hip/hip_runtime.h>

#include <cstdio>

constexpr int kIters = 4096;
constexpr int kUnroll = 16;

template <int BYTES>
__global__ __launch_bounds__(256) void lds_kernel(float* out, int stride_dwords, int rot) {
  __shared__ __attribute__((aligned(16))) float s[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) s[i] = (float)i;
  __syncthreads();
  // lane address (bytes), naturally aligned by the caller's choice of stride; the reads of an iteration go to
  // kUnroll different offsets (multiples of 64 dwords: same banks) held in registers -- no address arithmetic in
  // the timed loop
  int addr[kUnroll];
#pragma unroll
  for (int u = 0; u < kUnroll; ++u) addr[u] = (((threadIdx.x & 63) * stride_dwords + u * 64 * (1 + rot)) & 8191) * 4;
  float acc = 0.f;
#pragma unroll 1
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      if constexpr (BYTES == 4) {
        float v;
        asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr[u]));
        acc += v;
      } else if constexpr (BYTES == 8) {
        float2 v;
        asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr[u]));
        acc += v.x;
      } else {
        float4 v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr[u]));
        acc += v.x;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (acc == 123.456f) out[0] = acc;
}

#define CK(x)                                                         \
  do {                                                                \
    hipError_t e = (x);                                               \
    if (e != hipSuccess) {                                            \
      printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); \
      return 1;                                                       \
    }                                                                 \
  } while (0)

template <int BYTES>
int run(float* out, int cus, double ghz) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int align = BYTES / 4;
  for (int stride : {1, 2, 3, 4, 5, 8, 9, 16, 17, 32, 33, 64, 65}) {
    if (stride % align) continue;  // keep the access naturally aligned
    hipLaunchKernelGGL(lds_kernel<BYTES>, dim3(cus), dim3(256), 0, 0, out, stride, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(lds_kernel<BYTES>, dim3(cus), dim3(256), 0, 0, out, stride, 0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double ns = ms * 1e6 / ((double)kIters * kUnroll * 4);  // 4 waves per CU share the LDS
    printf("ds_read_b%-3d lane stride %2d dwords: %6.2f ns = %5.1f clk per wave instruction per CU\n", BYTES * 8, stride, ns, ns * ghz);
  }
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const double ghz = prop.clockRate * 1e-6;
  printf("%s: %d CUs, clockRate %.2f GHz; 4 waves per CU, %d reads per wave and iteration\n", prop.name, prop.multiProcessorCount, ghz, kUnroll);
  float* out;
  CK(hipMalloc(&out, 4));
  if (run<4>(out, prop.multiProcessorCount, ghz)) return 1;
  if (run<8>(out, prop.multiProcessorCount, ghz)) return 1;
  if (run<16>(out, prop.multiProcessorCount, ghz)) return 1;
  return 0;
}
