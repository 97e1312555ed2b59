// Calibration micro-benchmark (round 3; tools/ldsbench.hip is the round-2 stride sweep at 4 waves per CU): issue cost of the LDS access patterns the stencil kernels are made of, in SHADER
// CYCLES per wave64 instruction per COMPUTE UNIT (the LDS is one per CU), at 4 ... 32 resident waves per CU.
// Timed inside the kernel with s_memtime like tools/valubench.hip.  The questions:
//   * what does a ds_read_b128 cost (1024 bytes per wave: 8 cycles at 128 B/clk)?  a ds_write_b128?
//   * what does the scalar neighbour read of the stencil kernels cost -- ds_read_b32 with a 16-byte lane stride, a
//     4-way bank conflict -- against a conflict-free ds_read_b32?
//   * do LDS instructions and VALU instructions of the same waves overlap (MIX: one ds_read_b128 per 8 FMAs)?
// With these, SQ_INSTS_LDS-style instruction counts of a kernel turn into LDS-busy cycles, the way valubench's
// 2 cycles per VALU instruction turn SQ_INSTS_VALU into VALU-busy cycles.
// build: hipcc -O3 --offload-arch=gfx950 tools/lds_issue_bench.hip -o tools/lds_issue_bench.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { OP_R128 = 0, OP_R32 = 1, OP_R32_S16 = 2, OP_R64 = 3, OP_W128 = 4, OP_W32 = 5, OP_R128_FMA8 = 6, OP_R32_S16_FMA4 = 7, OP_R128_ROW = 8 };

struct Stamp {
  unsigned long long cycles, real;
};

constexpr int kBatch = 8;  // independent LDS instructions in flight per wave before the wait

template <int OP>
__global__ __launch_bounds__(256) void lds_kernel(float* out, Stamp* stamps, int iters, float a, float b) {
  __shared__ __attribute__((aligned(16))) float lds[256 * 4 + 64];
  for (int i = threadIdx.x; i < 256 * 4 + 64; i += 256) lds[i] = 0.001f * i;
  __syncthreads();
  // byte addresses per lane.  16-byte lane stride (vector per lane); 4-byte lane stride (conflict-free scalars)
  const unsigned a16 = (unsigned)(size_t)(lds) + threadIdx.x * 16;
  const unsigned a4 = (unsigned)(size_t)(lds) + threadIdx.x * 4;
  const unsigned a8 = (unsigned)(size_t)(lds) + threadIdx.x * 8;
  f32x4 v[kBatch];
  f32x2 w2[kBatch];
  float s[kBatch];
  float x[8];
#pragma unroll
  for (int c = 0; c < kBatch; ++c) {
    v[c] = f32x4{1.f, 2.f, 3.f, 4.f};
    w2[c] = f32x2{1.f, 2.f};
    s[c] = 1.f;
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) x[c] = 1.0f + 0.001f * (threadIdx.x + c);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < kBatch; ++c) {
      if constexpr (OP == OP_R128) asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(v[c]) : "v"(a16));
      if constexpr (OP == OP_R128_ROW) asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(v[c]) : "v"(a16));
      if constexpr (OP == OP_R32) asm volatile("ds_read_b32 %0, %1 offset:0" : "=v"(s[c]) : "v"(a4));
      if constexpr (OP == OP_R32_S16) asm volatile("ds_read_b32 %0, %1 offset:12" : "=v"(s[c]) : "v"(a16));
      if constexpr (OP == OP_R64) asm volatile("ds_read_b64 %0, %1 offset:0" : "=v"(w2[c]) : "v"(a8));
      if constexpr (OP == OP_W128) asm volatile("ds_write_b128 %0, %1 offset:0" : : "v"(a16), "v"(v[c]));
      if constexpr (OP == OP_W32) asm volatile("ds_write_b32 %0, %1 offset:0" : : "v"(a4), "v"(s[c]));
      if constexpr (OP == OP_R128_FMA8) {
        asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(v[c]) : "v"(a16));
#pragma unroll
        for (int q = 0; q < 8; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[q]) : "v"(a), "v"(b));
      }
      if constexpr (OP == OP_R32_S16_FMA4) {
        asm volatile("ds_read_b32 %0, %1 offset:12" : "=v"(s[c]) : "v"(a16));
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[q]) : "v"(a), "v"(b));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    const int wv = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    stamps[wv] = Stamp{t1 - t0, r1 - r0};
  }
  float acc = 0;
#pragma unroll
  for (int c = 0; c < kBatch; ++c) acc += v[c][0] + v[c][3] + w2[c][1] + s[c];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc += x[c];
  if (acc == 12345.678f) out[0] = acc + lds[threadIdx.x];
}

#define CK(x)                                                         \
  do {                                                                \
    hipError_t e = (x);                                               \
    if (e != hipSuccess) {                                            \
      printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); \
      return 1;                                                       \
    }                                                                 \
  } while (0)

template <int OP>
int run(const char* name, float* out, Stamp* stamps, int cus, double nominal_cycles) {
  for (int wgs_per_cu : {1, 2, 4, 6, 8}) {
    // one 256-thread block = 4 waves, one per SIMD of a CU
    const int blocks = cus * wgs_per_cu;
    const int iters = (int)(2.0e9 * 1.0e-3 / (nominal_cycles * kBatch * 4 * wgs_per_cu)) + 1;
    hipLaunchKernelGGL(lds_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, stamps, iters / 8 + 1, 1.0001f, 0.0001f);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(lds_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, stamps, iters, 1.0001f, 0.0001f);
    CK(hipDeviceSynchronize());
    const int nw = blocks * 4;
    std::vector<Stamp> h(nw);
    CK(hipMemcpy(h.data(), stamps, sizeof(Stamp) * nw, hipMemcpyDeviceToHost));
    std::vector<double> cyc(nw), ghz(nw);
    for (int i = 0; i < nw; ++i) {
      cyc[i] = (double)h[i].cycles;
      ghz[i] = (double)h[i].cycles / (double)h[i].real * 0.1;
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(ghz.begin(), ghz.end());
    const double insts = (double)iters * kBatch;  // LDS instructions per wave
    const double med = cyc[nw / 2];
    printf("%-16s waves/CU %2d: %7.0f us body, %6.2f cycles per wave-LDS-instruction per CU (median; p10 %6.2f p90 %6.2f), "
           "clock held %.2f GHz\n",
           name, 4 * wgs_per_cu, med / (ghz[nw / 2] * 1e3), med / (insts * 4 * wgs_per_cu), cyc[nw / 10] / (insts * 4 * wgs_per_cu),
           cyc[nw * 9 / 10] / (insts * 4 * wgs_per_cu), ghz[nw / 2]);
  }
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("%s: %d CUs; cycles are s_memtime ticks measured in-kernel; every wave keeps %d LDS instructions in flight\n", prop.name, cus,
         kBatch);
  float* out;
  Stamp* stamps;
  CK(hipMalloc(&out, 4));
  CK(hipMalloc(&stamps, sizeof(Stamp) * cus * 8 * 4));
  if (run<OP_R128>("read_b128", out, stamps, cus, 8)) return 1;
  if (run<OP_R128_ROW>("read_b128+16", out, stamps, cus, 8)) return 1;
  if (run<OP_R64>("read_b64", out, stamps, cus, 4)) return 1;
  if (run<OP_R32>("read_b32", out, stamps, cus, 2)) return 1;
  if (run<OP_R32_S16>("read_b32 stride16", out, stamps, cus, 8)) return 1;
  if (run<OP_W128>("write_b128", out, stamps, cus, 8)) return 1;
  if (run<OP_W32>("write_b32", out, stamps, cus, 2)) return 1;
  if (run<OP_R128_FMA8>("read_b128+8fma", out, stamps, cus, 16)) return 1;
  if (run<OP_R32_S16_FMA4>("b32s16+4fma", out, stamps, cus, 8)) return 1;
  return 0;
}
