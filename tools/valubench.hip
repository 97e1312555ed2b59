// Calibration micro-benchmark (round 3 rewrite): issue cost of the VALU instruction classes the stencil
// kernels are made of, in SHADER CYCLES per wave64 instruction per SIMD, at 1 ... 8 resident waves per SIMD.
//
// Timed INSIDE the kernel: every wave stamps s_memtime (shader-clock ticks) and s_memrealtime (100 MHz) around its
// loop, so neither a launch nor an assumed clock enters the figure (round 2 divided a launch-inclusive wall time of
// 60-300 us bodies by an assumed 2.4 GHz).  Bodies run >= 1 ms.  Printed per class and occupancy:
//   cycles per wave-instruction per SIMD = median over waves of  d(s_memtime) / (instructions of one wave x waves per SIMD)
//   the clock the chip held              = d(s_memtime) / d(s_memrealtime) x 100 MHz
// The question: does a SIMD-32 issue one wave64 VALU instruction every 2 clocks once several waves are resident
// (MI355X_MICROARCH.md: "v_fma_f32 (wave64) 2 cyc; one wave alone: 4"), or every 3.5-4 (what round 2 read off wall
// times)?  MIX = 3 fma : 1 log per wave: do transcendental costs add?
// build: hipcc -O3 --offload-arch=gfx950 tools/valubench.hip -o tools/valubench.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kChains = 8;  // independent dependency chains per lane (a chain's next instruction is 8 issues away)

enum { OP_FMA = 0, OP_PK_FMA = 1, OP_LOG = 2, OP_RCP = 3, OP_MUL_LO = 4, OP_MUL_HI = 5, OP_ADD = 6, OP_MIX = 7,
       OP_FMA64 = 8, OP_ADD64 = 9, OP_RCP64 = 10, OP_PK_ADD = 11, OP_CNDMASK = 12, OP_MOV = 13 };

struct Stamp {
  unsigned long long cycles, real;
};

template <int OP>
__global__ __launch_bounds__(256) void valu_kernel(float* out, Stamp* stamps, int iters, float a, float b, unsigned m) {
  float x[kChains];
  f32x2 p[kChains];
  unsigned u[kChains];
  double d[kChains];
  const double da = a, db = b;
#pragma unroll
  for (int c = 0; c < kChains; ++c) {
    d[c] = 1.0 + 0.001 * (threadIdx.x + c);
    x[c] = 1.0f + 0.001f * (threadIdx.x + c);
    p[c] = f32x2{x[c], x[c] + 0.5f};
    u[c] = threadIdx.x * 2654435761u + c;
  }
  const f32x2 pa{a, a}, pb{b, b};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
      for (int c = 0; c < kChains; ++c) {
        if constexpr (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
        if constexpr (OP == OP_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[c]) : "v"(b));
        if constexpr (OP == OP_PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[c]) : "v"(pa), "v"(pb));
        if constexpr (OP == OP_LOG) asm volatile("v_log_f32 %0, %0" : "+v"(x[c]));
        if constexpr (OP == OP_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[c]));
        if constexpr (OP == OP_MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[c]) : "v"(m));
        if constexpr (OP == OP_MUL_HI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[c]) : "v"(m));
        if constexpr (OP == OP_FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[c]) : "v"(da), "v"(db));
        if constexpr (OP == OP_ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(db));
        if constexpr (OP == OP_RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[c]));
        if constexpr (OP == OP_PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[c]) : "v"(pb));
        if constexpr (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(b));
        if constexpr (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(x[c]) : "v"(b));
        if constexpr (OP == OP_MIX) {
          if (c % 4 == 3)
            asm volatile("v_log_f32 %0, %0" : "+v"(x[c]));
          else
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    stamps[w] = Stamp{t1 - t0, r1 - r0};
  }
  float s = 0;
#pragma unroll
  for (int c = 0; c < kChains; ++c) s += x[c] + p[c][0] + p[c][1] + (float)u[c] + (float)d[c];
  if (s == 12345.678f) out[0] = s;  // keep the chains alive
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
  for (int waves_per_simd : {1, 2, 3, 4, 6, 8}) {
    // one 256-thread block = one wave per SIMD of a CU; waves_per_simd blocks per CU.  The body is sized for >= 1 ms at
    // the nominal cost, the same total per SIMD at every occupancy.
    const int blocks = cus * waves_per_simd;
    const int per_iter = 4 * kChains;
    const int iters = (int)(2.4e9 * 1.2e-3 / (nominal_cycles * per_iter * waves_per_simd)) + 1;
    hipLaunchKernelGGL(valu_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, stamps, iters / 8 + 1, 1.0001f, 0.0001f, 2654435761u);
    CK(hipDeviceSynchronize());  // warm: clocks ramp up
    hipLaunchKernelGGL(valu_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, stamps, iters, 1.0001f, 0.0001f, 2654435761u);
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
    const double insts = (double)iters * per_iter;  // per wave
    const double med = cyc[nw / 2];
    printf("%-10s waves/SIMD %d: %7.0f us body, %5.2f cycles per wave-instruction per SIMD (median; p10 %5.2f p90 %5.2f), one wave's "
           "own rate %5.2f cycles per instruction, clock held %.2f GHz\n",
           name, waves_per_simd, med / (ghz[nw / 2] * 1e3), med / (insts * waves_per_simd), cyc[nw / 10] / (insts * waves_per_simd),
           cyc[nw * 9 / 10] / (insts * waves_per_simd), med / insts, ghz[nw / 2]);
  }
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("%s: %d CUs, clockRate %.2f GHz (nominal); cycles below are s_memtime ticks measured in-kernel\n", prop.name, cus,
         prop.clockRate * 1e-6);
  float* out;
  Stamp* stamps;
  CK(hipMalloc(&out, 4));
  CK(hipMalloc(&stamps, sizeof(Stamp) * cus * 8 * 4));
  if (run<OP_FMA>("fma_f32", out, stamps, cus, 4)) return 1;
  if (run<OP_ADD>("add_f32", out, stamps, cus, 4)) return 1;
  if (run<OP_MOV>("mov_b32", out, stamps, cus, 4)) return 1;
  if (run<OP_CNDMASK>("cndmask", out, stamps, cus, 4)) return 1;
  if (run<OP_PK_FMA>("pk_fma_f32", out, stamps, cus, 6)) return 1;
  if (run<OP_PK_ADD>("pk_add_f32", out, stamps, cus, 6)) return 1;
  if (run<OP_LOG>("log_f32", out, stamps, cus, 9)) return 1;
  if (run<OP_RCP>("rcp_f32", out, stamps, cus, 9)) return 1;
  if (run<OP_MUL_LO>("mul_lo_u32", out, stamps, cus, 5)) return 1;
  if (run<OP_MUL_HI>("mul_hi_u32", out, stamps, cus, 5)) return 1;
  if (run<OP_MIX>("3fma:1log", out, stamps, cus, 5)) return 1;
  if (run<OP_FMA64>("fma_f64", out, stamps, cus, 5)) return 1;
  if (run<OP_ADD64>("add_f64", out, stamps, cus, 5)) return 1;
  if (run<OP_RCP64>("rcp_f64", out, stamps, cus, 17)) return 1;
  return 0;
}
