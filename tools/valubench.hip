// Calibration micro-benchmark: per-SIMD issue cost of the VALU instruction classes the stencil
// kernels are made of (fp32 fma, packed fp32 fma, transcendental log/rcp, 32-bit integer mul_lo /
// mul_hi), at 1 ... 8 waves per SIMD.  Prints cycles per wave-instruction per SIMD.  The question it settles
// (VERDICT r1 weak #4): does a SIMD issue one wave64 VALU instruction every 2 clocks once several waves are
// resident (the 157.3 TF figure), or every 4 (that figure being the PACKED fp32 rate)?  MIX = 3 fma : 1 log per
// wave: do transcendentals co-issue with the main pipe, or do the costs add?
// build: hipcc -O3 --offload-arch=gfx950 tools/valubench.hip -o gpurun_out/valubench
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kIters = 2048;
constexpr int kChains = 8;  // independent dependency chains per lane

enum { OP_FMA = 0, OP_PK_FMA = 1, OP_LOG = 2, OP_RCP = 3, OP_MUL_LO = 4, OP_MUL_HI = 5, OP_ADD = 6, OP_MIX = 7,
       OP_FMA64 = 8, OP_ADD64 = 9, OP_RCP64 = 10, OP_PK_ADD = 11 };

template <int OP>
__global__ __launch_bounds__(256) void valu_kernel(float* out, float a, float b, unsigned m) {
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
#pragma unroll 1
  for (int it = 0; it < kIters; ++it) {
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
      if constexpr (OP == OP_MIX) {
        if (c % 4 == 3)
          asm volatile("v_log_f32 %0, %0" : "+v"(x[c]));
        else
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
      }
    }
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
int run(const char* name, float* out, int cus, double ghz_guess) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int waves_per_simd : {1, 2, 3, 4, 5, 6, 8}) {
    // one 256-thread block = one wave per SIMD of a CU; waves_per_simd blocks per CU
    const int blocks = cus * waves_per_simd;
    hipLaunchKernelGGL(valu_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.0001f, 2654435761u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r)
      hipLaunchKernelGGL(valu_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.0001f, 2654435761u);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double t = ms * 1e-3 / 5;
    const double instr_per_simd = (double)kIters * kChains * waves_per_simd;
    printf("%-10s waves/SIMD %d: %7.1f us  %5.2f ns per wave-instruction per SIMD (= %4.1f cycles at %.1f GHz)\n", name,
           waves_per_simd, t * 1e6, t * 1e9 / instr_per_simd, t * 1e9 / instr_per_simd * ghz_guess, ghz_guess);
  }
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate * 1e-6;
  printf("%s: %d CUs, clockRate %.2f GHz\n", prop.name, cus, ghz);
  float* out;
  CK(hipMalloc(&out, 4));
  if (run<OP_FMA>("fma_f32", out, cus, ghz)) return 1;
  if (run<OP_ADD>("add_f32", out, cus, ghz)) return 1;
  if (run<OP_PK_FMA>("pk_fma_f32", out, cus, ghz)) return 1;
  if (run<OP_LOG>("log_f32", out, cus, ghz)) return 1;
  if (run<OP_RCP>("rcp_f32", out, cus, ghz)) return 1;
  if (run<OP_MUL_LO>("mul_lo_u32", out, cus, ghz)) return 1;
  if (run<OP_MUL_HI>("mul_hi_u32", out, cus, ghz)) return 1;
  if (run<OP_MIX>("3fma:1log", out, cus, ghz)) return 1;
  if (run<OP_PK_ADD>("pk_add_f32", out, cus, ghz)) return 1;
  if (run<OP_FMA64>("fma_f64", out, cus, ghz)) return 1;
  if (run<OP_ADD64>("add_f64", out, cus, ghz)) return 1;
  if (run<OP_RCP64>("rcp_f64", out, cus, ghz)) return 1;
  return 0;
}
