// semantics check of the gfx9 whole-wave DPP shifts used for neighbour exchange
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
  const int lane = threadIdx.x;
  const float x = (float)lane;
  // old value (first arg) is what a lane with no source keeps
  const float shr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -1.0f), __builtin_bit_cast(int, x), 0x138, 0xf, 0xf, false));
  const float shl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -1.0f), __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, false));
  out[lane] = shr;
  out[64 + lane] = shl;
}
int main() {
  float* d; hipMalloc((void**)&d, 128 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("wave_shr:1  lane0=%g lane1=%g lane15=%g lane16=%g lane32=%g lane63=%g\n", h[0], h[1], h[15], h[16], h[32], h[63]);
  printf("wave_shl:1  lane0=%g lane1=%g lane15=%g lane16=%g lane31=%g lane62=%g lane63=%g\n", h[64], h[65], h[79], h[80], h[95], h[126], h[127]);
  return 0;
}
