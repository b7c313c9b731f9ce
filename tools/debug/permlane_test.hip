// v_permlane32_swap semantics check (fused_fwd32.hip: xhalf_sum)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* o) {
  const float v = (float)threadIdx.x;
  {
    const auto s = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    o[threadIdx.x] = __builtin_bit_cast(float, s[0]);
    o[64 + threadIdx.x] = __builtin_bit_cast(float, s[1]);
  }
  {
    float v2 = v;
    asm volatile("" : "+v"(v2));
    const auto s = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v2), false, false);
    o[128 + threadIdx.x] = __builtin_bit_cast(float, s[0]);
    o[192 + threadIdx.x] = __builtin_bit_cast(float, s[1]);
  }
  {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    o[384 + threadIdx.x] = a + b;
  }
  {
    const float w = 100.f + v;
    const auto s = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, w), false, false);
    o[256 + threadIdx.x] = __builtin_bit_cast(float, s[0]);
    o[320 + threadIdx.x] = __builtin_bit_cast(float, s[1]);
  }
}
int main() {
  float* d; hipMalloc(&d, 448 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[448]; hipMemcpy(h, d, 448 * 4, hipMemcpyDeviceToHost);
  for (int i : {0, 1, 31, 32, 33, 63}) printf("lane %2d: same-reg s0 %3.0f s1 %3.0f | opaque copy s0 %3.0f s1 %3.0f | (v, 100+v) s0 %3.0f s1 %3.0f | asm sum %3.0f (want %d)\n", i, h[i], h[64 + i], h[128 + i], h[192 + i], h[256 + i], h[320 + i], h[384 + i], 2 * (i % 32) + 32);
  return 0;
}
