// bf16 matrix-core rate (v_mfma_f32_32x32x16_bf16) and whether f32 VALU work hides behind it (it does not behind the f32 MFMA).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int N>
__global__ void k(float* out, int iters, float a, float b) {
  f32x16 acc = {0};
  bf16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(a + i); y[i] = (__bf16)(b - i); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < N; ++j) v[j % 8] = v[j % 8] * a + b;
    }
  }
  float s = 0;
  for (int j = 0; j < 16; ++j) s += acc[j];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int N>
void run(int threads) {
  float* d; hipMalloc(&d, 4 << 20);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<N>, dim3(256), dim3(threads), 0, 0, d, 10, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<N>, dim3(256), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 16;
  printf("threads/CU=%d  %2d v_fma per MFMA: %.2f ns per bf16 MFMA per wave (%.0f TFLOP/s)\n", threads, N, ms * 1e6 / n,
         n * 256 * (threads / 64) * 32.0 * 32 * 16 * 2 / (ms * 1e-3) / 1e12);
  hipFree(d);
}
int main() { run<0>(256); run<2>(256); run<4>(256); run<8>(256); run<0>(512); run<4>(512); return 0; }
