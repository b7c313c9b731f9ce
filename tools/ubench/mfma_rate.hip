// Sustained v_mfma_f32_32x32x2_f32 issue rate on every CU at once (development micro-benchmark).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = {0};
  long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = (float)(t1 - t0);
}
template <int NACC>
void run(int blocks, int threads) {
  float* d; hipMalloc(&d, 4 << 20);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, d, 100, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 8 * NACC;           // MFMAs per wave
  const double waves = (double)blocks * threads / 64;
  printf("nacc=%d blocks=%d threads=%d: %.2f ns per MFMA per wave, %.1f TFLOP/s total\n", NACC, blocks, threads, ms * 1e6 / n, n * waves * 4096 / (ms * 1e-3) / 1e12);
  hipFree(d);
}
int main() {
  run<1>(256, 256); run<4>(256, 256); run<4>(256, 512); run<4>(512, 256); run<1>(256, 64); run<4>(1, 256);
  return 0;
}
