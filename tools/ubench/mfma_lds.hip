// MFMA rate when the operands come from LDS (NT tile GEMM inner loop of the fused kernels), one wave per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kLd = 68;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  extern __shared__ float lds[];
  float* As = lds; float* Bs = lds + 64 * kLd;
  for (int i = threadIdx.x; i < 2 * 64 * kLd; i += 256) lds[i] = (float)(i % 7) * 0.01f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5, wr = wave & 1, wc = wave >> 1;
  f32x16 acc = {0};
  long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {          // NT: both operands as b128 row reads
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float4 a = *reinterpret_cast<const float4*>(&As[(32 * wr + r) * kLd + 8 * c + 4 * h]);
        const float4 b = *reinterpret_cast<const float4*>(&Bs[(32 * wc + r) * kLd + 8 * c + 4 * h]);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
      }
    } else if (MODE == 1) {   // NN: A rows, B columns (scalar reads)
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float4 a = *reinterpret_cast<const float4*>(&As[(32 * wr + r) * kLd + 8 * c + 4 * h]);
        const float* wp = &Bs[(8 * c + 4 * h) * kLd + 32 * wc + r];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wp[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wp[kLd], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wp[2 * kLd], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wp[3 * kLd], acc, 0, 0, 0);
      }
    } else {                  // TN: both scalar column reads
#pragma unroll
      for (int m = 0; m < 32; ++m) {
        const int t = 2 * m + h;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[t * kLd + 32 * wr + r], Bs[t * kLd + 32 * wc + r], acc, 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  long long t1 = wall_clock64();
  float s = 0;
  for (int j = 0; j < 16; ++j) s += acc[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = (float)(t1 - t0);
}
template <int MODE>
void run(const char* name) {
  float* d; hipMalloc(&d, 4 << 20);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 2 * 64 * kLd * 4, 0, d, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 2 * 64 * kLd * 4, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%s: %.2f ns per MFMA per wave\n", name, ms * 1e6 / ((double)iters * 32));
  hipFree(d);
}
int main() { run<0>("NT (b128 + b128)"); run<1>("NN (b128 + 4 x b32)"); run<2>("TN (b32 + b32)"); return 0; }
