// MFMA rate when the weight operand is STREAMED FROM L2 in fragment-major order (one coalesced 1 KB load per wave and
// 4 MFMAs) by independent single-wave workgroups -- the question behind the wave-independent fused forward:
// can 1 / 2 / 3 decoupled waves per SIMD keep the matrix pipe busy without any LDS staging or workgroup barrier?
//   A operand (tokens) in registers for the whole run, rolling window of WIN float4 weight fragments per lane, two accumulators.
// Weights: 34 matrices of 64 x 64 floats (544 KB: L2-resident, shared by every wave), each stored as [wc][c][lane] float4.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMats = 34;
constexpr int kFragPerMat = 16;           // (wc, c) pairs; one float4 per lane each

template <int WIN>
__global__ __launch_bounds__(64) void k(const f32x4* __restrict__ W, float* out, int tiles) {
  extern __shared__ float lds_pad[];      // occupancy control only
  const int lane = threadIdx.x;
  f32x4 afr[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) afr[c] = (f32x4){0.01f * lane, 0.02f, 0.03f * c, 0.04f};
  f32x16 acc0 = {0}, acc1 = {0};
  f32x4 win[WIN];
  const f32x4* p = W + lane;
  const int total = tiles * kMats * kFragPerMat;
#pragma unroll
  for (int i = 0; i < WIN; ++i) win[i] = p[(int64_t)i * 64];
  int nxt = WIN;
  for (int f = 0; f < total; f += WIN) {
#pragma unroll
    for (int i = 0; i < WIN; ++i) {
      const f32x4 b = win[i];
      const int c = i & 7;
      // refill this slot for the fragment WIN ahead (wraps inside the 544 KB weight set)
      int q = nxt + i;
      q = q % (kMats * kFragPerMat);
      win[i] = p[(int64_t)q * 64];
      if ((i >> 3) & 1) {
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, afr[c].x, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, afr[c].y, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, afr[c].z, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, afr[c].w, acc1, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, afr[c].x, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, afr[c].y, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, afr[c].z, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, afr[c].w, acc0, 0, 0, 0);
      }
    }
    nxt = (nxt + WIN) % (kMats * kFragPerMat);
  }
  float s = 0;
  for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
  out[(int64_t)blockIdx.x * 64 + lane] = s;
}

template <int WIN>
void run(int waves_per_simd, int lds_bytes) {
  f32x4* W; float* d;
  const size_t wbytes = (size_t)kMats * kFragPerMat * 64 * sizeof(f32x4);
  hipMalloc(&W, wbytes); hipMemset(W, 0, wbytes);
  const int nwg = 256 * 4 * waves_per_simd;
  hipMalloc(&d, (size_t)nwg * 64 * 4);
  const int tiles = 40;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<WIN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<WIN>, dim3(nwg), dim3(64), lds_bytes, 0, W, d, 2);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<WIN>, dim3(nwg), dim3(64), lds_bytes, 0, W, d, tiles);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_wave = (double)tiles * kMats * kFragPerMat * 4;
  // per SIMD: waves_per_simd waves share the pipe; 27 ns per MFMA is the pipe's own rate (mfma_rate.hip)
  printf("window %2d float4, %d wave(s)/SIMD (LDS %3d KB/wg): %.2f ns per MFMA per SIMD  (%.1f %% of the 27 ns pipe rate), L2 stream %.2f TB/s\n", WIN,
         waves_per_simd, lds_bytes >> 10, ms * 1e6 / (mfma_per_wave * waves_per_simd), 100.0 * 27.0 / (ms * 1e6 / (mfma_per_wave * waves_per_simd)),
         (double)nwg * mfma_per_wave / 4 * 1024 / (ms * 1e-3) / 1e12);
  hipFree(W); hipFree(d);
}

int main() {
  // LDS per single-wave workgroup sets the occupancy: 160 KB / (4 x waves per SIMD)
  run<8>(1, 36 << 10); run<8>(2, 18 << 10); run<8>(3, 12 << 10);
  run<16>(1, 36 << 10); run<16>(2, 18 << 10); run<16>(3, 12 << 10);
  run<4>(2, 18 << 10);
  return 0;
}
