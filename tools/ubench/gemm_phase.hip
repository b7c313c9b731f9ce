// Micro-benchmark of the fused kernels' "three projections" phase: Q, K, V = X . W^T (64x64x64 each, one 32x32 quadrant per
// wave) with results stored to LDS tiles, one wave per SIMD.  Variants of the code pattern, ns per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kLd = 68, kTile = 64 * kLd;

__device__ __forceinline__ f32x16 gemm_nt(f32x16 acc, const float* As, const float* Bs, int wr, int wc, int r, int h) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float4 a = *reinterpret_cast<const float4*>(&As[(32 * wr + r) * kLd + 8 * c + 4 * h]);
    const float4 b = *reinterpret_cast<const float4*>(&Bs[(32 * wc + r) * kLd + 8 * c + 4 * h]);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  }
  return acc;
}
__device__ __forceinline__ void quad_store(float* Ts, const f32x16& acc, const float* bias, int wr, int wc, int r, int h) {
  const int col = 32 * wc + r;
  const float bv = bias[col];
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) Ts[(32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h) * kLd + col] = acc[reg] + bv;
}

// acc += A . B^T while the PREVIOUS accumulator is written out, two elements per k-step, in the shadow of the MFMAs
__device__ __forceinline__ f32x16 gemm_nt_store_prev(f32x16 acc, const float* As, const float* Bs, float* Ts, const f32x16& prev, const float* bias,
                                                     int wr, int wc, int r, int h) {
  const int col = 32 * wc + r;
  const float bv = bias[col];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float4 a = *reinterpret_cast<const float4*>(&As[(32 * wr + r) * kLd + 8 * c + 4 * h]);
    const float4 b = *reinterpret_cast<const float4*>(&Bs[(32 * wc + r) * kLd + 8 * c + 4 * h]);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    { const int reg = 2 * c; Ts[(32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h) * kLd + col] = prev[reg] + bv; }
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    { const int reg = 2 * c + 1; Ts[(32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h) * kLd + col] = prev[reg] + bv; }
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  return acc;
}

// Transposed product: the WEIGHT rows are the MFMA row operand and the token rows the column operand, so a lane's four
// consecutive accumulator registers are four consecutive FEATURES of one token -> one ds_write_b128 per group.
__device__ __forceinline__ f32x16 gemm_nt_T(f32x16 acc, const float* As, const float* Bs, int wr, int wc, int r, int h) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float4 a = *reinterpret_cast<const float4*>(&As[(32 * wr + r) * kLd + 8 * c + 4 * h]);
    const float4 b = *reinterpret_cast<const float4*>(&Bs[(32 * wc + r) * kLd + 8 * c + 4 * h]);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a.w, acc, 0, 0, 0);
  }
  return acc;
}
__device__ __forceinline__ void quad_store_T(float* Ts, const f32x16& acc, const float* bias, int wr, int wc, int r, int h) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int f0 = 32 * wc + 8 * g + 4 * h;
    const float4 bv = *reinterpret_cast<const float4*>(&bias[f0]);
    *reinterpret_cast<float4*>(&Ts[(32 * wr + r) * kLd + f0]) = make_float4(acc[4 * g] + bv.x, acc[4 * g + 1] + bv.y, acc[4 * g + 2] + bv.z, acc[4 * g + 3] + bv.w);
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  extern __shared__ float lds[];
  float *Xs = lds, *Wq = lds + kTile, *Wk = lds + 2 * kTile, *Wv = lds + 3 * kTile, *Qs = lds + 4 * kTile, *Ks = lds + 5 * kTile, *Vs = lds + 6 * kTile;
  float* cb = lds + 7 * kTile;
  for (int i = threadIdx.x; i < 7 * kTile + 192; i += 256) lds[i] = (float)(i % 13) * 0.01f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5, wr = wave & 1, wc = wave >> 1;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {                 // as in fused_bwd today: three sequential GEMM + store
      { f32x16 a = {0}; a = gemm_nt(a, Xs, Wq, wr, wc, r, h); quad_store(Qs, a, cb, wr, wc, r, h); }
      { f32x16 a = {0}; a = gemm_nt(a, Xs, Wk, wr, wc, r, h); quad_store(Ks, a, cb + 64, wr, wc, r, h); }
      { f32x16 a = {0}; a = gemm_nt(a, Xs, Wv, wr, wc, r, h); quad_store(Vs, a, cb + 128, wr, wc, r, h); }
    } else if (MODE == 1) {          // no stores (upper bound of the pattern)
      f32x16 a = {0};
      a = gemm_nt(a, Xs, Wq, wr, wc, r, h); a = gemm_nt(a, Xs, Wk, wr, wc, r, h); a = gemm_nt(a, Xs, Wv, wr, wc, r, h);
      if (a[0] == 123.f) Qs[threadIdx.x] = a[1];
      asm volatile("" ::: "memory");      // keep the operand loads inside the loop (they are loop-invariant otherwise)
    } else if (MODE == 2) {          // shared A fragment, three accumulators, stores at the end
      f32x16 q = {0}, kk = {0}, v = {0};
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float4 a = *reinterpret_cast<const float4*>(&Xs[(32 * wr + r) * kLd + 8 * c + 4 * h]);
        const float4 bq = *reinterpret_cast<const float4*>(&Wq[(32 * wc + r) * kLd + 8 * c + 4 * h]);
        const float4 bk = *reinterpret_cast<const float4*>(&Wk[(32 * wc + r) * kLd + 8 * c + 4 * h]);
        const float4 bv = *reinterpret_cast<const float4*>(&Wv[(32 * wc + r) * kLd + 8 * c + 4 * h]);
        q = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq.x, q, 0, 0, 0); kk = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bk.x, kk, 0, 0, 0); v = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bv.x, v, 0, 0, 0);
        q = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq.y, q, 0, 0, 0); kk = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bk.y, kk, 0, 0, 0); v = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bv.y, v, 0, 0, 0);
        q = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq.z, q, 0, 0, 0); kk = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bk.z, kk, 0, 0, 0); v = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bv.z, v, 0, 0, 0);
        q = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq.w, q, 0, 0, 0); kk = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bk.w, kk, 0, 0, 0); v = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bv.w, v, 0, 0, 0);
      }
      quad_store(Qs, q, cb, wr, wc, r, h); quad_store(Ks, kk, cb + 64, wr, wc, r, h); quad_store(Vs, v, cb + 128, wr, wc, r, h);
    } else if (MODE == 5) {          // transposed accumulators, b128 stores
      { f32x16 a = {0}; a = gemm_nt_T(a, Xs, Wq, wr, wc, r, h); quad_store_T(Qs, a, cb, wr, wc, r, h); }
      { f32x16 a = {0}; a = gemm_nt_T(a, Xs, Wk, wr, wc, r, h); quad_store_T(Ks, a, cb + 64, wr, wc, r, h); }
      { f32x16 a = {0}; a = gemm_nt_T(a, Xs, Wv, wr, wc, r, h); quad_store_T(Vs, a, cb + 128, wr, wc, r, h); }
    } else if (MODE == 4) {          // each store interleaved into the NEXT GEMM's MFMA stream
      f32x16 q = {0}, kk = {0}, v = {0};
      q = gemm_nt(q, Xs, Wq, wr, wc, r, h);
      __builtin_amdgcn_sched_barrier(0);
      kk = gemm_nt_store_prev(kk, Xs, Wk, Qs, q, cb, wr, wc, r, h);
      v = gemm_nt_store_prev(v, Xs, Wv, Ks, kk, cb + 64, wr, wc, r, h);
      quad_store(Vs, v, cb + 128, wr, wc, r, h);
    } else {                         // sequential GEMMs, each store delayed behind the next GEMM's MFMAs
      f32x16 q = {0}, kk = {0}, v = {0};
      q = gemm_nt(q, Xs, Wq, wr, wc, r, h);
      __builtin_amdgcn_sched_barrier(0);
      kk = gemm_nt(kk, Xs, Wk, wr, wc, r, h);
      quad_store(Qs, q, cb, wr, wc, r, h);
      __builtin_amdgcn_sched_barrier(0);
      v = gemm_nt(v, Xs, Wv, wr, wc, r, h);
      quad_store(Ks, kk, cb + 64, wr, wc, r, h);
      __builtin_amdgcn_sched_barrier(0);
      quad_store(Vs, v, cb + 128, wr, wc, r, h);
    }
    __syncthreads();
  }
  out[blockIdx.x * 256 + threadIdx.x] = Qs[threadIdx.x] + Ks[threadIdx.x] + Vs[threadIdx.x];
}
template <int MODE>
void run(const char* name) {
  float* d; hipMalloc(&d, 4 << 20);
  const int iters = 5000;
  const size_t lds = (7 * kTile + 192) * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), lds, 0, d, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), lds, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-70s %.2f ns per MFMA (%.2f us per phase)\n", name, ms * 1e6 / ((double)iters * 96), ms * 1e3 / iters);
  hipFree(d);
}
int main() {
  run<0>("0: three sequential GEMM + store (fused_bwd today)");
  run<1>("1: no stores");
  run<2>("2: shared A fragment, three accumulators, stores at the end");
  run<3>("3: sequential GEMMs, stores delayed behind the next GEMM");
  run<4>("4: stores interleaved into the next GEMM's MFMA stream");
  run<5>("5: transposed accumulators (4 consecutive features per lane), ds_write_b128");
  return 0;
}
