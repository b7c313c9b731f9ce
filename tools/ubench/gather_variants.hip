// Embedding-row gather variants on HBM-resident tables (development micro-benchmark behind DESIGN.md §5 "gather").
//   v0  one row per 16-lane group, one 16-B load per lane and chunk, plain stores (the round-1 gather_rows_kernel)
//   v1  R rows in flight per 16-lane group (R independent loads before the first store), plain stores
//   v2  as v1 with non-temporal stores
//   v3  read-only: rows are summed into a register and ONE float per 16 rows is written (the read ceiling of the access pattern)
//   v4  as v2 with int32 ids
// Row bytes = 4d (d = 64: 256 B, d = 256: 1 KB); ids uniform random; table sizes 1 M and 16 M rows.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NCH, int R, int MODE, typename IdT>
__global__ __launch_bounds__(256) void gather_k(const IdT* __restrict__ ids, int64_t T, const float* __restrict__ table,
                                                float* __restrict__ rows, float* __restrict__ sink) {
  constexpr int d = 64 * NCH;
  const int s = threadIdx.x & 15;
  const int64_t g = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);     // 16-lane group index
  const int64_t t0 = g * R;
  float4 v[R][NCH];
  int64_t id[R];
#pragma unroll
  for (int r = 0; r < R; ++r) id[r] = (t0 + r < T) ? (int64_t)ids[t0 + r] : 0;
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < NCH; ++c) v[r][c] = *reinterpret_cast<const float4*>(table + id[r] * d + 64 * c + 4 * s);
  if (MODE == 3) {
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc += v[r][c].x + v[r][c].y + v[r][c].z + v[r][c].w;
    if (acc == 123.456f) sink[g] = acc;      // practically never taken; keeps the loads alive
    return;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (t0 + r >= T) break;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4* dst = reinterpret_cast<f4*>(rows + (t0 + r) * d + 64 * c + 4 * s);
      const f4 val = {v[r][c].x, v[r][c].y, v[r][c].z, v[r][c].w};
      if (MODE == 2) __builtin_nontemporal_store(val, dst);
      else *dst = val;
    }
  }
}

template <int NCH, int R, int MODE, typename IdT>
static void run(const char* name, const IdT* ids, int64_t T, const float* table, int64_t N, float* rows, float* sink) {
  constexpr int d = 64 * NCH;
  const int64_t groups = (T + R - 1) / R;
  dim3 grid((unsigned)((groups + 15) / 16));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((gather_k<NCH, R, MODE, IdT>), grid, dim3(256), 0, 0, ids, T, table, rows, sink);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 10;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gather_k<NCH, R, MODE, IdT>), grid, dim3(256), 0, 0, ids, T, table, rows, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  const double rd = (double)T * (4.0 * d + sizeof(IdT)) / (us * 1e-6);
  const double tot = (double)T * ((MODE == 3 ? 4.0 : 8.0) * d + sizeof(IdT)) / (us * 1e-6);
  printf("%-28s N=%9lld d=%3d T=%9lld R=%d: %8.1f us  read %6.0f GB/s = %4.1f %% of 8 TB/s   read+write %6.0f GB/s\n", name, (long long)N, d,
         (long long)T, R, us, rd / 1e9, rd / 8e12 * 100, tot / 1e9);
}

template <int NCH>
static void suite(int64_t N, int64_t T) {
  constexpr int d = 64 * NCH;
  float *table, *rows, *sink;
  int64_t* ids; int32_t* ids32;
  CK(hipMalloc(&table, (size_t)(N + 1) * d * 4));
  CK(hipMalloc(&rows, (size_t)T * d * 4));
  CK(hipMalloc(&sink, (size_t)T * 4));
  CK(hipMalloc(&ids, (size_t)T * 8));
  CK(hipMalloc(&ids32, (size_t)T * 4));
  CK(hipMemset(table, 0, (size_t)(N + 1) * d * 4));
  std::vector<int64_t> h(T);
  std::vector<int32_t> h32(T);
  uint64_t s = 88172645463325252ull;
  for (int64_t i = 0; i < T; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    h[i] = 1 + (int64_t)(s % (uint64_t)N);
    h32[i] = (int32_t)h[i];
  }
  CK(hipMemcpy(ids, h.data(), (size_t)T * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(ids32, h32.data(), (size_t)T * 4, hipMemcpyHostToDevice));
  run<NCH, 1, 0, int64_t>("v0 1 row/group plain", ids, T, table, N, rows, sink);
  run<NCH, 2, 0, int64_t>("v1 2 rows in flight", ids, T, table, N, rows, sink);
  run<NCH, 4, 0, int64_t>("v1 4 rows in flight", ids, T, table, N, rows, sink);
  run<NCH, 1, 2, int64_t>("v2 1 row nt stores", ids, T, table, N, rows, sink);
  run<NCH, 2, 2, int64_t>("v2 2 rows nt stores", ids, T, table, N, rows, sink);
  run<NCH, 4, 2, int64_t>("v2 4 rows nt stores", ids, T, table, N, rows, sink);
  if (NCH == 1) run<NCH, 8, 2, int64_t>("v2 8 rows nt stores", ids, T, table, N, rows, sink);
  run<NCH, 1, 3, int64_t>("v3 read-only 1 row", ids, T, table, N, rows, sink);
  run<NCH, 2, 3, int64_t>("v3 read-only 2 rows", ids, T, table, N, rows, sink);
  run<NCH, 4, 3, int64_t>("v3 read-only 4 rows", ids, T, table, N, rows, sink);
  if (NCH == 1) run<NCH, 8, 3, int64_t>("v3 read-only 8 rows", ids, T, table, N, rows, sink);
  run<NCH, 4, 2, int32_t>("v4 4 rows nt int32 ids", ids32, T, table, N, rows, sink);
  run<NCH, 4, 3, int32_t>("v4 read-only 4 rows int32", ids32, T, table, N, rows, sink);
  CK(hipFree(table)); CK(hipFree(rows)); CK(hipFree(sink)); CK(hipFree(ids)); CK(hipFree(ids32));
  printf("\n");
}

int main() {
  suite<1>(1 << 20, 1 << 24);       // 256 MB table, d = 64
  suite<1>(1 << 24, 1 << 24);       // 4 GB table, d = 64 (far beyond the 256 MiB Infinity Cache)
  suite<4>(1 << 20, 1 << 22);       // C5: 1 GB table, d = 256
  suite<2>(1 << 22, 1 << 23);       // 2 GB table, d = 128
  return 0;
}
