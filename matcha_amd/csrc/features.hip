// Feature construction for the adj front end from contact maps.
//
//   pixels_to_adj   process.py:144-172   cooler pixels (bin1, bin2, count) -> intra / inter adjacency [N, N] float64,
//                                        both triangles, NaN counts skipped, pixels of unmapped bins skipped
//   corrcoef_block  main.py:571-575      np.corrcoef of one chromosome's intra block: float32 in, float64 arithmetic
//                                        (row means, centred X X^T on the f64 MFMA, / (n - 1), / sd_i, / sd_j, clip), float32 out,
//                                        NaN -> 0
//   zscore_rows     Modules.py:146-152   per row of the inter matrix: z-score (ddof 0) of its strictly positive entries, in
//                                        place, NaN -> 0
//
// Rooflines.  pixels_to_adj: 24 B read per pixel + two 8 B atomic adds into a matrix that is far larger than L2 at 100 kb
// bins (N^2 * 8 B = 7.4 GB) -- HBM atomic-rate bound.  corrcoef_block: 2 n^3 flops on the f64 MFMA (v_mfma_f64_16x16x4), LDS
// tiled 64 x 64 per workgroup; n <= 2491 at 100 kb bins, so the whole genome is ~0.3 TFLOP.  zscore_rows: three passes over
// a row that stays in L2 (<= 121 kB): 4 B read + 4 B written per entry from HBM.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.hpp"

namespace matcha {
namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block_sum for 256 threads (4 waves); every thread gets the total
__device__ __forceinline__ double block_sum_f64(double v, double* sh) {
  v = wave_sum_f64(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// ---- pixels -> adjacency ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pixels_to_adj_kernel(const int64_t* __restrict__ bin1, const int64_t* __restrict__ bin2,
                                                            const double* __restrict__ count, int64_t n_pixels,
                                                            const int32_t* __restrict__ index2node, int64_t n_index,
                                                            const int32_t* __restrict__ node2chrom, int64_t N, double* __restrict__ intra,
                                                            double* __restrict__ inter) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pixels; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b1 = bin1[i], b2 = bin2[i];
    if (b1 < 0 || b1 >= n_index || b2 < 0 || b2 >= n_index) continue;
    const int n1 = index2node[b1], n2 = index2node[b2];               // node ids start at 1; < 1 = bin of a chromosome not in chrom_list
    if (n1 < 1 || n2 < 1 || n1 > N || n2 > N) continue;
    const double c = count[i];
    if (c != c) continue;                                              // process.py:161 `if not np.isnan(count)`
    double* dst = node2chrom[n1] == node2chrom[n2] ? intra : inter;
    const int64_t r = n1 - 1, s = n2 - 1;
    atomicAdd(&dst[r * N + s], c);                                     // :167-168 / :170-171: both triangles (the diagonal twice)
    atomicAdd(&dst[s * N + r], c);
  }
}

// ---- corrcoef -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void row_mean_kernel(const float* __restrict__ adj, int64_t ld, int n, double* __restrict__ mean) {
  const int row = blockIdx.x;
  const float* p = adj + (int64_t)row * ld;
  double s = 0.0;
  for (int k = threadIdx.x; k < n; k += 64) s += (double)p[k];
  s = wave_sum_f64(s);
  if (threadIdx.x == 0) mean[row] = s / (double)n;
}

constexpr int kCT = 64;        // output tile
constexpr int kCK = 16;        // contraction chunk
constexpr int kCLd = kCK + 1;  // LDS row stride in doubles (odd: the 16 rows a fragment load touches land in different banks)

// C[i][j] = sum_k (x[i][k] - m[i]) (x[j][k] - m[j]); workgroup = one 64 x 64 tile, wave w = the 32 x 32 quadrant (w >> 1, w & 1)
// as 2 x 2 MFMA tiles of v_mfma_f64_16x16x4: lane l supplies A[l % 16][l / 16] and B[l / 16][l % 16], and holds
// D[4 r + l / 16][l % 16] in register r (measured: tools/ubench/mfma_f64_layout.hip; the f32 16x16x4 uses 4 (l / 16) + r).  Only tiles with tj >= ti are computed; the mirror is written from the same values.
__global__ __launch_bounds__(256) void centred_gram_kernel(const float* __restrict__ adj, int64_t ld, int n, const double* __restrict__ mean,
                                                           double* __restrict__ C) {
  const int ti = blockIdx.y, tj = blockIdx.x;
  if (tj < ti) return;
  __shared__ double As[kCT * kCLd], Bs[kCT * kCLd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int l16 = lane & 15, lk = lane >> 4;
  f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (f64x4){0.0, 0.0, 0.0, 0.0};
  // staging: thread t loads row t / 4, columns 4 (t % 4) .. + 3 of both operand tiles
  const int sr = tid >> 2, sc = (tid & 3) * 4;
  const int ra = ti * kCT + sr, rb = tj * kCT + sr;
  const double ma = ra < n ? mean[ra] : 0.0, mb = rb < n ? mean[rb] : 0.0;
  const float* pa = adj + (int64_t)(ra < n ? ra : 0) * ld;
  const float* pb = adj + (int64_t)(rb < n ? rb : 0) * ld;
  for (int k0 = 0; k0 < n; k0 += kCK) {
    float va[4], vb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + sc + u;
      const int kc = k < n ? k : n - 1;
      va[u] = pa[kc];
      vb[u] = pb[kc];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool in = k0 + sc + u < n;
      As[sr * kCLd + sc + u] = (in && ra < n) ? (double)va[u] - ma : 0.0;
      Bs[sr * kCLd + sc + u] = (in && rb < n) ? (double)vb[u] - mb : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < kCK; kk += 4) {
      double af[2], bf[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) af[a] = As[(32 * wi + 16 * a + l16) * kCLd + kk + lk];
#pragma unroll
      for (int b = 0; b < 2; ++b) bf[b] = Bs[(32 * wj + 16 * b + l16) * kCLd + kk + lk];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ti * kCT + 32 * wi + 16 * a + 4 * r + lk;
        const int j = tj * kCT + 32 * wj + 16 * b + l16;
        if (i < n && j < n) {
          C[(int64_t)i * n + j] = acc[a][b][r];
          if (tj > ti) C[(int64_t)j * n + i] = acc[a][b][r];
        }
      }
}

__global__ __launch_bounds__(256) void corr_normalise_kernel(const double* __restrict__ C, int n, float* __restrict__ out) {
#pragma clang fp contract(off)
  const int64_t total = (int64_t)n * n;
  const double fact = 1.0 / (double)(n - 1);                           // np.cov: c *= true_divide(1, n - ddof)
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / n), j = (int)(e - (int64_t)i * n);
    const double sdi = sqrt(C[(int64_t)i * n + i] * fact), sdj = sqrt(C[(int64_t)j * n + j] * fact);
    double v = C[e] * fact;
    v = v / sdi;                                                       // np.corrcoef: c /= stddev[:, None]; c /= stddev[None, :]
    v = v / sdj;
    v = v > 1.0 ? 1.0 : (v < -1.0 ? -1.0 : v);                         // np.clip(c, -1, 1); NaN compares false and stays NaN
    const float f = (float)v;
    out[e] = f != f ? 0.f : f;                                         // main.py:575 temp[np.isnan(temp)] = 0
  }
}

// ---- row z-score ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void zscore_rows_kernel(float* __restrict__ M, int64_t cols) {
  __shared__ double sh[4];
  float* row = M + (int64_t)blockIdx.x * cols;
  double s = 0.0, c = 0.0;
  for (int64_t k = threadIdx.x; k < cols; k += 256) {
    const float v = row[k];
    if (v > 0.f) { s += (double)v; c += 1.0; }
  }
  s = block_sum_f64(s, sh);
  c = block_sum_f64(c, sh);
  if (c == 0.0) {                                                      // no positive entry: only NaN -> 0 remains (Modules.py:152)
    for (int64_t k = threadIdx.x; k < cols; k += 256) { const float v = row[k]; if (v != v) row[k] = 0.f; }
    return;
  }
  const double mean = s / c;
  double q = 0.0;
  for (int64_t k = threadIdx.x; k < cols; k += 256) {
    const float v = row[k];
    if (v > 0.f) { const double d = (double)v - mean; q += d * d; }
  }
  q = block_sum_f64(q, sh);
  const double sd = sqrt(q / c);
  for (int64_t k = threadIdx.x; k < cols; k += 256) {
    const float v = row[k];
    float z = v;
    if (v > 0.f) z = (float)(((double)v - mean) / sd);                 // sd == 0 (a single positive entry): 0 / 0 = NaN -> 0
    row[k] = z != z ? 0.f : z;
  }
}

}  // namespace
}  // namespace matcha

using namespace matcha;

extern "C" int matcha_pixels_to_adj(const int64_t* bin1, const int64_t* bin2, const double* count, int64_t n_pixels, const int32_t* index2node,
                                    int64_t n_index, const int32_t* node2chrom, int32_t n_nodes, double* intra, double* inter,
                                    matcha_stream_t stream) {
  MATCHA_CHECK_ARG(bin1 && bin2 && count && index2node && node2chrom && intra && inter, "matcha_pixels_to_adj: null pointer");
  MATCHA_CHECK_ARG(n_pixels >= 0 && n_index >= 1 && n_nodes >= 1, "matcha_pixels_to_adj: bad sizes");
  if (n_pixels == 0) return MATCHA_OK;
  int64_t blocks = cdiv(n_pixels, 256);
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(pixels_to_adj_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, bin1, bin2, count, n_pixels, index2node,
                     n_index, node2chrom, (int64_t)n_nodes, intra, inter);
  MATCHA_CHECK_LAUNCH("pixels_to_adj_kernel");
  return MATCHA_OK;
}

extern "C" size_t matcha_corrcoef_workspace_bytes(int32_t n) {
  if (n < 1) return 0;
  return align_up((size_t)n * sizeof(double), 256) + (size_t)n * n * sizeof(double);
}

extern "C" int matcha_corrcoef_block(const float* adj, int64_t ld, int32_t n, float* out, void* ws, size_t ws_bytes, matcha_stream_t stream) {
  MATCHA_CHECK_ARG(adj && out && ws, "matcha_corrcoef_block: null pointer");
  MATCHA_CHECK_ARG(n >= 1 && ld >= n, "matcha_corrcoef_block: need 1 <= n <= ld");
  MATCHA_CHECK_ARG(ws_bytes >= matcha_corrcoef_workspace_bytes(n), "matcha_corrcoef_block: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  double* mean = (double*)ws;
  double* C = (double*)((char*)ws + align_up((size_t)n * sizeof(double), 256));
  hipLaunchKernelGGL(row_mean_kernel, dim3(n), dim3(64), 0, st, adj, ld, n, mean);
  const unsigned t = (unsigned)cdiv(n, kCT);
  hipLaunchKernelGGL(centred_gram_kernel, dim3(t, t), dim3(256), 0, st, adj, ld, n, mean, C);
  int64_t blocks = cdiv((int64_t)n * n, 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(corr_normalise_kernel, dim3((unsigned)blocks), dim3(256), 0, st, C, n, out);
  MATCHA_CHECK_LAUNCH("corrcoef kernels");
  return MATCHA_OK;
}

extern "C" int matcha_zscore_rows(float* matrix, int64_t rows, int64_t cols, matcha_stream_t stream) {
  MATCHA_CHECK_ARG(matrix, "matcha_zscore_rows: null pointer");
  MATCHA_CHECK_ARG(rows >= 0 && cols >= 1 && rows < ((int64_t)1 << 31), "matcha_zscore_rows: bad sizes");
  if (rows == 0) return MATCHA_OK;
  hipLaunchKernelGGL(zscore_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, matrix, cols);
  MATCHA_CHECK_LAUNCH("zscore_rows_kernel");
  return MATCHA_OK;
}
