// Positive-weight preprocessing: the uniform quantile transform the reference applies to the k-mer
// frequencies before its cut-offs (Code/main.py:555, :653 -- sklearn QuantileTransformer(n_quantiles=1000,
// output_distribution='uniform').fit_transform on one float32 column), fitted on EVERY row (scikit-learn fits on a random
// 10 000-row subsample above that size; this is its subsample=None behaviour and the only deliberate difference).
//
//   1. rocPRIM radix sort of the column                                   (HBM bound: a few passes over 4n bytes)
//   2. quantile_fit_kernel: Q <= 4096 landmarks = numpy 'linear' percentiles of the sorted column at linspace(0, 1, Q),
//      float64, running maximum                                            (one workgroup, Q threads; negligible)
//   3. quantile_transform_kernel: per row two binary searches over the landmarks held in LDS -- np.interp ascending and
//      on the negated, reversed arrays -- averaged, float64 -> float32, exact hits on the extreme landmarks -> 0 / 1.
//      8 bytes of HBM traffic per row; the 2 * log2(Q) LDS probes per row are what it costs.
//
// Every float64 expression is evaluated in numpy's order with contraction off, so the result is bit-identical to the
// restatement in oracle/positives.py (and to scikit-learn 1.7.2 wherever that is deterministic).
#include <hip/hip_runtime.h>

#include <cstdint>

#include <rocprim/device/device_radix_sort.hpp>

#include "kernels.hpp"

namespace matcha {
namespace {

constexpr int kMaxQuantiles = 4096;

__device__ __forceinline__ double reference_at(int j, int Q) {          // np.linspace(0, 1, Q)[j]
#pragma clang fp contract(off)
  if (Q == 1) return 0.0;
  if (j == Q - 1) return 1.0;
  const double step = 1.0 / (double)(Q - 1);
  return (double)j * step;
}

__global__ __launch_bounds__(1024) void quantile_fit_kernel(const float* __restrict__ sorted, int64_t n, int Q, double* __restrict__ quantiles) {
#pragma clang fp contract(off)
  __shared__ double q[kMaxQuantiles];
  for (int j = threadIdx.x; j < Q; j += blockDim.x) {
    const double pct = reference_at(j, Q) * 100.0;
    const double frac = pct / 100.0;                                    // numpy: true_divide(q, float32(100)) in float64
    const double v = (double)(n - 1) * frac;                            // 'linear': virtual index (n - 1) q
    int64_t lo = (int64_t)floor(v);
    if (lo < 0) lo = 0;
    if (lo > n - 1) lo = n - 1;
    const int64_t hi = lo + 1 < n ? lo + 1 : n - 1;
    const double g = v - (double)lo;
    const float a = sorted[lo], b = sorted[hi];
    const double diff = (double)(b - a);                                // float32 subtract, float64 from here on
    q[j] = g >= 0.5 ? (double)b - diff * (1.0 - g) : (double)a + diff * g;
  }
  __syncthreads();
  if (threadIdx.x == 0) {                                               // np.maximum.accumulate (Q <= 4096: a serial pass is microseconds)
    double m = q[0];
    for (int j = 0; j < Q; ++j) { m = q[j] > m ? q[j] : m; q[j] = m; }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < Q; j += blockDim.x) quantiles[j] = q[j];
}

__global__ __launch_bounds__(256) void quantile_transform_kernel(const float* __restrict__ x, int64_t n, int Q, const double* __restrict__ quantiles,
                                                                 float* __restrict__ out) {
#pragma clang fp contract(off)
  __shared__ double q[kMaxQuantiles];
  for (int j = threadIdx.x; j < Q; j += blockDim.x) q[j] = quantiles[j];
  __syncthreads();
  const double q_lo = q[0], q_hi = q[Q - 1];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double v = (double)x[i];
    int lb = 0, ub = 0;                                                 // lb = #(q < v), ub = #(q <= v)
    {
      int lo = 0, hi = Q;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (q[mid] < v) lo = mid + 1; else hi = mid; }
      lb = lo;
      hi = Q;                                                           // ub >= lb: continue from lb
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (q[mid] <= v) lo = mid + 1; else hi = mid; }
      ub = lo;
    }
    // np.interp(v, q, refs): j = ub - 1
    double fwd;
    {
      const int j = ub - 1;
      if (j < 0) fwd = reference_at(0, Q);
      else if (v > q_hi || j >= Q - 1) fwd = reference_at(Q - 1, Q);
      else if (q[j] == v) fwd = reference_at(j, Q);
      else {
        const double slope = (reference_at(j + 1, Q) - reference_at(j, Q)) / (q[j + 1] - q[j]);
        fwd = slope * (v - q[j]) + reference_at(j, Q);
      }
    }
    // np.interp(-v, -q[::-1], -refs[::-1]): position j' = Q - lb - 1 of the reversed arrays is landmark lb
    double bwd;
    {
      if (lb >= Q) bwd = -reference_at(Q - 1, Q);                       // -v below the first reversed landmark
      else if (v < q_lo || lb == 0) bwd = -reference_at(0, Q);          // above / at the last one
      else if (q[lb] == v) bwd = -reference_at(lb, Q);
      else {
        const double slope = (-reference_at(lb - 1, Q) - -reference_at(lb, Q)) / (-q[lb - 1] - -q[lb]);
        bwd = slope * (-v - -q[lb]) + -reference_at(lb, Q);
      }
    }
    float r = (float)(0.5 * (fwd - bwd));
    if (v == q_hi) r = 1.f;
    if (v == q_lo) r = 0.f;
    out[i] = r;
  }
}

struct QPlan {
  size_t off_sorted, off_quant, off_tmp, tmp_bytes, total;
};

int make_qplan(int64_t n, QPlan& pl) {
  size_t t_sort = 0;
  float* fn = nullptr;
  if (rocprim::radix_sort_keys(nullptr, t_sort, fn, fn, (size_t)n, 0, 32, (hipStream_t)0) != hipSuccess) return MATCHA_EHIP;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += align_up(bytes, 256); return o; };
  pl.off_sorted = take((size_t)n * sizeof(float));
  pl.off_quant = take((size_t)kMaxQuantiles * sizeof(double));
  pl.off_tmp = take(t_sort);
  pl.tmp_bytes = t_sort;
  pl.total = off;
  return MATCHA_OK;
}

}  // namespace
}  // namespace matcha

using namespace matcha;

extern "C" size_t matcha_quantile_workspace_bytes(int64_t n) {
  if (n < 1 || n > ((int64_t)1 << 31) - 1) return 0;
  QPlan pl;
  if (make_qplan(n, pl) != MATCHA_OK) return 0;
  return pl.total;
}

extern "C" int matcha_quantile_uniform(const float* freq, int64_t n, int32_t n_quantiles, float* out, double* quantiles_out, void* ws,
                                       size_t ws_bytes, matcha_stream_t stream) {
  MATCHA_CHECK_ARG(freq && out && ws, "matcha_quantile_uniform: null pointer");
  MATCHA_CHECK_ARG(n >= 1 && n <= ((int64_t)1 << 31) - 1, "matcha_quantile_uniform: n out of range");
  MATCHA_CHECK_ARG(n_quantiles >= 1 && n_quantiles <= kMaxQuantiles, "matcha_quantile_uniform: n_quantiles must be in [1, 4096]");
  QPlan pl;
  MATCHA_TRY(make_qplan(n, pl));
  MATCHA_CHECK_ARG(ws_bytes >= pl.total, "matcha_quantile_uniform: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  char* w = (char*)ws;
  float* sorted = (float*)(w + pl.off_sorted);
  double* quant = (double*)(w + pl.off_quant);
  size_t tb = pl.tmp_bytes;
  if (rocprim::radix_sort_keys(w + pl.off_tmp, tb, freq, sorted, (size_t)n, 0, 32, st) != hipSuccess) {
    set_error("quantile: radix sort failed");
    return MATCHA_EHIP;
  }
  const int Q = (int64_t)n_quantiles < n ? n_quantiles : (int)n;       // sklearn clamps n_quantiles to the row count
  hipLaunchKernelGGL(quantile_fit_kernel, dim3(1), dim3(1024), 0, st, sorted, n, Q, quant);
  int64_t blocks = cdiv(n, 256 * 4);
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(quantile_transform_kernel, dim3((unsigned)blocks), dim3(256), 0, st, freq, n, Q, quant, out);
  MATCHA_CHECK_LAUNCH("quantile_transform_kernel");
  if (quantiles_out && hipMemcpyAsync(quantiles_out, quant, (size_t)Q * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess) {
    set_error("quantile: copy of the landmarks failed");
    return MATCHA_EHIP;
  }
  return MATCHA_OK;
}
