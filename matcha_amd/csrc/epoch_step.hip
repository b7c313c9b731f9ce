// The bookkeeping of one training step of the reference's epoch loop (main.py:155-188) as two launches, so that an epoch can replay ONE
// captured step with everything that changes from step to step living in device memory:
//   matcha_step_select   before the step: this step's positives out of the epoch's shuffled list (main.py:160-161: the slice
//                        [i * batch, (i + 1) * batch) of edges / edge_weight), their weights, the step's reconstruction chromosome;
//                        the counter-RNG seeds of the sampler and of the dropout masks + 1
//   matcha_step_record   after the step: sigmoid of the logits into the epoch's prediction buffer (main.py:58, :185-186), the size of
//                        every hyperedge (main.py:449-451), the running sums of the two losses (main.py:187-188), step counter + 1
// As torch ops this was fourteen tiny kernels per step (a fifth of the launches of a 384-row step); the arithmetic is index copies, a
// popcount per row and expf.
#include "kernels.hpp"

namespace matcha {

__global__ __launch_bounds__(256) void step_select_kernel(const int64_t* __restrict__ pos, const float* __restrict__ w, int64_t n_rows, int L,
                                                          const int64_t* __restrict__ it, int P, int64_t* __restrict__ x,
                                                          float* __restrict__ ww, const int32_t* __restrict__ chroms, int64_t n_chroms,
                                                          int32_t* __restrict__ cell, uint64_t* __restrict__ seed0, uint64_t* __restrict__ seed1) {
  const int64_t step = *it;
  const int i = blockIdx.x * 256 + threadIdx.x;             // one thread per (row, slot)
  if (i == 0) {
    if (cell) cell[0] = chroms[step < n_chroms ? step : n_chroms - 1];
    if (seed0) seed0[0] += 1;                               // the sampler's and the trainer's counter-RNG seeds: new draws every step
    if (seed1) seed1[0] += 1;
  }
  if (i >= P * L) return;
  const int row = i / L, l = i - row * L;
  const int64_t src = (step * P + row) % n_rows;            // (past the end of the list the selection wraps around: never out of bounds)
  x[(int64_t)row * L + l] = pos[src * L + l];
  if (l == 0) ww[row] = w[src];
}

// one workgroup (the driver's batch is 384 rows; a larger B just loops): every thread reads the counter, a barrier, then thread 0 advances it
__global__ __launch_bounds__(1024) void step_record_kernel(const float* __restrict__ logits, const float* __restrict__ losses,
                                                           const int64_t* __restrict__ x, int B, int L, int64_t* __restrict__ it,
                                                           int64_t n_steps, float* __restrict__ sums, float* __restrict__ preds,
                                                           int64_t* __restrict__ sizes) {
  int64_t step = *it;
  __syncthreads();                                          // every thread holds the counter before thread 0 advances it
  if (threadIdx.x == 0) {
    it[0] = step + 1;
    sums[0] += losses[0];
    sums[1] += losses[1];
  }
  if (step >= n_steps) step = n_steps - 1;                  // (a replay past the epoch's end overwrites the last row: never out of bounds)
  for (int b = threadIdx.x; b < B; b += 1024) {
    const float z = logits[b];
    preds[step * B + b] = 1.0f / (1.0f + expf(-z));         // torch.sigmoid's expression (main.py:58)
    int k = 0;
    for (int l = 0; l < L; ++l) k += x[(int64_t)b * L + l] != 0 ? 1 : 0;
    sizes[step * B + b] = k;
  }
}

}  // namespace matcha

using namespace matcha;

extern "C" int matcha_step_select(const int64_t* pos, const float* w, int64_t n_rows, int32_t L, const int64_t* it, int32_t P, int64_t* x,
                                  float* ww, const int32_t* chroms, int64_t n_chroms, int32_t* cell, uint64_t* seed0, uint64_t* seed1,
                                  matcha_stream_t stream) {
  MATCHA_CHECK_ARG(pos && w && it && x && ww && n_rows > 0 && L >= 1 && L <= MATCHA_MAX_L && P >= 1, "matcha_step_select: bad argument");
  MATCHA_CHECK_ARG(!cell || (chroms && n_chroms > 0), "matcha_step_select: cell without chroms");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(step_select_kernel, dim3((unsigned)cdiv((int64_t)P * L, 256)), dim3(256), 0, st, pos, w, n_rows, (int)L, it, (int)P, x, ww, chroms,
                     n_chroms, cell, seed0, seed1);
  MATCHA_CHECK_LAUNCH("step_select_kernel");
  return MATCHA_OK;
}

extern "C" int matcha_step_record(const float* logits, const float* losses, const int64_t* x, int64_t B, int32_t L, int64_t* it, int64_t n_steps,
                                  float* sums, float* preds, int64_t* sizes, matcha_stream_t stream) {
  MATCHA_CHECK_ARG(logits && losses && x && it && sums && preds && sizes && B >= 1 && n_steps >= 1 && L >= 1 && L <= MATCHA_MAX_L,
                   "matcha_step_record: bad argument");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(step_record_kernel, dim3(1), dim3(1024), 0, st, logits, losses, x, (int)B, (int)L, it, n_steps, sums, preds, sizes);
  MATCHA_CHECK_LAUNCH("step_record_kernel");
  return MATCHA_OK;
}
