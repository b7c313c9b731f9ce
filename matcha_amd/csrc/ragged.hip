// Ragged execution plan: compact the real (non-padding) token slots of x [B, L] into a CSR list.
//
// The reference computes every padding slot like a real token (pads are attended as keys/values, SURVEY.md headline
// fact 7), but all padding slots hold the SAME vector -- tanh(next_w(0 + attribute_nn.bias)) -- so their K/V rows are
// one constant per step, their queries are masked out downstream (Modules.py:614, :309) and nothing else depends on
// them.  The HIP path therefore runs the token-level layers on  Tr real tokens + ONE shared padding token  (index Tr)
// and the attention kernels add the (L - k)-fold padding-key term in closed form.  Results are identical to the padded
// computation; the work drops by the padding fraction (30 % for k uniform in {2..5} padded to 5).
//
//   row_off [B+1]   first compact token of hyperedge b (exclusive prefix sum of k_b); row_off[B] = Tr
//   tok_slot [T+1]  compact token -> original slot b*L + l (dropout counters follow the original slots, so masks are
//                   the ones the oracle generates for the padded layout); tok_slot[Tr] = B*L
//   tok_id [T+1]    node id of the compact token; tok_id[Tr] = 0 (padding id)
//   count [2]       {Tr + 1, Tr}  -- device-side row counts consumed by every kernel through m_dev / t_dev
//   tok_pos [T+1]   position of the compact token inside its hyperedge | k << 8 (fused kernels: token -> hyperedge rows)
//   tile_b0, tile_meta   tiles of whole hyperedges (<= 63 tokens) for the fused d = 64 kernels
#include "kernels.hpp"

namespace matcha {

constexpr int kRowsPerBlock = 1024;

__device__ __forceinline__ int block_exclusive_scan_256(int v, int* lds4, int* total) {
  // 256 threads: inclusive scan inside each wave, then the 4 wave totals
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(incl, o, 64);
    if (lane >= o) incl += u;
  }
  if (lane == 63) lds4[wave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += lds4[w];
  if (total) *total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
  return base + incl - v;
}

__global__ __launch_bounds__(256) void row_count_kernel(const int64_t* __restrict__ x, int64_t B, int L, int32_t* __restrict__ blk_sum) {
  __shared__ int lds4[4];
  const int64_t b0 = (int64_t)blockIdx.x * kRowsPerBlock + threadIdx.x * 4;
  int cnt = 0;
  for (int i = 0; i < 4; ++i)
    if (b0 + i < B)
      for (int l = 0; l < L; ++l) cnt += x[(b0 + i) * L + l] != 0 ? 1 : 0;
  int total;
  (void)block_exclusive_scan_256(cnt, lds4, &total);
  if (threadIdx.x == 0) blk_sum[blockIdx.x] = total;
}

// exclusive scan of the block sums in place; count = {Tr + 1, Tr}
__global__ __launch_bounds__(1024) void row_scan_kernel(int32_t* __restrict__ blk_sum, int nblk, int32_t* __restrict__ count) {
  __shared__ int part[1024];
  const int chunk = (nblk + 1023) / 1024;
  const int b0 = threadIdx.x * chunk, b1 = (b0 + chunk < nblk) ? b0 + chunk : nblk;
  int local = 0;
  for (int b = b0; b < b1; ++b) local += blk_sum[b];
  part[threadIdx.x] = local;
  __syncthreads();
  // simple two-level: thread 0 of each 32-group is not needed -- nblk is small (B/1024); a serial pass is fine
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < 1024; ++i) { const int v = part[i]; part[i] = run; run += v; }
    count[0] = run + 1;
    count[1] = run;
  }
  __syncthreads();
  int run = part[threadIdx.x];
  for (int b = b0; b < b1; ++b) { const int v = blk_sum[b]; blk_sum[b] = run; run += v; }
}

__global__ __launch_bounds__(256) void row_fill_kernel(const int64_t* __restrict__ x, int64_t B, int L, const int32_t* __restrict__ blk_base,
                                                       const int32_t* __restrict__ count, int32_t* __restrict__ row_off,
                                                       int32_t* __restrict__ tok_slot, int64_t* __restrict__ tok_id,
                                                       int32_t* __restrict__ tok_pos) {
  __shared__ int lds4[4];
  const int64_t b0 = (int64_t)blockIdx.x * kRowsPerBlock + threadIdx.x * 4;
  int cnt = 0;
  for (int i = 0; i < 4; ++i)
    if (b0 + i < B)
      for (int l = 0; l < L; ++l) cnt += x[(b0 + i) * L + l] != 0 ? 1 : 0;
  int pos = blk_base[blockIdx.x] + block_exclusive_scan_256(cnt, lds4, nullptr);
  for (int i = 0; i < 4; ++i) {
    const int64_t b = b0 + i;
    if (b >= B) break;
    row_off[b] = pos;
    int k = 0;
    for (int l = 0; l < L; ++l) k += x[b * L + l] != 0 ? 1 : 0;
    int nth = 0;
    for (int l = 0; l < L; ++l) {
      const int64_t id = x[b * L + l];
      if (id != 0) { tok_slot[pos] = (int32_t)(b * L + l); tok_id[pos] = id; tok_pos[pos] = nth | (k << 8); ++pos; ++nth; }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int tr = count[1];
    row_off[B] = tr;
    tok_slot[tr] = (int32_t)(B * L);
    tok_id[tr] = 0;
    tok_pos[tr] = 0;
  }
}

// Tiles of the fused kernels: tile w owns the hyperedges whose first token index lies in [w*win, (w+1)*win), win = 64 - L
// (<= 63 real tokens per tile).  tile_b0[w] = first hyperedge of tile w, tile_b0[ntiles] = B.  Consecutive hyperedges
// start at most L < win tokens apart, so the window index grows by at most one per hyperedge.
__global__ __launch_bounds__(256) void tile_plan_kernel(const int32_t* __restrict__ row_off, int64_t B, int win, int ntiles,
                                                        int32_t* __restrict__ tile_b0) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b < B) {
    const int w = row_off[b] / win;
    const int wprev = b > 0 ? row_off[b - 1] / win : -1;
    if (w != wprev && w <= ntiles) tile_b0[w] = (int32_t)b;
  }
  if (b == B - 1 || (B == 0 && b == 0)) {
    const int wlast = B > 0 ? row_off[B - 1] / win : -1;
    for (int w = wlast + 1; w <= ntiles; ++w) tile_b0[w] = (int32_t)B;
  }
}

// tile_meta[w] = {t0, n_real, b0, n_h}; entries past the last real tile are zero (n_real = 0: skipped by the kernels)
__global__ __launch_bounds__(256) void tile_meta_kernel(const int32_t* __restrict__ row_off, const int32_t* __restrict__ tile_b0, int ntiles,
                                                        int32_t* __restrict__ meta) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= ntiles + 2) return;
  int4 m = make_int4(0, 0, 0, 0);
  if (w < ntiles) {
    const int b0 = tile_b0[w], b1 = tile_b0[w + 1];
    if (b1 > b0) {
      const int t0 = row_off[b0];
      m = make_int4(t0, row_off[b1] - t0, b0, b1 - b0);
    }
  }
  reinterpret_cast<int4*>(meta)[w] = m;
}

size_t ragged_bytes(int64_t B, int L) {
  const int64_t T = B * L;
  size_t n = 0;
  n += align_up((size_t)(B + 1) * 4, 256);        // row_off
  n += align_up((size_t)(T + 1) * 4, 256);        // tok_slot
  n += align_up((size_t)(T + 1) * 8, 256);        // tok_id
  n += 256;                                        // count
  n += align_up((size_t)cdiv(B, kRowsPerBlock) * 4, 256);
  n += align_up((size_t)(cdiv(T + 1, 64 - L) + 2) * 4, 256);   // tile_b0
  n += align_up((size_t)(T + 1) * 4, 256);        // tok_pos
  n += align_up((size_t)(cdiv(T + 1, 64 - L) + 2) * 16, 256);  // tile_meta
  return n;
}

void ragged_carve(int64_t B, int L, char* base, Ragged& r) {
  const int64_t T = B * L;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base + off; off += align_up(bytes, 256); return p; };
  r.row_off = (int32_t*)take((size_t)(B + 1) * 4);
  r.tok_slot = (int32_t*)take((size_t)(T + 1) * 4);
  r.tok_id = (int64_t*)take((size_t)(T + 1) * 8);
  r.count = (int32_t*)take(256);
  r.blk_sum = (int32_t*)take((size_t)cdiv(B, kRowsPerBlock) * 4);
  r.nblk = (int)cdiv(B, kRowsPerBlock);
  r.ntiles = (int)cdiv(T + 1, 64 - L);
  r.tile_b0 = (int32_t*)take((size_t)(r.ntiles + 2) * 4);
  r.tok_pos = (int32_t*)take((size_t)(T + 1) * 4);
  r.tile_meta = (int32_t*)take((size_t)(r.ntiles + 2) * 16);
}

int launch_ragged_plan(const int64_t* x, int64_t B, int L, const Ragged& r, hipStream_t st) {
  hipLaunchKernelGGL(row_count_kernel, dim3(r.nblk), dim3(256), 0, st, x, B, L, r.blk_sum);
  MATCHA_CHECK_LAUNCH("row_count_kernel");
  hipLaunchKernelGGL(row_scan_kernel, dim3(1), dim3(1024), 0, st, r.blk_sum, r.nblk, r.count);
  MATCHA_CHECK_LAUNCH("row_scan_kernel");
  hipLaunchKernelGGL(row_fill_kernel, dim3(r.nblk), dim3(256), 0, st, x, B, L, r.blk_sum, r.count, r.row_off, r.tok_slot, r.tok_id, r.tok_pos);
  MATCHA_CHECK_LAUNCH("row_fill_kernel");
  hipLaunchKernelGGL(tile_plan_kernel, dim3((unsigned)cdiv(B, 256)), dim3(256), 0, st, r.row_off, B, 64 - L, r.ntiles, r.tile_b0);
  MATCHA_CHECK_LAUNCH("tile_plan_kernel");
  hipLaunchKernelGGL(tile_meta_kernel, dim3((unsigned)cdiv(r.ntiles + 2, 256)), dim3(256), 0, st, r.row_off, r.tile_b0, r.ntiles, r.tile_meta);
  MATCHA_CHECK_LAUNCH("tile_meta_kernel");
  return MATCHA_OK;
}

}  // namespace matcha
