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
//   count [3]       {Tr + 1, Tr, number of tiles}  -- device-side counts consumed by every kernel through m_dev / t_dev
//   tok_pos [T+1]   position of the compact token inside its hyperedge | k << 8 (fused kernels: token -> hyperedge rows)
//   tile_meta       tiles of whole hyperedges (<= 63 tokens) for the fused d = 64 kernels
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

// Tiles of the fused kernels: runs of whole consecutive hyperedges with at most 63 tokens (+ the shared padding token = 64
// rows), packed greedily so the MFMA tiles are ~96 % full (fixed windows of 64 - L first-token indices gave 90 %).
// Greedy packing is sequential, so it runs per SUPERBLOCK (the hyperedges whose first token lies in a window of kSuperTok
// tokens, ~32 tiles): one wavefront per superblock loads 64 row lengths at a time and walks them with scalar code
// (v_readlane); only the last tile of a superblock is partial.  A second kernel compacts the per-superblock lists.
// tile_meta[w] = {first token t0, tokens, first hyperedge b0, hyperedges}; entries past the tile count are zero.
constexpr int kTileTok = 63;
constexpr int kSuperTok = 63 * 32;

__device__ __forceinline__ int lower_bound_off(const int32_t* __restrict__ row_off, int64_t B, int target) {
  int64_t lo = 0, hi = B;                              // first b in [0, B] with row_off[b] >= target
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (row_off[mid] < target) lo = mid + 1; else hi = mid;
  }
  return (int)lo;
}

__global__ __launch_bounds__(64) void tile_pack_kernel(const int32_t* __restrict__ row_off, int64_t B, int nsb, int cap_per_sb,
                                                       int32_t* __restrict__ sb_tiles, int32_t* __restrict__ sb_cnt) {
  const int s = blockIdx.x, lane = threadIdx.x;
  const int b_lo = lower_bound_off(row_off, B, s * kSuperTok);
  const int b_hi = (s + 1 == nsb) ? (int)B : lower_bound_off(row_off, B, (s + 1) * kSuperTok);
  int4* out = reinterpret_cast<int4*>(sb_tiles) + (int64_t)s * cap_per_sb;
  int tile_b0 = b_lo, tile_t0 = b_lo < B ? row_off[b_lo] : 0, cur = 0, nt = 0;
  for (int base = b_lo; base < b_hi; base += 64) {
    const int kk = (base + lane < b_hi) ? row_off[base + lane + 1] - row_off[base + lane] : 0;
    const int n = (b_hi - base < 64) ? b_hi - base : 64;
    for (int i = 0; i < n; ++i) {
      const int k = __builtin_amdgcn_readlane(kk, i);
      if (cur + k > kTileTok) {                        // close the tile in front of hyperedge base + i
        if (lane == 0 && nt < cap_per_sb) out[nt] = make_int4(tile_t0, cur, tile_b0, base + i - tile_b0);
        ++nt;
        tile_b0 = base + i; tile_t0 += cur; cur = 0;
      }
      cur += k;
    }
  }
  if (b_hi > tile_b0) {
    if (lane == 0 && nt < cap_per_sb) out[nt] = make_int4(tile_t0, cur, tile_b0, b_hi - tile_b0);
    ++nt;
  }
  if (lane == 0) sb_cnt[s] = nt < cap_per_sb ? nt : cap_per_sb;
}

// exclusive scan of the superblock tile counts (one block; nsb is small), compaction, zero fill; count[2] = number of tiles
__global__ __launch_bounds__(1024) void tile_compact_kernel(const int32_t* __restrict__ sb_tiles, int32_t* __restrict__ sb_cnt, int nsb,
                                                            int cap_per_sb, int ntiles_cap, int32_t* __restrict__ meta, int32_t* __restrict__ count) {
  __shared__ int total;
  if (threadIdx.x == 0) {
    int run = 0;
    for (int s = 0; s < nsb; ++s) { const int c = sb_cnt[s]; sb_cnt[s] = run; run += c; }
    total = run < ntiles_cap ? run : ntiles_cap;
    count[2] = total;
  }
  __syncthreads();
  const int4* src = reinterpret_cast<const int4*>(sb_tiles);
  int4* dst = reinterpret_cast<int4*>(meta);
  for (int i = threadIdx.x; i < nsb * cap_per_sb; i += 1024) {
    const int s = i / cap_per_sb, j = i - s * cap_per_sb;
    const int lo = sb_cnt[s], hi = (s + 1 < nsb) ? sb_cnt[s + 1] : total;
    if (lo + j < hi && lo + j < ntiles_cap) dst[lo + j] = src[i];
  }
  for (int i = total + threadIdx.x; i < ntiles_cap + 2; i += 1024) dst[i] = make_int4(0, 0, 0, 0);
}

static inline int super_blocks(int64_t T) { return (int)cdiv(T + 1, kSuperTok); }
static inline int super_cap(int L) { return (int)cdiv(kSuperTok + L, 64 - L) + 2; }
static inline int tiles_cap(int64_t T, int L) { return (int)cdiv(T + 1, 64 - L) + super_blocks(T); }   // every tile but a superblock's last holds > 63 - L tokens

int ragged_tiles_cap(int64_t B, int L) { return tiles_cap(B * L, L); }

size_t ragged_bytes(int64_t B, int L) {
  const int64_t T = B * L;
  size_t n = 0;
  n += align_up((size_t)(B + 1) * 4, 256);        // row_off
  n += align_up((size_t)(T + 1) * 4, 256);        // tok_slot
  n += align_up((size_t)(T + 1) * 8, 256);        // tok_id
  n += 256;                                        // count
  n += align_up((size_t)cdiv(B, kRowsPerBlock) * 4, 256);
  n += align_up((size_t)(T + 1) * 4, 256);        // tok_pos
  n += align_up((size_t)(tiles_cap(T, L) + 2) * 16, 256);                    // tile_meta
  n += align_up((size_t)super_blocks(T) * super_cap(L) * 16, 256);           // sb_tiles
  n += align_up((size_t)super_blocks(T) * 4, 256);                           // sb_cnt
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
  r.ntiles = tiles_cap(T, L);
  r.tok_pos = (int32_t*)take((size_t)(T + 1) * 4);
  r.tile_meta = (int32_t*)take((size_t)(r.ntiles + 2) * 16);
  r.nsb = super_blocks(T);
  r.sb_cap = super_cap(L);
  r.sb_tiles = (int32_t*)take((size_t)r.nsb * r.sb_cap * 16);
  r.sb_cnt = (int32_t*)take((size_t)r.nsb * 4);
}

int launch_ragged_plan(const int64_t* x, int64_t B, int L, const Ragged& r, hipStream_t st) {
  hipLaunchKernelGGL(row_count_kernel, dim3(r.nblk), dim3(256), 0, st, x, B, L, r.blk_sum);
  MATCHA_CHECK_LAUNCH("row_count_kernel");
  hipLaunchKernelGGL(row_scan_kernel, dim3(1), dim3(1024), 0, st, r.blk_sum, r.nblk, r.count);
  MATCHA_CHECK_LAUNCH("row_scan_kernel");
  hipLaunchKernelGGL(row_fill_kernel, dim3(r.nblk), dim3(256), 0, st, x, B, L, r.blk_sum, r.count, r.row_off, r.tok_slot, r.tok_id, r.tok_pos);
  MATCHA_CHECK_LAUNCH("row_fill_kernel");
  hipLaunchKernelGGL(tile_pack_kernel, dim3(r.nsb), dim3(64), 0, st, r.row_off, B, r.nsb, r.sb_cap, r.sb_tiles, r.sb_cnt);
  MATCHA_CHECK_LAUNCH("tile_pack_kernel");
  hipLaunchKernelGGL(tile_compact_kernel, dim3(1), dim3(1024), 0, st, r.sb_tiles, r.sb_cnt, r.nsb, r.sb_cap, r.ntiles, r.tile_meta, r.count);
  MATCHA_CHECK_LAUNCH("tile_compact_kernel");
  return MATCHA_OK;
}

}  // namespace matcha
