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
//   tok_id [T+1]    node id of the compact token; tok_id[Tr] = 0 (padding id).  Ids outside [0, n_nodes] are flagged in the
//                   caller's status word and replaced by 0, so that no later kernel indexes out of bounds
//   tok_key [T+1]   the same ids as int32 with 0 in every unused slot (the (id, gradient row) list of table_grad.hip and of
//                   the row-sparse data-parallel exchange)
//   count [4]       {Tr + 1, Tr, number of tiles, number of half tiles}  -- device-side counts consumed by every kernel through m_dev / t_dev
//   tok_pos [T+1]   position of the compact token inside its hyperedge | k << 8 (fused kernels: token -> hyperedge rows)
//   tile_meta       tiles of whole hyperedges (<= 63 tokens) for the fused d = 64 kernels (fused_bwd.hip walks these)
//   half_meta       the same token stream packed into HALF tiles of whole hyperedges with <= 31 tokens (+ the shared padding token =
//                   32 rows: one wavefront of the wave-independent fused forward, fused_fwd32.hip); count[3] = number of half tiles
//   tok_tile [T+1]  compact token -> (tile << 6) | row inside that tile: where the forward's wavefront stores the token's Q / K / V
//                   rows for the backward kernel, whose 64-row tiles cut the stream at other places than the half tiles
#include "kernels.hpp"

namespace matcha {

constexpr int kRpt = 1;                      // rows per thread of the counting / filling passes (round 6: 1 -- four rows per thread left 64 workgroups for the
                                             // bench's 65 536 rows, a quarter of the CUs, each thread a serial chain of ~56 scattered stores: 12.9 us; now 256 workgroups)
constexpr int kRowsPerBlock = 256 * kRpt;

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
  const int64_t b0 = (int64_t)blockIdx.x * kRowsPerBlock + threadIdx.x * kRpt;
  int cnt = 0;
  for (int i = 0; i < kRpt; ++i)
    if (b0 + i < B)
      for (int l = 0; l < L; ++l) cnt += x[(b0 + i) * L + l] != 0 ? 1 : 0;
  int total;
  (void)block_exclusive_scan_256(cnt, lds4, &total);
  if (threadIdx.x == 0) blk_sum[blockIdx.x] = total;
}

// exclusive scan of the block sums in place; count = {Tr + 1, Tr}
__global__ __launch_bounds__(1024) void row_scan_kernel(int32_t* __restrict__ blk_sum, int nblk, int32_t* __restrict__ count,
                                                        int32_t* __restrict__ sb_first, int nsb, int32_t n_rows) {
  __shared__ int part[1024];
  for (int i = threadIdx.x; i <= nsb; i += 1024) sb_first[i] = n_rows;       // superblocks in which no hyperedge starts: empty range
  const int chunk = (nblk + 1023) / 1024;
  const int b0 = threadIdx.x * chunk, b1 = (b0 + chunk < nblk) ? b0 + chunk : nblk;
  int local = 0;
  for (int b = b0; b < b1; ++b) local += blk_sum[b];
  // exclusive scan of the 1024 per-thread sums: inclusive scan inside each wavefront (six shuffle steps), then the sixteen wave totals
  // (a serial pass of thread 0 over the 1024 entries cost 8 us of this 10 us kernel at every batch size)
  __shared__ int wtot[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = local;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(incl, o, 64);
    if (lane >= o) incl += u;
  }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  int wbase = 0, total = 0;
#pragma unroll
  for (int w2 = 0; w2 < 16; ++w2) { const int v = wtot[w2]; if (w2 < wave) wbase += v; total += v; }
  part[threadIdx.x] = wbase + incl - local;
  if (threadIdx.x == 0) {
    count[0] = total + 1;
    count[1] = total;
  }
  int run = part[threadIdx.x];
  for (int b = b0; b < b1; ++b) { const int v = blk_sum[b]; blk_sum[b] = run; run += v; }
}

__global__ __launch_bounds__(256) void row_fill_kernel(const int64_t* __restrict__ x, int64_t B, int L, const int32_t* __restrict__ blk_base,
                                                       const int32_t* __restrict__ count, int32_t* __restrict__ row_off,
                                                       int32_t* __restrict__ tok_slot, int64_t* __restrict__ tok_id,
                                                       int32_t* __restrict__ tok_pos, int32_t* __restrict__ sb_first, int super_tok,
                                                       int32_t* __restrict__ tok_key, int64_t n_nodes, int32_t* __restrict__ status) {
  __shared__ int lds4[4];
  const int64_t b0 = (int64_t)blockIdx.x * kRowsPerBlock + threadIdx.x * kRpt;
  // this thread's rows (and the row in front of them) in registers: ONE round of loads, all in flight together, on clamped
  // addresses with the masks applied to the values (the first version read x three times behind dependent waits)
  int64_t v[kRpt][MATCHA_MAX_L], vp[MATCHA_MAX_L];
#pragma unroll
  for (int i = 0; i < kRpt; ++i) {
    const int64_t bc = b0 + i < B ? b0 + i : B - 1;
#pragma unroll
    for (int l = 0; l < MATCHA_MAX_L; ++l) v[i][l] = x[bc * L + (l < L ? l : L - 1)];
  }
  {
    const int64_t bp = b0 > 0 ? (b0 - 1 < B ? b0 - 1 : B - 1) : 0;
#pragma unroll
    for (int l = 0; l < MATCHA_MAX_L; ++l) vp[l] = x[bp * L + (l < L ? l : L - 1)];
  }
  int kk[kRpt], cnt = 0;
#pragma unroll
  for (int i = 0; i < kRpt; ++i) {
    int k = 0;
#pragma unroll
    for (int l = 0; l < MATCHA_MAX_L; ++l) {
      v[i][l] = (b0 + i < B && l < L) ? v[i][l] : 0;
      k += v[i][l] != 0 ? 1 : 0;
    }
    kk[i] = k;
    cnt += k;
  }
  int kprev = 0;
#pragma unroll
  for (int l = 0; l < MATCHA_MAX_L; ++l) kprev += (b0 > 0 && l < L && vp[l] != 0) ? 1 : 0;
  int pos = blk_base[blockIdx.x] + block_exclusive_scan_256(cnt, lds4, nullptr);
#pragma unroll
  for (int i = 0; i < kRpt; ++i) {
    const int64_t b = b0 + i;
    if (b < B) {
      row_off[b] = pos;
      const int k = kk[i];
      // first hyperedge of a planning superblock: its first token lies in another window of super_tok tokens than its predecessor's
      if (b == 0 || pos / super_tok != (pos - kprev) / super_tok) sb_first[pos / super_tok] = (int32_t)b;
      kprev = k;
      int nth = 0;
#pragma unroll
      for (int l = 0; l < MATCHA_MAX_L; ++l) {
        int64_t id = v[i][l];
        if (id != 0) {
          if (id < 0 || id > n_nodes) {                  // the reference raises IndexError here (nn.Embedding, Modules.py:34)
            if (status) atomicOr(status, MATCHA_STATUS_BAD_ID);
            id = 0;
          }
          tok_slot[pos] = (int32_t)(b * L + l); tok_id[pos] = id; tok_key[pos] = (int32_t)id; tok_pos[pos] = nth | (k << 8); ++pos; ++nth;
        }
      }
    }
  }
  const int tr = count[1];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    row_off[B] = tr;
    tok_slot[tr] = (int32_t)(B * L);
    tok_id[tr] = 0;
    tok_pos[tr] = 0;
  }
  // tok_key of every slot behind the real tokens is 0 (the padding token's included): written here instead of a memset of the whole
  // array in front of the plan (the real tokens' keys are written above by their owners; nobody else touches slots >= tr)
  for (int64_t i = tr + (int64_t)blockIdx.x * 256 + threadIdx.x; i < B * L + 1; i += (int64_t)gridDim.x * 256) tok_key[i] = 0;
}

// Tiles of the fused kernels: runs of whole consecutive hyperedges with at most 63 tokens (+ the shared padding token = 64
// rows), packed greedily so the MFMA tiles are ~96 % full (fixed windows of 64 - L first-token indices gave 90 %).
// Greedy packing is sequential, so it runs per SUPERBLOCK (the hyperedges whose first token lies in a window of kSuperTok
// tokens, ~32 tiles): one wavefront per superblock loads 64 row offsets at a time; a ballot finds the first hyperedge that
// ends beyond the open tile, which closes it (about 4 closes per 64 hyperedges); only a superblock's last tile is partial.  A second kernel compacts the per-superblock lists.
// tile_meta[w] = {first token t0, tokens, first hyperedge b0, hyperedges}; entries past the tile count are zero.
constexpr int kFullTok = 63;
constexpr int kHalfTok = 31;
constexpr int kSuperTok = 63 * 32;

// one wavefront packs superblock s of one list (half = the <= 31-token list); returns the number of tiles written to out[0 .. cap_per_sb)
__device__ __forceinline__ int tile_pack_wave(const int32_t* __restrict__ row_off, int64_t B, int nsb, int s, bool half, int cap_per_sb,
                                              const int32_t* __restrict__ sb_first, int4* __restrict__ out, int lane) {
  const int kTileTok = half ? kHalfTok : kFullTok;
  const int b_lo = sb_first[s];                        // marked by the fill pass (B: nothing starts here)
  int b_hi = (int)B;
  for (int q = s + 1; q < nsb; ++q)                    // next superblock that has a first hyperedge (normally s + 1)
    if (sb_first[q] < (int)B) { b_hi = sb_first[q]; break; }
  int nt = 0;
  if (b_lo < b_hi) {
    int tile_b0 = b_lo;
    int tile_tok0 = __builtin_amdgcn_readfirstlane(row_off[b_lo]);
    for (int base = b_lo; base < b_hi; base += 64) {
      const int idx = base + lane;
      const bool valid = idx < b_hi;
      const int so = valid ? row_off[idx] : 0;         // first token of hyperedge idx
      const int eo = valid ? row_off[idx + 1] : 0;     // one past its last token (monotone over idx)
      for (;;) {
        const uint64_t over = __ballot(valid && eo - tile_tok0 > kTileTok);   // hyperedges that end beyond the open tile
        if (over == 0) break;
        const int i = __ffsll((long long)over) - 1;    // the first of them closes the tile and opens the next one
        const int start = __builtin_amdgcn_readlane(so, i);
        if (lane == 0 && nt < cap_per_sb) out[nt] = make_int4(tile_tok0, start - tile_tok0, tile_b0, base + i - tile_b0);
        ++nt;
        tile_b0 = base + i;
        tile_tok0 = start;
      }
    }
    const int end = __builtin_amdgcn_readfirstlane(row_off[b_hi]);
    if (lane == 0 && nt < cap_per_sb) out[nt] = make_int4(tile_tok0, end - tile_tok0, tile_b0, b_hi - tile_b0);
    ++nt;
  }
  return nt < cap_per_sb ? nt : cap_per_sb;
}

// blockIdx.y = 0: tiles of <= 63 tokens; 1: half tiles of <= 31 tokens (same superblocks, own lists)
__global__ __launch_bounds__(64) void tile_pack_kernel(const int32_t* __restrict__ row_off, int64_t B, int nsb, int cap_per_sb0, int cap_per_sb1,
                                                       const int32_t* __restrict__ sb_first, int32_t* __restrict__ sb_tiles0,
                                                       int32_t* __restrict__ sb_cnt0, int32_t* __restrict__ sb_tiles1, int32_t* __restrict__ sb_cnt1, int y0) {
  const int s = blockIdx.x, lane = threadIdx.x;
  const bool half = (int)blockIdx.y + y0 != 0;
  const int cap_per_sb = half ? cap_per_sb1 : cap_per_sb0;
  int32_t* sb_tiles = half ? sb_tiles1 : sb_tiles0;
  int32_t* sb_cnt = half ? sb_cnt1 : sb_cnt0;
  const int nt = tile_pack_wave(row_off, B, nsb, s, half, cap_per_sb, sb_first, reinterpret_cast<int4*>(sb_tiles) + (int64_t)s * cap_per_sb, lane);
  if (lane == 0) sb_cnt[s] = nt;
}

// exclusive scan of the superblock tile counts (one block, 1024 counts per pass), compaction, zero fill; count[2] = tiles
struct CompactArgs {
  const int32_t* sb_tiles[2]; int32_t* sb_cnt[2]; int cap_per_sb[2]; int ntiles_cap[2]; int32_t* meta[2];
};
// blockIdx.x = 0: tiles -> tile_meta, count[2];  1: half tiles -> half_meta, count[3]
__global__ __launch_bounds__(1024) void tile_compact_kernel(CompactArgs a, int nsb, int32_t* __restrict__ count, int x0) {
  const int which = (int)blockIdx.x + x0;
  if (x0 != 0 && threadIdx.x == 0) count[2] = 0;       // half tiles only: no 64-row tile list this time
  const int32_t* __restrict__ sb_tiles = a.sb_tiles[which];
  int32_t* __restrict__ sb_cnt = a.sb_cnt[which];
  const int cap_per_sb = a.cap_per_sb[which], ntiles_cap = a.ntiles_cap[which];
  int32_t* __restrict__ meta = a.meta[which];
  __shared__ int wtot[16];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base < nsb; base += 1024) {
    const int i = base + threadIdx.x;
    const int c = i < nsb ? sb_cnt[i] : 0;
    // inclusive scan inside each wavefront (six shuffle steps), then the sixteen wave totals (a Hillis-Steele pass over LDS took
    // twenty barriers per 1024 counts)
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o, 64);
      if (lane >= o) incl += u;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int wbase = 0, tot = 0;
#pragma unroll
    for (int w2 = 0; w2 < 16; ++w2) { const int v = wtot[w2]; if (w2 < wave) wbase += v; tot += v; }
    if (i < nsb) sb_cnt[i] = carry + wbase + incl - c;
    __syncthreads();
    if (threadIdx.x == 0) carry += tot;
    __syncthreads();
  }
  const int total = carry < ntiles_cap ? carry : ntiles_cap;
  if (threadIdx.x == 0) count[2 + which] = total;
  __syncthreads();                                     // sb_cnt (now offsets) written by this block: visible after the barrier
  const int4* __restrict__ src = reinterpret_cast<const int4*>(sb_tiles);
  int4* __restrict__ dst = reinterpret_cast<int4*>(meta);
  __shared__ int offs[1025];
  if (nsb <= 1024) {
    // the offsets of the superblocks in LDS; every thread then moves tiles i = tid, tid + 1024, ... of the COMPACT list, finding each one's
    // superblock by bisection over the offsets -- independent loads, four in flight per thread (a wavefront per superblock was a chain of
    // nsb / 16 dependent round trips)
    for (int i = threadIdx.x; i < nsb; i += 1024) offs[i] = sb_cnt[i];
    if (threadIdx.x == 0) offs[nsb] = total;
    __syncthreads();
    for (int i0 = threadIdx.x; i0 < total; i0 += 4 * 1024) {
      int4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 1024 < total ? i0 + u * 1024 : total - 1;
        int lo = 0, hi = nsb;                          // the last superblock whose offset is <= i (empty ones in front of it share the offset)
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (offs[mid] <= i) lo = mid; else hi = mid; }
        v[u] = src[(int64_t)lo * cap_per_sb + (i - offs[lo])];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i0 + u * 1024 < total) dst[i0 + u * 1024] = v[u];
    }
  } else {
    // one wavefront per superblock: its (few) tiles move as consecutive int4
    for (int s = wave; s < nsb; s += 16) {
      const int lo = sb_cnt[s], hi = (s + 1 < nsb) ? sb_cnt[s + 1] : total;
      for (int j = lane; lo + j < hi && j < cap_per_sb; j += 64)
        if (lo + j < ntiles_cap) dst[lo + j] = src[(int64_t)s * cap_per_sb + j];
    }
  }
  for (int i = total + threadIdx.x; i < ntiles_cap + 2; i += 1024) dst[i] = make_int4(0, 0, 0, 0);
}

// tok_tile[t] = (tile << 6) | row for every token of every planned tile (one 64-thread block per tile slot; slots past the count are zero)
__global__ __launch_bounds__(64) void tok_tile_kernel(const int32_t* __restrict__ meta, int32_t* __restrict__ tok_tile) {
  const int4 m = reinterpret_cast<const int4*>(meta)[blockIdx.x];
  if ((int)threadIdx.x < m.y) tok_tile[m.x + threadIdx.x] = (int32_t)((blockIdx.x << 6) | threadIdx.x);
}

// ---- the whole plan in ONE launch for small batches (B <= 1024 rows, <= 8 planning superblocks: the reference's own 384-row step) ------------
// The five kernels above are five dependent launches of a few microseconds each -- at 384 rows a tenth of the step (main.py:527-528).  Here
// one workgroup of 1024 threads does the same work in the same order and writes the SAME plan bit for bit (tests/test_hip_kernels.py compares
// both paths with oracle/c/ragged_plan.c): thread b owns hyperedge b (count, block scan, token lists), wavefronts pack the superblocks of the
// two lists side by side, the lists are compacted by offset, the token -> tile map follows.
constexpr int kSmallRows = 1024;
constexpr int kSmallSb = 8;
struct PlanSmallArgs {
  const int64_t* x; int64_t B; int L; int64_t n_nodes; int32_t* status;
  int32_t *row_off, *tok_slot; int64_t* tok_id; int32_t *tok_pos, *tok_key, *count, *sb_first;
  int nsb, level;
  int32_t* sb_tiles[2]; int cap_per_sb[2]; int ntiles_cap[2]; int32_t* meta[2];
  int32_t* tok_tile;
};
__global__ __launch_bounds__(1024) void plan_small_kernel(PlanSmallArgs a) {
  __shared__ int wtot[16];
  __shared__ int sbf[kSmallSb + 1];
  __shared__ int cnt[2][kSmallSb];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = a.L;
  const int64_t B = a.B;
  const bool on = tid < B;
  int64_t v[MATCHA_MAX_L];
  int k = 0;
#pragma unroll
  for (int l = 0; l < MATCHA_MAX_L; ++l) {
    v[l] = (on && l < L) ? a.x[(int64_t)tid * L + l] : 0;
    k += v[l] != 0 ? 1 : 0;
  }
  if (tid <= a.nsb && tid <= kSmallSb) sbf[tid] = (int)B;              // superblocks in which no hyperedge starts: empty range
  int incl = k;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(incl, o, 64);
    if (lane >= o) incl += u;
  }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  int wbase = 0, total = 0;
#pragma unroll
  for (int w2 = 0; w2 < 16; ++w2) { const int t = wtot[w2]; if (w2 < wave) wbase += t; total += t; }
  int pos = wbase + incl - k;
  const int kprev = __shfl_up(k, 1, 64);
  int kp = lane > 0 ? kprev : 0;
  // (the row in front of a wave's first row: its k through LDS)
  __shared__ int klast[16];
  if (lane == 63) klast[wave] = k;
  __syncthreads();
  if (lane == 0 && wave > 0) kp = klast[wave - 1];
  if (on) {
    a.row_off[tid] = pos;
    if (tid == 0 || pos / kSuperTok != (pos - kp) / kSuperTok) sbf[pos / kSuperTok] = tid;
    int nth = 0, p = pos;
#pragma unroll
    for (int l = 0; l < MATCHA_MAX_L; ++l) {
      int64_t id = v[l];
      if (id != 0) {
        if (id < 0 || id > a.n_nodes) {                  // the reference raises IndexError here (nn.Embedding, Modules.py:34)
          if (a.status) atomicOr(a.status, MATCHA_STATUS_BAD_ID);
          id = 0;
        }
        a.tok_slot[p] = (int32_t)(tid * L + l); a.tok_id[p] = id; a.tok_key[p] = (int32_t)id; a.tok_pos[p] = nth | (k << 8); ++p; ++nth;
      }
    }
  }
  if (tid == 0) {
    a.row_off[B] = total;
    a.tok_slot[total] = (int32_t)(B * L);
    a.tok_id[total] = 0;
    a.tok_pos[total] = 0;
    a.count[0] = total + 1;
    a.count[1] = total;
    if (a.level <= 0) { a.count[2] = 0; a.count[3] = 0; }
  }
  for (int64_t i = total + tid; i < B * L + 1; i += 1024) a.tok_key[i] = 0;
  __syncthreads();                                       // row_off (global) and sbf (LDS) are complete
  if (tid <= a.nsb) a.sb_first[tid] = sbf[tid < kSmallSb ? tid : kSmallSb];
  if (a.level <= 0) return;
  const int first = a.level >= 2 ? 0 : 1;                // list 0: 64-row tiles, list 1: half tiles
  for (int task = wave; task < a.nsb * (2 - first); task += 16) {
    const int which = first + task / a.nsb, sb = task - (task / a.nsb) * a.nsb;
    const int nt = tile_pack_wave(a.row_off, B, a.nsb, sb, which != 0, a.cap_per_sb[which], sbf,
                                  reinterpret_cast<int4*>(a.sb_tiles[which]) + (int64_t)sb * a.cap_per_sb[which], lane);
    if (lane == 0) cnt[which][sb] = nt;
  }
  __syncthreads();                                       // the per-superblock lists (global scratch) and their counts
  for (int which = first; which < 2; ++which) {
    int off[kSmallSb + 1];
    int run = 0;
    for (int sb = 0; sb < a.nsb; ++sb) { off[sb] = run; run += cnt[which][sb]; }
    const int tot = run < a.ntiles_cap[which] ? run : a.ntiles_cap[which];
    if (tid == 0) a.count[2 + which] = tot;
    const int4* src = reinterpret_cast<const int4*>(a.sb_tiles[which]);
    int4* dst = reinterpret_cast<int4*>(a.meta[which]);
    for (int sb = wave; sb < a.nsb; sb += 16)
      for (int j = lane; j < cnt[which][sb]; j += 64)
        if (off[sb] + j < a.ntiles_cap[which]) dst[off[sb] + j] = src[(int64_t)sb * a.cap_per_sb[which] + j];
    for (int i = tot + tid; i < a.ntiles_cap[which] + 2; i += 1024) dst[i] = make_int4(0, 0, 0, 0);
  }
  if (first != 0 && tid == 0) a.count[2] = 0;            // half tiles only: no 64-row tile list this time
  if (a.level >= 2) {
    __syncthreads();                                     // tile_meta complete
    const int4* tm = reinterpret_cast<const int4*>(a.meta[0]);
    for (int t = wave; t < a.ntiles_cap[0]; t += 16) {
      const int4 m = tm[t];
      if (lane < m.y) a.tok_tile[m.x + lane] = (int32_t)((t << 6) | lane);
    }
  }
}

static inline int super_blocks(int64_t T) { return (int)cdiv(T + 1, kSuperTok); }
static inline int super_cap(int L) { return (int)cdiv(kSuperTok + L, 64 - L) + 2; }
static inline int tiles_cap(int64_t T, int L) { return (int)cdiv(T + 1, 64 - L) + super_blocks(T); }   // every tile but a superblock's last holds > 63 - L tokens
static inline int super_hcap(int L) { return (int)cdiv(kSuperTok + L, 32 - L) + 2; }
static inline int halves_cap(int64_t T, int L) { return (int)cdiv(T + 1, 32 - L) + super_blocks(T); }  // ... > 31 - L tokens

int ragged_tiles_cap(int64_t B, int L) { return tiles_cap(B * L, L); }
int ragged_halves_cap(int64_t B, int L) { return halves_cap(B * L, L); }

size_t ragged_bytes(int64_t B, int L) {
  const int64_t T = B * L;
  size_t n = 0;
  n += align_up((size_t)(B + 1) * 4, 256);        // row_off
  n += align_up((size_t)(T + 1) * 4, 256);        // tok_slot
  n += align_up((size_t)(T + 1) * 8, 256);        // tok_id
  n += 256;                                        // count
  n += align_up((size_t)cdiv(B, kRowsPerBlock) * 4, 256);
  n += align_up((size_t)(T + 1) * 4, 256);        // tok_pos
  n += align_up((size_t)(T + 1) * 4, 256);        // tok_key
  n += align_up((size_t)(tiles_cap(T, L) + 2) * 16, 256);                    // tile_meta
  n += align_up((size_t)super_blocks(T) * super_cap(L) * 16, 256);           // sb_tiles
  n += align_up((size_t)super_blocks(T) * 4, 256);                           // sb_cnt
  n += align_up((size_t)(super_blocks(T) + 1) * 4, 256);                     // sb_first
  n += align_up((size_t)(halves_cap(T, L) + 2) * 16, 256);                   // half_meta
  n += align_up((size_t)super_blocks(T) * super_hcap(L) * 16, 256);          // sb_htiles
  n += align_up((size_t)super_blocks(T) * 4, 256);                           // sb_hcnt
  n += align_up((size_t)(T + 1) * 4, 256);                                   // tok_tile
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
  r.tok_key = (int32_t*)take((size_t)(T + 1) * 4);
  r.tile_meta = (int32_t*)take((size_t)(r.ntiles + 2) * 16);
  r.nsb = super_blocks(T);
  r.sb_cap = super_cap(L);
  r.sb_tiles = (int32_t*)take((size_t)r.nsb * r.sb_cap * 16);
  r.sb_cnt = (int32_t*)take((size_t)r.nsb * 4);
  r.sb_first = (int32_t*)take((size_t)(r.nsb + 1) * 4);
  r.nhalves = halves_cap(T, L);
  r.sb_hcap = super_hcap(L);
  r.half_meta = (int32_t*)take((size_t)(r.nhalves + 2) * 16);
  r.sb_htiles = (int32_t*)take((size_t)r.nsb * r.sb_hcap * 16);
  r.sb_hcnt = (int32_t*)take((size_t)r.nsb * 4);
  r.tok_tile = (int32_t*)take((size_t)(T + 1) * 4);
}

// level 2: everything; 1: no 64-row tile list and no token -> tile map (the fused kernels that run work on half tiles only); 0: rows and
// tokens only (no fused kernel will read a tile list)
int launch_ragged_plan(const int64_t* x, int64_t B, int L, int64_t n_nodes, int32_t* status, const Ragged& r, hipStream_t st, int level) {
  if (B <= kSmallRows && r.nsb <= kSmallSb && !options().disable_small_batch) {
    PlanSmallArgs a;
    a.x = x; a.B = B; a.L = L; a.n_nodes = n_nodes; a.status = status;
    a.row_off = r.row_off; a.tok_slot = r.tok_slot; a.tok_id = r.tok_id; a.tok_pos = r.tok_pos; a.tok_key = r.tok_key; a.count = r.count; a.sb_first = r.sb_first;
    a.nsb = r.nsb; a.level = level;
    a.sb_tiles[0] = r.sb_tiles; a.cap_per_sb[0] = r.sb_cap; a.ntiles_cap[0] = r.ntiles; a.meta[0] = r.tile_meta;
    a.sb_tiles[1] = r.sb_htiles; a.cap_per_sb[1] = r.sb_hcap; a.ntiles_cap[1] = r.nhalves; a.meta[1] = r.half_meta;
    a.tok_tile = r.tok_tile;
    hipLaunchKernelGGL(plan_small_kernel, dim3(1), dim3(1024), 0, st, a);
    MATCHA_CHECK_LAUNCH("plan_small_kernel");
    return MATCHA_OK;
  }
  hipLaunchKernelGGL(row_count_kernel, dim3(r.nblk), dim3(256), 0, st, x, B, L, r.blk_sum);
  MATCHA_CHECK_LAUNCH("row_count_kernel");
  hipLaunchKernelGGL(row_scan_kernel, dim3(1), dim3(1024), 0, st, r.blk_sum, r.nblk, r.count, r.sb_first, r.nsb, (int32_t)B);
  MATCHA_CHECK_LAUNCH("row_scan_kernel");
  hipLaunchKernelGGL(row_fill_kernel, dim3(r.nblk), dim3(256), 0, st, x, B, L, r.blk_sum, r.count, r.row_off, r.tok_slot, r.tok_id, r.tok_pos, r.sb_first, kSuperTok,
                     r.tok_key, n_nodes, status);
  MATCHA_CHECK_LAUNCH("row_fill_kernel");
  if (level <= 0) return MATCHA_OK;
  const int first = level >= 2 ? 0 : 1;                // list 0: 64-row tiles, list 1: half tiles
  hipLaunchKernelGGL(tile_pack_kernel, dim3(r.nsb, 2 - first), dim3(64), 0, st, r.row_off, B, r.nsb, r.sb_cap, r.sb_hcap, r.sb_first, r.sb_tiles, r.sb_cnt,
                     r.sb_htiles, r.sb_hcnt, first);
  MATCHA_CHECK_LAUNCH("tile_pack_kernel");
  CompactArgs ca;
  ca.sb_tiles[0] = r.sb_tiles; ca.sb_cnt[0] = r.sb_cnt; ca.cap_per_sb[0] = r.sb_cap; ca.ntiles_cap[0] = r.ntiles; ca.meta[0] = r.tile_meta;
  ca.sb_tiles[1] = r.sb_htiles; ca.sb_cnt[1] = r.sb_hcnt; ca.cap_per_sb[1] = r.sb_hcap; ca.ntiles_cap[1] = r.nhalves; ca.meta[1] = r.half_meta;
  hipLaunchKernelGGL(tile_compact_kernel, dim3(2 - first), dim3(1024), 0, st, ca, r.nsb, r.count, first);
  MATCHA_CHECK_LAUNCH("tile_compact_kernel");
  if (level >= 2) {
    hipLaunchKernelGGL(tok_tile_kernel, dim3(r.ntiles), dim3(64), 0, st, r.tile_meta, r.tok_tile);
    MATCHA_CHECK_LAUNCH("tok_tile_kernel");
  }
  return MATCHA_OK;
}

}  // namespace matcha

using namespace matcha;

extern "C" size_t matcha_ragged_plan_bytes(int64_t B, int32_t L) {
  if (B < 1 || L < 1 || L > MATCHA_MAX_L || B * (int64_t)L >= (1ll << 31) - 2) return 0;
  return ragged_bytes(B, L);
}

extern "C" int matcha_ragged_plan(const int64_t* x, int64_t B, int32_t L, int32_t n_nodes, int32_t* status, void* ws, size_t ws_bytes,
                                  matcha_ragged_view* view, matcha_stream_t stream) {
  MATCHA_CHECK_ARG(x && ws && view, "matcha_ragged_plan: null pointer");
  MATCHA_CHECK_ARG(B >= 1 && L >= 1 && L <= MATCHA_MAX_L && B * (int64_t)L < (1ll << 31) - 2, "matcha_ragged_plan: B=%lld L=%d", (long long)B, L);
  MATCHA_CHECK_ARG(n_nodes >= 1, "matcha_ragged_plan: n_nodes=%d", n_nodes);
  MATCHA_CHECK_ARG(((uintptr_t)ws) % 256 == 0, "matcha_ragged_plan: workspace must be 256-byte aligned");
  if (ws_bytes < ragged_bytes(B, L)) { set_error("matcha_ragged_plan: workspace %zu < %zu bytes", ws_bytes, ragged_bytes(B, L)); return MATCHA_ENOMEM; }
  Ragged r;
  ragged_carve(B, L, (char*)ws, r);
  view->row_off = r.row_off; view->tok_slot = r.tok_slot; view->tok_id = r.tok_id; view->tok_key = r.tok_key; view->tok_pos = r.tok_pos;
  view->count = r.count; view->tile_meta = r.tile_meta; view->tiles_cap = r.ntiles;
  view->half_meta = r.half_meta; view->halves_cap = r.nhalves; view->tok_tile = r.tok_tile;
  return launch_ragged_plan(x, B, L, n_nodes, status, r, (hipStream_t)stream, 2);
}
