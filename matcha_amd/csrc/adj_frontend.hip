// adj-mode front end: MultipleEmbedding.forward (Modules.py:176-201) and its backward on the GPU.
//
//   per chromosome i:  rows = feats_i[x - lo_i]  (SparseEmbedding :67, the gather)  -> dropout(0.2) (:186)
//                      -> W0_i (d x n_i, no bias) -> tanh -> W1_i (d x d)  (TiedAutoEncoder :104-113) -> final[sel] (:188)
//   recon branch:      for tokens outside chromosome r (r drawn by the caller, :192) and not padding:
//                      100 * mean( (inter[x-1, cols of r] - Linear_r(tanh(final)))^2 )   (:194-199)
//
// GPU formulation: token slots are bucket-sorted by chromosome once per step (stable counting sort -> `order`,
// `seg`), so each chromosome owns a contiguous run of sorted rows and the per-chromosome Python loop becomes
//   * one gather-GEMM launch (feature rows gathered by node id straight into LDS tiles, dropout applied while
//     staging, f32 MFMA against W0_i, tanh)            -> Hs [sorted rows, d]
//   * one grouped launch of the generic LDS GEMM (per-group weight W1_i, C rows scattered back to token slots).
// A workgroup whose 128-row tile straddles a chromosome boundary simply processes both sub-ranges.
// Weight gradients of the per-chromosome tensors are accumulated with float atomics (row-contiguous 128-B
// segments); they are the only non-bitwise-reproducible sums of the adj path.
#include <string.h>

#include "adj_common.hpp"

namespace matcha {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kSortTok = 1024;   // tokens per workgroup in the counting sort
constexpr int kLdA = 68;

static size_t adj_carve(const matcha_shape& s, int64_t T, char* base, AdjWs& w) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  const int C = s.n_chrom;
  w.nblk = (int)cdiv(T, kSortTok);
  w.nr_pad = align_up((size_t)(s.max_bins > 0 ? s.max_bins : 1), 4);
  w.order = (int32_t*)take(T * 4);
  w.other_map = (int32_t*)take(T * 4);
  w.seg = (int32_t*)take((C + 2) * 4);
  w.counts = (int32_t*)take(16);
  w.hist = (int32_t*)take((size_t)w.nblk * (C + 1) * 4);
  w.base = (int32_t*)take((size_t)w.nblk * (C + 1) * 4);
  w.Hs = (float*)take((size_t)T * s.d * 4);
  w.TH = (float*)take((size_t)T * s.d * 4);
  w.rec = (float*)take((size_t)T * w.nr_pad * 4);
  w.dTH = (float*)take((size_t)T * s.d * 4);
  w.dZ = (float*)take((size_t)T * s.d * 4);
  w.lossslab = (float*)take((size_t)(cdiv(T, 64) + 8) * 4);
  w.rgrad = (float*)take((size_t)w.nr_pad * (s.d + 1) * 4);
  w.total = off;
  return off;
}

size_t adj_workspace_bytes(const matcha_shape& s, int64_t T) {
  AdjWs w;
  return adj_carve(s, T, nullptr, w);
}

__device__ __forceinline__ int chrom_of(int64_t id, const int32_t* __restrict__ bounds, int C) {
  if (id <= 0 || id > bounds[C]) return C;        // padding bucket; ids outside [1, N] land there too (flagged by the caller's status word), so nothing indexes out of bounds
  int c = 0;
  while (c + 1 < C && id > bounds[c + 1]) ++c;   // ids of chromosome c: bounds[c]+1 .. bounds[c+1]
  return c;
}

// ---- stable counting sort of token slots by chromosome -------------------------------------------------------
__global__ __launch_bounds__(256) void adj_hist_kernel(const int64_t* __restrict__ x, int64_t T, const int32_t* __restrict__ bounds, int C,
                                                       int32_t* __restrict__ hist, const int32_t* __restrict__ t_dev) {
  __shared__ int cnt[kMaxChrom + 1];
  if (t_dev) T = *t_dev;                 // ragged layout: token count on the device; slots beyond it are ignored
  if (threadIdx.x <= C) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t t0 = (int64_t)blockIdx.x * kSortTok + threadIdx.x * 4;
  for (int i = 0; i < 4; ++i)
    if (t0 + i < T) atomicAdd(&cnt[chrom_of(x[t0 + i], bounds, C)], 1);
  __syncthreads();
  if (threadIdx.x <= C) hist[(int64_t)blockIdx.x * (C + 1) + threadIdx.x] = cnt[threadIdx.x];
}

// one workgroup: seg[] (segment starts), base[blk][c] (first sorted position of block blk's tokens of bucket c),
// counts[0] = number of "other" tokens (non-pad, not in chromosome r), counts[1] = non-pad tokens; touched flags
// 1024 threads: 16 lanes per bucket, each lane owns a contiguous chunk of sort blocks (local sums -> 16-lane scan).
// The histogram is staged in LDS with coalesced loads when it fits (lds_ints >= nblk * (C + 1): up to ~350 K tokens at 23 chromosomes) and
// the prefix sums are written back the same way -- walking it in global memory was two chains of strided dependent loads (23 us at 65 536
// rows); larger batches keep that path.
__global__ __launch_bounds__(1024) void adj_scan_kernel(const int32_t* __restrict__ hist, int nblk, int C, int r_chrom, int32_t* __restrict__ base,
                                                        int32_t* __restrict__ seg, int32_t* __restrict__ counts, int32_t* __restrict__ touched, int lds_ints,
                                                        const int32_t* __restrict__ r_dev, float* __restrict__ zero_buf, int zero_n) {
  extern __shared__ int hs[];
  // the fused reconstruction kernel's gradient scratch, zeroed here instead of by a memset in front of it (one launch less; and a memset
  // NODE of a captured step was seen to run unordered with its consumer -- tools/debug/graph_vs_eager.py)
  for (int i = threadIdx.x; i < zero_n; i += 1024) zero_buf[i] = 0.f;
  if (r_dev) { const int rv = *r_dev; r_chrom = (rv >= 0 && rv < C) ? rv : -1; }     // opts->random_chrom_dev
  __shared__ int tot[kMaxChrom + 2];
  __shared__ int segs[kMaxChrom + 2];
  const int n = nblk * (C + 1);
  const bool staged = lds_ints >= n;
  if (staged) {
    for (int i = threadIdx.x; i < n; i += 1024) hs[i] = hist[i];
    __syncthreads();
  }
  const int c = threadIdx.x >> 4, sl = threadIdx.x & 15;
  const int chunk = (nblk + 15) / 16;
  const int b0 = sl * chunk, b1 = (b0 + chunk < nblk) ? b0 + chunk : nblk;
  int local = 0;
  if (c <= C)
    for (int b = b0; b < b1; ++b) local += staged ? hs[b * (C + 1) + c] : hist[(int64_t)b * (C + 1) + c];
  int incl = local;                                  // inclusive scan over the 16 lanes of the bucket
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
    const int v = __shfl_up(incl, o, 16);
    if (sl >= o) incl += v;
  }
  const int my_start = incl - local;
  if (c <= C && sl == 15) tot[c] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int k = 0; k <= C; ++k) { segs[k] = run; seg[k] = run; run += tot[k]; }
    seg[C + 1] = run;
    const int nonpad = segs[C];
    const int in_r = (r_chrom >= 0 && r_chrom < C) ? tot[r_chrom] : 0;
    counts[0] = (r_chrom >= 0) ? nonpad - in_r : 0;
    counts[1] = nonpad;
    if (touched) {
      touched[0] = 1;
      touched[1] = 0;
      for (int k = 0; k < C; ++k) {
        touched[2 + k] = tot[k] > 0 ? 1 : 0;                                  // wstack[k] ran (Modules.py:182-183)
        touched[2 + C + k] = (k == r_chrom && nonpad - in_r > 0) ? 1 : 0;     // recon[r] ran (:195)
      }
    }
  }
  __syncthreads();
  if (c <= C) {
    int run = segs[c] + my_start;
    for (int b = b0; b < b1; ++b) {
      if (staged) { const int v = hs[b * (C + 1) + c]; hs[b * (C + 1) + c] = run; run += v; }
      else { const int v = hist[(int64_t)b * (C + 1) + c]; base[(int64_t)b * (C + 1) + c] = run; run += v; }
    }
  }
  if (staged) {
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) base[i] = hs[i];
  }
}

// touched flags only (backward): which per-chromosome tensors received a gradient this step
__global__ void adj_flags_kernel(const int32_t* __restrict__ seg, const int32_t* __restrict__ counts, int C, int r_chrom, int32_t* __restrict__ touched,
                                 const int32_t* __restrict__ r_dev) {
  if (r_dev) { const int rv = *r_dev; r_chrom = (rv >= 0 && rv < C) ? rv : -1; }
  const int k = threadIdx.x;
  if (k == 0) { touched[0] = 1; touched[1] = 0; }
  if (k < C) {
    touched[2 + k] = seg[k + 1] > seg[k] ? 1 : 0;
    touched[2 + C + k] = (k == r_chrom && counts[0] > 0) ? 1 : 0;
  }
}

__global__ __launch_bounds__(256) void adj_scatter_kernel(const int64_t* __restrict__ x, int64_t T, const int32_t* __restrict__ bounds, int C,
                                                          int r_chrom, const int32_t* __restrict__ base, const int32_t* __restrict__ seg,
                                                          int32_t* __restrict__ order, int32_t* __restrict__ other_map,
                                                          const int32_t* __restrict__ t_dev, const int32_t* __restrict__ r_dev) {
  extern __shared__ int tc[];
  if (r_dev) { const int rv = *r_dev; r_chrom = (rv >= 0 && rv < C) ? rv : -1; }
  if (t_dev) T = *t_dev;                 // [C+1][256] per-thread counts -> exclusive prefix
  const int tid = threadIdx.x;
  for (int k = 0; k <= C; ++k) tc[k * 256 + tid] = 0;
  const int64_t t0 = (int64_t)blockIdx.x * kSortTok + tid * 4;
  int ch[4];
  for (int i = 0; i < 4; ++i) {
    ch[i] = (t0 + i < T) ? chrom_of(x[t0 + i], bounds, C) : -1;
    if (ch[i] >= 0) tc[ch[i] * 256 + tid] += 1;
  }
  __syncthreads();
  // exclusive prefix of every bucket's 256 per-thread counts: one wavefront per bucket in turn, four consecutive counts per lane, one
  // 64-lane shuffle scan (a single thread walking the 256 counts of its bucket was a chain of 512 dependent LDS accesses: 13 us)
  {
    const int lane = tid & 63, wave = tid >> 6;
    for (int k = wave; k <= C; k += 4) {
      int* row = tc + k * 256 + 4 * lane;
      const int v0 = row[0], v1 = row[1], v2 = row[2], v3 = row[3];
      const int sum = (v0 + v1) + (v2 + v3);
      int incl = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(incl, o, 64);
        if (lane >= o) incl += u;
      }
      const int base = incl - sum;
      row[0] = base; row[1] = base + v0; row[2] = base + v0 + v1; row[3] = base + v0 + v1 + v2;
    }
  }
  __syncthreads();
  const int r0 = (r_chrom >= 0 && r_chrom < C) ? seg[r_chrom] : 0, r1 = (r_chrom >= 0 && r_chrom < C) ? seg[r_chrom + 1] : 0;
  const int nonpad = seg[C];
  for (int i = 0; i < 4; ++i) {
    if (ch[i] < 0) continue;
    const int pos = base[(int64_t)blockIdx.x * (C + 1) + ch[i]] + tc[ch[i] * 256 + tid];
    tc[ch[i] * 256 + tid] += 1;
    order[pos] = (int32_t)(t0 + i);
    if (r_chrom >= 0 && pos < nonpad && !(pos >= r0 && pos < r1)) other_map[pos < r0 ? pos : pos - (r1 - r0)] = (int32_t)(t0 + i);
  }
}

// The same sort for SMALL batches (T <= 4096 token slots: the reference's own 384-row step) in ONE launch: one workgroup of 512 threads,
// eight consecutive slots per thread; per-thread bucket counts in LDS, their exclusive prefix per bucket by one wavefront (eight counts
// per lane + a 64-lane shuffle scan), the segment starts by thread 0, then every thread places its slots.  Same outputs as
// hist + scan + scatter (order, other_map, seg, counts, touched; the reconstruction scratch zeroed), which it replaces at that size.
constexpr int kSortSmallTok = 4096;
__global__ __launch_bounds__(512) void adj_sort_small_kernel(const int64_t* __restrict__ x, int64_t T, const int32_t* __restrict__ bounds, int C, int r_chrom,
                                                             int32_t* __restrict__ seg, int32_t* __restrict__ counts, int32_t* __restrict__ touched,
                                                             int32_t* __restrict__ order, int32_t* __restrict__ other_map,
                                                             const int32_t* __restrict__ t_dev, const int32_t* __restrict__ r_dev,
                                                             float* __restrict__ zero_buf, int zero_n) {
  extern __shared__ int tc[];                       // [C + 1][512]
  __shared__ int tot[kMaxChrom + 2];
  __shared__ int segs[kMaxChrom + 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < zero_n; i += 512) zero_buf[i] = 0.f;
  if (r_dev) { const int rv = *r_dev; r_chrom = (rv >= 0 && rv < C) ? rv : -1; }
  if (t_dev) T = *t_dev;
  for (int k = 0; k <= C; ++k) tc[k * 512 + tid] = 0;
  const int64_t t0 = (int64_t)tid * 8;
  int ch[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    ch[i] = (t0 + i < T) ? chrom_of(x[t0 + i], bounds, C) : -1;
    if (ch[i] >= 0) tc[ch[i] * 512 + tid] += 1;
  }
  __syncthreads();
  for (int k = wave; k <= C; k += 8) {
    int* row = tc + k * 512 + 8 * lane;
    int v[8], sum = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = row[i]; sum += v[i]; }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o, 64);
      if (lane >= o) incl += u;
    }
    int run = incl - sum;
#pragma unroll
    for (int i = 0; i < 8; ++i) { row[i] = run; run += v[i]; }
    if (lane == 63) tot[k] = incl;
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int k = 0; k <= C; ++k) { segs[k] = run; seg[k] = run; run += tot[k]; }
    seg[C + 1] = run;
    segs[C + 1] = run;
    const int nonpad = segs[C];
    const int in_r = (r_chrom >= 0 && r_chrom < C) ? tot[r_chrom] : 0;
    counts[0] = (r_chrom >= 0) ? nonpad - in_r : 0;
    counts[1] = nonpad;
    if (touched) {
      touched[0] = 1;
      touched[1] = 0;
      for (int k = 0; k < C; ++k) {
        touched[2 + k] = tot[k] > 0 ? 1 : 0;
        touched[2 + C + k] = (k == r_chrom && nonpad - in_r > 0) ? 1 : 0;
      }
    }
  }
  __syncthreads();
  const int r0 = (r_chrom >= 0 && r_chrom < C) ? segs[r_chrom] : 0, r1 = (r_chrom >= 0 && r_chrom < C) ? segs[r_chrom + 1] : 0;
  const int nonpad = segs[C];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (ch[i] < 0) continue;
    const int pos = segs[ch[i]] + tc[ch[i] * 512 + tid];
    tc[ch[i] * 512 + tid] += 1;
    order[pos] = (int32_t)(t0 + i);
    if (r_chrom >= 0 && pos < nonpad && !(pos >= r0 && pos < r1)) other_map[pos < r0 ? pos : pos - (r1 - r0)] = (int32_t)(t0 + i);
  }
}

// ---- gather-GEMM: Hs[p] = tanh( (feats_c[x - lo_c] * dropmask) . W0_c^T ) for sorted rows p ---------------------
struct AdjEncArgs {
  const int64_t* x;
  const int32_t *order, *seg, *bounds;
  const int64_t* feat_off;
  const float *feats, *w0;
  float* Hs;
  int64_t T;
  int C, d;
  int feat_pad;                // matcha_frozen.feat_row_pad
  int splits;                  // workgroups per chromosome (grid = C * splits); a workgroup takes every splits-th 128-row step
  const uint64_t* seed;
  float p_drop;
  const int32_t* slot_map;     // token index -> original slot (dropout counter); null = identity
};

template <int NT>   // NT = ceil(d / 32) column tiles per wave
__global__ __launch_bounds__(256) void adj_encode_fwd_kernel(AdjEncArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                              // [128][68]
  float* Bs = lds + 128 * kLdA;                 // [32*NT][68]
  __shared__ int64_t rowoff[128];               // element offset of the gathered feature row inside feats
  __shared__ int rowslot[128];                  // token slot (dropout counter), -1 = row not in this sub-range
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  // chromosome-major: workgroup (c, sp) takes the 128-row steps sp, sp + splits, ... of chromosome c, so a small batch
  // spreads over C workgroups instead of walking several chromosomes serially inside a few
  const int c = blockIdx.x / g.splits, sp = blockIdx.x - c * g.splits;
  const bool drop = g.p_drop > 0.f;
  uint32_t key = 0, thr = 0;
  float keep_scale = 1.f;
  if (drop) { key = rng_key(*g.seed, kStreamDropAdj); thr = dropout_threshold(g.p_drop); keep_scale = 1.f / (1.f - g.p_drop); }
  const int64_t c_lo = g.seg[c], c_hi = g.seg[c + 1];
  for (int64_t m0 = c_lo + (int64_t)sp * 128; m0 < c_hi; m0 += (int64_t)g.splits * 128) {
    const int64_t row_lo = m0;
    const int64_t row_hi = m0 + 128 < c_hi ? m0 + 128 : c_hi;
    const int lo = g.bounds[c], n_c = g.bounds[c + 1] - g.bounds[c];
    const float* W0 = g.w0 + (int64_t)g.d * lo;                 // [d, n_c] row-major
    __syncthreads();
    if (tid < 128) {
      const int64_t p = m0 + tid;
      const bool in = p >= row_lo && p < row_hi;
      const int64_t pc = in ? p : row_lo;                         // out-of-range rows alias a valid row (never stored)
      const int slot = g.order[pc];
      rowoff[tid] = g.feat_off[c] + (g.x[slot] - lo - 1) * (int64_t)feat_ld(n_c, g.feat_pad);
      rowslot[tid] = in ? slot : -1;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x16){0};
    // contraction in chunks of 64 feature columns, software-pipelined: the loads of chunk kc + 64 (8 feature-row windows and 2 NT weight-row
    // windows per lane) are in flight while the MFMAs of chunk kc run.  A lane reads four consecutive floats, a wave-instruction 64 floats of
    // each of FOUR rows; the wave's 32 rows are 8 instructions.
    const int sub = lane >> 4, c4 = (lane & 15) * 4;
    f4u rawa[8], raww[2 * NT];
#define ENC_GLOAD(KC)                                                                                    \
  do {                                                                                                   \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) rawa[u] = row4_load(g.feats + rowoff[wave + 4 * (4 * u + sub)], (KC) + c4, n_c); \
    _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                       \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                    \
        const int j = wave + 32 * t + 4 * (4 * u + sub);                                                 \
        raww[2 * t + u] = row4_load(W0 + (int64_t)(j < g.d ? j : g.d - 1) * n_c, (KC) + c4, n_c);        \
      }                                                                                                  \
  } while (0)
    __syncthreads();                                  // rowoff / rowslot of this row step are in LDS
    ENC_GLOAD(0);
    for (int kc = 0; kc < n_c; kc += 64) {
      const int col0 = kc + c4;
      __syncthreads();                                // the previous chunk's MFMAs are done with the tiles
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int rr = wave + 4 * (4 * u + sub);
        float e[4];
        row4_fix(rawa[u], col0, n_c, e);
        if (drop) {
          int slot = rowslot[rr] < 0 ? 0 : rowslot[rr];
          if (g.slot_map) slot = g.slot_map[slot];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            e[j] = (rng_u32(key, (uint32_t)slot, (uint32_t)(col0 + j)) >= thr) ? e[j] * keep_scale : 0.f;   // counter = (token slot, column)
        }
        *reinterpret_cast<float4*>(&As[rr * kLdA + c4]) = make_float4(e[0], e[1], e[2], e[3]);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int j = wave + 32 * t + 4 * (4 * u + sub);
          float e[4];
          row4_fix(raww[2 * t + u], col0, n_c, e);
          *reinterpret_cast<float4*>(&Bs[j * kLdA + c4]) = (j < g.d) ? make_float4(e[0], e[1], e[2], e[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      __syncthreads();
      if (kc + 64 < n_c) ENC_GLOAD(kc + 64);          // in flight during this chunk's MFMAs
      const float* arow = &As[(32 * wave + r) * kLdA + 4 * h];
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) {
        const float4 a = *reinterpret_cast<const float4*>(arow + 8 * cc);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float4 b = *reinterpret_cast<const float4*>(&Bs[(32 * t + r) * kLdA + 8 * cc + 4 * h]);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[t], 0, 0, 0);
        }
      }
    }
#undef ENC_GLOAD
    const int64_t mrow0 = m0 + 32 * wave + 4 * h;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = 32 * t + r;
      if (col >= g.d) continue;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int64_t p = mrow0 + (reg & 3) + 8 * (reg >> 2);
        if (p < row_lo || p >= row_hi) continue;
        g.Hs[p * g.d + col] = tanhf(acc[t][reg]);
      }
    }
  }
}

constexpr size_t kAdjTnLds = (size_t)2 * 128 * 68 * sizeof(float);

// ---- grouped weight gradients with atomics -----------------------------------------------------------------------
//   MODE 0: dW1_c[j][k] += sum_p dnode[order[p]][j] * Hs[p][k]
//   MODE 1: dW0_c[j][col] += sum_p dZ[p][j] * feats_c[x_p - lo_c][col] * dropmask(slot_p, col)
struct AdjTnArgs {
  const float* A;            // MODE 0: dnode [T,d] (token-slot order);  MODE 1: dZ [T,d] (sorted order)
  const float* Bd;           // MODE 0: Hs [T,d]
  const int64_t* x;
  const int32_t *order, *seg, *bounds;
  const int64_t* feat_off;
  const float* feats;
  float* out;                // MODE 0: adj_w1 grads [C,d,d];  MODE 1: adj_w0 grads (chromosome c at d*bounds[c], row stride n_c)
  int C, d, rows_per_block;
  int chrom_parallel, feat_pad;
  const uint64_t* seed;
  float p_drop;
  const int32_t* slot_map;
};

// LDS-staged: per 128 sorted rows the workgroup resolves the row indices once (order -> slot -> feature row), stages the
// A rows (64 columns of d) and the B rows (64 feature columns, dropout applied) with coalesced 256-B row reads, and every
// wave accumulates its own 32x32 quadrant of the 64x64 output tile with the row index as the MFMA contraction index; a
// chromosome's tile is added to the gradient with atomics straight from the accumulators (waves own distinct quadrants).
template <int MODE>
__global__ __launch_bounds__(256, 2) void adj_tn_kernel(AdjTnArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                              // [128][kLdA]
  float* Bs = lds + 128 * kLdA;                 // [128][kLdA]
  __shared__ int64_t rowoff2[2][128];           // MODE 1: element offset of the feature row;  MODE 0: unused.  Two sets: the next step's
  __shared__ int rowslot2[2][128];              // token slot of the sorted row, -1 = past the range        indices are fetched during this step
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5, wr = wave & 1, wc = wave >> 1;
  const int mo0 = blockIdx.x * 64;             // output row tile (j)
  const int no0 = blockIdx.y * 64;             // output column tile (k or feature column)
  // blockIdx.z = a window of g.rows_per_block consecutive SORTED rows (a multiple of 128), whatever chromosomes it touches: one pass of the
  // loop below per chromosome in the window, each with its own accumulators and atomics.  (Equal shares of every chromosome -- the first
  // version -- gave the workgroups of chromosome 1 five times the rows of chromosome 21's and the kernel the duration of the longest.)
  // (Small batches, g.chrom_parallel: blockIdx.z = window * C + chromosome -- the few chromosomes of a window side by side, not in turn.)
  const int zwin = g.chrom_parallel ? (int)blockIdx.z / g.C : (int)blockIdx.z;
  const int csel = g.chrom_parallel ? (int)blockIdx.z - zwin * g.C : -1;
  const int64_t win_lo = (int64_t)zwin * g.rows_per_block;
  const int64_t win_hi = win_lo + g.rows_per_block;
  const bool drop = MODE == 1 && g.p_drop > 0.f;
  uint32_t key = 0, thr = 0;
  float keep_scale = 1.f;
  if (drop) { key = rng_key(*g.seed, kStreamDropAdj); thr = dropout_threshold(g.p_drop); keep_scale = 1.f / (1.f - g.p_drop); }
  const int acol = mo0 + lane, acolc = acol < g.d ? acol : g.d - 1;
  const float amask = acol < g.d ? 1.f : 0.f;
  if (win_lo >= g.seg[g.C]) return;
  for (int c = csel >= 0 ? csel : 0; c < (csel >= 0 ? csel + 1 : g.C); ++c) {
    const int64_t c_lo = g.seg[c], c_hi = g.seg[c + 1];
    if (c_hi <= win_lo) continue;
    if (c_lo >= win_hi) break;
    const int64_t p_lo = c_lo > win_lo ? c_lo : win_lo;
    const int64_t p_hi = c_hi < win_hi ? c_hi : win_hi;
    if (p_lo >= p_hi) continue;
    const int lo = g.bounds[c], n_c = g.bounds[c + 1] - g.bounds[c];
    const int ncols = MODE == 0 ? g.d : n_c;
    if (no0 >= ncols) continue;
    __syncthreads();                            // the previous chromosome's last step is done with the tiles and the index sets
    const int bcol = no0 + lane, bcolc = bcol < ncols ? bcol : ncols - 1;
    const float bmask = bcol < ncols ? 1.f : 0.f;
    f32x16 acc = {0};
    // index pipeline: sorted row -> token slot -> node id -> feature-row offset are two dependent global round trips; the indices of step
    // p0 + 128 are fetched while step p0 stages and multiplies (set `cur ^ 1`), so only the first step waits for them
    auto fetch_idx = [&](int64_t q0, int& slot_out, int64_t& off_out) {
      const int64_t p = q0 + tid;
      const bool in = p < p_hi;
      const int slot = g.order[in ? p : p_hi - 1];
      slot_out = in ? slot : -1;
      off_out = MODE == 1 ? g.feat_off[c] + (g.x[slot] - lo - 1) * (int64_t)feat_ld(n_c, g.feat_pad) : 0;
    };
    int cur = 0;
    if (tid < 128) {
      int sl0; int64_t of0;
      fetch_idx(p_lo, sl0, of0);
      rowslot2[0][tid] = sl0; rowoff2[0][tid] = of0;
    }
    for (int64_t p0 = p_lo; p0 < p_hi; p0 += 128) {
      __syncthreads();                          // previous step's MFMAs are done with the tiles; this step's index set is complete
      const int* rowslot = rowslot2[cur];
      const int64_t* rowoff = rowoff2[cur];
      int nsl = -1; int64_t nof = 0;
      const bool more = p0 + 128 < p_hi;
      if (more && tid < 128) fetch_idx(p0 + 128, nsl, nof);
      // a lane reads four consecutive floats, a wave-instruction 64 floats of each of FOUR rows; the wave's 32 rows = 8 instructions per
      // operand, all issued before the first is consumed (one float per lane and 8 rows in flight ran at a third of this)
      {
        const int sub = lane >> 4, c4 = (lane & 15) * 4;
        const int acol0 = mo0 + c4, bcol0 = no0 + c4;
        f4u ra[8], rb[8];
        int sl[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int rr = wave + 4 * (4 * u + sub);
          const int slot = rowslot[rr];
          sl[u] = slot;
          const int slc = slot >= 0 ? slot : 0;
          const int64_t p = p0 + rr < p_hi ? p0 + rr : p_hi - 1;
          const int64_t arow = MODE == 0 ? (int64_t)slc : p;
          ra[u] = row4_load(g.A + arow * g.d, acol0, g.d);
          rb[u] = MODE == 0 ? row4_load(g.Bd + p * g.d, bcol0, ncols) : row4_load(g.feats + rowoff[rr], bcol0, ncols);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int rr = wave + 4 * (4 * u + sub);
          const float rowm = sl[u] >= 0 ? 1.f : 0.f;
          const int slc = sl[u] >= 0 ? sl[u] : 0;
          float ea[4], eb[4];
          row4_fix(ra[u], acol0, g.d, ea);
          row4_fix(rb[u], bcol0, ncols, eb);
          if (drop) {
            const uint32_t slot = (uint32_t)(g.slot_map ? g.slot_map[slc] : slc);
#pragma unroll
            for (int j = 0; j < 4; ++j) eb[j] = (rng_u32(key, slot, (uint32_t)(bcol0 + j)) >= thr) ? eb[j] * keep_scale : 0.f;
          }
          *reinterpret_cast<float4*>(&As[rr * kLdA + c4]) = make_float4(ea[0] * rowm, ea[1] * rowm, ea[2] * rowm, ea[3] * rowm);
          *reinterpret_cast<float4*>(&Bs[rr * kLdA + c4]) = make_float4(eb[0] * rowm, eb[1] * rowm, eb[2] * rowm, eb[3] * rowm);
        }
      }
      __syncthreads();
#pragma unroll 8
      for (int m = 0; m < 64; ++m) {
        const int t = 2 * m + h;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[t * kLdA + 32 * wr + r], Bs[t * kLdA + 32 * wc + r], acc, 0, 0, 0);
      }
      if (more && tid < 128) { rowslot2[cur ^ 1][tid] = nsl; rowoff2[cur ^ 1][tid] = nof; }
      cur ^= 1;
    }
    float* out = MODE == 0 ? g.out + (int64_t)c * g.d * g.d : g.out + (int64_t)g.d * lo;
    const int col = no0 + 32 * wc + r;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = mo0 + 32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (row < g.d && col < ncols) atomicAdd(out + (int64_t)row * ncols + col, acc[reg]);
    }
  }
}

// ---- recon branch ------------------------------------------------------------------------------------------------
// TH[j] = tanh(node[other_map[j]]) for j < m
__global__ __launch_bounds__(256) void adj_tanh_gather_kernel(const float* __restrict__ node, const int32_t* __restrict__ other_map,
                                                              const int32_t* __restrict__ counts, int d, float* __restrict__ TH) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t j = i / d;
  if (j >= counts[0]) return;
  const int col = (int)(i - j * d);
  TH[i] = tanhf(node[(int64_t)other_map[j] * d + col]);
}

// D[j][col] = (rec[j][col] - inter[x_j - 1][bounds[r] + col]) * 200 / (m n_r)  (in place; pad columns zeroed): the gradient of
// the recon loss w.r.t. rec up to the upstream factor g (beta or *drecon), which the backward pass folds into the outputs of
// its two linear consumers.  Per-block sums of the unscaled squares for the loss itself.
__global__ __launch_bounds__(256) void adj_recon_loss_kernel(float* __restrict__ rec, int64_t nr_pad, int n_r, int col0, const float* __restrict__ inter,
                                                             int64_t n_nodes, const int64_t* __restrict__ x, const int32_t* __restrict__ other_map,
                                                             const int32_t* __restrict__ counts, float* __restrict__ slab) {
  __shared__ float red[256];
  const int m = counts[0];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float gs = m > 0 ? 200.f / ((float)m * (float)n_r) : 0.f;
  float s = 0.f;
  // one wave per row, 64 rows per workgroup
  for (int rr = wave; rr < 64; rr += 4) {
    const int64_t j = (int64_t)blockIdx.x * 64 + rr;
    if (j >= m) break;
    const int64_t node = x[other_map[j]];
    const float* trow = inter + (node - 1) * n_nodes + col0;
    float* rrow = rec + j * nr_pad;
    for (int col = lane; col < nr_pad; col += 64) {
      float dv = 0.f;
      if (col < n_r) { dv = rrow[col] - trow[col]; s += dv * dv; }
      rrow[col] = dv * gs;
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) slab[blockIdx.x] = red[0];
}

// recon_loss = 100 * sum / (m * n_r)  (mean over columns, mean over rows, * 100; Modules.py:199), 0 when m == 0 (:195)
__global__ __launch_bounds__(256) void adj_recon_final_kernel(const float* __restrict__ slab, int nslab, const int32_t* __restrict__ counts, int n_r,
                                                              float* __restrict__ out) {
  __shared__ float red[256];
  const int m = counts[0];
  const int used = (m + 63) / 64;
  float s = 0.f;
  for (int i = threadIdx.x; i < used && i < nslab; i += 256) s += slab[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = m > 0 ? 100.f * red[0] / ((float)m * (float)n_r) : 0.f;
    out[1] = (float)m;                 // losses[2]: rows of the mean (data-parallel weighting)
  }
}

// dnode[other_map[j]] += g * dTH[j] * (1 - TH[j]^2)      (g = *drecon from autograd, or beta)
__global__ __launch_bounds__(256) void adj_recon_dnode_kernel(const float* __restrict__ dTH, const float* __restrict__ TH,
                                                              const int32_t* __restrict__ other_map, const int32_t* __restrict__ counts, int d,
                                                              float* __restrict__ dnode, const float* __restrict__ drecon, float beta) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t j = i / d;
  if (j >= counts[0]) return;
  const int col = (int)(i - j * d);
  const float t = TH[i];
  dnode[(int64_t)other_map[j] * d + col] += (drecon ? drecon[0] : beta) * dTH[i] * (1.f - t * t);
}

// ---- host orchestration ----------------------------------------------------------------------------------------------
static int check_adj(const matcha_shape& s, const matcha_tensors& p, const matcha_frozen& f) {
  MATCHA_CHECK_ARG(s.n_chrom >= 1 && s.n_chrom <= kMaxChrom, "adj mode: n_chrom=%d outside 1..%d", s.n_chrom, kMaxChrom);
  MATCHA_CHECK_ARG(s.max_bins >= 1, "adj mode: max_bins must be set");
  MATCHA_CHECK_ARG(p.adj_w0 && p.adj_w1 && f.bounds && f.feats && f.feat_off, "adj mode: null tensor (adj_w0/adj_w1/bounds/feats/feat_off)");
  return MATCHA_OK;
}

static int sort_tokens(const matcha_shape& s, const matcha_frozen& f, const int64_t* x, int64_t T, int r_chrom, AdjWs& w, int32_t* touched,
                       const int32_t* t_dev, hipStream_t st, const int32_t* r_dev = nullptr, bool zero_rgrad = false) {
  const int C = s.n_chrom;
  if (T <= kSortSmallTok && !options().disable_small_batch) {
    const size_t lds = (size_t)(C + 1) * 512 * sizeof(int);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(adj_sort_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(adj_sort_small_kernel, dim3(1), dim3(512), lds, st, x, T, f.bounds, C, r_chrom, w.seg, w.counts, touched, w.order, w.other_map, t_dev,
                       r_dev, zero_rgrad ? w.rgrad : nullptr, zero_rgrad ? (int)(w.nr_pad * (s.d + 1)) : 0);
    MATCHA_CHECK_LAUNCH("adj_sort_small_kernel");
    return MATCHA_OK;
  }
  hipLaunchKernelGGL(adj_hist_kernel, dim3(w.nblk), dim3(256), 0, st, x, T, f.bounds, C, w.hist, t_dev);
  MATCHA_CHECK_LAUNCH("adj_hist_kernel");
  const int scan_ints = w.nblk * (C + 1) <= 12288 ? w.nblk * (C + 1) : 0;        // <= 48 KB of LDS for the staged histogram
  hipLaunchKernelGGL(adj_scan_kernel, dim3(1), dim3(1024), (size_t)scan_ints * sizeof(int), st, w.hist, w.nblk, C, r_chrom, w.base, w.seg, w.counts,
                     touched, scan_ints, r_dev, zero_rgrad ? w.rgrad : nullptr, zero_rgrad ? (int)(w.nr_pad * (s.d + 1)) : 0);
  MATCHA_CHECK_LAUNCH("adj_scan_kernel");
  hipLaunchKernelGGL(adj_scatter_kernel, dim3(w.nblk), dim3(256), (size_t)(C + 1) * 256 * sizeof(int), st, x, T, f.bounds, C, r_chrom, w.base,
                     w.seg, w.order, w.other_map, t_dev, r_dev);
  MATCHA_CHECK_LAUNCH("adj_scatter_kernel");
  return MATCHA_OK;
}

// fused_x0 / fused_X (embed_dim 64, adj_fused_eligible): the attribute path and next_w run in the same kernel and the encoder's input rows come
// back instead of node_out; `save` = a backward pass (adj_backward with fused = true) will follow on this workspace
int adj_forward(const matcha_shape& s, const matcha_tensors& p, const matcha_frozen& f, const matcha_step_opts& o, const int64_t* x, int64_t T,
                float* node_out, float* recon_out, void* ws, size_t ws_bytes, hipStream_t st, const int32_t* t_dev, const int32_t* slot_map,
                float* fused_x0, float* fused_X, bool save, bool fused_node) {
  MATCHA_TRY(check_adj(s, p, f));
  MATCHA_CHECK_ARG(ws && ((uintptr_t)ws) % 256 == 0, "adj_forward: workspace missing or misaligned");
  AdjWs w;
  const size_t need = adj_carve(s, T, (char*)ws, w);
  if (ws_bytes < need) { set_error("adj_forward: workspace %zu < %zu bytes", ws_bytes, need); return MATCHA_ENOMEM; }
  const int C = s.n_chrom, d = s.d;
  const bool fused_here = adj_fused_eligible(s, f) && (fused_X || (fused_node && node_out && !recon_out));
  MATCHA_CHECK_ARG(!o.random_chrom_dev || fused_here, "adj_forward: opts->random_chrom_dev needs the fused adj front end (embed_dim 64, feat_row_pad 64)");
  const int32_t* r_dev = recon_out ? o.random_chrom_dev : nullptr;
  // with a device-side chromosome the host only knows that SOME chromosome will be drawn: r = 0 stands for "the branch runs"
  const int r = r_dev ? 0 : ((recon_out && o.random_chrom >= 0 && o.random_chrom < C) ? o.random_chrom : -1);
  MATCHA_TRY(sort_tokens(s, f, x, T, r, w, nullptr, t_dev, st, r_dev, fused_here && save && recon_out && r >= 0));
  // fused_node: a node-rows-only call that no backward pass follows (matcha_node_embeddings) may take the fused kernel too
  if (adj_fused_eligible(s, f) && (fused_X || (fused_node && node_out && !recon_out))) {
    MATCHA_CHECK_ARG(!(o.training != 0 && o.p_drop_adj > 0.f) || o.seed, "adj_forward: dropout needs a seed");
    return adj_fused_forward(s, p, f, o, x, T, w, r, save, fused_X ? nullptr : node_out, fused_x0, fused_X, recon_out, st, slot_map);
  }
  MATCHA_CHECK_ARG(!fused_X, "adj_forward: the fused front end was asked for a shape it does not cover");
  const bool train = o.training != 0 && o.p_drop_adj > 0.f;
  MATCHA_CHECK_ARG(!train || o.seed, "adj_forward: dropout needs a seed");
  // layer 1: gather-GEMM + tanh -> Hs (sorted rows)
  {
    AdjEncArgs a;
    a.x = x; a.order = w.order; a.seg = w.seg; a.bounds = f.bounds; a.feat_off = f.feat_off; a.feats = f.feats; a.w0 = p.adj_w0;
    a.Hs = w.Hs; a.T = T; a.C = C; a.d = d; a.feat_pad = f.feat_row_pad; a.seed = o.seed; a.p_drop = train ? o.p_drop_adj : 0.f; a.slot_map = slot_map;
    const int nt = (int)cdiv(d, 32);
    a.splits = (int)cdiv(T, (int64_t)C * 128);
    if (a.splits < 1) a.splits = 1;
    dim3 grid((unsigned)(C * a.splits));
    ProfScope ps(MATCHA_PROF_ADJ_ENCODE, 0.0, st);
    if (nt <= 1) hipLaunchKernelGGL((adj_encode_fwd_kernel<1>), grid, dim3(256), (size_t)(128 + 32) * kLdA * 4, st, a);
    else if (nt == 2) hipLaunchKernelGGL((adj_encode_fwd_kernel<2>), grid, dim3(256), (size_t)(128 + 64) * kLdA * 4, st, a);
    else if (nt <= 4) hipLaunchKernelGGL((adj_encode_fwd_kernel<4>), grid, dim3(256), (size_t)(128 + 128) * kLdA * 4, st, a);
    else hipLaunchKernelGGL((adj_encode_fwd_kernel<8>), grid, dim3(256), (size_t)(128 + 256) * kLdA * 4, st, a);
    MATCHA_CHECK_LAUNCH("adj_encode_fwd_kernel");
  }
  // layer 2: node[order[p]] = Hs[p] . W1_c^T ; padding slots stay 0 (Modules.py:178)
  MATCHA_TRY(zero_async(node_out, (size_t)T * d * sizeof(float), st));
  {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A[0] = w.Hs; g.B[0] = p.adj_w1; g.C[0] = node_out; g.batch = 1;
    g.M = T; g.N = d; g.K = d; g.lda = d; g.ldb = d; g.ldc = d; g.aux_scale = 1.f;
    g.c_row_map = w.order; g.seg = w.seg; g.n_groups = C; g.b_group_stride = (int64_t)d * d;
    g.m_dev = w.seg + C;
    MATCHA_TRY(launch_gemm_rm(false, g, st));
  }
  if (!recon_out) return MATCHA_OK;
  if (r < 0) {
    MATCHA_TRY(zero_async(recon_out, 2 * sizeof(float), st));
    return MATCHA_OK;
  }
  // recon branch (Modules.py:192-199)
  MATCHA_CHECK_ARG(p.recon_w && p.recon_b && f.inter && f.bounds_host, "adj_forward: recon tensors / bounds_host missing");
  const int lo_r = f.bounds_host[r], n_r = f.bounds_host[r + 1] - f.bounds_host[r];
  hipLaunchKernelGGL(adj_tanh_gather_kernel, dim3((unsigned)cdiv(T * d, 256)), dim3(256), 0, st, node_out, w.other_map, w.counts, d, w.TH);
  MATCHA_CHECK_LAUNCH("adj_tanh_gather_kernel");
  {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A[0] = w.TH; g.B[0] = p.recon_w + (int64_t)d * lo_r; g.C[0] = w.rec; g.batch = 1;
    g.M = T; g.N = n_r; g.K = d; g.lda = d; g.ldb = d; g.ldc = w.nr_pad; g.aux_scale = 1.f;
    g.flags = MATCHA_EPI_BIAS; g.bias[0] = p.recon_b + lo_r; g.m_dev = w.counts;
    MATCHA_TRY(launch_gemm_rm(false, g, st));
  }
  const int nslab = (int)cdiv(T, 64);
  hipLaunchKernelGGL(adj_recon_loss_kernel, dim3(nslab), dim3(256), 0, st, w.rec, w.nr_pad, n_r, lo_r, f.inter, (int64_t)s.n_nodes, x, w.other_map,
                     w.counts, w.lossslab);
  MATCHA_CHECK_LAUNCH("adj_recon_loss_kernel");
  hipLaunchKernelGGL(adj_recon_final_kernel, dim3(1), dim3(256), 0, st, w.lossslab, nslab, w.counts, n_r, recon_out);
  MATCHA_CHECK_LAUNCH("adj_recon_final_kernel");
  return MATCHA_OK;
}

int adj_backward(const matcha_shape& s, const matcha_tensors& p, const matcha_frozen& f, const matcha_step_opts& o, const int64_t* x, int64_t T,
                 float* dnode, const float* drecon, matcha_tensors& g_, int32_t* touched, void* ws, size_t ws_bytes, void* gemm_ws,
                 size_t gemm_ws_bytes, hipStream_t st, const int32_t* slot_map, bool fused) {
  MATCHA_TRY(check_adj(s, p, f));
  AdjWs w;
  const size_t need = adj_carve(s, T, (char*)ws, w);
  if (ws_bytes < need) { set_error("adj_backward: workspace %zu < %zu bytes", ws_bytes, need); return MATCHA_ENOMEM; }
  MATCHA_CHECK_ARG(g_.adj_w0 && g_.adj_w1, "adj_backward: gradient buffers missing");
  const int C = s.n_chrom, d = s.d;
  MATCHA_CHECK_ARG(!o.random_chrom_dev || fused, "adj_backward: opts->random_chrom_dev needs the fused adj front end");
  const int r = o.random_chrom_dev ? 0 : ((o.random_chrom >= 0 && o.random_chrom < C) ? o.random_chrom : -1);
  const bool train = o.training != 0 && o.p_drop_adj > 0.f;
  // order/seg/other_map of the forward are still in the workspace; only the `touched` flags are (re)written -- by a block of the fused
  // backward kernel, or by a kernel of their own on the layer-wise path
  if (fused) return adj_fused_backward(s, p, f, o, x, T, w, r, dnode, drecon, g_, st, slot_map, touched);
  if (touched) {
    hipLaunchKernelGGL(adj_flags_kernel, dim3(1), dim3(64), 0, st, w.seg, w.counts, C, r, touched, o.random_chrom_dev);
    MATCHA_CHECK_LAUNCH("adj_flags_kernel");
  }
  // ---- recon branch: d loss / d rec = g * 200/(m n_r) * (rec - target) ----
  if (r >= 0 && (drecon || o.beta != 0.f)) {
    MATCHA_CHECK_ARG(g_.recon_w && g_.recon_b && f.bounds_host, "adj_backward: recon gradient buffers missing");
    const int lo_r = f.bounds_host[r], n_r = f.bounds_host[r + 1] - f.bounds_host[r];
    // w.rec holds D = (rec - target) * 200/(m n_r) since the forward pass; the upstream factor g scales the outputs
    // dWr[n_r, d] += g D^T TH ; dbr += g colsum(D)
    MATCHA_TRY(launch_gemm_tn(w.rec, w.TH, g_.recon_w + (int64_t)d * lo_r, g_.recon_b + lo_r, n_r, d, T, w.nr_pad, d, nullptr, true, gemm_ws,
                              gemm_ws_bytes, st, w.counts, drecon, drecon ? 1.f : o.beta));
    // dTH = D . Wr   (K = n_r)
    {
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      g.A[0] = w.rec; g.B[0] = p.recon_w + (int64_t)d * lo_r; g.C[0] = w.dTH; g.batch = 1;
      g.M = T; g.N = d; g.K = n_r; g.lda = w.nr_pad; g.ldb = d; g.ldc = d; g.aux_scale = 1.f; g.m_dev = w.counts;
      MATCHA_TRY(launch_gemm_rm(true, g, st));
    }
    hipLaunchKernelGGL(adj_recon_dnode_kernel, dim3((unsigned)cdiv(T * d, 256)), dim3(256), 0, st, w.dTH, w.TH, w.other_map, w.counts, d, dnode, drecon, o.beta);
    MATCHA_CHECK_LAUNCH("adj_recon_dnode_kernel");
  }
  // ---- encoder ----
  // windows of sorted rows (whole 128-row steps): ~320 windows of up to 1024 rows at large batches (every window ends in 64 x 64 float atomics per
  // column tile onto a small gradient array: more, smaller windows lose to their contention), one step per window at small ones
  int steps_per_win = (int)(T / (128 * 320));
  steps_per_win = steps_per_win < 1 ? 1 : (steps_per_win > 8 ? 8 : steps_per_win);
  const int rpb = 128 * steps_per_win;
  const int64_t nwin = cdiv(T, (int64_t)rpb);
  const int cpar = nwin * C <= 2048 ? 1 : 0;         // few windows: one workgroup per (window, chromosome)
  const unsigned zblocks = (unsigned)(cpar ? nwin * C : nwin);
  {
    AdjTnArgs a;
    memset(&a, 0, sizeof(a));
    a.A = dnode; a.Bd = w.Hs; a.x = x; a.order = w.order; a.seg = w.seg; a.bounds = f.bounds; a.feat_off = f.feat_off; a.feats = f.feats;
    a.out = g_.adj_w1; a.C = C; a.d = d; a.feat_pad = f.feat_row_pad; a.rows_per_block = rpb; a.chrom_parallel = cpar; a.seed = o.seed; a.p_drop = 0.f;
    auto k0 = adj_tn_kernel<0>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAdjTnLds) != hipSuccess) {
      set_error("adj_tn_kernel: cannot raise the dynamic LDS limit"); return MATCHA_EHIP;
    }
    hipLaunchKernelGGL(k0, dim3((unsigned)cdiv(d, 64), (unsigned)cdiv(d, 64), zblocks), dim3(256), kAdjTnLds, st, a);
    MATCHA_CHECK_LAUNCH("adj_tn_kernel<0>");
  }
  {   // dZ[p] = (dnode[order[p]] . W1_c) * (1 - Hs[p]^2)
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A[0] = dnode; g.B[0] = p.adj_w1; g.C[0] = w.dZ; g.batch = 1;
    g.M = T; g.N = d; g.K = d; g.lda = d; g.ldb = d; g.ldc = d; g.aux_scale = 1.f;
    g.flags = MATCHA_EPI_DTANH; g.aux = w.Hs;
    g.a_row_map = w.order; g.seg = w.seg; g.n_groups = C; g.b_group_stride = (int64_t)d * d;
    g.m_dev = w.seg + C;     // sorted rows beyond the non-padding tokens do not exist (order[] is only filled up to the token count)
    MATCHA_TRY(launch_gemm_rm(true, g, st));
  }
  {
    AdjTnArgs a;
    memset(&a, 0, sizeof(a));
    a.A = w.dZ; a.x = x; a.order = w.order; a.seg = w.seg; a.bounds = f.bounds; a.feat_off = f.feat_off; a.feats = f.feats;
    a.out = g_.adj_w0; a.C = C; a.d = d; a.feat_pad = f.feat_row_pad; a.rows_per_block = rpb; a.chrom_parallel = cpar; a.seed = o.seed; a.p_drop = train ? o.p_drop_adj : 0.f; a.slot_map = slot_map;
    auto k1 = adj_tn_kernel<1>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAdjTnLds) != hipSuccess) {
      set_error("adj_tn_kernel: cannot raise the dynamic LDS limit"); return MATCHA_EHIP;
    }
    hipLaunchKernelGGL(k1, dim3((unsigned)cdiv(d, 64), (unsigned)cdiv(s.max_bins, 64), zblocks), dim3(256), kAdjTnLds, st, a);
    MATCHA_CHECK_LAUNCH("adj_tn_kernel<1>");
  }
  return MATCHA_OK;
}

}  // namespace matcha
