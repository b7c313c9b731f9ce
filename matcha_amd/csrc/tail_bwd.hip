// Backward of pff_n1's two convolutions at embed_dim 64 (Modules.py:353-376) for LARGE batches, as a kernel of its own behind the fused
// forward (round 5).
//
// Inside fused_fwd32_kernel the tail's backward is one wavefront's text per half tile (fused_fwd32_tail.hpp).  Its LayerNorm part works on
// values the wavefront holds anyway; its convolution part -- dZ1 = (dH2 conv1) x mask x tanh', d dyn = dZ1 conv0 + dH2, dW1 += dH2^T H1,
// dW0 += dZ1^T Y -- streams two more weight matrices from L2 per half tile, runs the two weight gradients as f32 MFMA column walks and writes
// 32 KB of them per half tile (259 MB per 65 536-row step, summed by a 60 us reduction).  For large batches the forward kernel stops behind the
// LayerNorm backward and leaves dH2 rows next to the Y and H1 rows it parks anyway; this kernel does the rest the way fused_bwd.hip runs the
// attention block's backward: persistent workgroups of four wavefronts (two per CU) walk the half tiles, conv1^T / conv0^T stay in registers as
// bf16-plane fragments, the two weight gradients accumulate in MFMA accumulators for the whole walk (ONE slab per workgroup), every GEMM
// operand lives in LDS as bf16 planes (bf16x3.hpp):
//
//   stage     dH2, H1, Y rows -> planes (+ dH2 and H1 in f32: the residual of d dyn, the tanh values)
//   GEMM      dZ1^T = conv1^T dH2^T, x dropout mask x tanh'            (row fragments)
//   GEMM      d dyn^T = conv0^T dZ1^T + dH2^T, x dropout mask          (row fragments) -> ddyn0 (out)
//   TN        dW1 += dH2^T H1,  dW0 += dZ1^T Y                          (column fragments: ds_read_b64_tr_b16)
//
// Measured at 65 536 rows: the forward kernel 0.439 -> 0.346 ms, this kernel 0.088, the slab reductions 0.066 -> 0.035: 0.036 ms off the step.
// In-situ ablations of this kernel (94 us on that box): without the operand splits of the staging -23, without the weight-gradient products
// -20, without the dropout hashes -11, without the d dyn stores -4; all four off 39.  (A variant that also ran the LayerNorm backward here -- 7
// more vector accumulators -- spilled 71 registers and took 236 us.)
// The small-batch step keeps the whole tail in the forward kernel (fused_fwd32h_kernel): there the forward is one wave of workgroups anyway.
#include "bf16x3.hpp"
#include "kernels.hpp"
#include "loss_reduce.hpp"

namespace matcha {

namespace {

constexpr int kLd = 68;
constexpr int kTileF = 32 * kLd;
constexpr int kPS = 80;                 // 40-dword plane rows: conflict-free row and transposed reads (fused_bwd.hip)
constexpr int kPlane = 32 * kPS;
constexpr int kPT = 3 * kPlane;
constexpr float kEps = 1e-5f;
constexpr int kSlab = 2 * 4096 + 10 * 64;          // dW1 | dW0 (row-major [out][in]) | gp bp g1 b1 g2 b2 wc pff1_b pff0_b bc  (tail_slab_reduce's vector order)
constexpr int kVec = 2 * 4096;
typedef Planes<kPS, 16> PL;

struct TailBwdArgs {
  const float* dH2; const float* Y; const float* H1;  // [T][64] rows left by the forward kernel
  const int32_t* count; const int32_t* half_meta; const int32_t* tok_slot;
  int nhalves;
  const float* W0; const float* W1;                 // pff_n1 conv0 / conv1 weights [64 out][64 in]
  const uint64_t* seed; float p_fc1, p_pff;
  float* ddyn0; float* slab;                        // slab [gridDim.x][kSlab]: dW1 | dW0 | the ten vector slots
  float* zero_rows;                                 // [T][64] or null: the buffer the attention block's backward adds its d x_hat into with float atomics --
                                                    // zeroed here row by row (a 13 us launch of its own otherwise)
  const float* row_loss; int64_t B; float* losses; int zero_recon;     // losses != null: ONE extra block (the last) reduces the forward's per-hyperedge losses
                                                    // to their mean (loss_reduce.hpp) -- a 7 us launch of its own otherwise
  const float* vslab;                               // the forward kernel's per-half-tile slabs: their vector slots 0-6 and 9 (LayerNorm / classifier
                                                    // gradients) are summed along the walk into this workgroup's slab -- ONE reduction over 512 slabs behind it
};
constexpr size_t kLdsBytes = (size_t)2 * kTileF * 4 + (size_t)4 * kPT * 2;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void tail_bwd64_kernel(TailBwdArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Hs = lds;                                   // H1 f32 (the tanh values, for tanh')
  float* Ds = lds + kTileF;                          // dH2 f32 (the residual of d dyn)
  short* Dp = reinterpret_cast<short*>(lds + 2 * kTileF);   // dH2 planes
  short* Hp = Dp + kPT;                              // H1 planes
  short* Yp = Hp + kPT;                              // Y planes
  short* Zp = Yp + kPT;                              // dZ1 planes

  const int tid = threadIdx.x;
  const int nwg = (int)gridDim.x - (g.losses ? 1 : 0);
  if ((int)blockIdx.x == nwg) {
    loss_reduce_role<256>(g.row_loss, g.B, g.losses, g.zero_recon, lds);
    return;
  }
  const int tr = g.count[1];
  int nh = g.count[3];
  if (nh > g.nhalves) nh = g.nhalves;
  const int per = (nh + nwg - 1) / nwg;
  const int tile_lo = blockIdx.x * per;
  const int tile_hi = tile_lo + per < nh ? tile_lo + per : nh;

  if (blockIdx.x == 0 && tid < 16)                  // the shared padding token receives no gradient from the tail (its rows are masked)
    *reinterpret_cast<f32x4*>(g.ddyn0 + (int64_t)tr * 64 + 4 * tid) = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool drop1 = g.p_fc1 > 0.f, drop2 = g.p_pff > 0.f;
  uint32_t key1 = 0, key2 = 0, thr1 = 0, thr2 = 0;
  float ks1 = 1.f, ks2 = 1.f;
  if (drop1 || drop2) {
    const uint64_t seed = *g.seed;
    key1 = rng_key(seed, kStreamDropFc1); key2 = rng_key(seed, kStreamDropPff);
    if (drop1) { thr1 = dropout_threshold(g.p_fc1); ks1 = 1.f / (1.f - g.p_fc1); }
    if (drop2) { thr2 = dropout_threshold(g.p_pff); ks2 = 1.f / (1.f - g.p_pff); }
  }
  const float unscale = drop2 ? 1.f - g.p_pff : 1.f;

  // accumulators of the whole walk
  f32x4 aw1[4], aw0[4];                              // dW1 / dW0: rows 16 i + 4 kq + reg, column fb + c16
#pragma unroll
  for (int i = 0; i < 4; ++i) { aw1[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; aw0[i] = aw1[i]; }
  float va0 = 0.f, va1 = 0.f, va2 = 0.f;            // vector slots of the forward's slabs: elements tid, tid + 256, and d bc (slot 9) in thread 64
  V8 a_c1 = zero8();                                 // column sums of dH2 (conv1's bias gradient), this thread's 8 features of its staging row
  f32x4 a_c0 = {0.f, 0.f, 0.f, 0.f};                 // column sums of dZ1 in the GEMM's output layout (features fb + 4 kq + {0..3})

  // conv1^T and conv0^T as A operands: lane (c16, kq), step s holds W[32 s + 8 kq + {0..7}][fb + c16]
  Frag3 W1f[2], W0f[2];
  {
    const int lane = tid & 63, wave = tid >> 6, c16 = lane & 15, kq = lane >> 4, fb = 16 * wave;
    const float* p1 = g.W1 + (8 * kq) * 64 + fb + c16;
    const float* p0 = g.W0 + (8 * kq) * 64 + fb + c16;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float v1[8], v0[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { v1[j] = p1[(32 * s + j) * 64]; v0[j] = p0[(32 * s + j) * 64]; }
      W1f[s] = split8(v1); W0f[s] = split8(v0);
    }
  }

  const int4* meta = reinterpret_cast<const int4*>(g.half_meta);
  const int4 mzero = make_int4(0, 0, 0, 0);
  int4 mc = tile_lo < tile_hi ? meta[tile_lo] : mzero;
  int4 mn = tile_lo + 1 < tile_hi ? meta[tile_lo + 1] : mzero;
  V8 yn, hn, dn;
  uint32_t sn0 = 0, sn1 = 0;
#define TB_GLOAD(M)                                                                                      \
  do {                                                                                                   \
    const int row__ = tid >> 3, sub__ = tid & 7;                                                         \
    const int64_t tok__ = (M).x + (row__ < (M).y ? row__ : ((M).y > 0 ? (M).y - 1 : 0));                  \
    const int64_t o__ = tok__ * 64 + 8 * sub__;                                                          \
    yn = ld8(g.Y + o__); hn = ld8(g.H1 + o__); dn = ld8(g.dH2 + o__);                                    \
    if (drop1 || drop2) {     /* the dropout counters of the two tokens this lane finishes in the GEMM epilogues */ \
      const int c16__ = tid & 15;                                                                        \
      sn0 = (uint32_t)g.tok_slot[(M).x + (c16__ < (M).y ? c16__ : 0)];                                   \
      sn1 = (uint32_t)g.tok_slot[(M).x + (16 + c16__ < (M).y ? 16 + c16__ : 0)];                         \
    }                                                                                                    \
  } while (0)
  TB_GLOAD(mc);
  V8 yq = yn, hq = hn, dq = dn;                      // two tiles of rows in flight: HBM latency under load exceeds one tile's time
  uint32_t sq0 = sn0, sq1 = sn1;
  {
    const V8 ty = yn, th = hn, td = dn; const uint32_t t0_ = sn0, t1_ = sn1;
    TB_GLOAD(mn);                                     // tile_lo + 1
    yq = yn; hq = hn; dq = dn; sq0 = sn0; sq1 = sn1;
    yn = ty; hn = th; dn = td; sn0 = t0_; sn1 = t1_;
  }

  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int4 mnn = tile + 2 < tile_hi ? meta[tile + 2] : mzero;
    const int t0 = mc.x, n_real = mc.y;
    const float* vs = g.vslab + (int64_t)tile * kSlab + kVec;          // (added at the end of the trip: in flight during the tile's work)
    const float vl0 = vs[tid], vl1 = vs[tid + 256], vl2 = tid == 64 ? vs[9 * 64] : 0.f;      // slots 0-3, 4-7, the scalar of slot 9
    int tid_ = tid;
    asm volatile("" : "+v"(tid_));
    const int lane = tid_ & 63, wave = tid_ >> 6;
    const int c16 = lane & 15, kq = lane >> 4, fb = 16 * wave;
    const int row = tid_ >> 3, sub = tid_ & 7;
    // ---- stage: this thread's 8 features of its row -> planes (rows past the tokens: zero gradient rows) ----
    {
      const V8 dh2 = scale8(row < n_real ? 1.f : 0.f, dn);
      if (g.zero_rows && row < n_real) st8(g.zero_rows + ((int64_t)t0 + row) * 64 + 8 * sub, zero8());
      add8(a_c1, dh2);
      st8(&Ds[row * kLd + 8 * sub], dh2);
      st8(&Hs[row * kLd + 8 * sub], hn);
      PL::store(Dp + row * kPS + 8 * sub, split8(dh2));
      PL::store(Hp + row * kPS + 8 * sub, split8(hn));
      PL::store(Yp + row * kPS + 8 * sub, split8(yn));
    }
    const uint32_t slot0 = sn0, slot1 = sn1;
    __syncthreads();
    // (yn ..) <- the next tile's rows (in flight since the previous trip); their registers take the loads of the tile after it
    yn = yq; hn = hq; dn = dq; sn0 = sq0; sn1 = sq1;
    {
      const V8 ty = yn, th = hn, td = dn; const uint32_t t0_ = sn0, t1_ = sn1;
      TB_GLOAD(mnn);
      yq = yn; hq = hn; dq = dn; sq0 = sn0; sq1 = sn1;
      yn = ty; hn = th; dn = td; sn0 = t0_; sn1 = t1_;
    }
    // ---- dZ1^T = conv1^T dH2^T, x dropout mask x tanh': lane (c16, kq) ends with token c16 (+ 16), features fb + 4 kq + {0..3} ----
    {
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      const short* dp = Dp + c16 * kPS + 8 * kq;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const Frag3 b0 = PL::row(dp + 32 * s), b1 = PL::row(dp + 16 * kPS + 32 * s);
        acc0 = mma6(acc0, W1f[s], b0); acc1 = mma6(acc1, W1f[s], b1);
      }
      const f32x4 hv0 = *reinterpret_cast<const f32x4*>(&Hs[c16 * kLd + fb + 4 * kq]);
      const f32x4 hv1 = *reinterpret_cast<const f32x4*>(&Hs[(16 + c16) * kLd + fb + 4 * kq]);
      const uint32_t hr0 = lowbias32(slot0 ^ key2), hr1 = lowbias32(slot1 ^ key2);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t f = (uint32_t)(fb + 4 * kq + e);
        float v0 = acc0[e], v1 = acc1[e];
        if (drop2) {
          v0 = lowbias32(f ^ hr0) >= thr2 ? v0 * ks2 : 0.f;
          v1 = lowbias32(f ^ hr1) >= thr2 ? v1 * ks2 : 0.f;
        }
        const float t0_ = hv0[e] * unscale, t1_ = hv1[e] * unscale;       // the tanh values (0 where dropped)
        acc0[e] = v0 * (1.f - t0_ * t0_); acc1[e] = v1 * (1.f - t1_ * t1_);
      }
      a_c0 += acc0 + acc1;                                                 // conv0's bias gradient (rows past the tokens are zero: dH2 = 0)
      {
        const P3 p0 = split2(acc0[0], acc0[1]), p1 = split2(acc0[2], acc0[3]);
        short* d = Zp + c16 * kPS + fb + 4 * kq;
        *reinterpret_cast<u32x2*>(d) = (u32x2){p0.h, p1.h}; *reinterpret_cast<u32x2*>(d + kPlane) = (u32x2){p0.m, p1.m};
        *reinterpret_cast<u32x2*>(d + 2 * kPlane) = (u32x2){p0.l, p1.l};
      }
      {
        const P3 p0 = split2(acc1[0], acc1[1]), p1 = split2(acc1[2], acc1[3]);
        short* d = Zp + (16 + c16) * kPS + fb + 4 * kq;
        *reinterpret_cast<u32x2*>(d) = (u32x2){p0.h, p1.h}; *reinterpret_cast<u32x2*>(d + kPlane) = (u32x2){p0.m, p1.m};
        *reinterpret_cast<u32x2*>(d + 2 * kPlane) = (u32x2){p0.l, p1.l};
      }
    }
    __syncthreads();
    // ---- d dyn^T = conv0^T dZ1^T + dH2^T (H2 = conv1(H1) + Y), x dropout mask x row mask ----
    {
      f32x4 acc0 = *reinterpret_cast<const f32x4*>(&Ds[c16 * kLd + fb + 4 * kq]);
      f32x4 acc1 = *reinterpret_cast<const f32x4*>(&Ds[(16 + c16) * kLd + fb + 4 * kq]);
      const short* zp = Zp + c16 * kPS + 8 * kq;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const Frag3 b0 = PL::row(zp + 32 * s), b1 = PL::row(zp + 16 * kPS + 32 * s);
        acc0 = mma6(acc0, W0f[s], b0); acc1 = mma6(acc1, W0f[s], b1);
      }
      if (drop1) {
        const uint32_t hr0 = lowbias32(slot0 ^ key1), hr1 = lowbias32(slot1 ^ key1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t f = (uint32_t)(fb + 4 * kq + e);
          acc0[e] = lowbias32(f ^ hr0) >= thr1 ? acc0[e] * ks1 : 0.f;
          acc1[e] = lowbias32(f ^ hr1) >= thr1 ? acc1[e] * ks1 : 0.f;
        }
      }
      float* out = g.ddyn0 + ((int64_t)t0 + c16) * 64 + fb + 4 * kq;
      if (c16 < n_real) *reinterpret_cast<f32x4*>(out) = acc0;
      if (16 + c16 < n_real) *reinterpret_cast<f32x4*>(out + 16 * 64) = acc1;
    }
    // ---- weight gradients: dW1[n][k] += sum_t dH2[t][n] H1[t][k];  dW0[n][k] += sum_t dZ1[t][n] Y[t][k]  (one 32-token step) ----
    {
      const int blk = ((4 * kq + ((lane & 15) >> 2)) * kPS) + 4 * (lane & 3);
      const Frag3 hb = PL::col(Hp + blk + fb), yb = PL::col(Yp + blk + fb);
#pragma unroll
      for (int i = 0; i < 4; ++i) aw1[i] = mma6(aw1[i], PL::col(Dp + blk + 16 * i), hb);
#pragma unroll
      for (int i = 0; i < 4; ++i) aw0[i] = mma6(aw0[i], PL::col(Zp + blk + 16 * i), yb);
    }
    __syncthreads();                                  // the GEMMs are done with every tile
    va0 += vl0; va1 += vl1; va2 += vl2;
    mc = mn; mn = mnn;
  }
#undef TB_GLOAD

  // ---- workgroup slab ----
  const int lane = tid & 63, wave = tid >> 6, c16 = lane & 15, kq = lane >> 4, fb = 16 * wave;
  float* slab = g.slab + (int64_t)blockIdx.x * kSlab;
  {
    const int col = fb + c16;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int rw = 16 * i + 4 * kq + reg;
        slab[rw * 64 + col] = aw1[i][reg];
        slab[4096 + rw * 64 + col] = aw0[i][reg];
      }
  }
  // conv1's bias gradient: the 8 lanes with equal `sub` of a wave (fixed xor tree), then the 4 waves in order
  float* red = lds;                     // [4 waves][64]
  {
    const int sub = tid & 7;
    const float t[8] = {a_c1.a.x, a_c1.a.y, a_c1.b.x, a_c1.b.y, a_c1.c.x, a_c1.c.y, a_c1.d.x, a_c1.d.y};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = t[e];
      x += __shfl_xor(x, 8, 64); x += __shfl_xor(x, 16, 64); x += __shfl_xor(x, 32, 64);
      if (lane < 8) red[wave * 64 + 8 * sub + e] = x;
    }
  }
  // conv0's bias gradient: features fb + 4 kq + e, summed over the 16 token lanes c16
  {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = a_c0[e];
      x += __shfl_xor(x, 1, 64); x += __shfl_xor(x, 2, 64); x += __shfl_xor(x, 4, 64); x += __shfl_xor(x, 8, 64);
      if (c16 == 0) slab[kVec + 8 * 64 + fb + 4 * kq + e] = x;
    }
  }
  // the forward's vector slots (element e of [10][64]; slots 7 and 8 are this kernel's own, written around this)
  {
    slab[kVec + tid] = va0;                                              // slots 0-3
    if (tid < 192) slab[kVec + 256 + tid] = va1;                          // slots 4-6 (7 is this kernel's)
    if (tid == 64) slab[kVec + 9 * 64] = va2;                             // d bc
  }
  __syncthreads();
  if (tid < 64) slab[kVec + 7 * 64 + tid] = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
}

}  // namespace

int tail_bwd_grid() { return 2 * device_cu_count(); }
size_t tail_bwd_slab_floats() { return (size_t)tail_bwd_grid() * kSlab; }

// After a training forward that ran fused_fwd32_kernel with `tail_split` set (it stopped behind the LayerNorm backward and left dH2 rows):
// ddyn0 and one slab of parameter-gradient partials per workgroup (tail_bwd_slab_floats() floats at `slab`: the convolutions' gradients + the
// forward's per-half-tile vector slots of `vslab` summed along the walk; launch_tail_reduce(..., n_slabs = tail_bwd_grid(), rowmajor) sums them)
int launch_tail_bwd64(const matcha_tensors& p, const float* dH2, const float* Y, const float* H1, const Ragged& rg, const uint64_t* seed, float p_fc1,
                      float p_pff, float* ddyn0, float* slab, const float* vslab, float* zero_rows, hipStream_t st, const float* row_loss, int64_t B,
                      float* losses, bool zero_recon) {
  TailBwdArgs g;
  g.row_loss = row_loss; g.B = B; g.losses = losses; g.zero_recon = zero_recon ? 1 : 0;
  g.vslab = vslab; g.zero_rows = zero_rows;
  g.dH2 = dH2; g.Y = Y; g.H1 = H1; g.count = rg.count; g.half_meta = rg.half_meta; g.tok_slot = rg.tok_slot; g.nhalves = rg.nhalves;
  g.W0 = p.pff0_w; g.W1 = p.pff1_w; g.seed = seed; g.p_fc1 = p_fc1; g.p_pff = p_pff;
  g.ddyn0 = ddyn0; g.slab = slab;
  // timed with the forward kernel's class (bench.py): the tail's backward was inside that kernel before, its flops were never counted as work
  ProfScope ps(MATCHA_PROF_FUSED_FWD, 0.0, st);
  auto kfn = tail_bwd64_kernel;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(kfn, dim3(tail_bwd_grid() + (losses ? 1 : 0)), dim3(256), kLdsBytes, st, g);
  MATCHA_CHECK_LAUNCH("tail_bwd64_kernel");
  return MATCHA_OK;
}

}  // namespace matcha
