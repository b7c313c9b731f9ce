// The tail of the fused embed_dim-64 forward -- pff_n1 (Modules.py:353-376), the LayerNorms and the classifier (:290-311), the weighted BCE
// (main.py:56) and, in a training step, the backward of all of that -- as ONE WAVEFRONT computes it for its half tile, everything in
// registers (layout FL).  Included verbatim into the body of fused_fwd32_kernel (one wavefront per workgroup) and of
// fused_fwd32h_kernel (eight wavefronts per half tile, one per head; wavefront 0 runs the tail alone).  ONE wavefront executes this text in
// both kernels, so no workgroup barrier appears in it: F32_TAIL_SYNC is a wave-local ordering point (s_waitcnt lgkmcnt(0) + a compiler
// memory barrier; a wavefront's LDS operations execute in order).  The including kernel defines it, the locals the tail works on
// (g, lane, r, h, n, n_h, b0, t0, real, xh, rx, dyn, TK, TV, outs, douts, krow, wp, W_, F32_BIAS, F32_XH_RELOAD) and
// where Y, H1 and hh are PARKED between the forward and the backward half of the tail -- F32_PARK_Y / _H1 / _HH (store this lane's row) and
// F32_UNPARK_Y / _H1 / _HH (an FL loaded back through a laundered pointer, so that the values do not stay in registers): the
// single-wave kernel parks them in the workspace rows the layer-wise backward reads (Y, H1: L2-resident, 2 x 256 B per token), the
// eight-wave kernel in two of the heads' dead LDS tiles.  Round 5: with Y and H1 in registers across the LayerNorm backward both kernels
// spilled ~75 VGPRs (260 B of scratch per lane); parked, and with H2 / U replaced by their normalised forms as soon as those exist, the
// tail's live set is x_hat, hh, uh, df + two temporaries.
  // ---- what only the tail needs is set up here, not in front of the head loop (registers held across it were spilled) ----
  // dropout: keep <=> lowbias32(col ^ lowbias32(slot ^ key)) >= threshold (threshold 0 = keep everything: no branches below)
  uint32_t thr1 = 0, thr2 = 0, hrow1 = 0, hrow2 = 0;
  float ks1 = 1.f, ks2 = 1.f;
  const bool drop1 = g.p_fc1 > 0.f, drop2 = g.p_pff > 0.f;
  if (drop1 || drop2) {
    const uint32_t slot = (uint32_t)g.tok_slot[F32_TOK()];
    const uint64_t seed = *g.seed;
    hrow1 = lowbias32(slot ^ rng_key(seed, kStreamDropFc1));
    hrow2 = lowbias32(slot ^ rng_key(seed, kStreamDropPff));
    if (drop1) { thr1 = dropout_threshold(g.p_fc1); ks1 = 1.f / (1.f - g.p_fc1); }
    if (drop2) { thr2 = dropout_threshold(g.p_pff); ks2 = 1.f / (1.f - g.p_pff); }
  }
  // the tail's seven parameter vectors -> TV [7][64]: gp bp g1 b1 g2 b2 wc  (the biases of fc1 / conv0 / conv1 come with the weight stream)
  F32_TAIL_SYNC();                                    // the last head's P V reads of TV are done
  for (int i4 = lane; i4 < 112; i4 += 64) {
    const int v = i4 >> 4;
    const float* src = (v == 0 ? g.hp.gp : v == 1 ? g.hp.bp : v == 2 ? g.hp.g1 : v == 3 ? g.hp.b1 : v == 4 ? g.hp.g2 : v == 5 ? g.hp.b2 : g.hp.wc) + 4 * (i4 & 15);
    *reinterpret_cast<f32x4*>(TV + 4 * i4) = *reinterpret_cast<const f32x4*>(src);
  }
  float* T1 = TK;
  float* T2 = TV;
  float* myrow = krow;
  const float* tpar = T2 + 4 * h;                     // this lane's feature offset inside a 64-float vector
  uint32_t keep1 = 0, keep2 = 0;
  FL y;
  {
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      const int f = 32 * (e >> 4) + 8 * ((e >> 2) & 3) + (e & 3);     // + 4 h
      float v = e < 16 ? dyn.lo[e] : dyn.hi[e - 16];
      const bool kp = lowbias32((uint32_t)(f + 4 * h) ^ hrow1) >= thr1;
      keep1 |= kp ? (1u << e) : 0u;
      v = (kp && real) ? v * ks1 : 0.f;               // the padding token's row is masked (Modules.py:614)
      if (e < 16) y.lo[e] = v; else y.hi[e - 16] = v;
    }
    F32_PARK_Y(y);
  }
  FL h1 = F32_BIAS(kBiasConv0);
  W32_CHAIN(h1, y, true);                             // conv0 (+ bias)
  {
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      const int f = 32 * (e >> 4) + 8 * ((e >> 2) & 3) + (e & 3);
      float v = fast_tanh(e < 16 ? h1.lo[e] : h1.hi[e - 16]);
      const bool kp = lowbias32((uint32_t)(f + 4 * h) ^ hrow2) >= thr2;
      keep2 |= kp ? (1u << e) : 0u;
      v = kp ? v * ks2 : 0.f;
      if (e < 16) h1.lo[e] = v; else h1.hi[e - 16] = v;
    }
    F32_PARK_H1(h1);
  }
  FL h2 = y;                                          // residual (+ bias) as the accumulator's initial value
  {
    const FL b1v = F32_BIAS(kBiasConv1);
#pragma unroll
    for (int e = 0; e < 16; ++e) { h2.lo[e] += b1v.lo[e]; h2.hi[e] += b1v.hi[e]; }
  }
  W32_CHAIN(h2, h1, false);                           // conv1; the window is primed again before the backward GEMMs
  if (g.H2 && !g.ddyn0 && r <= n) fl_store_global(g.H2 + F32_ROW(), h2);      // (with the tail's backward in this kernel the H2 rows park hh instead)
  FF_T(7);
  F32_TAIL_SYNC();                                    // the parameter vectors in T2 are visible
  // ---- out_t = sum_f (LN1(LN_pff(H2)) - LN2(X))_f^2 wc_f + bc ----
  // only the NORMALISED rows stay live: hh = LN_pff's x_hat (H2 itself is dead from here), uh = layer_norm1's (U = hh gp + bp likewise)
  float rh, ru;
  FL uh;
  {
    FL hh;
    {
      float mh;
      fl_stats(h2, mh, rh);
#pragma unroll
      for (int e = 0; e < 16; ++e) { hh.lo[e] = (h2.lo[e] - mh) * rh; hh.hi[e] = (h2.hi[e] - mh) * rh; }
    }
    FL u;                                             // LN_pff output (before layer_norm1)
    const FL Gp = fl_vec(tpar + 0 * 64), Bp = fl_vec(tpar + 1 * 64);
#pragma unroll
    for (int e = 0; e < 16; ++e) { u.lo[e] = hh.lo[e] * Gp.lo[e] + Bp.lo[e]; u.hi[e] = hh.hi[e] * Gp.hi[e] + Bp.hi[e]; }
    if (g.ddyn0) F32_PARK_HH(hh);                     // needed again only at the end of the LayerNorm backward: parked like Y and H1
    float mu;
    fl_stats(u, mu, ru);
#pragma unroll
    for (int e = 0; e < 16; ++e) { uh.lo[e] = (u.lo[e] - mu) * ru; uh.hi[e] = (u.hi[e] - mu) * ru; }
  }
  F32_XH_RELOAD();
  FL df;                                              // dynamic - static
  {
    const FL G1 = fl_vec(tpar + 2 * 64), B1 = fl_vec(tpar + 3 * 64);
    const FL G2 = fl_vec(tpar + 4 * 64), B2 = fl_vec(tpar + 5 * 64);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      df.lo[e] = (uh.lo[e] * G1.lo[e] + B1.lo[e]) - (xh.lo[e] * G2.lo[e] + B2.lo[e]);
      df.hi[e] = (uh.hi[e] * G1.hi[e] + B1.hi[e]) - (xh.hi[e] * G2.hi[e] + B2.hi[e]);
    }
  }
  {
    const FL Wc = fl_vec(tpar + 6 * 64);
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { s0 += df.lo[e] * df.lo[e] * Wc.lo[e]; s1 += df.hi[e] * df.hi[e] * Wc.hi[e]; }
    const float o = xhalf_sum(s0 + s1) + g.hp.bc[0];
    if (h == 0) outs[r] = real ? o : 0.f;
  }
  F32_TAIL_SYNC();
  // ---- per-hyperedge masked mean -> logit (+ BCE term, + its gradient): lane e takes hyperedge b0 + e (+ 64, ... only with many all-padding rows) ----
  for (int e = lane; e < n_h; e += 64) {
    const int64_t b = b0 + e;
    const int lo_g = g.row_off[b];
    const int lo = lo_g - t0, kk = g.row_off[b + 1] - lo_g;
    float yb = 0.f, wb = 0.f;
    if (g.row_loss) { yb = g.y[b]; wb = g.w[b]; }
    float tot = 0.f;
    for (int i = 0; i < kk; ++i) tot += outs[lo + i];
    const float z = tot / ((float)kk + 1e-15f);
    g.logits[b] = z;
    if (g.row_loss) g.row_loss[b] = wb * (fmaxf(z, 0.f) - z * yb + log1pf(expf(-fabsf(z))));
    if (g.ddyn0) {                                    // main.py:56 backward: d bce / d z = w (sigmoid(z) - y) / B  (x alpha, main.py:166)
      const float dz = g.alpha_over_B * wb * (1.f / (1.f + expf(-z)) - yb);
      const float dout = dz / ((float)kk + 1e-15f);
      for (int i = 0; i < kk; ++i) douts[lo + i] = dout;
    }
  }
  FF_T(8);
  if (!g.ddyn0 || (F32_ABL & 4)) return;

  // =========================== backward of the tail and of pff_n1 (Modules.py:290-311, :353-376) ===========================
  F32_TAIL_SYNC();
  const float dout = real ? douts[r] : 0.f;
  float* tsl = g.tslab + (int64_t)blockIdx.x * kTailSlab32;
  // cross-token sums of a per-token FL quantity: through T1 as [token][feature], one lane per feature column
#define F32_COLSUM(V, SLOT)                                                                              \
  do {                                                                                                   \
    F32_TAIL_SYNC();                                                                                     \
    fl_store(myrow, V);                                                                                  \
    F32_TAIL_SYNC();                                                                                     \
    float c0__ = 0.f, c1__ = 0.f, c2__ = 0.f, c3__ = 0.f;                                                \
    _Pragma("unroll") for (int t__ = 0; t__ < 32; t__ += 4) {                                            \
      c0__ += T1[t__ * kLdH + lane]; c1__ += T1[(t__ + 1) * kLdH + lane];                                \
      c2__ += T1[(t__ + 2) * kLdH + lane]; c3__ += T1[(t__ + 3) * kLdH + lane];                          \
    }                                                                                                    \
    cs_last = (c0__ + c1__) + (c2__ + c3__);                                                             \
    tsl[kTailVec32 + (SLOT) * 64 + lane] = cs_last;                                                      \
  } while (0)
  float cs_last = 0.f, db1 = 0.f;
  FL dh2;
  {
    // d(dynamic) = 2 df wc dout;  d(static) = - d(dynamic)
    FL ddn;
    {
      const FL Wc = fl_vec(tpar + 6 * 64);
      FL aw;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        aw.lo[e] = df.lo[e] * df.lo[e] * dout; aw.hi[e] = df.hi[e] * df.hi[e] * dout;
        ddn.lo[e] = 2.f * df.lo[e] * Wc.lo[e] * dout; ddn.hi[e] = 2.f * df.hi[e] * Wc.hi[e] * dout;
      }
      F32_COLSUM(aw, 6);                              // d wc
    }
    // layer_norm2 (static branch) -> gradient into X; its affine gradients
    {
      const FL G2 = fl_vec(tpar + 4 * 64);
      FL t;
#pragma unroll
      for (int e = 0; e < 16; ++e) { t.lo[e] = -ddn.lo[e] * xh.lo[e]; t.hi[e] = -ddn.hi[e] * xh.hi[e]; }
      F32_COLSUM(t, 4);                               // d g2
#pragma unroll
      for (int e = 0; e < 16; ++e) { t.lo[e] = -ddn.lo[e] * G2.lo[e]; t.hi[e] = -ddn.hi[e] * G2.hi[e]; }      // d x_hat
      const float a = xhalf_sum(fl_sum(t)) * (1.f / 64.f), b = xhalf_sum(fl_dot(t, xh)) * (1.f / 64.f);
#pragma unroll
      for (int e = 0; e < 16; ++e) { t.lo[e] = rx * (t.lo[e] - a - xh.lo[e] * b); t.hi[e] = rx * (t.hi[e] - a - xh.hi[e] * b); }
      if (r <= n) fl_store_global(g.dXs + F32_ROW(), t);      // the padding token's row is zero (dout = 0)
    }
    // layer_norm1 (dynamic branch)
    {
      FL t;
#pragma unroll
      for (int e = 0; e < 16; ++e) { t.lo[e] = ddn.lo[e] * uh.lo[e]; t.hi[e] = ddn.hi[e] * uh.hi[e]; }
      F32_COLSUM(t, 2);                               // d g1
      F32_COLSUM(ddn, 3);                             // d b1   (d b2 = - d b1: written below)
      db1 = cs_last;
    }
    FL du;
    {
      const FL G1 = fl_vec(tpar + 2 * 64);
#pragma unroll
      for (int e = 0; e < 16; ++e) { du.lo[e] = ddn.lo[e] * G1.lo[e]; du.hi[e] = ddn.hi[e] * G1.hi[e]; }
      const float a = xhalf_sum(fl_sum(du)) * (1.f / 64.f), b = xhalf_sum(fl_dot(du, uh)) * (1.f / 64.f);
#pragma unroll
      for (int e = 0; e < 16; ++e) { du.lo[e] = ru * (du.lo[e] - a - uh.lo[e] * b); du.hi[e] = ru * (du.hi[e] - a - uh.hi[e] * b); }
    }
    // pff_n1.layer_norm
    const FL hh = F32_UNPARK_HH();
    {
      FL t;
#pragma unroll
      for (int e = 0; e < 16; ++e) { t.lo[e] = du.lo[e] * hh.lo[e]; t.hi[e] = du.hi[e] * hh.hi[e]; }
      F32_COLSUM(t, 0);                               // d gp
      F32_COLSUM(du, 1);                              // d bp
    }
    {
      const FL Gp = fl_vec(tpar + 0 * 64);
#pragma unroll
      for (int e = 0; e < 16; ++e) { dh2.lo[e] = du.lo[e] * Gp.lo[e]; dh2.hi[e] = du.hi[e] * Gp.hi[e]; }
      const float a = xhalf_sum(fl_sum(dh2)) * (1.f / 64.f), b = xhalf_sum(fl_dot(dh2, hh)) * (1.f / 64.f);
#pragma unroll
      for (int e = 0; e < 16; ++e) { dh2.lo[e] = rh * (dh2.lo[e] - a - hh.lo[e] * b); dh2.hi[e] = rh * (dh2.hi[e] - a - hh.hi[e] * b); }
    }
  }
  // d b2 = - d b1;  d bc = sum of dout over the tokens
  {
    tsl[kTailVec32 + 5 * 64 + lane] = -db1;
    float sd = (h == 0) ? dout : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sd += __shfl_xor(sd, o, 64);
    if (lane == 0) tsl[kTailVec32 + 9 * 64] = sd;
  }
  if (g.tail_dh2) {
    // large batches: the convolutions' backward (dZ1, d dyn, the two weight gradients) is a kernel of its own (tail_bwd.hip) that reads this
    // row next to the parked Y and H1 rows
    if (r <= n) fl_store_global(g.tail_dh2 + F32_ROW(), dh2);      // (rows past the tokens: dout = 0 made them zero)
    return;
  }
  // the weight stream resumes at conv1^T (the window was not refilled across the end of conv1)
  FF_T(9);
  W32_PRIME_AT(18);
  // ---- conv1: dW1[n][k] = sum_t dH2[t][n] H1[t][k];  d b1 = column sums of dH2 ----
  F32_TAIL_SYNC();
  fl_store(myrow, dh2);                               // T1 = dH2 [token][feature] (rows past the tokens are zero: dout = 0)
  {
    const FL h1b = F32_UNPARK_H1();
    fl_store(T2 + r * kLdH + 4 * h, h1b);             // T2 = H1 (the parameter vectors are dead)
  }
  F32_TAIL_SYNC();
#define F32_TN(A_T, B_T, SLAB, CS_SLOT)                                                                  \
  do {                                                                                                   \
    float cs__[2];                                                                                       \
    _Pragma("unroll") for (int wr__ = 0; wr__ < 2; ++wr__) {                                             \
      f32x16 a0__ = {0}, a1__ = {0};                                                                     \
      float s__ = 0.f;                                                                                   \
      _Pragma("unroll 8") for (int m__ = 0; m__ < 16; ++m__) {                                           \
        const int t__ = 2 * m__ + h;                                                                     \
        const float ga__ = (A_T)[t__ * kLdH + 32 * wr__ + r];                                            \
        s__ += ga__;                                                                                     \
        a0__ = MFMA32(ga__, (B_T)[t__ * kLdH + r], a0__);                                                \
        a1__ = MFMA32(ga__, (B_T)[t__ * kLdH + 32 + r], a1__);                                           \
      }                                                                                                  \
      cs__[wr__] = xhalf_sum(s__);                                                                       \
      f32x4* s0__ = reinterpret_cast<f32x4*>(SLAB) + ((0 * 2 + wr__) * 64 + lane) * 4;                   \
      f32x4* s1__ = reinterpret_cast<f32x4*>(SLAB) + ((1 * 2 + wr__) * 64 + lane) * 4;                   \
      _Pragma("unroll") for (int q__ = 0; q__ < 4; ++q__) {                                              \
        s0__[q__] = (f32x4){a0__[4 * q__], a0__[4 * q__ + 1], a0__[4 * q__ + 2], a0__[4 * q__ + 3]};     \
        s1__[q__] = (f32x4){a1__[4 * q__], a1__[4 * q__ + 1], a1__[4 * q__ + 2], a1__[4 * q__ + 3]};     \
      }                                                                                                  \
    }                                                                                                    \
    tsl[kTailVec32 + (CS_SLOT) * 64 + lane] = h == 0 ? cs__[0] : cs__[1];                                \
  } while (0)
  F32_TN(T1, T2, tsl, 7);
  FF_T(10);
  // ---- dZ1^T = W1^T . dH2^T, x dropout mask x tanh' ----
  FL dz = fl_zero();
  W32_CHAIN(dz, dh2, true);
  {
    const float unscale = drop2 ? 1.f - g.p_pff : 1.f;
    const FL h1b = fl_load(T2 + r * kLdH + 4 * h);    // this lane's own H1 row, still in T2 (overwritten with dZ1 below)
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      const float hval = (e < 16 ? h1b.lo[e] : h1b.hi[e - 16]) * unscale;     // tanh value (0 where dropped)
      float v = e < 16 ? dz.lo[e] : dz.hi[e - 16];
      if (drop2) v = ((keep2 >> e) & 1u) ? v * ks2 : 0.f;
      v *= 1.f - hval * hval;
      if (e < 16) dz.lo[e] = v; else dz.hi[e - 16] = v;
    }
  }
  // ---- conv0: dW0[n][k] = sum_t dZ1[t][n] Y[t][k];  d b0 = column sums of dZ1 ----
  F32_TAIL_SYNC();                                    // the column walks over dH2 and H1 are done
  fl_store(T2 + r * kLdH + 4 * h, dz);
  {
    const FL yb = F32_UNPARK_Y();
    fl_store(myrow, yb);
  }
  F32_TAIL_SYNC();
  FF_T(11);
  F32_TN(T2, T1, tsl + 4096, 8);
  FF_T(12);
  // ---- d dyn^T = (W0^T . dZ1^T + dH2^T) x dropout mask x row mask ----
  FL dd = dh2;                                        // residual: H2 = conv1(H1) + Y
  W32_CHAIN(dd, dz, false);
#pragma unroll
  for (int e = 0; e < 32; ++e) {
    float v = e < 16 ? dd.lo[e] : dd.hi[e - 16];
    if (drop1) v = ((keep1 >> e) & 1u) ? v * ks1 : 0.f;
    v = real ? v : 0.f;
    if (e < 16) dd.lo[e] = v; else dd.hi[e - 16] = v;
  }
  if (r <= n) fl_store_global(g.ddyn0 + F32_ROW(), dd);     // the padding token's row: zeros (every half tile writes the same)
  FF_T(13);
