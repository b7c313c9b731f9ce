/* CPU twins (plain C) of the byte / index / optimizer entry points of include/matcha_hip.h -- TEST INFRASTRUCTURE ONLY (oracle/__init__.py:
 * nothing under matcha_amd/ links or calls this; the product has no CPU path).  SURVEY.md section 8 b2 asks for `*_cpu` twins of the
 * minimum export set with the same signatures (minus the stream): they restate the REFERENCE's semantics for those ops and are
 * what tests/test_cpu_twins.py checks against numpy / torch on the host and tests/test_hip_kernels.py against the HIP kernels.
 *   matcha_gather_rows_cpu        nn.Embedding forward / Wrap_Embedding.forward (Modules.py:29-34): rows[t] = table[ids[t]]; an id outside
 *                                 [0, n_nodes] is flagged (status[0] |= 1) and read as row 0 (the reference raises IndexError)
 *   matcha_embed_scatter_bwd_cpu  nn.Embedding backward with padding_idx = 0: dtable[x[t]] += dx0[t] in token order, row 0 skipped
 *   matcha_adamw_step_cpu         torch.optim.AdamW(lr) as main.py:630 builds it over one flat buffer with per-tensor segments: a segment
 *                                 whose group was not touched is skipped entirely ("grad is None": no decay, no step count); otherwise
 *                                 p *= 1 - lr wd; m = lerp(m, g, 1 - b1); v = b2 v + (1 - b2) g g; p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps);
 *                                 g = 0 (opt.zero_grad, main.py:175-176).  Scalars rounded to float exactly where torch / the kernel round them.
 *   matcha_hashset_contains_cpu   exact membership of zero-padded ascending rows in a list of known hyperedges (build_hash utils.py:75-97 is
 *                                 a Bloom filter per size; the HIP path and this twin are exact): linear scan, for small fixtures. */
#include <math.h>
#include <stdint.h>
#include <string.h>

int matcha_gather_rows_cpu(const int64_t* ids, int64_t T, int32_t d, const float* table, int64_t n_nodes, float* rows, int32_t* status) {
  for (int64_t t = 0; t < T; ++t) {
    int64_t id = ids[t];
    if (id < 0 || id > n_nodes) { if (status) status[0] |= 1; id = 0; }
    memcpy(rows + t * (int64_t)d, table + id * (int64_t)d, (size_t)d * sizeof(float));
  }
  return 0;
}

int matcha_embed_scatter_bwd_cpu(const int64_t* x, int64_t T, int32_t d, const float* dx0, float* dtable) {
  for (int64_t t = 0; t < T; ++t) {
    const int64_t id = x[t];
    if (id == 0) continue;                               /* padding_idx = 0 receives no gradient */
    for (int j = 0; j < d; ++j) dtable[id * (int64_t)d + j] += dx0[t * (int64_t)d + j];
  }
  return 0;
}

int matcha_adamw_step_cpu(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, const int64_t* seg_off, int32_t n_seg,
                          const int32_t* seg_group, const int32_t* touched, int32_t* seg_step, double lr, double beta1, double beta2, double eps,
                          double weight_decay, double grad_scale) {
  const float decay = (float)(1.0 - lr * weight_decay), omb1 = (float)(1.0 - beta1), b2 = (float)beta2, omb2 = (float)(1.0 - beta2);
  const float epsf = (float)eps, gscale = (float)grad_scale;
  for (int32_t s = 0; s < n_seg; ++s) {
    const int active = touched ? (touched[seg_group ? seg_group[s] : 0] != 0) : 1;
    if (!active) continue;
    const int t = ++seg_step[s];
    const float step_size = (float)(lr / (1.0 - pow(beta1, (double)t)));
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow(beta2, (double)t)));
    const int64_t hi = seg_off[s + 1] < n ? seg_off[s + 1] : n;
    for (int64_t i = seg_off[s]; i < hi; ++i) {
      const float gg = grads[i] * gscale;
      float p = params[i] * decay;
      const float m = exp_avg[i] + (gg - exp_avg[i]) * omb1;
      const float v = exp_avg_sq[i] * b2 + omb2 * gg * gg;
      const float denom = sqrtf(v) * inv_sqrt_bc2 + epsf;
      p -= step_size * (m / denom);
      params[i] = p; exp_avg[i] = m; exp_avg_sq[i] = v; grads[i] = 0.f;
    }
  }
  return 0;
}

int matcha_hashset_contains_cpu(const int64_t* edges, int64_t n_edges, int32_t L, const int64_t* rows, int64_t n_rows, int32_t Lr, uint8_t* out) {
  for (int64_t r = 0; r < n_rows; ++r) {
    out[r] = 0;
    for (int64_t e = 0; e < n_edges && !out[r]; ++e) {
      int same = 1;
      const int Lm = L > Lr ? L : Lr;
      for (int j = 0; j < Lm && same; ++j) {
        const int64_t a = j < L ? edges[e * L + j] : 0, b = j < Lr ? rows[r * Lr + j] : 0;
        same = a == b;
      }
      out[r] = (uint8_t)same;
    }
  }
  return 0;
}
