/* CPU twin (plain C) of matcha_forward for the TABLE front end in eval mode -- TEST INFRASTRUCTURE ONLY (oracle/__init__.py).
 * Statement for statement the live path of Classifier.forward (Modules.py:278-318) with the structs of include/matcha_hip.h (host
 * pointers here): the same function as oracle/hypersagnn.py::classifier_forward, in C, so that logits of the HIP path can be held to it
 * at thousands of rows without PyTorch; pinned to the real reference by the G2 goldens (tests/test_cpu_twins.py).
 *   get_embedding :261-276      x0 = table[id] + attribute_nn(attr_table[id]);  X = tanh(next_w(x0))
 *   MultiHeadAttention :513-575 q/k/v = LayerNorm_{1,2,3}(X) W_{q,k,v}^T per head (d_k = d_v = d_model), scores / sqrt(d), ONLY the diagonal
 *                               masked (-1e32): padding slots are attended like any key (SURVEY.md headline fact 7), softmax, . V,
 *                               heads concatenated, fc1 + bias
 *   PositionwiseFeedForward :353-376 (pff_n1)  y = dyn * non_pad; h = conv1(tanh(conv0(y))) + y; LayerNorm; * non_pad
 *   tail :290-311               (LayerNorm1(dynamic) - LayerNorm2(X))^2 . cls_w + cls_b, mean over the non-padding slots
 * Sums accumulate in double (closer to exact than either fp32 implementation). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/matcha_hip.h"

#define NH MATCHA_N_HEAD

static void layer_norm(const float* x, int d, const float* g, const float* b, float* out) {
  double m = 0.0, v = 0.0;
  for (int i = 0; i < d; ++i) m += x[i];
  m /= d;
  for (int i = 0; i < d; ++i) v += (x[i] - m) * (x[i] - m);
  v /= d;                                               /* biased variance, eps 1e-5 (nn.LayerNorm defaults) */
  const double rs = 1.0 / sqrt(v + 1e-5);
  for (int i = 0; i < d; ++i) out[i] = (float)((x[i] - m) * rs * g[i] + b[i]);
}
static void linear(const float* x, const float* W, const float* b, int n_out, int n_in, float* out) {      /* out = W x + b, W [n_out, n_in] */
  for (int o = 0; o < n_out; ++o) {
    double s = b ? b[o] : 0.0;
    for (int i = 0; i < n_in; ++i) s += (double)W[(int64_t)o * n_in + i] * x[i];
    out[o] = (float)s;
  }
}

int matcha_forward_cpu(const matcha_shape* shp, const matcha_tensors* p, const matcha_frozen* f, const int64_t* x, int64_t B, int32_t L,
                       float* logits) {
  if (!shp || !p || !f || !x || !logits || shp->mode != 0 || L < 1 || L > MATCHA_MAX_L || f->attr_mode != 0) return -22;
  const int d = shp->d, na = shp->n_attr, ald = f->attr_ld > 0 ? f->attr_ld : na;
  const double inv_temp = 1.0 / sqrt((double)d);
  float* buf = (float*)malloc(sizeof(float) * ((size_t)L * d * 3 + (size_t)L * NH * d * 4 + 8 * (size_t)d));
  if (!buf) return -12;
  float* X = buf; float* dyn = X + (size_t)L * d; float* dynamic = dyn + (size_t)L * d;
  float* q = dynamic + (size_t)L * d; float* k = q + (size_t)L * NH * d; float* v = k + (size_t)L * NH * d; float* o = v + (size_t)L * NH * d;
  float* t0 = o + (size_t)L * NH * d; float* t1 = t0 + d; float* t2 = t1 + d; float* t3 = t2 + d;
  for (int64_t b = 0; b < B; ++b) {
    const int64_t* row = x + b * L;
    for (int l = 0; l < L; ++l) {
      int64_t id = row[l];
      if (id < 0 || id > shp->n_nodes) id = 0;
      linear(f->attr_table + id * ald, p->attr_w, p->attr_b, d, na, t0);
      for (int i = 0; i < d; ++i) t0[i] += p->table[id * d + i];                                        /* x0 */
      linear(t0, p->next_w, p->next_b, d, d, t1);
      for (int i = 0; i < d; ++i) X[(size_t)l * d + i] = tanhf(t1[i]);
      layer_norm(X + (size_t)l * d, d, p->ln_q_g, p->ln_q_b, t0); linear(t0, p->w_q, NULL, NH * d, d, q + (size_t)l * NH * d);
      layer_norm(X + (size_t)l * d, d, p->ln_k_g, p->ln_k_b, t0); linear(t0, p->w_k, NULL, NH * d, d, k + (size_t)l * NH * d);
      layer_norm(X + (size_t)l * d, d, p->ln_v_g, p->ln_v_b, t0); linear(t0, p->w_v, NULL, NH * d, d, v + (size_t)l * NH * d);
    }
    for (int h = 0; h < NH; ++h)
      for (int i = 0; i < L; ++i) {
        double s[MATCHA_MAX_L], mx = -1e300, den = 0.0;
        for (int j = 0; j < L; ++j) {
          double a = 0.0;
          for (int c = 0; c < d; ++c) a += (double)q[((size_t)i * NH + h) * d + c] * k[((size_t)j * NH + h) * d + c];
          s[j] = (i == j) ? -1e32 : a * inv_temp;
          if (s[j] > mx) mx = s[j];
        }
        for (int j = 0; j < L; ++j) { s[j] = exp(s[j] - mx); den += s[j]; }
        for (int c = 0; c < d; ++c) {
          double a = 0.0;
          for (int j = 0; j < L; ++j) a += s[j] / den * v[((size_t)j * NH + h) * d + c];
          o[((size_t)i * NH + h) * d + c] = (float)a;
        }
      }
    double num = 0.0, cnt = 0.0;
    for (int l = 0; l < L; ++l) {
      const float np = row[l] != 0 ? 1.f : 0.f;
      linear(o + (size_t)l * NH * d, p->fc1_w, p->fc1_b, d, NH * d, t0);                                /* dyn */
      for (int i = 0; i < d; ++i) t0[i] *= np;                                                          /* y */
      linear(t0, p->pff0_w, p->pff0_b, d, d, t1);
      for (int i = 0; i < d; ++i) t1[i] = tanhf(t1[i]);
      linear(t1, p->pff1_w, p->pff1_b, d, d, t2);
      for (int i = 0; i < d; ++i) t2[i] += t0[i];
      layer_norm(t2, d, p->pff_ln_g, p->pff_ln_b, t3);
      for (int i = 0; i < d; ++i) t3[i] *= np;                                                          /* dynamic */
      layer_norm(t3, d, p->ln1_g, p->ln1_b, t0);
      layer_norm(X + (size_t)l * d, d, p->ln2_g, p->ln2_b, t1);
      double out = p->cls_b[0];
      for (int i = 0; i < d; ++i) out += (double)(t0[i] - t1[i]) * (t0[i] - t1[i]) * p->cls_w[i];
      num += out * np; cnt += np;
    }
    logits[b] = (float)(num / (cnt + 1e-15));
  }
  free(buf);
  return 0;
}
