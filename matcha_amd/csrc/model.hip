// Orchestration of one classifier step behind the C ABI: Classifier.forward (Modules.py:278-318) and its
// backward as a fixed sequence of kernel launches on the caller's stream -- no allocation, no host
// synchronisation, so the whole step is hipGraph-capturable.
//
// Token layout: ragged (ragged.hip).  The token-level layers run on Tn = Tr + 1 rows -- the Tr real tokens of the
// batch plus ONE shared padding token -- instead of B*L slots; Tr is only known on the device, so every launch is
// sized for the upper bound B*L + 1 and reads the true count from `count` (m_dev / r_dev / t_dev).
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <unordered_map>

#include "kernels.hpp"
#include "tail_reduce.hpp"

namespace matcha {

thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int device_cu_count() {
  static int cache[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
  if (dev >= 0 && dev < 64 && cache[dev] > 0) return cache[dev];
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); return 256; }
  if (dev >= 0 && dev < 64) cache[dev] = n;
  return n;
}

// ---- per-kernel-class event timing -------------------------------------------------------------------
int g_prof_class = 0;
static hipEvent_t* g_prof_ev = nullptr;     // pairs (start, stop)
static int g_prof_cap = 0, g_prof_used = 0;
static double g_prof_work = 0.0;
void prof_record(bool start, double work, hipStream_t st) {
  if (start) {
    if (g_prof_used + 2 > g_prof_cap) {
      const int ncap = g_prof_cap ? 2 * g_prof_cap : 1024;
      hipEvent_t* ne = (hipEvent_t*)realloc(g_prof_ev, sizeof(hipEvent_t) * ncap);
      if (!ne) return;
      g_prof_ev = ne;
      for (int i = g_prof_cap; i < ncap; ++i)
        if (hipEventCreate(&g_prof_ev[i]) != hipSuccess) { g_prof_cap = i & ~1; return; }
      g_prof_cap = ncap;
    }
    g_prof_work += work;
    (void)hipEventRecord(g_prof_ev[g_prof_used], st);
  } else if (g_prof_used + 1 < g_prof_cap) {
    (void)hipEventRecord(g_prof_ev[g_prof_used + 1], st);
    g_prof_used += 2;
  }
}

// ---- launch log: kernel name -> number of launches since matcha_launch_log(1) -------------------------
int g_launch_log = 0;
namespace {
struct LaunchCount { const char* name; int64_t n; };
constexpr int kLaunchNames = 128;
LaunchCount g_launches[kLaunchNames];
int g_launch_names = 0;
}  // namespace
void note_launch(const char* name) {
  for (int i = 0; i < g_launch_names; ++i)
    if (g_launches[i].name == name || strcmp(g_launches[i].name, name) == 0) { ++g_launches[i].n; return; }
  if (g_launch_names < kLaunchNames) g_launches[g_launch_names++] = LaunchCount{name, 1};
}

// adj_frontend.hip
int adj_forward(const matcha_shape& s, const matcha_tensors& p, const matcha_frozen& f, const matcha_step_opts& o,
                const int64_t* x, int64_t T, float* node_out, float* recon_out, void* ws, size_t ws_bytes, hipStream_t st,
                const int32_t* t_dev, const int32_t* slot_map, float* fused_x0 = nullptr, float* fused_X = nullptr, bool save = false,
                bool fused_node = false);
int adj_backward(const matcha_shape& s, const matcha_tensors& p, const matcha_frozen& f, const matcha_step_opts& o,
                 const int64_t* x, int64_t T, float* dnode, const float* drecon, matcha_tensors& g, int32_t* touched,
                 void* ws, size_t ws_bytes, void* gemm_ws, size_t gemm_ws_bytes, hipStream_t st, const int32_t* slot_map, bool fused = false);
size_t adj_workspace_bytes(const matcha_shape& s, int64_t T);
bool adj_fused_eligible(const matcha_shape& s, const matcha_frozen& f);     // adj_fused.hip

struct Workspace {
  Ragged rg;
  // saved by forward
  float *x0, *X, *qin, *kin, *vin, *stats, *Q, *K, *V, *P, *O, *Y, *H1, *H2, *row_loss, *logits, *node;
  // backward temporaries
  float *dH2, *dXs, *dZ1, *ddyn0, *dO, *dQ, *dK, *dV, *dqin, *dkin, *dvin, *dZ0, *dX0;
  float* slab;      size_t slab_bytes;    // column-sum slabs (LayerNorm / tail parameter gradients, attention pad-token grads)
  float* gemm_ws;   size_t gemm_ws_bytes; // TN GEMM slabs
  void* adj_ws;     size_t adj_ws_bytes;
  float* folded;                          // LayerNorm-folded projection weights of the fused d = 64 kernels
  float* lwB; float* lwM; float* lwdB; float* lwdM;   // embed_dim >= 128, merged heads: B_all [8d, d], M_all [d, 8d] and their gradients
  float* enc;                             // embed_dim 128: the fused attention block's folded weights, fragments and slabs (enc128.hip)
  float* enc_rec; size_t enc_rec_floats;  // ... and its forward records, over the Q / K / V / P / O buffers the fused path does not use
  float* merged;                          // merged per-head matrices B_h = W'k^T W'q, M_h = Wfc1_h W'v (two products per head instead of four)
  float* frag;                            // the same weights + fc1 / pff_n1 blocks in MFMA-fragment order (fused_fwd32.hip streams them from L2)
  float* fb_ws;                           // fused backward: workgroup slabs + reduction partials
  float* tslab;                           // training forward: per-tile partials of the tail / pff_n1 parameter gradients
  float* tpart;                           // their two-pass reduction's split partials
  float* tslab2;                          // large batches: one slab of weight-gradient partials per workgroup of tail_bwd64_kernel
  float* qkv;                             // training forward -> fused backward: Q, K, V tiles of every (tile, head), 384 KB per tile
  float* front_ws;                        // fused front-end backward: workgroup slabs
  void* tg_ws;      size_t tg_ws_bytes;   // table mode: sort scratch of the deterministic table gradient (table_grad.hip)
  size_t total;
};

// A/B switches for tests and profiling (include/matcha_hip.h, matcha_set_option): ONE process-wide struct whose initial values
// come from the environment (MATCHA_DISABLE_FUSED, ...) when it is first touched; no entry point calls getenv per call.
//   disable_fused            1: layer-by-layer kernels everywhere (the d != 64 path, at embed_dim 64 too); 2: only the FRONT END (gather +
//                            attribute_nn + next_w; its backward; the fused adj kernels) as separate kernels, the encoder stays fused
//                            (the tail's backward inside the forward kernel is a per-call choice: opts->loss_in_forward)
//   disable_merged           the REFERENCE formulation of the heads (Q, K, V, fc1: four products per head forward, eight backward) instead
//                            of the merged two / four.  It lives on the layer-by-layer kernels only (attention.hip + the GEMMs), so at
//                            embed_dim 64 the switch implies disable_fused; the fused four-product kernels of rounds 1-3 (four-wave
//                            forward, eight-wave / 64-row-tile / recompute backward, their five switches) are gone since round 4
//   disable_small_batch      the kernels picked for small batches by size -- one workgroup of eight wavefronts (one per head) per half tile
//                            in the forward, the ragged plan as one launch -- replaced by the large-batch kernels
//   disable_wide_gemm        embed_dim >= 128: the 64-wide GEMM / attention kernels (gemm_lds.hip, gemm_f32.hip, attention.hip; four-product
//                            heads) instead of gemm_wide.hip / attention_wide.hip
struct OptionName { const char* name; int Options::*field; };
static const OptionName kOptionNames[] = {
    {"disable_fused", &Options::disable_fused}, {"disable_merged", &Options::disable_merged}, {"disable_small_batch", &Options::disable_small_batch},
    {"disable_wide_gemm", &Options::disable_wide_gemm}, {"debug_nan", &Options::debug_nan}, {"fused_dbg", &Options::fused_dbg}};
Options& options() {
  static Options o = [] {
    Options v;
    memset(&v, 0, sizeof(v));
    for (const OptionName& n : kOptionNames) {
      char env[64] = "MATCHA_";
      size_t k = strlen(env);
      for (const char* c = n.name; *c && k + 1 < sizeof(env); ++c) env[k++] = (char)((*c >= 'a' && *c <= 'z') ? *c - 32 : *c);
      env[k] = 0;
      const char* e = getenv(env);
      if (e) v.*(n.field) = (*e == 0) ? 1 : atoi(e);       // "MATCHA_X=" (set, empty) counts as on, like the old getenv != NULL test
      if (e && v.*(n.field) == 0 && strcmp(e, "0") != 0) v.*(n.field) = 1;
    }
    return v;
  }();
  return o;
}
// embed_dim >= 128: the layer-by-layer path with merged heads -- r = qin B_all^T (one projection instead of three), every head attends the
// SAME key / value rows kin = LN_k(x), vin = LN_v(x) ([T, d], not [T, 8d]), dyn = Z M_all^T with M_all[:, h] = Wfc1_h W_v[h]; the
// backward computes dB_all, dM_all and applies the chain rule per head (merged_chain).  Shapes the wide attention kernels take.
static bool merged_layerwise_shape(const matcha_shape& s) { return s.d >= 128 && s.d % 64 == 0; }
static bool merged_layerwise(const matcha_shape& s) { return merged_layerwise_shape(s) && attn_wide_eligible(s.d) && !options().disable_merged; }
// embed_dim 128: attention block as one fused forward / backward kernel pair (enc128.hip).  The heads' d x_hat meet through float atomics, so the
// callers that need a bitwise reproducible embedding gradient stay on the layer-by-layer kernels.
static bool enc128_enabled(const matcha_shape& s, const matcha_step_opts& o) {
  return enc128_shape(s.d) && merged_layerwise(s) && options().disable_fused != 1 && !o.deterministic && !o.sparse_table_grad;
}
static bool fused_enabled(const matcha_shape& s) { return s.d == 64 && options().disable_fused != 1 && !options().disable_merged; }
static bool fused_front_enabled() { return (options().disable_fused & 2) == 0; }
static bool loss_in_forward(const matcha_shape& s, const matcha_step_opts& o, const float* y, const float* w) {
  return o.loss_in_forward && !o.forward_only && y && w && fused_enabled(s);
}
// Which formulation the forward that last ran on a workspace used.  matcha_backward keys off THIS record, not off the option: flipping
// disable_merged between a forward and its backward (two separate calls on the autograd path) would otherwise make the merged backward
// consume records nobody wrote.  Host-side record per workspace pointer (the decision picks a kernel, so it cannot live in device memory
// without a synchronisation); bounded, evicted oldest-first, guarded by a mutex.  A backward on a workspace WITHOUT a record is refused.
static std::mutex g_fwd_mu;
static std::unordered_map<const void*, std::pair<int, uint64_t>> g_fwd_state;     // ws -> (bits, age); bit 0: merged heads, bit 1: fused d = 64 forward, bit 2: fused d = 128 attention block, bit 3: the tail's backward ran as tail_bwd64_kernel, bit 4: the forward zeroed the d x_hat rows of the backward kernel
static uint64_t g_fwd_clock = 0;
static void note_forward(const void* ws, bool merged, bool fused, bool enc = false, bool split_tail = false, bool dx_zeroed = false) {
  std::lock_guard<std::mutex> lk(g_fwd_mu);
  if (g_fwd_state.size() >= 4096 && g_fwd_state.find(ws) == g_fwd_state.end()) {
    auto old = g_fwd_state.begin();
    for (auto it = g_fwd_state.begin(); it != g_fwd_state.end(); ++it)
      if (it->second.second < old->second.second) old = it;
    g_fwd_state.erase(old);
  }
  g_fwd_state[ws] = std::make_pair((merged ? 1 : 0) | (fused ? 2 : 0) | (enc ? 4 : 0) | (split_tail ? 8 : 0) | (dx_zeroed ? 16 : 0), ++g_fwd_clock);
}
static int ws_state(const void* ws) {       // -1: no forward on record for this workspace
  std::lock_guard<std::mutex> lk(g_fwd_mu);
  auto it = g_fwd_state.find(ws);
  return it != g_fwd_state.end() ? it->second.first : -1;
}
static bool fwd_ran_merged(const void* ws) { return ws_state(ws) > 0 && (ws_state(ws) & 1) != 0; }

// B_all[h d + a][b] = sum_m W_k[h d + m][a] W_q[h d + m][b];   M_all[n][h d + b] = sum_m Wfc1[n][h d + m] W_v[h d + m][b]   (16 small GEMMs)
static int merged_weights(const matcha_shape& s, const matcha_tensors& p, Workspace& w, hipStream_t st) {
  const int64_t d = s.d, hd = (int64_t)MATCHA_N_HEAD * d;
  // both products, all heads, one launch (bmm_heads.hip; every embed_dim the merged layer-wise path takes is a multiple of 64)
  const BmmProduct pr[2] = {
      {p.w_k, 1, d, d * d, p.w_q, d, 1, d * d, w.lwB, d, d * d, 0},            // B_h[a][b] = sum_m W_k[h d + m][a] W_q[h d + m][b]
      {p.fc1_w, hd, 1, d, p.w_v, d, 1, d * d, w.lwM, hd, d, 0}};               // M_all[n][h d + b] = sum_m Wfc1[n][h d + m] W_v[h d + m][b]
  return launch_bmm_heads(pr, 2, s.d, st);
}
// chain rule from dB_all, dM_all to the projections (accumulating):  dW_q[h] += W_k[h] dB_h;  dW_k[h] += W_q[h] dB_h^T;
//   dWfc1[:, h] += dM_h W_v[h]^T;  dW_v[h] += Wfc1[:, h]^T dM_h
static int merged_chain(const matcha_shape& s, const matcha_tensors& p, matcha_tensors& g_, Workspace& w, hipStream_t st) {
  const int64_t d = s.d, hd = (int64_t)MATCHA_N_HEAD * d;
  const BmmProduct pr[4] = {
      {p.w_k, d, 1, d * d, w.lwdB, d, 1, d * d, g_.w_q, d, d * d, 1},          // dW_q[h][m][b] += sum_a W_k[h][m][a] dB_h[a][b]
      {p.w_q, d, 1, d * d, w.lwdB, 1, d, d * d, g_.w_k, d, d * d, 1},          // dW_k[h][m][a] += sum_b W_q[h][m][b] dB_h[a][b]
      {w.lwdM, hd, 1, d, p.w_v, 1, d, d * d, g_.fc1_w, hd, d, 1},              // dWfc1[n][h d + m] += sum_b dM[n][h d + b] W_v[h d + m][b]
      {p.fc1_w, 1, hd, d, w.lwdM, hd, 1, d, g_.w_v, d, d * d, 1}};             // dW_v[h][m][b] += sum_n Wfc1[n][h d + m] dM[n][h d + b]
  return launch_bmm_heads(pr, 4, s.d, st);
}

// `compact`: layout of a forward that will not be differentiated and runs the fused kernels (d = 64): only the ragged plan,
// x0, X, the adj front end's buffers, the folded weights and the per-row outputs exist; everything else has size 0.
static size_t carve(const matcha_shape& s, int64_t B, int L, char* base, Workspace& w, bool compact = false) {
  const int64_t Tn = B * L + 1, d = s.d, hd = (int64_t)MATCHA_N_HEAD * d;      // upper bound of token rows
  size_t off = 0;
  auto take_always = [&](size_t n_floats) {
    float* p = base ? (float*)(base + off) : nullptr;
    off += align_up(n_floats * sizeof(float), 256);
    return p;
  };
  auto take = [&](size_t n_floats) { return compact ? (float*)nullptr : take_always(n_floats); };
  {
    const size_t rb = ragged_bytes(B, L);
    if (base) ragged_carve(B, L, base + off, w.rg);
    off += align_up(rb, 256);
  }
  w.x0 = take_always(Tn * d); w.X = take_always(Tn * d);
  w.qin = take(Tn * d); w.kin = take(Tn * d); w.vin = take(Tn * d);
  w.stats = take(Tn * 2);
  w.Q = take(Tn * hd); w.K = take(Tn * hd); w.V = take(Tn * hd);
  w.P = take(B * MATCHA_N_HEAD * L * L);
  w.O = take(Tn * hd);
  w.Y = take(Tn * d); w.H1 = take(Tn * d); w.H2 = take(Tn * d);
  w.row_loss = take_always(B); w.logits = take_always(B);
  w.node = take_always(s.mode == 1 ? Tn * d : 0);
  w.dH2 = take(Tn * d); w.dXs = take(Tn * d); w.dZ1 = take(Tn * d); w.ddyn0 = take(Tn * d);
  // The attention block's gradients REPLACE the activations they are computed from (layer-by-layer path; the 8d-wide tensors are
  // 32 of the ~80 d floats a token cost): dO overwrites O once the fc1 weight gradient has read it; attn_bwd reads every V chunk
  // of a hyperedge before it writes its first dV chunk, and per chunk loads all K rows before it stores dQ and all Q rows before
  // it stores dK -- so dV lives in V, dQ in K's buffer and dK in Q's (attention.hip / attention_wide.hip keep that order);
  // dqin / dkin / dvin overwrite qin / kin / vin after the projection weight gradients have read them.  With embed_dim 64 the
  // fused kernels use dO as the 8 per-head d x_hat slabs and never touch Q/K/V/O, so there the alias is harmless too.
  w.dO = w.O; w.dQ = w.K; w.dK = w.Q; w.dV = w.V;
  w.dqin = w.qin; w.dkin = w.kin; w.dvin = w.vin;
  w.dZ0 = take(Tn * d); w.dX0 = take(Tn * d);
  size_t sb = colsum_slab_bytes(Tn, 6, (int)d);
  size_t sb2 = colsum_slab_bytes(B, 7, (int)d);
  if (sb2 > sb) sb = sb2;
  sb2 = attn_bwd_slab_bytes(B, (int)d);
  if (sb2 > sb) sb = sb2;
  w.slab_bytes = sb; w.slab = take(sb / sizeof(float));
  size_t gb = gemm_tn_ws_bytes(hd, d, Tn);                       // dWq/dWk/dWv
  size_t g2 = gemm_tn_ws_bytes(d, hd, Tn); if (g2 > gb) gb = g2; // dfc1
  g2 = gemm_tn_ws_bytes(d, d, Tn); if (g2 > gb) gb = g2;
  g2 = gemm_tn_ws_bytes(d, s.n_attr, Tn); if (g2 > gb) gb = g2;
  if (s.mode == 1) {   // recon head gradient [n_r, d] for any chromosome r: the slab count depends on ceil(n_r/64)
    for (int64_t m = 64;; m += 64) {
      const int64_t mm = m < s.max_bins ? m : s.max_bins;
      g2 = gemm_tn_ws_bytes(mm, d, Tn); if (g2 > gb) gb = g2;
      if (mm >= s.max_bins) break;
    }
  }
  w.gemm_ws_bytes = gb; w.gemm_ws = take(gb / sizeof(float));
  w.adj_ws_bytes = (s.mode == 1) ? adj_workspace_bytes(s, Tn) : 0;
  w.adj_ws = take_always(w.adj_ws_bytes / sizeof(float));
  w.folded = take_always(s.d == 64 ? fused_fold_floats() : 0);
  w.frag = take_always(s.d == 64 ? fused_frag_floats() : 0);
  w.merged = take_always(s.d == 64 ? fused_merged_floats() : 0);
  {
    const size_t nm = merged_layerwise_shape(s) ? (size_t)hd * d : 0;
    w.lwB = take_always(nm); w.lwM = take_always(nm);
    w.lwdB = take(nm); w.lwdM = take(nm);
  }
  w.enc = take_always(enc128_shape((int)d) ? enc128_ws_floats() : 0);
  w.enc_rec = w.Q;                                        // [Q, K, V, P, O] are consecutive: Tn * 32 d + B * 8 L^2 floats
  w.enc_rec_floats = (w.Q && w.O) ? (size_t)((w.O + Tn * hd) - w.Q) : 0;
  w.fb_ws = take(s.d == 64 ? fused_bwd_ws_floats(B, L) : 0);
  w.tpart = take(s.d == 64 ? fused_tail_partial_floats() : 0);
  w.tslab = take(s.d == 64 ? fused_tail_slab32_floats(B, L) : 0);   // one slab per HALF tile (>= the four-wave kernel's per-tile need)
  w.tslab2 = take(s.d == 64 ? tail_bwd_slab_floats() : 0);
  w.qkv = take(s.d == 64 ? fused_qkv_floats(B, L) : 0);        // reserved whatever the option table says: the layout must not depend on a switch read per call
  w.front_ws = take(front_bwd_supported(s.d, s.n_attr) ? front_bwd_ws_floats() : 0);
  w.tg_ws_bytes = (s.mode == 0 && !compact) ? table_grad_ws_bytes(Tn, s.n_nodes) : 0;
  w.tg_ws = take(w.tg_ws_bytes / sizeof(float));
  w.total = off;
  return off;
}

// Table mode, after the backward pass left one gradient row per compact token in w.dX0 (ids in w.rg.tok_key): add them into the
// dense table gradient deterministically (table_grad.hip; opts.deterministic) -- unless the caller asked for the row-sparse form
// (opts.sparse_table_grad: the list stays in the workspace for matcha_table_grad_rows / the data-parallel exchange).
static int table_gradient(const matcha_shape& s, const matcha_step_opts& o, const Workspace& w, int64_t Tn, matcha_tensors& g, hipStream_t st) {
  if (o.sparse_table_grad) return MATCHA_OK;
  if (!o.deterministic) return MATCHA_OK;                 // the front-end kernel already added the rows with float atomics
  return launch_table_grad(w.rg.tok_key, w.dX0, Tn, s.d, s.n_nodes, g.table, w.tg_ws, w.tg_ws_bytes, st);
}

// opts->encoder_done_event: every gradient from ln_q_g to cls_b is final at this point of the stream (include/matcha_hip.h)
static int encoder_done(const matcha_step_opts& o, hipStream_t st) {
  if (o.encoder_done_event && hipEventRecord((hipEvent_t)o.encoder_done_event, st) != hipSuccess) {
    set_error("matcha_backward: recording encoder_done_event failed"); return MATCHA_EHIP;
  }
  return MATCHA_OK;
}

static int check_shape(const matcha_shape* s, int64_t B, int32_t L) {
  MATCHA_CHECK_ARG(s, "null shape");
  MATCHA_CHECK_ARG(s->d >= 8 && s->d <= 256 && s->d % 8 == 0 && (s->d <= 64 || s->d % 64 == 0),
                   "embed_dim d=%d unsupported (multiples of 8 up to 64, then 128, 192, 256)", s->d);
  MATCHA_CHECK_ARG(s->d % 4 == 0, "d must be a multiple of 4");
  MATCHA_CHECK_ARG(L >= 1 && L <= MATCHA_MAX_L, "L=%d outside 1..%d", L, MATCHA_MAX_L);
  MATCHA_CHECK_ARG(B >= 1 && B * (int64_t)L < (1ll << 31) - 2, "B=%lld must be >= 1 and B*L < 2^31", (long long)B);
  // the layer-by-layer kernels (every embed_dim but 64) put token tiles on grid.y (<= 65 535 tiles of 128 tokens) and launch
  // T * d / 256 workgroups on grid.x: 8.38 M token rows is what they support
  MATCHA_CHECK_ARG(s->d == 64 || B * (int64_t)L + 1 <= 65535ll * 128, "B*L=%lld exceeds the %lld token rows the layer-by-layer kernels (embed_dim != 64) launch",
                   (long long)(B * (int64_t)L), 65535ll * 128);
  MATCHA_CHECK_ARG(s->n_nodes >= 1, "n_nodes=%d must be >= 1", s->n_nodes);
  MATCHA_CHECK_ARG(s->mode == 0 || s->mode == 1, "mode=%d must be 0 (table) or 1 (adj)", s->mode);
  MATCHA_CHECK_ARG(s->n_attr >= 1 && (size_t)s->n_attr * s->d * 4 <= 160 * 1024, "n_attr=%d does not fit the LDS staging", s->n_attr);
  return MATCHA_OK;
}

static GemmArgs gemm1(const Workspace& w, const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t K, bool b_kn) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A[0] = A; g.B[0] = B; g.C[0] = C; g.batch = 1;
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = b_kn ? N : K; g.ldc = N;
  g.aux_scale = 1.f;
  g.m_dev = w.rg.count;            // true token count (Tr + 1) lives on the device
  g.rng_row_map = w.rg.tok_slot;   // dropout counters follow the original [B, L] slots
  return g;
}

// MATCHA_DEBUG_NAN=1: after selected stages count the non-finite values in the VALID rows of a buffer and print them
// (synchronises; development only -- it found reads of uninitialised workspace rows)
__global__ void nan_count_kernel(const float* __restrict__ p, const int32_t* __restrict__ rows_dev, int64_t rows, int64_t width, int* __restrict__ out) {
  if (rows_dev) rows = *rows_dev;
  int bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows * width; i += (int64_t)gridDim.x * blockDim.x)
    bad += isfinite(p[i]) ? 0 : 1;
  if (bad) atomicAdd(out, bad);
}
static void nan_check(const char* name, const float* p, const int32_t* rows_dev, int64_t rows, int64_t width, hipStream_t st) {
  if (!options().debug_nan || !p) return;
  int* d = nullptr;
  int h = 0;
  if (hipMalloc(&d, sizeof(int)) != hipSuccess) return;
  (void)hipMemsetAsync(d, 0, sizeof(int), st);
  hipLaunchKernelGGL(nan_count_kernel, dim3(256), dim3(256), 0, st, p, rows_dev, rows, width, d);
  (void)hipMemcpyAsync(&h, d, sizeof(int), hipMemcpyDeviceToHost, st);
  (void)hipStreamSynchronize(st);
  (void)hipFree(d);
  fprintf(stderr, "[matcha nan-check] %-10s %d non-finite values in the valid rows\n", name, h);
}

}  // namespace matcha

using namespace matcha;

extern "C" int matcha_abi_version(void) { return MATCHA_ABI_VERSION; }
extern "C" const char* matcha_last_error(void) { return g_err; }
extern "C" int matcha_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

extern "C" int matcha_profile_select(int32_t kernel_class) {
  g_prof_class = kernel_class;
  g_prof_used = 0;
  g_prof_work = 0.0;
  return MATCHA_OK;
}

extern "C" int matcha_launch_log(int32_t on) {
  if (on) g_launch_names = 0;
  g_launch_log = on ? 1 : 0;
  return MATCHA_OK;
}

extern "C" int matcha_launch_log_read(char* out, size_t cap) {
  MATCHA_CHECK_ARG(out && cap > 0, "matcha_launch_log_read: null buffer");
  size_t n = 0;
  out[0] = 0;
  for (int i = 0; i < g_launch_names; ++i) {
    const int k = snprintf(out + n, cap - n, "%s %lld\n", g_launches[i].name, (long long)g_launches[i].n);
    if (k < 0 || (size_t)k >= cap - n) { set_error("matcha_launch_log_read: buffer too small"); return MATCHA_ENOMEM; }
    n += (size_t)k;
  }
  return MATCHA_OK;
}

extern "C" int matcha_profile_read(double* total_ms, int64_t* launches, double* work) {
  double ms = 0.0;
  for (int i = 0; i + 1 < g_prof_used; i += 2) {
    if (hipEventSynchronize(g_prof_ev[i + 1]) != hipSuccess) { set_error("matcha_profile_read: event sync failed"); return MATCHA_EHIP; }
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_prof_ev[i], g_prof_ev[i + 1]) != hipSuccess) { set_error("matcha_profile_read: elapsed failed"); return MATCHA_EHIP; }
    ms += t;
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = g_prof_used / 2;
  if (work) *work = g_prof_work;
  g_prof_used = 0;
  g_prof_work = 0.0;
  return MATCHA_OK;
}

extern "C" size_t matcha_workspace_bytes(const matcha_shape* shp, int64_t B, int32_t L) {
  if (check_shape(shp, B, L) != MATCHA_OK) return 0;
  Workspace w;
  return carve(*shp, B, L, nullptr, w);
}

// a forward with opts->forward_only set needs no more than this (equal to matcha_workspace_bytes when the shape has no fused path)
static bool compact_forward(const matcha_shape& s, const matcha_step_opts& o) { return o.forward_only && fused_enabled(s); }
extern "C" size_t matcha_workspace_bytes_forward(const matcha_shape* shp, int64_t B, int32_t L) {
  if (check_shape(shp, B, L) != MATCHA_OK) return 0;
  Workspace w;
  return carve(*shp, B, L, nullptr, w, fused_enabled(*shp));
}

// force_layerwise: run the separate kernels whatever the shape (matcha_get_embedding needs H2 and X in HBM);
// stop_before_head: return after pff_n1 (no logits, no loss)
static int forward_impl(const matcha_shape* shp, const matcha_tensors* params, const matcha_frozen* frozen,
                        const matcha_step_opts* opts, const int64_t* x, int64_t B, int32_t L, const float* y,
                        const float* w_bce, float* logits, float* losses, void* ws, size_t ws_bytes,
                        matcha_stream_t stream, bool force_layerwise, bool stop_before_head) {
  MATCHA_TRY(check_shape(shp, B, L));
  MATCHA_CHECK_ARG(params && frozen && opts && x && ws, "matcha_forward: null pointer");
  MATCHA_CHECK_ARG(((uintptr_t)ws) % 256 == 0, "matcha_forward: workspace must be 256-byte aligned");
  const matcha_shape& s = *shp;
  const matcha_tensors& p = *params;
  hipStream_t st = (hipStream_t)stream;
  Workspace w;
  const size_t need = carve(s, B, L, (char*)ws, w, !force_layerwise && compact_forward(s, *opts));
  if (ws_bytes < need) { set_error("matcha_forward: workspace %zu < %zu bytes", ws_bytes, need); return MATCHA_ENOMEM; }
  const int64_t Tn = B * L + 1;                 // upper bound; the true count is *w.rg.count
  const int d = s.d;
  const int64_t hd = (int64_t)MATCHA_N_HEAD * d;
  const bool train = opts->training != 0;
  MATCHA_CHECK_ARG(!train || opts->seed || (opts->p_drop_adj <= 0 && opts->p_drop_fc1 <= 0 && opts->p_drop_pff <= 0),
                   "matcha_forward: training with dropout needs opts->seed");
  MATCHA_CHECK_ARG((frozen->attr_table || frozen->attr_mode == 1) && p.attr_w && p.attr_b && p.next_w && p.next_b && p.w_q && p.w_k && p.w_v && p.fc1_w &&
                       p.fc1_b && p.pff0_w && p.pff0_b && p.pff1_w && p.pff1_b && p.pff_ln_g && p.pff_ln_b && p.ln1_g && p.ln1_b &&
                       p.ln2_g && p.ln2_b && p.cls_w && p.cls_b && p.ln_q_g && p.ln_q_b && p.ln_k_g && p.ln_k_b && p.ln_v_g && p.ln_v_b,
                   "matcha_forward: a parameter pointer is null");
  const int32_t* cnt = w.rg.count;
  const int64_t* ids = w.rg.tok_id;

  // which fused kernels will run on this workspace (decided here: the plan, the front end and the saved records depend on it)
  const bool fused_path = !force_layerwise && fused_enabled(s);
  const bool lif = fused_path && loss_in_forward(s, *opts, y, w_bce);      // the tail's backward runs in the forward kernel: nothing saved
  // a training forward leaves r rows + probabilities per (half tile, head) for fused_bwdh_kernel (merged heads: fused_fwd32.hip)
  const bool keep_rimg = !opts->forward_only;
  // CSR plan: real tokens + one shared padding token; the fused kernels walk the HALF tiles (level 1: no 64-row tile list, no token -> tile
  // map -- matcha_ragged_plan still builds those for callers that ask)
  // embed_dim 128: the attention block as one kernel (enc128.hip) -- when a training forward's records fit where the layer-wise path keeps
  // Q / K / V / P / O: the records are sized per HALF TILE (at least two, 8 x 4352 floats each), so a batch of one to three rows does not
  // fit and runs layer by layer (it used to fail with MATCHA_ENOMEM)
  const bool enc = !force_layerwise && enc128_enabled(s, *opts) && (opts->forward_only || enc128_rec_floats(w.rg) <= w.enc_rec_floats);
  const int plan_level = (fused_enabled(s) || enc) ? 1 : 0;
  MATCHA_TRY(launch_ragged_plan(x, B, L, s.n_nodes, opts->status, w.rg, st, plan_level));
  // front end: node rows (K1) + attribute path (K6) + add (Modules.py:263-269)
  float* recon_out = losses ? losses + 1 : nullptr;
  const bool front = !force_layerwise && fused_enabled(s) && front_bwd_supported(s.d, s.n_attr) && fused_front_enabled();
  // table front end: the two reconstruction-loss slots are zero; loss_reduce_kernel writes them when it runs anyway
  const bool recon_zero_in_loss = s.mode == 0 && recon_out && fused_path && y && w_bce && losses;
  // adj front end at embed_dim 64: gather-GEMM, W1, attribute path and next_w in ONE kernel over the chromosome-sorted rows (adj_fused.hip)
  const bool adj_fused = front && fused_path && s.mode == 1 && adj_fused_eligible(s, *frozen);
  if (s.mode == 0) {
    MATCHA_CHECK_ARG(p.table, "matcha_forward: table mode without table");
    if (recon_out && !recon_zero_in_loss) MATCHA_TRY(zero_async(recon_out, 2 * sizeof(float), st));
  } else if (adj_fused) {
    MATCHA_TRY(adj_forward(s, p, *frozen, *opts, ids, Tn, nullptr, recon_out, w.adj_ws, w.adj_ws_bytes, st, cnt, w.rg.tok_slot,
                           opts->forward_only ? nullptr : w.x0, w.X, !opts->forward_only));
  } else {
    MATCHA_TRY(adj_forward(s, p, *frozen, *opts, ids, Tn, w.node, recon_out, w.adj_ws, w.adj_ws_bytes, st, cnt, w.rg.tok_slot));
  }
  bool prep_in_front = false;
  if (adj_fused) {
    // x0 and X are already there
  } else if (front) {
    // gather (or the adj front end's rows) + attribute path + add + next_w + tanh in one kernel (Modules.py:263-270)
    // (fused encoder behind it: the launch also builds the step's weight forms -- folded / merged / fragment-order -- in blocks of their own)
    const PrepSpec prep = {&p, w.folded, w.merged, w.frag};
    prep_in_front = fused_path;
    MATCHA_TRY(launch_front_fwd(p, ids, s.mode == 0 ? p.table : nullptr, s.mode == 0 ? nullptr : w.node, *frozen, s.n_attr, w.rg, Tn,
                                opts->forward_only ? nullptr : w.x0, w.X, st, prep_in_front ? &prep : nullptr));     // x0 (pre-activation) is only read by the backward pass
  } else {
    MATCHA_TRY(launch_embed_fwd(ids, Tn, d, s.mode == 0 ? p.table : nullptr, s.mode == 0 ? nullptr : w.node, *frozen, s.n_attr, p.attr_w,
                                p.attr_b, w.x0, st, cnt));
    // X = tanh(next_w(x0))   (Modules.py:270)
    GemmArgs g = gemm1(w, w.x0, p.next_w, w.X, Tn, d, d, false);
    g.flags = MATCHA_EPI_BIAS | MATCHA_EPI_TANH; g.bias[0] = p.next_b;
    MATCHA_TRY(launch_gemm_rm(false, g, st));
  }
  if (fused_path) {
    // everything from X to the logits in one kernel; a forward that will be differentiated saves Y, H1, H2 (768 B per
    // token); every training forward also leaves its Q/K/V tiles and attention probabilities for the fused backward (w.qkv)
    const bool save = !opts->forward_only && !lif;
    if (!prep_in_front) MATCHA_TRY(launch_prep_heads(p, w.folded, w.merged, w.frag, st));      // folded / merged / fragment-order weights of this step
    // the logits go straight to the caller's buffer when no later kernel reads them from the workspace (a differentiated forward
    // keeps them there for head_bwd): one device-to-device copy less per step / per inference call
    float* lg_out = (logits && !save) ? logits : w.logits;
    // large batches: the tail's backward inside the forward kernel stops behind its LayerNorms; pff_n1's two convolutions (dZ1, d dyn, the
    // weight gradients) are a kernel of their own right behind it (tail_bwd.hip).  Development switch fused_dbg bit 0: all of it in-kernel.
    const bool split_tail = lif && !fused_small_batch(w.rg) && (options().fused_dbg & 1) == 0;
    // the backward kernel's heads ADD their d x_hat rows (float atomics) unless the sum has to be reproducible: the rows are zeroed by a launch
    // that runs anyway -- tail_bwd64_kernel, or for small batches the loss reduction's -- and the record says so
    const bool zero_dx = lif && !opts->deterministic && !opts->sparse_table_grad && (split_tail || fused_small_batch(w.rg)) && y && w_bce && losses;
    note_forward(ws, true, true, false, split_tail, zero_dx);
    // (with the tail's backward in the forward kernel, Y and H1 are still handed over: the single-wave kernel PARKS the two rows there
    // (and the normalised H2 row in H2's place) between the tail's forward and backward halves instead of holding 96 registers per lane -- fused_fwd32_tail.hpp)
    MATCHA_TRY(launch_fused_fwd32(p, w.folded, w.frag, w.X, w.rg, B, L, y, w_bce, (save || lif) ? w.Y : nullptr, (save || lif) ? w.H1 : nullptr, (save || lif) ? w.H2 : nullptr,
                                  lg_out, w.row_loss, opts->seed, train ? opts->p_drop_fc1 : 0.f, train ? opts->p_drop_pff : 0.f, st,
                                  lif ? w.ddyn0 : nullptr, w.dXs, w.tslab, opts->alpha, keep_rimg ? w.qkv : nullptr, split_tail ? w.dH2 : nullptr));
    if (split_tail)
      MATCHA_TRY(launch_tail_bwd64(p, w.dH2, w.Y, w.H1, w.rg, opts->seed, train ? opts->p_drop_fc1 : 0.f, train ? opts->p_drop_pff : 0.f, w.ddyn0, w.tslab2, w.tslab,
                                   zero_dx ? w.dO : nullptr, st, w.row_loss, B, (y && w_bce) ? losses : nullptr, recon_zero_in_loss));      // ... zeroes the backward's d x_hat rows, reduces the loss
    const bool zero_in_loss = zero_dx && !split_tail;
    if (y && w_bce && losses && !split_tail)
      MATCHA_TRY(launch_loss_reduce(w.row_loss, B, losses, st, recon_zero_in_loss, zero_in_loss ? w.dO : nullptr, zero_in_loss ? (size_t)Tn * 64 * sizeof(float) : 0));
    if (logits && lg_out != logits && hipMemcpyAsync(logits, w.logits, B * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
      set_error("logits copy failed"); return MATCHA_EHIP;
    }
    return MATCHA_OK;
  }
  nan_check("x0", w.x0, cnt, 0, d, st);
  nan_check("X", w.X, cnt, 0, d, st);
  const bool mlw = merged_layerwise(s);
  note_forward(ws, mlw, false, enc);                 // the backward pass on this workspace must use the same formulation
  if (enc) {
    // LayerNorms, merged heads, attention, fc1 + dropout + mask: X -> Y in one kernel; a training forward leaves r rows + probabilities
    MATCHA_TRY(merged_weights(s, p, w, st));
    const bool keep = !opts->forward_only;
    if (keep && enc128_rec_floats(w.rg) > w.enc_rec_floats) { set_error("matcha_forward: the fused d = 128 records do not fit the workspace"); return MATCHA_ENOMEM; }
    MATCHA_TRY(launch_enc128_fwd(p, w.lwB, w.lwM, w.X, w.rg, B, L, w.Y, keep ? w.enc_rec : nullptr, w.enc, w.rg.tok_slot, opts->seed,
                                 (train && opts->p_drop_fc1 > 0.f) ? opts->p_drop_fc1 : 0.f, st));
  } else {
  // three LayerNorms on the same row (Modules.py:519-521), then Q/K/V projections (:527-529), one batched launch
  MATCHA_TRY(launch_ln3_fwd(w.X, Tn, d, p.ln_q_g, p.ln_q_b, p.ln_k_g, p.ln_k_b, p.ln_v_g, p.ln_v_b, w.qin, w.kin, w.vin, w.stats, st, cnt));
  if (mlw) {
    MATCHA_TRY(merged_weights(s, p, w, st));
    GemmArgs g = gemm1(w, w.qin, w.lwB, w.Q, Tn, hd, d, false);                 // r = qin B_all^T  (in Q's buffer)
    MATCHA_TRY(launch_gemm_rm(false, g, st));
  } else {
    GemmArgs g = gemm1(w, w.qin, p.w_q, w.Q, Tn, hd, d, false);
    g.A[1] = w.kin; g.B[1] = p.w_k; g.C[1] = w.K;
    g.A[2] = w.vin; g.B[2] = p.w_v; g.C[2] = w.V;
    g.batch = 3;
    MATCHA_TRY(launch_gemm_rm(false, g, st));
  }
  nan_check("qin", w.qin, cnt, 0, d, st);
  nan_check("Q", w.Q, cnt, 0, hd, st);
  nan_check("K", w.K, cnt, 0, hd, st);
  nan_check("V", w.V, cnt, 0, hd, st);
  if (mlw) MATCHA_TRY(launch_attn_fwd(w.Q, w.kin, w.vin, w.rg.row_off, B, L, d, w.O, w.P, st, true));     // O = Z = P . vin per head
  else MATCHA_TRY(launch_attn_fwd(w.Q, w.K, w.V, w.rg.row_off, B, L, d, w.O, w.P, st));
  nan_check("O", w.O, cnt + 1, 0, hd, st);
  // Y = (dropout(fc1(O))) * non_pad_mask    (Modules.py:572, :614); the mask only zeroes the shared padding token's row
  {
    GemmArgs g = gemm1(w, w.O, mlw ? w.lwM : p.fc1_w, w.Y, Tn, d, hd, false);
    g.flags = MATCHA_EPI_BIAS | MATCHA_EPI_ROWMASK; g.bias[0] = p.fc1_b; g.row_ids = ids;
    if (train && opts->p_drop_fc1 > 0.f) { g.flags |= MATCHA_EPI_DROPOUT; g.seed = opts->seed; g.stream_id = kStreamDropFc1; g.p_drop = opts->p_drop_fc1; }
    MATCHA_TRY(launch_gemm_rm(false, g, st));
  }
  }
  // pff_n1: H1 = dropout(tanh(conv0(Y)));  H2 = conv1(H1) + Y      (Modules.py:353-371)
  {
    GemmArgs g = gemm1(w, w.Y, p.pff0_w, w.H1, Tn, d, d, false);
    g.flags = MATCHA_EPI_BIAS | MATCHA_EPI_TANH; g.bias[0] = p.pff0_b;
    if (train && opts->p_drop_pff > 0.f) { g.flags |= MATCHA_EPI_DROPOUT; g.seed = opts->seed; g.stream_id = kStreamDropPff; g.p_drop = opts->p_drop_pff; }
    MATCHA_TRY(launch_gemm_rm(false, g, st));
  }
  {
    GemmArgs g = gemm1(w, w.H1, p.pff1_w, w.H2, Tn, d, d, false);
    g.flags = MATCHA_EPI_BIAS | MATCHA_EPI_RESIDUAL; g.bias[0] = p.pff1_b; g.residual = w.Y;
    MATCHA_TRY(launch_gemm_rm(false, g, st));
  }
  nan_check("Y", w.Y, cnt, 0, d, st);
  nan_check("H1", w.H1, cnt, 0, d, st);
  nan_check("H2", w.H2, cnt, 0, d, st);
  if (stop_before_head) return MATCHA_OK;
  // LayerNorms, (dynamic-static)^2, Conv1d(d->1), masked mean, weighted BCE   (Modules.py:373-374, :290-311; main.py:56)
  HeadParams hp = {p.pff_ln_g, p.pff_ln_b, p.ln1_g, p.ln1_b, p.ln2_g, p.ln2_b, p.cls_w, p.cls_b};
  MATCHA_TRY(launch_head_fwd(w.rg.row_off, w.H2, w.X, B, L, d, hp, y, w_bce, w.logits, w.row_loss, losses, st));
  if (logits && hipMemcpyAsync(logits, w.logits, B * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
    set_error("logits copy failed"); return MATCHA_EHIP;
  }
  return MATCHA_OK;
}

extern "C" int matcha_forward(const matcha_shape* shp, const matcha_tensors* params, const matcha_frozen* frozen,
                              const matcha_step_opts* opts, const int64_t* x, int64_t B, int32_t L, const float* y,
                              const float* w_bce, float* logits, float* losses, void* ws, size_t ws_bytes,
                              matcha_stream_t stream) {
  return forward_impl(shp, params, frozen, opts, x, B, L, y, w_bce, logits, losses, ws, ws_bytes, stream, false, false);
}

extern "C" int matcha_get_embedding(const matcha_shape* shp, const matcha_tensors* params, const matcha_frozen* frozen,
                                    const matcha_step_opts* opts, const int64_t* x, int64_t B, int32_t L, float* dynamic,
                                    float* static_, float* attn, float* losses, void* ws, size_t ws_bytes, matcha_stream_t stream) {
  MATCHA_CHECK_ARG(dynamic && static_ && opts, "matcha_get_embedding: null pointer");
  matcha_step_opts o = *opts;
  o.forward_only = 0;                  // full workspace layout: the layer-by-layer path keeps H2 and X in HBM
  o.loss_in_forward = 0;
  MATCHA_TRY(forward_impl(shp, params, frozen, &o, x, B, L, nullptr, nullptr, nullptr, losses, ws, ws_bytes, stream, true, true));
  Workspace w;
  carve(*shp, B, L, (char*)ws, w);
  if (attn && hipMemcpyAsync(attn, w.P, (size_t)B * MATCHA_N_HEAD * L * L * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
    set_error("matcha_get_embedding: attention copy failed"); return MATCHA_EHIP;
  }
  return launch_expand_embedding(x, B, L, shp->d, w.rg.row_off, w.H2, w.X, params->pff_ln_g, params->pff_ln_b, dynamic, static_, (hipStream_t)stream);
}

// Development aid, not part of include/matcha_hip.h: the workspace layout as text, one "name offset bytes" line per buffer, so that
// tools/debug/ws_diff.py can say WHICH buffer a differing workspace byte belongs to (round 5: which output the failing forward variant lost).
extern "C" int matcha_debug_layout(const matcha_shape* shp, int64_t B, int32_t L, char* out, size_t cap) {
  MATCHA_TRY(check_shape(shp, B, L));
  MATCHA_CHECK_ARG(out && cap > 0, "matcha_debug_layout: null buffer");
  Workspace w;
  char* const base = reinterpret_cast<char*>((uintptr_t)1 << 20);
  const size_t total = carve(*shp, B, L, base, w);
  struct { const char* name; const void* p; } f[] = {
      {"plan", w.rg.row_off}, {"x0", w.x0}, {"X", w.X}, {"qin", w.qin}, {"kin", w.kin}, {"vin", w.vin}, {"stats", w.stats}, {"Q", w.Q}, {"K", w.K}, {"V", w.V},
      {"P", w.P}, {"O_dxh", w.O}, {"Y", w.Y}, {"H1", w.H1}, {"H2", w.H2}, {"row_loss", w.row_loss}, {"logits", w.logits}, {"node", w.node}, {"dH2", w.dH2},
      {"dXs", w.dXs}, {"dZ1", w.dZ1}, {"ddyn0", w.ddyn0}, {"dZ0", w.dZ0}, {"dX0", w.dX0}, {"slab", w.slab}, {"gemm_ws", w.gemm_ws}, {"adj_ws", w.adj_ws},
      {"folded", w.folded}, {"frag", w.frag}, {"merged", w.merged}, {"lwB", w.lwB}, {"lwM", w.lwM}, {"lwdB", w.lwdB}, {"lwdM", w.lwdM}, {"enc", w.enc}, {"fb_ws", w.fb_ws},
      {"tpart", w.tpart}, {"tslab", w.tslab}, {"tslab2", w.tslab2}, {"qkv_records", w.qkv}, {"front_ws", w.front_ws}, {"tg_ws", w.tg_ws}};
  size_t n = 0;
  for (const auto& e : f) {
    if (!e.p) continue;
    const int k = snprintf(out + n, cap - n, "%s %zu\n", e.name, (size_t)((const char*)e.p - base));
    if (k < 0 || (size_t)k >= cap - n) { set_error("matcha_debug_layout: buffer too small"); return MATCHA_ENOMEM; }
    n += (size_t)k;
  }
  const int k = snprintf(out + n, cap - n, "end %zu\n", total);
  if (k < 0 || (size_t)k >= cap - n) { set_error("matcha_debug_layout: buffer too small"); return MATCHA_ENOMEM; }
  return MATCHA_OK;
}

extern "C" int matcha_random_chrom_dev_supported(const matcha_shape* shp, const matcha_frozen* frozen) {
  if (!shp || !frozen) return 0;
  if (shp->mode == 0) return 1;
  return adj_fused_eligible(*shp, *frozen) && fused_enabled(*shp) && front_bwd_supported(shp->d, shp->n_attr) ? 1 : 0;
}

extern "C" int matcha_set_option(const char* name, int32_t value) {
  MATCHA_CHECK_ARG(name, "matcha_set_option: null name");
  for (const OptionName& n : kOptionNames)
    if (strcmp(n.name, name) == 0) { options().*(n.field) = value; return MATCHA_OK; }
  set_error("matcha_set_option: unknown option '%s'", name);
  return MATCHA_EINVAL;
}

extern "C" int32_t matcha_get_option(const char* name) {
  if (!name) return -1;
  for (const OptionName& n : kOptionNames)
    if (strcmp(n.name, name) == 0) return options().*(n.field);
  return -1;
}

extern "C" int matcha_table_grad_rows(const matcha_shape* shp, int64_t B, int32_t L, void* ws, size_t ws_bytes,
                                      const int32_t** ids, const float** rows, const int32_t** n_tokens, int64_t* cap) {
  MATCHA_TRY(check_shape(shp, B, L));
  MATCHA_CHECK_ARG(ws && ids && rows && cap, "matcha_table_grad_rows: null pointer");
  MATCHA_CHECK_ARG(shp->mode == 0, "matcha_table_grad_rows: table mode only");
  Workspace w;
  const size_t need = carve(*shp, B, L, (char*)ws, w);
  if (ws_bytes < need) { set_error("matcha_table_grad_rows: workspace %zu < %zu bytes", ws_bytes, need); return MATCHA_ENOMEM; }
  *ids = w.rg.tok_key;
  *rows = w.dX0;
  if (n_tokens) *n_tokens = w.rg.count + 1;
  *cap = B * (int64_t)L + 1;
  return MATCHA_OK;
}

extern "C" int matcha_backward(const matcha_shape* shp, const matcha_tensors* params, const matcha_frozen* frozen,
                               const matcha_step_opts* opts, const int64_t* x, int64_t B, int32_t L, const float* y,
                               const float* w_bce, const float* dlogits, const float* drecon, matcha_tensors* grads,
                               int32_t* touched, void* ws, size_t ws_bytes, matcha_stream_t stream) {
  MATCHA_TRY(check_shape(shp, B, L));
  MATCHA_CHECK_ARG(params && frozen && opts && x && ws && grads, "matcha_backward: null pointer");
  MATCHA_CHECK_ARG(dlogits || (y && w_bce), "matcha_backward: need dlogits or (y, w)");
  const matcha_shape& s = *shp;
  const matcha_tensors& p = *params;
  matcha_tensors& g_ = *grads;
  hipStream_t st = (hipStream_t)stream;
  Workspace w;
  const size_t need = carve(s, B, L, (char*)ws, w);
  if (ws_bytes < need) { set_error("matcha_backward: workspace %zu < %zu bytes", ws_bytes, need); return MATCHA_ENOMEM; }
  const int64_t Tn = B * L + 1;
  const int d = s.d;
  const int64_t hd = (int64_t)MATCHA_N_HEAD * d;
  const bool train = opts->training != 0;
  const bool drop_fc1 = train && opts->p_drop_fc1 > 0.f, drop_pff = train && opts->p_drop_pff > 0.f;
  const int32_t* cnt = w.rg.count;             // the plan (row_off, tok_id, tok_slot, count) is still in the workspace
  const int64_t* ids = w.rg.tok_id;

  const int fwd_state = ws_state(ws);
  MATCHA_CHECK_ARG(fwd_state >= 0, "matcha_backward: no matcha_forward on record for this workspace (one backward per forward, same ws pointer)");
  const bool fused_fwd = (fwd_state & 2) != 0;          // which kernels the forward ran is what decides, not the option table now
  const bool lif = fused_fwd && loss_in_forward(s, *opts, y, w_bce);
  MATCHA_CHECK_ARG(!(lif && dlogits), "matcha_backward: opts->loss_in_forward excludes an explicit dlogits");
  bool dx_zeroed = false, tail_in_bwd = false;
  TailReduceArgs tail_args;
  if (lif) {
    // ddyn0 and dXs were produced by matcha_forward; only the per-half-tile parameter-gradient partials remain to be summed
    // (fused_fwd32.hip).  Small batches: one launch, which also zeroes the buffer the backward kernel's heads add their d x_hat into
    const bool small = fused_small_batch(w.rg);
    dx_zeroed = (fwd_state & 16) != 0 && !opts->deterministic && !opts->sparse_table_grad;      // tail_bwd64_kernel / the loss reduction zeroed the rows
    if ((fwd_state & 8) != 0) {
      // one row-major slab per workgroup of tail_bwd64_kernel: the convolutions' gradients and the half tiles' LayerNorm / classifier vectors
      // (summed along its walk): 512 slabs.  Summed by blocks of the launch that sums the backward kernel's slabs (fused_bwd.hip)
      tail_reduce_args(w.tslab2, w.rg, L, g_, true, tail_bwd_grid(), true, tail_args);
      tail_in_bwd = true;
    } else if (small) {
      tail_reduce_args(w.tslab, w.rg, L, g_, true, -1, false, tail_args);     // one slab per half tile (a few dozen)
      tail_in_bwd = true;
    } else
      MATCHA_TRY(launch_tail_reduce(w.tslab, w.rg, L, g_, st, true, w.tpart));
  } else {
  // tail: dH2, dXs and the gradients of pff_n1.layer_norm, layer_norm1/2, pff_classifier
  HeadParams hp = {p.pff_ln_g, p.pff_ln_b, p.ln1_g, p.ln1_b, p.ln2_g, p.ln2_b, p.cls_w, p.cls_b};
  HeadParams ghp = {g_.pff_ln_g, g_.pff_ln_b, g_.ln1_g, g_.ln1_b, g_.ln2_g, g_.ln2_b, g_.cls_w, g_.cls_b};
  MATCHA_TRY(launch_head_bwd(w.rg.row_off, w.H2, w.X, B, L, d, hp, y, w_bce, w.logits, dlogits, opts->alpha, w.dH2, w.dXs, w.slab, ghp, st));
  // pff_n1 conv1: dW1 += dH2^T H1 ; db1 += colsum(dH2) ; dZ1 = (dH2 W1) * dropmask * (1 - tanh^2)
  MATCHA_TRY(launch_gemm_tn(w.dH2, w.H1, g_.pff1_w, g_.pff1_b, d, d, Tn, d, d, nullptr, true, w.gemm_ws, w.gemm_ws_bytes, st, cnt));
  {
    GemmArgs g = gemm1(w, w.dH2, p.pff1_w, w.dZ1, Tn, d, d, true);
    g.flags = MATCHA_EPI_DTANH; g.aux = w.H1;
    if (drop_pff) { g.flags |= MATCHA_EPI_DROPOUT; g.seed = opts->seed; g.stream_id = kStreamDropPff; g.p_drop = opts->p_drop_pff; g.aux_scale = 1.f - opts->p_drop_pff; }
    MATCHA_TRY(launch_gemm_rm(true, g, st));
  }
  // pff_n1 conv0: dW0 += dZ1^T Y ; ddyn0 = (dZ1 W0 + dH2[residual]) * dropmask_fc1 * non_pad
  MATCHA_TRY(launch_gemm_tn(w.dZ1, w.Y, g_.pff0_w, g_.pff0_b, d, d, Tn, d, d, nullptr, true, w.gemm_ws, w.gemm_ws_bytes, st, cnt));
  {
    GemmArgs g = gemm1(w, w.dZ1, p.pff0_w, w.ddyn0, Tn, d, d, true);
    g.flags = MATCHA_EPI_RESIDUAL | MATCHA_EPI_ROWMASK; g.residual = w.dH2; g.row_ids = ids;
    if (drop_fc1) { g.flags |= MATCHA_EPI_DROPOUT; g.seed = opts->seed; g.stream_id = kStreamDropFc1; g.p_drop = opts->p_drop_fc1; }
    MATCHA_TRY(launch_gemm_rm(true, g, st));
  }
  }
  if (fused_fwd) {
    // attention block (fc1, attention, Q/K/V projections, the three LayerNorms) from X and ddyn0 in one head-major kernel;
    // the forward pass left the folded weights in w.folded.  w.dO doubles as the 8 per-head d x_hat slabs.
    const bool front = front_bwd_supported(s.d, s.n_attr) && fused_front_enabled();
    // the eight heads add their d x_hat into ONE buffer with float atomics (they meet in L2; per-head slabs are 8 x 256 B per token written
    // and read back).  `deterministic` and the row-sparse table gradient (whose sum is bitwise reproducible) keep the slabs and their
    // fixed summation order
    const bool dx_atomic = !opts->deterministic && !opts->sparse_table_grad;
    MATCHA_TRY(launch_fused_bwd_merged(p, w.folded, w.merged, w.X, w.ddyn0, w.dXs, w.rg, B, L, w.dO, w.fb_ws, g_, front ? nullptr : w.dZ0, st, w.qkv, dx_atomic, dx_zeroed,
                                       tail_in_bwd ? &tail_args : nullptr));
    MATCHA_TRY(encoder_done(*opts, st));
    if (front) {
      // LayerNorm backward of the summed partials + next_w + attribute_nn backward + embedding scatter in one kernel
      if (s.mode == 0) MATCHA_CHECK_ARG(g_.table, "matcha_backward: table mode without a table gradient buffer");
      const bool rows_out = s.mode == 1 || opts->deterministic || opts->sparse_table_grad;     // dX0 rows instead of float atomics
      MATCHA_TRY(launch_front_bwd(p, w.X, w.dO, dx_atomic ? 1 : MATCHA_N_HEAD, Tn, fused_bwd_dxpad(w.fb_ws), w.dXs, w.x0, ids, *frozen, s.n_attr, w.rg,
                                  rows_out ? w.dX0 : nullptr, rows_out ? nullptr : g_.table, w.front_ws, g_, st, s.mode == 0 ? touched : nullptr));
      if (s.mode == 0) {
        MATCHA_TRY(table_gradient(s, *opts, w, Tn, g_, st));       // (the two `touched` flags were set by front_slab_reduce_kernel)
      } else {
        MATCHA_TRY(adj_backward(s, p, *frozen, *opts, ids, Tn, w.dX0, drecon, g_, touched, w.adj_ws, w.adj_ws_bytes, w.gemm_ws, w.gemm_ws_bytes, st,
                                w.rg.tok_slot, adj_fused_eligible(s, *frozen)));
      }
      return MATCHA_OK;
    }
  } else {
  if ((fwd_state & 4) != 0) {
    // embed_dim 128, fused attention block: dB_all / dM_all (LayerNorm affines taken back out) + d x_hat in one kernel, then the chain rule to
    // w_qs / w_ks / w_vs / fc1 and the LayerNorm backward (enc128.hip); dxh lives in qin's buffer
    MATCHA_TRY(launch_enc128_bwd(p, w.lwB, w.lwM, w.X, w.ddyn0, w.dXs, w.rg, B, L, w.dqin, w.enc_rec, w.enc, w.lwdB, w.lwdM, g_, w.dZ0, st));
    MATCHA_TRY(merged_chain(s, p, g_, w, st));
    MATCHA_TRY(encoder_done(*opts, st));
  } else {
  if (fwd_ran_merged(ws)) {
    // merged heads: dM_all = ddyn0^T Z ; d fc1_b += colsum ; dZ = ddyn0 M_all   (the scratch gradients start from zero: the TN GEMM
    // accumulates C and the column sums alike, and fc1_b's gradient must accumulate)
    MATCHA_TRY(zero_async(w.lwdM, (size_t)hd * d * sizeof(float), st));
    MATCHA_TRY(zero_async(w.lwdB, (size_t)hd * d * sizeof(float), st));
    MATCHA_TRY(launch_gemm_tn(w.ddyn0, w.O, w.lwdM, g_.fc1_b, d, hd, Tn, d, hd, nullptr, true, w.gemm_ws, w.gemm_ws_bytes, st, cnt));
    {
      GemmArgs g = gemm1(w, w.ddyn0, w.lwM, w.dO, Tn, hd, d, true);
      MATCHA_TRY(launch_gemm_rm(true, g, st));
    }
    // attention backward on the shared key / value rows: dR in K's (unused) buffer, dK / dV per head in Q's and V's buffers
    // d kin / d vin = the sums over the heads, added inside the kernel (kin / vin themselves are read by it: separate buffers)
    MATCHA_TRY(launch_attn_bwd(w.Q, w.kin, w.vin, w.P, w.dO, w.rg.row_off, B, L, d, w.dQ, w.dK, w.dV, w.slab, st, true, w.dkin, w.dvin));
    // dB_all = dR^T qin ; dqin = dR B_all ; then the chain rule to w_qs / w_ks / w_vs / fc1
    MATCHA_TRY(launch_gemm_tn(w.dQ, w.qin, w.lwdB, nullptr, hd, d, Tn, hd, d, nullptr, true, w.gemm_ws, w.gemm_ws_bytes, st, cnt));
    {
      GemmArgs g = gemm1(w, w.dQ, w.lwB, w.dqin, Tn, d, hd, true);
      MATCHA_TRY(launch_gemm_rm(true, g, st));
    }
    MATCHA_TRY(merged_chain(s, p, g_, w, st));
  } else {
  // fc1: dW += ddyn0^T O ; db += colsum ; dO = ddyn0 Wfc1
  MATCHA_TRY(launch_gemm_tn(w.ddyn0, w.O, g_.fc1_w, g_.fc1_b, d, hd, Tn, d, hd, nullptr, true, w.gemm_ws, w.gemm_ws_bytes, st, cnt));
  {
    GemmArgs g = gemm1(w, w.ddyn0, p.fc1_w, w.dO, Tn, hd, d, true);
    MATCHA_TRY(launch_gemm_rm(true, g, st));
  }
  MATCHA_TRY(launch_attn_bwd(w.Q, w.K, w.V, w.P, w.dO, w.rg.row_off, B, L, d, w.dQ, w.dK, w.dV, w.slab, st));
  // Q/K/V projections: dW += dQ^T qin ; dqin = dQ Wq  (batched x3)
  MATCHA_TRY(launch_gemm_tn(w.dQ, w.qin, g_.w_q, nullptr, hd, d, Tn, hd, d, nullptr, true, w.gemm_ws, w.gemm_ws_bytes, st, cnt));
  MATCHA_TRY(launch_gemm_tn(w.dK, w.kin, g_.w_k, nullptr, hd, d, Tn, hd, d, nullptr, true, w.gemm_ws, w.gemm_ws_bytes, st, cnt));
  MATCHA_TRY(launch_gemm_tn(w.dV, w.vin, g_.w_v, nullptr, hd, d, Tn, hd, d, nullptr, true, w.gemm_ws, w.gemm_ws_bytes, st, cnt));
  {
    GemmArgs g = gemm1(w, w.dQ, p.w_q, w.dqin, Tn, d, hd, true);
    g.A[1] = w.dK; g.B[1] = p.w_k; g.C[1] = w.dkin;
    g.A[2] = w.dV; g.B[2] = p.w_v; g.C[2] = w.dvin;
    g.batch = 3;
    MATCHA_TRY(launch_gemm_rm(true, g, st));
  }
  }
  // LayerNorm x3 backward + static-branch gradient + tanh'
  MATCHA_TRY(launch_ln3_bwd(w.X, w.dqin, w.dkin, w.dvin, w.dXs, Tn, d, p.ln_q_g, p.ln_k_g, p.ln_v_g, w.dZ0, w.slab, g_.ln_q_g,
                            g_.ln_q_b, g_.ln_k_g, g_.ln_k_b, g_.ln_v_g, g_.ln_v_b, st, cnt));
  MATCHA_TRY(encoder_done(*opts, st));
  }
  }
  // next_w: dW += dZ0^T x0 ; db += colsum ; dX0 = dZ0 Wn
  MATCHA_TRY(launch_gemm_tn(w.dZ0, w.x0, g_.next_w, g_.next_b, d, d, Tn, d, d, nullptr, true, w.gemm_ws, w.gemm_ws_bytes, st, cnt));
  {
    GemmArgs g = gemm1(w, w.dZ0, p.next_w, w.dX0, Tn, d, d, true);
    MATCHA_TRY(launch_gemm_rm(true, g, st));
  }
  // attribute_nn: dWa += dX0^T attr_table[id] ; dba += colsum(dX0)
  MATCHA_CHECK_ARG(frozen->attr_table, "matcha_backward: the layer-by-layer attribute_nn backward gathers attr_table rows (also under attr_mode 1)");
  MATCHA_TRY(launch_gemm_tn(w.dX0, frozen->attr_table, g_.attr_w, g_.attr_b, d, s.n_attr, Tn, d, frozen->attr_ld > 0 ? frozen->attr_ld : s.n_attr, ids, true,
                            w.gemm_ws, w.gemm_ws_bytes, st, cnt));
  // node embedding
  if (s.mode == 0) {
    MATCHA_CHECK_ARG(g_.table, "matcha_backward: table mode without a table gradient buffer");
    if (opts->deterministic || opts->sparse_table_grad) MATCHA_TRY(table_gradient(s, *opts, w, Tn, g_, st));
    else MATCHA_TRY(launch_embed_scatter(ids, Tn, d, w.dX0, g_.table, st, cnt));
    if (touched) MATCHA_TRY(launch_fill_i32(touched, 2, 1, st));
  } else {
    MATCHA_TRY(adj_backward(s, p, *frozen, *opts, ids, Tn, w.dX0, drecon, g_, touched, w.adj_ws, w.adj_ws_bytes, w.gemm_ws, w.gemm_ws_bytes, st,
                            w.rg.tok_slot));
  }
  return MATCHA_OK;
}

extern "C" int matcha_node_embeddings(const matcha_shape* shp, const matcha_tensors* params, const matcha_frozen* frozen,
                                      const int64_t* ids, int64_t T, float* rows, void* ws, size_t ws_bytes,
                                      int32_t* status, matcha_stream_t stream) {
  MATCHA_CHECK_ARG(shp && params && frozen && ids && rows, "matcha_node_embeddings: null pointer");
  MATCHA_CHECK_ARG(shp->d >= 4 && shp->d <= 256 && shp->d % 4 == 0, "matcha_node_embeddings: d=%d", shp->d);
  hipStream_t st = (hipStream_t)stream;
  if (shp->mode == 0) {
    // Wrap_Embedding: plain row gather (Modules.py:33-34)
    MATCHA_CHECK_ARG(params->table, "matcha_node_embeddings: null table");
    return launch_gather_rows(ids, T, shp->d, params->table, shp->n_nodes, rows, status, st);
  }
  MATCHA_TRY(launch_check_ids(ids, T, shp->n_nodes, status, st));      // out-of-range ids land in the padding bucket (chrom_of)
  matcha_step_opts o;
  memset(&o, 0, sizeof(o));
  o.random_chrom = -1;   // no reconstruction branch
  MATCHA_CHECK_ARG(ws && ws_bytes >= adj_workspace_bytes(*shp, T), "matcha_node_embeddings: adj mode needs a workspace of matcha_workspace_bytes()");
  return adj_forward(*shp, *params, *frozen, o, ids, T, rows, nullptr, ws, ws_bytes, st, nullptr, nullptr, nullptr, nullptr, false, true);
}
