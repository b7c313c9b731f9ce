// Host-side checks of libmatcha_hip's NON-KERNEL logic under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md §5
// "sanitizers"; GPU ASan is not available on the MI355X pool, so this is where sanitizers run): every source file of the
// library is compiled for the host only (hipcc --offload-host-only -fsanitize=address,undefined), this driver is linked
// against those objects and executed by tests/test_cpu_host_sanitizers.py in the CPU container.  No kernel runs here -- launches
// fail with "no device" and the entry points must turn that into MATCHA_EHIP; what IS exercised, instrumented:
//   * argument validation and error reporting of every extern "C" entry point (null pointers, bad shapes, misaligned or
//     too-small workspaces);
//   * every workspace-sizing function over a sweep of shapes (monotone in B, 256-byte granules, no overflow);
//   * the workspace carving of model.hip / ragged.hip: every region inside the buffer, ordered, aligned -- checked by writing
//     the first and last byte of every region into an exactly-sized heap buffer (ASan traps an overrun);
//   * the option table (matcha_set_option / environment parsing);
//   * the C restatement of the ragged plan (oracle/c/ragged_plan.c), fuzzed with its invariants.
// model.hip is included as source so that its static functions (carve, check_shape) are reachable.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <random>
#include <vector>

#include "../../matcha_amd/csrc/model.hip"

extern "C" int64_t matcha_oracle_ragged_halves(const int32_t* row_off, int64_t B, const int32_t* tile_meta, int64_t n_tiles, int32_t* half_meta,
                                               int64_t halves_cap, int32_t* tok_tile);
extern "C" int64_t matcha_oracle_ragged_plan(const int64_t* x, int64_t B, int32_t L, int64_t n_nodes, int32_t* row_off, int32_t* tok_slot,
                                             int64_t* tok_id, int32_t* tok_key, int32_t* tok_pos, int32_t* count, int32_t* tile_meta,
                                             int64_t tiles_cap, int32_t* status);

static int g_fail = 0, g_checks = 0;
#define CHECK(cond)                                                              \
  do {                                                                           \
    ++g_checks;                                                                  \
    if (!(cond)) { ++g_fail; fprintf(stderr, "CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); } \
  } while (0)

static matcha_shape shape(int d, int n_attr, int n_nodes, int n_chrom, int mode, int max_bins) {
  matcha_shape s;
  s.d = d; s.n_attr = n_attr; s.n_nodes = n_nodes; s.n_chrom = n_chrom; s.mode = mode; s.max_bins = max_bins;
  return s;
}

static void touch(char* base, size_t total, const void* p, const void* next) {
  // first and last byte of the region [p, next): ASan traps if either lies outside the heap block
  if (!p) return;
  char* a = (char*)p;
  char* b = next ? (char*)next : base + total;
  CHECK(a >= base && a <= base + total && b >= a && b <= base + total);
  CHECK(((uintptr_t)(a - base)) % 256 == 0);
  if (b > a) { a[0] = 1; b[-1] = 1; }
}

static void check_carve(const matcha_shape& s, int64_t B, int L, bool compact) {
  Workspace w0;
  const size_t total = carve(s, B, L, nullptr, w0, compact);
  CHECK(total > 0 && total % 256 == 0);
  char* raw = (char*)malloc(total + 256);
  char* base = (char*)(((uintptr_t)raw + 255) / 256 * 256);          // the library requires 256-byte alignment
  const size_t slack = (size_t)(raw + total + 256 - (base + total));
  (void)slack;
  Workspace w;
  const size_t t2 = carve(s, B, L, base, w, compact);
  CHECK(t2 == total);
  // regions in carve order (null = size 0 in this layout)
  const void* order[] = {w.rg.row_off, w.rg.tok_slot, w.rg.tok_id, w.rg.count, w.rg.blk_sum, w.rg.tok_pos, w.rg.tok_key, w.rg.tile_meta,
                         w.rg.sb_tiles, w.rg.sb_cnt, w.rg.sb_first, w.x0, w.X, w.qin, w.kin, w.vin, w.stats, w.Q, w.K, w.V, w.P, w.O, w.Y, w.H1,
                         w.H2, w.row_loss, w.logits, w.node, w.dH2, w.dXs, w.dZ1, w.ddyn0, w.dZ0,
                         w.dX0, w.slab, w.gemm_ws, w.adj_ws, w.folded, w.fb_ws, w.tslab, w.qkv, w.front_ws, w.tg_ws};
  // the attention block's gradients live in the buffers of the activations they replace (model.hip::carve)
  CHECK(w.dO == w.O && w.dQ == w.K && w.dK == w.Q && w.dV == w.V && w.dqin == w.qin && w.dkin == w.kin && w.dvin == w.vin);
  const int n = (int)(sizeof(order) / sizeof(order[0]));
  const char* prev = base;
  for (int i = 0; i < n; ++i) {
    if (!order[i]) continue;
    CHECK((const char*)order[i] >= prev);                             // carve order is address order
    prev = (const char*)order[i];
    const void* next = nullptr;
    for (int j = i + 1; j < n; ++j)
      if (order[j] && order[j] != order[i]) { next = order[j]; break; }
    touch(base, total, order[i], next);
  }
  free(raw);
}

static void check_sizes_and_validation() {
  const int dims[] = {8, 16, 32, 64, 128, 192, 256};
  for (int d : dims)
    for (int mode = 0; mode < 2; ++mode)
      for (int L = 1; L <= 8; L += (L < 5 ? 1 : 3)) {
        matcha_shape s = mode ? shape(d, 24, 3067, 23, 1, 250) : shape(d, 24, 3067, 0, 0, 0);
        size_t prev = 0;
        for (int64_t B : {1ll, 2ll, 96ll, 384ll, 4097ll, 65536ll}) {
          const size_t full = matcha_workspace_bytes(&s, B, L), fwd = matcha_workspace_bytes_forward(&s, B, L);
          CHECK(full > 0 && full % 256 == 0 && fwd > 0 && fwd <= full && full >= prev);
          prev = full;
          if (B <= 4097) { check_carve(s, B, L, false); if (d == 64) check_carve(s, B, L, true); }
        }
      }
  // shapes the library must reject, with a message
  matcha_shape bad = shape(20, 24, 10, 0, 0, 0);
  CHECK(matcha_workspace_bytes(&bad, 4, 3) == 0 && strstr(matcha_last_error(), "embed_dim") != nullptr);
  matcha_shape ok = shape(64, 24, 3067, 0, 0, 0);
  CHECK(matcha_workspace_bytes(&ok, 4, 9) == 0);                       // L > 8
  CHECK(matcha_workspace_bytes(&ok, 0, 3) == 0);                       // B < 1
  CHECK(matcha_workspace_bytes(&ok, (1ll << 31), 2) == 0);             // B * L overflows int32 token indices
  CHECK(matcha_workspace_bytes(nullptr, 4, 3) == 0);
  matcha_shape wide = shape(128, 24, 3067, 0, 0, 0);
  CHECK(matcha_workspace_bytes(&wide, 65535ll * 128, 1) == 0);         // beyond what the layer-by-layer launches support
  matcha_shape nomode = shape(64, 24, 3067, 0, 2, 0);
  CHECK(matcha_workspace_bytes(&nomode, 4, 3) == 0);
  matcha_shape noattr = shape(64, 0, 3067, 0, 0, 0);
  CHECK(matcha_workspace_bytes(&noattr, 4, 3) == 0);
  // sizing helpers of the other entry points
  CHECK(matcha_hashset_bytes(0) >= 1024 * 4 && matcha_hashset_bytes(1000) >= 2 * 1000 * 4 && matcha_hashset_bytes(100000000) >= 2ull * 100000000 * 4);
  CHECK(matcha_hashset_bytes(-5) == matcha_hashset_bytes(0));
  CHECK(matcha_gemm_tn_workspace_bytes(512, 64, 100000) > 0);
  CHECK(matcha_attn_bwd_workspace_bytes(65536, 64) > 0);
  CHECK(matcha_scatter_rows_workspace_bytes(1 << 20, 256, 1000000) > 3ull * (1 << 20) * 4);
  CHECK(matcha_ragged_plan_bytes(65536, 5) > 0 && matcha_ragged_plan_bytes(65536, 9) == 0 && matcha_ragged_plan_bytes(0, 5) == 0);
  CHECK(matcha_corrcoef_workspace_bytes(2491) > 0);
  if (matcha_device_count() > 0) {        // these two size rocPRIM scratch by asking rocPRIM, which needs a device to pick its configuration
    CHECK(matcha_quantile_workspace_bytes(1000000) > 0);
    CHECK(matcha_kmer_workspace_bytes(1000000, 3, 3067) > 0);
  }
}

static void check_entry_point_errors() {
  matcha_shape s = shape(64, 24, 3067, 0, 0, 0);
  matcha_tensors p;
  memset(&p, 0, sizeof(p));
  matcha_frozen f;
  memset(&f, 0, sizeof(f));
  matcha_step_opts o;
  memset(&o, 0, sizeof(o));
  const size_t need = matcha_workspace_bytes(&s, 8, 3);
  char* raw = (char*)malloc(need + 512);
  char* ws = (char*)(((uintptr_t)raw + 255) / 256 * 256);
  int64_t x[24] = {1, 2, 3};
  float logits[8], losses[3];
  // null pointers / bad alignment / too small a workspace: rejected before anything is launched
  CHECK(matcha_forward(&s, nullptr, &f, &o, x, 8, 3, nullptr, nullptr, logits, losses, ws, need, nullptr) == MATCHA_EINVAL);
  CHECK(matcha_forward(&s, &p, &f, &o, x, 8, 3, nullptr, nullptr, logits, losses, ws + 4, need, nullptr) == MATCHA_EINVAL);
  CHECK(matcha_forward(&s, &p, &f, &o, x, 8, 3, nullptr, nullptr, logits, losses, ws, need / 2, nullptr) == MATCHA_ENOMEM);
  CHECK(strstr(matcha_last_error(), "workspace") != nullptr);
  CHECK(matcha_forward(&s, &p, &f, &o, x, 8, 3, nullptr, nullptr, logits, losses, ws, need, nullptr) == MATCHA_EINVAL);   // parameter pointers are null
  o.training = 1; o.p_drop_fc1 = 0.3f;
  CHECK(matcha_forward(&s, &p, &f, &o, x, 8, 3, nullptr, nullptr, logits, losses, ws, need, nullptr) == MATCHA_EINVAL);   // dropout without a seed
  matcha_tensors g;
  memset(&g, 0, sizeof(g));
  CHECK(matcha_backward(&s, &p, &f, &o, x, 8, 3, nullptr, nullptr, nullptr, nullptr, &g, nullptr, ws, need, nullptr) == MATCHA_EINVAL);   // neither dlogits nor (y, w)
  CHECK(matcha_backward(&s, &p, &f, &o, x, 8, 3, nullptr, nullptr, logits, nullptr, &g, nullptr, ws, need / 2, nullptr) == MATCHA_ENOMEM);
  CHECK(matcha_node_embeddings(&s, &p, &f, x, 3, logits, nullptr, 0, nullptr, nullptr) == MATCHA_EINVAL);                 // null table
  CHECK(matcha_get_embedding(&s, &p, &f, &o, x, 8, 3, nullptr, nullptr, nullptr, nullptr, ws, need, nullptr) == MATCHA_EINVAL);
  const int32_t *ids = nullptr, *nt = nullptr;
  const float* rows = nullptr;
  int64_t cap = 0;
  CHECK(matcha_table_grad_rows(&s, 8, 3, ws, need, &ids, &rows, &nt, &cap) == MATCHA_OK && cap == 25);
  CHECK((const char*)ids >= ws && (const char*)ids + 25 * 4 <= ws + need && (const char*)rows >= ws && (const char*)(rows + 25 * 64) <= ws + need);
  matcha_shape adj = shape(64, 24, 3067, 23, 1, 250);
  CHECK(matcha_table_grad_rows(&adj, 8, 3, ws, need, &ids, &rows, &nt, &cap) == MATCHA_EINVAL);
  CHECK(matcha_adamw_step(nullptr, nullptr, nullptr, nullptr, 10, nullptr, 1, nullptr, nullptr, nullptr, nullptr, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 1.0, nullptr) != MATCHA_OK);
  CHECK(matcha_hashset_build(ws, 16, x, 100, 3, nullptr) == MATCHA_EINVAL);                                               // set buffer too small
  CHECK(matcha_hashset_build(ws, need, x, 3, 9, nullptr) == MATCHA_EINVAL);
  CHECK(matcha_neg_sample(nullptr, nullptr, 0, 3, x, 8, 3, 3, 0, nullptr, 10, nullptr, 1, nullptr, x, nullptr, nullptr) == MATCHA_EINVAL);
  CHECK(matcha_scatter_rows(nullptr, nullptr, 10, 64, 100, nullptr, nullptr, 0, nullptr) == MATCHA_EINVAL);
  CHECK(matcha_scatter_rows((const int32_t*)x, logits, 10, 63, 100, logits, ws, need, nullptr) == MATCHA_EINVAL);        // d % 4
  matcha_ragged_view view;
  CHECK(matcha_ragged_plan(x, 8, 3, 100, nullptr, ws, 64, &view, nullptr) == MATCHA_ENOMEM);
  CHECK(matcha_ragged_plan(x, 8, 9, 100, nullptr, ws, need, &view, nullptr) == MATCHA_EINVAL);
  CHECK(matcha_gemm(7, logits, logits, logits, 8, 8, 8, nullptr, nullptr, nullptr, nullptr, 0, nullptr) != MATCHA_OK);
  CHECK(matcha_embed_fwd(nullptr, 8, 64, nullptr, nullptr, nullptr, 24, nullptr, nullptr, nullptr, nullptr) == MATCHA_EINVAL);
  CHECK(matcha_quantile_uniform(nullptr, 10, 1000, nullptr, nullptr, nullptr, 0, nullptr) != MATCHA_OK);
  CHECK(matcha_zscore_rows(nullptr, 4, 4, nullptr) != MATCHA_OK);
  // a valid call up to the first launch: without a device the launch fails and is REPORTED, not ignored
  if (matcha_device_count() == 0) {
    int32_t status[4] = {0, 0, 0, 0};
    const int rc = matcha_ragged_plan(x, 8, 3, 100, status, ws, need, &view, nullptr);
    CHECK(rc == MATCHA_EHIP);
    CHECK((const char*)view.row_off >= ws && (const char*)view.tile_meta < ws + need);
  }
  free(raw);
}

static void check_options() {
  CHECK(matcha_get_option("disable_fused") == 0 && matcha_get_option("no_such_option") == -1 && matcha_get_option(nullptr) == -1);
  CHECK(matcha_set_option("no_such_option", 1) == MATCHA_EINVAL && matcha_set_option(nullptr, 1) == MATCHA_EINVAL);
  CHECK(matcha_set_option("disable_merged", 1) == MATCHA_OK && matcha_get_option("disable_merged") == 1);
  CHECK(matcha_set_option("disable_merged", 0) == MATCHA_OK && matcha_get_option("disable_merged") == 0);
  // MATCHA_FUSED_DBG=3 was exported by the test before this process started: the environment is read once, at first use
  CHECK(matcha_get_option("fused_dbg") == 3);
  matcha_shape s = shape(64, 24, 3067, 0, 0, 0);
  const size_t a = matcha_workspace_bytes_forward(&s, 1024, 5);
  matcha_set_option("disable_fused", 1);
  const size_t b = matcha_workspace_bytes_forward(&s, 1024, 5);
  matcha_set_option("disable_fused", 0);
  CHECK(a < b && b == matcha_workspace_bytes(&s, 1024, 5));          // the compact layout exists only with the fused kernels
}

static void fuzz_plan_oracle() {
  std::mt19937_64 rng(7);
  for (int it = 0; it < 300; ++it) {
    const int L = 1 + (int)(rng() % 8);
    const int64_t B = 1 + (int64_t)(rng() % (it < 250 ? 300 : 20000));
    const int64_t T = B * L, N = 50;
    std::vector<int64_t> x(T);
    const int style = (int)(rng() % 4);
    for (int64_t i = 0; i < T; ++i) {
      const uint64_t r = rng();
      x[i] = (style == 0) ? (int64_t)(r % (N + 1)) : (style == 1 ? (int64_t)(1 + r % N) : (style == 2 ? ((r & 7) ? 0 : (int64_t)(1 + r % N)) : (int64_t)(r % (N + 3)) - 1));
    }
    const int64_t cap = (T + 1 + (64 - L) - 1) / (64 - L) + (T + 1 + 63 * 32 - 1) / (63 * 32) + 2;        // ragged.hip: tiles_cap
    std::vector<int32_t> row_off(B + 1), slot(T + 1), key(T + 1), pos(T + 1), meta(cap * 4);
    std::vector<int64_t> id(T + 1);
    int32_t count[3], status = 0;
    const int64_t nt = matcha_oracle_ragged_plan(x.data(), B, L, N, row_off.data(), slot.data(), id.data(), key.data(), pos.data(), count, meta.data(), cap, &status);
    CHECK(nt >= 1 && nt <= cap && count[2] == nt);
    const int64_t Tr = count[1];
    CHECK(count[0] == Tr + 1 && row_off[B] == Tr && slot[Tr] == T && id[Tr] == 0);
    int64_t real = 0;
    bool bad = false;
    for (int64_t i = 0; i < T; ++i) { real += x[i] != 0; bad |= (x[i] < 0 || x[i] > N); }
    CHECK(real == Tr && (status != 0) == bad);
    int64_t tok = 0, hy = 0;
    for (int64_t t = 0; t < nt; ++t) {
      CHECK(meta[4 * t] == tok && meta[4 * t + 2] == hy && meta[4 * t + 1] <= 63 && meta[4 * t + 3] >= 1);
      CHECK(row_off[meta[4 * t + 2]] == meta[4 * t] && row_off[meta[4 * t + 2] + meta[4 * t + 3]] == meta[4 * t] + meta[4 * t + 1]);
      tok += meta[4 * t + 1]; hy += meta[4 * t + 3];
    }
    CHECK(tok == Tr && hy == B);
    for (int64_t t = 0; t < Tr; ++t) {
      const int64_t b = slot[t] / L;
      CHECK(t >= row_off[b] && t < row_off[b + 1] && (pos[t] & 255) == t - row_off[b] && (pos[t] >> 8) == row_off[b + 1] - row_off[b]);
      CHECK(key[t] == (int32_t)id[t] && (id[t] == x[slot[t]] || (id[t] == 0 && (x[slot[t]] < 0 || x[slot[t]] > N))));
    }
    for (int64_t t = Tr; t <= T; ++t) CHECK(key[t] == 0);
    // half tiles (<= 31 tokens: one wavefront of the fused forward) + the token -> (tile, row) map
    const int64_t hcap = (T + 1 + (32 - L) - 1) / (32 - L) + (T + 1 + 63 * 32 - 1) / (63 * 32) + 2;       // ragged.hip: halves_cap
    std::vector<int32_t> half(hcap * 4), tt(T + 1, -1);
    const int64_t nh = matcha_oracle_ragged_halves(row_off.data(), B, meta.data(), nt, half.data(), hcap, tt.data());
    CHECK(nh >= nt && nh <= hcap);
    tok = 0; hy = 0;
    for (int64_t t = 0; t < nh; ++t) {
      CHECK(half[4 * t] == tok && half[4 * t + 2] == hy && half[4 * t + 1] <= 31 && half[4 * t + 3] >= 1);
      CHECK(row_off[half[4 * t + 2] + half[4 * t + 3]] == half[4 * t] + half[4 * t + 1]);
      tok += half[4 * t + 1]; hy += half[4 * t + 3];
    }
    CHECK(tok == Tr && hy == B);
    for (int64_t t = 0; t < Tr; ++t) {
      const int64_t w = tt[t] >> 6, row = tt[t] & 63;
      CHECK(tt[t] >= 0 && w < nt && meta[4 * w] + row == t && row < meta[4 * w + 1]);
    }
  }
}

int main() {
  check_options();                      // first: the option table must see the environment of the process start
  check_sizes_and_validation();
  check_entry_point_errors();
  fuzz_plan_oracle();
  if (g_fail) { fprintf(stderr, "%d of %d host checks FAILED\n", g_fail, g_checks); return 1; }
  printf("ALL HOST CHECKS PASSED (%d checks, device_count=%d)\n", g_checks, matcha_device_count());
  return 0;
}
