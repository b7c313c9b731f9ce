/*
 * matcha_hip.h -- C ABI of the MI355X (gfx950) hot path of MATCHA's hyperedge-classifier
 * training loop.  One shared library, libmatcha_hip.so; plain pointers and sizes only.
 *
 * Conventions
 *   - every buffer is a caller-owned DEVICE pointer (hipMalloc / torch .data_ptr()), contiguous,
 *     row-major; fp32 data, int64 node ids at the boundary (the reference's dtype, main.py:433);
 *   - no allocation, no ownership transfer, no hidden synchronisation: every call enqueues work
 *     on `stream` (a hipStream_t passed as void*) and returns; scratch comes from the caller via
 *     matcha_workspace_bytes();
 *   - return 0 on success, a negative errno-style code otherwise; matcha_last_error() returns a
 *     thread-local message for the last failure on the calling thread;
 *   - reentrant across streams; one process per GPU.
 *
 * The reference (ma-compbio/MATCHA, Code/) is pure Python/PyTorch and has no FFI of its own; each
 * entry point cites the reference lines whose ATen call sites it replaces.  The reference-side
 * binding (a ctypes stub for Code/Modules.py) is in INTEGRATION.md.
 */
#ifndef MATCHA_HIP_H
#define MATCHA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MATCHA_ABI_VERSION 7

#define MATCHA_OK 0
#define MATCHA_EINVAL (-22) /* bad argument (shape, alignment, null pointer) */
#define MATCHA_EHIP (-5)    /* a HIP runtime call or kernel launch failed    */
#define MATCHA_ENOMEM (-12) /* workspace too small                             */

#define MATCHA_MAX_L 8      /* widest hyperedge (BASELINE.json configs[4]: k in 2..8) */
#define MATCHA_N_HEAD 8     /* main.py:616 */

typedef void* matcha_stream_t; /* hipStream_t */

int matcha_abi_version(void);
const char* matcha_last_error(void);
/* number of visible HIP devices (0 on a CPU-only box; never fails) */
int matcha_device_count(void);

/* Optional live timing of ONE kernel class with HIP events recorded on the launch stream (used by bench.py
 * for the roofline of the dominant kernel, inside the timed region).  select(0) switches it off.  read()
 * synchronises on the recorded events and returns the summed duration [ms], the number of launches and the
 * summed algorithmic work (FLOP for the GEMM classes, bytes otherwise), then clears the accumulators.
 * Process-global, single-threaded use; do not use while a stream capture is active. */
#define MATCHA_PROF_GEMM_NT 1
#define MATCHA_PROF_GEMM_NN 2
#define MATCHA_PROF_GEMM_TN 3
#define MATCHA_PROF_ATTN_FWD 4
#define MATCHA_PROF_ATTN_BWD 5
#define MATCHA_PROF_EMBED_FWD 6
#define MATCHA_PROF_EMBED_SCATTER 7
#define MATCHA_PROF_LN3_FWD 8
#define MATCHA_PROF_LN3_BWD 9
#define MATCHA_PROF_HEAD_FWD 10
#define MATCHA_PROF_HEAD_BWD 11
#define MATCHA_PROF_ADAMW 12
#define MATCHA_PROF_NEG_SAMPLE 13
#define MATCHA_PROF_ADJ_ENCODE 14
#define MATCHA_PROF_GATHER_ROWS 15
#define MATCHA_PROF_FUSED_FWD 16
#define MATCHA_PROF_FUSED_BWD 17
#define MATCHA_PROF_FRONT_FWD 18   /* gather + attribute_nn + next_w + tanh (front_fused.hip) */
#define MATCHA_PROF_FRONT_BWD 19   /* LayerNorm backward of the d x_hat partials + next_w / attribute_nn backward + scatter */
#define MATCHA_PROF_ADJ_RECON 20   /* adj front end: reconstruction branch, forward (+ its backward in a training forward); FLOP       */
#define MATCHA_PROF_ADJ_BWD 21     /* adj front end: encoder backward (dW1, dZ, dW0); FLOP.  MATCHA_PROF_ADJ_ENCODE (14) = the fused
                                      forward (gather-GEMM, W1, attribute path, next_w); FLOP.  The adj classes count the EXPECTED work
                                      of uniformly drawn node ids: a token's feature row has sum_i n_i^2 / N columns on average         */
int matcha_profile_select(int32_t kernel_class);
int matcha_profile_read(double* total_ms, int64_t* launches, double* work);

/* Launch log (ABI 7): which kernels did a call run?  matcha_launch_log(1) clears the log and starts counting every kernel launch the
 * library makes (by the kernel's name, template instances together), matcha_launch_log(0) stops; matcha_launch_log_read writes one
 * "kernel_name count\n" line per kernel seen (MATCHA_ENOMEM if `cap` is too small).  The parity tests assert the kernel set with
 * it: the library picks kernels by batch size and embed_dim, and a test that is "about" a kernel must fail when a size rule moves
 * it onto another one.  A captured graph counts at capture time, not per replay.  Process-global, single-threaded use. */
int matcha_launch_log(int32_t on);
int matcha_launch_log_read(char* out, size_t cap);

/* A/B switches for tests and profiling.  Each option is read from the environment variable MATCHA_<NAME> (upper case) ONCE,
 * when the library is loaded, and can be changed afterwards only through matcha_set_option -- no entry point reads the
 * environment per call.  FOUR switches: "disable_fused" (1 = layer-by-layer kernels at every embed_dim, also instead of the fused
 * attention block of embed_dim 128; 2 = only the front end as separate kernels, the encoder stays fused), "disable_merged" (the reference's four products per attention head instead of the
 * merged two; they live on the layer-by-layer kernels, so at embed_dim 64 this implies disable_fused), "disable_small_batch" (the
 * large-batch forward and plan kernels at every size), "disable_wide_gemm" (embed_dim >= 128: the 64-wide GEMM / attention kernels
 * instead of the 128 x 128 ones); development: "debug_nan", "fused_dbg" (bit 0: the backward of
 * pff_n1's convolutions inside the forward kernel at every batch size, instead of tail_bwd64_kernel for large batches).  (Whether the tail's backward runs inside the forward
 * kernel is a per-call choice: matcha_step_opts.loss_in_forward.)  Returns MATCHA_EINVAL for an unknown name;
 * matcha_get_option returns -1 for one.
 * Process-global: flip them only while no call is in flight. */
int matcha_set_option(const char* name, int32_t value);
int32_t matcha_get_option(const char* name);

/* Device-side status word of the entry points that index memory by node id.  `status` is a caller-owned DEVICE int32[4],
 * zeroed by the caller; kernels OR / add into it and the caller inspects it whenever it synchronises anyway (no entry point
 * syncs).  Ids that would index out of bounds are replaced by 0 (the padding id) so that no kernel reads or writes outside
 * its buffers; the reference raises IndexError there (nn.Embedding, Modules.py:34/:67).
 *   status[0]  bit 0: a node id of x / ids was outside [0, n_nodes];  bit 1: the sampler met a node without a chromosome
 *              (node2chrom < 0 or >= n_chrom) and kept it unchanged
 *   status[1]  number of negatives whose 65 536 trials were exhausted (the row is returned equal to its positive; the
 *              reference would loop forever, main.py:392)
 *   status[2..3] reserved */
#define MATCHA_STATUS_BAD_ID 1
#define MATCHA_STATUS_BAD_CHROM 2

/* ------------------------------------------------------------------------------------------
 * Model description: shapes + one pointer per LIVE tensor of the reference's
 * Classifier.state_dict() (Modules.py:204-249).  The same struct type carries parameters
 * (`matcha_tensors` of weights) and gradients (`matcha_tensors` of grads; same shapes).
 * Dead tensors of the reference (encode2.*, fc2, pff_n2, decoder biases, node_embedding.next_w;
 * SURVEY.md headline fact 3) are never read and have no slot.
 * ------------------------------------------------------------------------------------------ */
typedef struct matcha_shape {
  int32_t d;        /* embed_dim = d_model = d_k = d_v = bottle_neck (main.py:517); multiple of 16, <= 256 */
  int32_t n_attr;   /* C+1 columns of the attribute table (main.py:497-512) */
  int32_t n_nodes;  /* N; node ids are 1..N, 0 = padding */
  int32_t n_chrom;  /* C (adj mode); may be 0 in table mode */
  int32_t mode;     /* 0 = table (Wrap_Embedding, Modules.py:29-34); 1 = adj (MultipleEmbedding :125-201) */
  int32_t max_bins; /* adj: max_i n_i */
} matcha_shape;

typedef struct matcha_tensors {
  /* front end -- table mode */
  float* table;      /* [N+1, d]       node_embedding.weight                                    */
  /* front end -- adj mode: per-chromosome tensors packed back to back, chromosome i at bounds[i] */
  float* adj_w0;     /* sum_i [d, n_i] node_embedding.Embedding_Linear{i}."tied weight_0"; offset d*bounds[i] */
  float* adj_w1;     /* [C, d, d]      node_embedding.Embedding_Linear{i}."tied weight_1"         */
  float* recon_w;    /* sum_i [n_i, d] node_embedding.Embedding_recon{i}.FF_Linear0.weight; offset d*bounds[i] */
  float* recon_b;    /* [N]            node_embedding.Embedding_recon{i}.FF_Linear0.bias; offset bounds[i]     */
  /* attribute path (Modules.py:243-249, :263-264) */
  float* attr_w;     /* [d, n_attr]    attribute_nn.weight */
  float* attr_b;     /* [d]            attribute_nn.bias   */
  float* next_w;     /* [d, d]         next_w.FF_Linear0.weight (Modules.py:270) */
  float* next_b;     /* [d] */
  /* encode1.mul_head_attn (Modules.py:463-575) */
  float* ln_q_g; float* ln_q_b;   /* layer_norm1 */
  float* ln_k_g; float* ln_k_b;   /* layer_norm2 */
  float* ln_v_g; float* ln_v_b;   /* layer_norm3 */
  float* w_q;        /* [8d, d] w_qs.weight */
  float* w_k;        /* [8d, d] w_ks.weight */
  float* w_v;        /* [8d, d] w_vs.weight */
  float* fc1_w;      /* [d, 8d] fc1.weight  */
  float* fc1_b;      /* [d] */
  /* encode1.pff_n1 (Modules.py:604-605, :353-376); Conv1d(k=1) weight [d,d,1] == [d,d] */
  float* pff0_w; float* pff0_b;
  float* pff1_w; float* pff1_b;
  float* pff_ln_g; float* pff_ln_b;
  /* Classifier tail (Modules.py:290-299) */
  float* ln1_g; float* ln1_b;     /* layer_norm1 (dynamic) */
  float* ln2_g; float* ln2_b;     /* layer_norm2 (static)  */
  float* cls_w;      /* [d]  pff_classifier.PWF_Conv0.weight [1,d,1] */
  float* cls_b;      /* [1] */
} matcha_tensors;

/* Frozen (non-trainable) inputs of the front end. */
typedef struct matcha_frozen {
  const float* attr_table;   /* [N+1, n_attr] attribute_dict_embedding.weight, row 0 zeros (main.py:508) */
  const int32_t* bounds;     /* [C+1] Modules.py:138 num_list = [0, n_0, n_0+n_1, ...] (adj; device)     */
  const float* feats;        /* adj: sum_i [n_i, n_i] feature matrices (main.py:571-577), chrom i at feat_off[i] */
  const int64_t* feat_off;   /* [C+1] element offsets into feats (device)                                 */
  const float* inter;        /* adj: [N, N] z-scored inter-chromosome matrix (Modules.py:146-154)         */
  const int32_t* bounds_host;/* [C+1] same as bounds, HOST pointer                                         */
  /* ---- ABI 6 (all zero = the layouts above) ---- */
  int32_t feat_row_pad;      /* adj: 0 = the rows of chromosome i are n_i floats, back to back; P > 0 (a multiple of 4) = every row is
                                padded with zeros to a multiple of P floats, `feats` is 16-byte aligned and every feat_off[i] a
                                multiple of 4 -- aligned 16-byte loads of feature rows; the fused embed_dim-64 kernels
                                (csrc/adj_fused.hip) require P = 64, other values run the layer-by-layer kernels               */
  int32_t attr_mode;         /* 0 = attribute rows are read from attr_table; 1 = the table has the structure main.py:497-512 builds
                                (one-hot chromosome || bin index inside the chromosome / attr_scale, row 0 zeros) and is NOT read:
                                a token's row is rebuilt from its node id, attr_bounds and attr_scale (one random row per token
                                instead of two; the caller has verified the structure bit for bit, attr_table may be NULL)     */
  int32_t attr_ld;           /* attr_mode 0: row stride of attr_table in floats (0 = n_attr); 32 = rows padded to one 128-byte
                                fetch unit                                                                                       */
  float attr_scale;          /* attr_mode 1: num[0] of main.py:503                                                               */
  const int32_t* attr_bounds;/* attr_mode 1: device int32 [n_attr] = 0, n_0, n_0+n_1, ..., N                                      */
} matcha_frozen;

/* Per-call options of the fused step. */
typedef struct matcha_step_opts {
  int32_t training;          /* 0: eval (no dropout); 1: train (dropout with the seeds below)            */
  int32_t random_chrom;      /* adj: the chromosome drawn at Modules.py:192 (caller draws it)             */
  float p_drop_adj;          /* Modules.py:174  (0.2) */
  float p_drop_fc1;          /* Modules.py:226  (0.3) */
  float p_drop_pff;          /* Modules.py:227  (0.4) */
  float alpha;               /* main.py:166 loss = bce*alpha + recon*beta */
  float beta;
  const uint64_t* seed;      /* DEVICE pointer to the 64-bit dropout seed of this step (graph-replay safe) */
  int32_t forward_only;      /* 1: matcha_backward will NOT be called on this workspace (inference / no_grad): the
                                forward may keep every intermediate on chip (fused kernel) and save nothing          */
  int32_t loss_in_forward;   /* 1 (training step with y, w given; embed_dim 64): matcha_forward also runs the backward of the
                                classifier tail for loss = alpha*bce (+ beta*recon) and keeps the result in the workspace;
                                matcha_backward must then be called with the SAME opts and dlogits == NULL.  0: the
                                backward pass starts from the saved activations (required for an arbitrary dlogits)   */
  int32_t* status;           /* optional DEVICE int32[4] status word (see above); NULL = not reported                   */
  int32_t sparse_table_grad; /* table mode, matcha_backward: 1 = do NOT add the table gradient into grads->table; leave it as a
                                per-token (node id, gradient row) list in the workspace for matcha_table_grad_rows (the
                                row-sparse data-parallel exchange); 0 = dense grads->table as usual      */
  int32_t deterministic;     /* table mode, matcha_backward: 1 = the embedding backward sorts the tokens by node id and adds each
                                node's rows in token order with ONE writer per table row (csrc/table_grad.hip: bitwise
                                reproducible, ~60 us per 65 536-row step at hg38 1 Mb sizes); 0 = float atomics from the
                                front-end backward kernel (order of additions varies from run to run, like
                                torch.nn.Embedding's CUDA backward; nothing else in the step is order-dependent)            */
  void* encoder_done_event;  /* optional hipEvent_t (NULL = off): matcha_backward records it on `stream` as soon as the gradients
                                of every parameter from ln_q_g to cls_b (the encoder, pff_n1 and the classifier tail -- the
                                contiguous tail of a flat gradient buffer laid out in matcha_tensors order) are final, i.e.
                                before the front-end backward and the embedding scatter run: a data-parallel caller starts the
                                all-reduce of that part on a second stream and overlaps it with the rest of the backward      */
  const int32_t* random_chrom_dev; /* ABI 6, optional (NULL = use random_chrom): DEVICE int32 holding the chromosome drawn at
                                Modules.py:192, read by the kernels when they run -- a step captured in a hipGraph then replays with a new
                                draw every time the caller rewrites the cell (random_chrom itself is a launch parameter and would be
                                frozen into the graph).  Honoured by the fused adj front end (embed_dim 64, feat_row_pad 64:
                                matcha_random_chrom_dev_supported); other shapes return MATCHA_EINVAL.  Valid range [0, n_chrom): any other
                                value in the cell means "no reconstruction branch" (loss 0, no gradient), like random_chrom = -1   */
} matcha_step_opts;

/* 1 if matcha_forward / matcha_backward honour opts->random_chrom_dev for this shape and these frozen inputs (table mode: always --
 * nothing reads the chromosome there). */
int matcha_random_chrom_dev_supported(const matcha_shape* shp, const matcha_frozen* frozen);

/* Scratch (bytes) the fused forward+backward needs for a [B,L] batch. */
size_t matcha_workspace_bytes(const matcha_shape* shp, int64_t B, int32_t L);
/* Scratch (bytes) that is enough for matcha_forward with opts->forward_only = 1 (inference: predict(), the pairwise sweep,
 * save_embeddings).  At embed_dim 64 only the ragged plan, two [tokens, d] buffers and the front end's buffers exist
 * (~1 KB per token instead of ~20 KB); for other shapes it equals matcha_workspace_bytes. */
size_t matcha_workspace_bytes_forward(const matcha_shape* shp, int64_t B, int32_t L);

/* Classifier.forward(x, return_recon=True) (Modules.py:278-318) + the weighted BCE of main.py:56.
 *   x      int64 [B,L], 0 = padding
 *   y, w   float [B] labels / BCE weights, may be NULL (then no loss is computed)
 *   logits float [B]  (the reference returns [B,1] logits; callers apply sigmoid themselves)
 *   losses float [3]  {bce (mean over B), recon_loss, m = rows the recon mean ran over (0 in table mode)} -- m lets
 *                     data-parallel callers weight each rank's recon gradient by m_rank / m_global
 * Activations needed by matcha_backward stay in `ws`. */
int matcha_forward(const matcha_shape* shp, const matcha_tensors* params, const matcha_frozen* frozen,
                   const matcha_step_opts* opts, const int64_t* x, int64_t B, int32_t L,
                   const float* y, const float* w, float* logits, float* losses,
                   void* ws, size_t ws_bytes, matcha_stream_t stream);

/* loss.backward() of main.py:179 for loss = bce*alpha + recon*beta, or, when `dlogits` is not NULL,
 * the vector-Jacobian product for an arbitrary upstream gradient on the logits (autograd glue).
 * Gradients are ACCUMULATED into `grads` (same layout as params; caller zeroes them -- the fused AdamW
 * does).  `touched` (device int32 [2+2C], may be NULL) receives 1 for each tensor group that received a
 * gradient this step: [0] always 1, [1] table, [2+i] adj encoder of chromosome i, [2+C+i] recon head i --
 * these are the tensors whose grad is not None in the reference (SURVEY.md §7 "AdamW semantics").
 * The call CONSUMES the activations its forward left in `ws` (gradients overwrite them): one backward per forward,
 * as loss.backward() without retain_graph. */
int matcha_backward(const matcha_shape* shp, const matcha_tensors* params, const matcha_frozen* frozen,
                    const matcha_step_opts* opts, const int64_t* x, int64_t B, int32_t L,
                    const float* y, const float* w, const float* dlogits, const float* drecon,
                    matcha_tensors* grads, int32_t* touched,
                    void* ws, size_t ws_bytes, matcha_stream_t stream);

/* model.get_node_embeddings(x) (Modules.py:252-259), eval mode: rows float [T,d] for ids int64 [T].
 * `status`: optional device status word (ids outside [0, n_nodes] are flagged and read as id 0). */
int matcha_node_embeddings(const matcha_shape* shp, const matcha_tensors* params, const matcha_frozen* frozen,
                           const int64_t* ids, int64_t T, float* rows, void* ws, size_t ws_bytes,
                           int32_t* status, matcha_stream_t stream);

/* Classifier.get_embedding(x) (Modules.py:261-276): the encoder's two outputs in the reference's padded layout,
 *   dynamic float [B,L,d] = pff_n1(...) * non_pad_mask (Modules.py:614; zero rows at padding slots)
 *   static  float [B,L,d] = tanh(next_w(node + attribute))  (:270; padding slots hold tanh(next_w(attribute_nn.bias)))
 *   attn    float [B,8,L,L], optional (may be NULL): attention probabilities in the ragged form of matcha_attn_fwd -- row i of
 *           hyperedge b and head h holds the probabilities of its k_b real keys in columns j < k_b and the probability of EACH
 *           padding slot in column k_b (all L - k_b padding keys are equal); rows i >= k_b (padding queries) are not computed
 * computed by the layer-by-layer kernels; opts as for matcha_forward (training = dropout with opts->seed).  `ws` needs
 * matcha_workspace_bytes(shp, B, L) bytes.  Not differentiable through this entry point. */
int matcha_get_embedding(const matcha_shape* shp, const matcha_tensors* params, const matcha_frozen* frozen,
                         const matcha_step_opts* opts, const int64_t* x, int64_t B, int32_t L, float* dynamic,
                         float* static_, float* attn, float* losses /* [3] as matcha_forward (bce unused), may be NULL */,
                         void* ws, size_t ws_bytes, matcha_stream_t stream);

/* Row-sparse view of the table gradient of the LAST matcha_backward on `ws` that ran with opts->sparse_table_grad = 1
 * (table mode): one (node id, gradient row) pair per token of the batch, NOT yet summed per node (at 1 M nodes a batch holds
 * few repeated ids, so the per-token list is what the data-parallel exchange ships; the receiver sums, matcha_scatter_rows).
 * Pointers into `ws` (valid until the next call on it):
 *   *ids      device int32 [cap]    node id of each list entry; 0 = unused entry (skip)
 *   *rows     device float [cap,d]  gradient row of each entry (rows of unused entries hold garbage)
 *   *n_tokens device int32 [1]      number of used entries (they are the first *n_tokens[0] ones); may be passed NULL
 *   *cap      = B*L + 1 entries (the host-known bound a fixed-size all-gather is sized for). */
int matcha_table_grad_rows(const matcha_shape* shp, int64_t B, int32_t L, void* ws, size_t ws_bytes,
                           const int32_t** ids, const float** rows, const int32_t** n_tokens, int64_t* cap);

/* Sum (ids, rows) lists into a dense gradient table deterministically: the n entries (ids int32 [n], 0 = unused entry;
 * rows float [n,d]) -- e.g. the all-gathered lists of every rank, concatenated in rank order -- are stably sorted by id, the
 * rows of equal ids are added in list order, and each sum is added to dtable[id] by ONE writer (plain stores, no atomics):
 * bitwise reproducible.  Rows of ids that do not occur are not touched.  `ws`: matcha_scatter_rows_workspace_bytes bytes,
 * 256-byte aligned.  This is also the embedding backward of matcha_backward in table mode (nn.Embedding, Modules.py:29-34). */
size_t matcha_scatter_rows_workspace_bytes(int64_t n, int32_t d, int32_t n_nodes);
int matcha_scatter_rows(const int32_t* ids, const float* rows, int64_t n, int32_t d, int32_t n_nodes, float* dtable,
                        void* ws, size_t ws_bytes, matcha_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * torch.optim.AdamW(lr=1e-3) as main.py:630/:671 builds it, fused over one flat buffer.
 * Segment s covers elements [seg_off[s], seg_off[s+1]) (device int64 [n_seg+1]) and is one tensor of the
 * reference: it has its own step count `seg_step[s]` (device int32, incremented here) and is skipped
 * entirely -- no decay, no step -- unless touched[seg_group[s]] != 0 (device int32 arrays; `touched` is what
 * matcha_backward wrote; NULL = every segment is active), which is how tensors with grad None behave.
 * Gradients are zeroed after use (opt.zero_grad of main.py:175-176).  `grad_scale` multiplies the gradient
 * first (1/world_size after an all-reduce SUM).  `seg_coef` is device scratch, 3*n_seg floats.
 * ------------------------------------------------------------------------------------------ */
int matcha_adamw_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                      const int64_t* seg_off, int32_t n_seg, const int32_t* seg_group, const int32_t* touched,
                      int32_t* seg_step, float* seg_coef, double lr, double beta1, double beta2, double eps,
                      double weight_decay, double grad_scale, matcha_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Negative sampling (main.py:361-459) against an exact device hash set of the positive hyperedges
 * (the reference uses a Bloom filter per size, utils.py:75-97; an exact set is strictly stronger).
 * ------------------------------------------------------------------------------------------ */
/* bytes for a set holding `n_edges` hyperedges */
size_t matcha_hashset_bytes(int64_t n_edges);
/* build: edges int64 [n_edges, L] zero-padded rows (k = number of non-zero entries, ascending) */
int matcha_hashset_build(void* set, size_t set_bytes, const int64_t* edges, int64_t n_edges, int32_t L,
                         matcha_stream_t stream);
/* out[i] = 1 if row i of `rows` [n,L] is in the set */
int matcha_hashset_contains(const void* set, const int64_t* edges, int32_t L_set, const int64_t* rows,
                            int64_t n, int32_t L, int32_t* out, matcha_stream_t stream);
/* generate_negative: for each positive row (int64 [P,L], zero-padded) emit `neg_num` corrupted copies into
 * neg int64 [P*neg_num, L] (row P*? layout: negatives of positive j at rows neg_num*j .. neg_num*j+neg_num-1,
 * main.py:383-428).  node2chrom int32 [n_nodes+1]; chrom_range int32 [n_chrom,2] = [start,end) ids; an EMPTY set
 * (n_set_edges == 0) reproduces the reference's phase-1 quirk: negatives == positives (main.py:589).
 * `status` (optional device int32[4]): nodes outside [1, n_nodes] or without a chromosome are kept unchanged and flagged;
 * status[1] counts the negatives whose trials were exhausted (returned equal to the positive). */
int matcha_neg_sample(const void* set, const int64_t* set_edges, int64_t n_set_edges, int32_t L_set,
                      const int64_t* pos, int64_t P, int32_t L, int32_t neg_num, int32_t min_dis,
                      const int32_t* node2chrom, int32_t n_nodes, const int32_t* chrom_range, int32_t n_chrom,
                      const uint64_t* seed, int64_t* neg, int32_t* status, matcha_stream_t stream);

/* The bookkeeping of one step of the reference's epoch loop (main.py:155-188) on the device, so that an epoch can replay one captured
 * step (matcha_amd/train.py).  `it` is a device int64 step counter.
 * matcha_step_select (main.py:160-161, Modules.py:192): x[0:P] = pos[it*P .. it*P+P) (rows of the epoch's shuffled positives,
 * int64 [n_rows, L]; indices wrap around modulo n_rows), ww[0:P] = w[it*P ..], and, when `cell` is given, cell[0] = chroms[it] (the step's reconstruction chromosome);
 * seed0 / seed1 (optional device uint64): counter-RNG seeds to advance by one (the sampler's and the dropout masks').
 * matcha_step_record (main.py:58, :185-188, :449-451): preds[it][b] = sigmoid(logits[b]), sizes[it][b] = non-zero entries of x[b]
 * (x int64 [B, L]), sums[0] += losses[0] (bce), sums[1] += losses[1] (recon), then it += 1.  preds / sizes are [n_steps, B]. */
int matcha_step_select(const int64_t* pos, const float* w, int64_t n_rows, int32_t L, const int64_t* it, int32_t P, int64_t* x,
                       float* ww, const int32_t* chroms, int64_t n_chroms, int32_t* cell, uint64_t* seed0, uint64_t* seed1,
                       matcha_stream_t stream);
int matcha_step_record(const float* logits, const float* losses, const int64_t* x, int64_t B, int32_t L, int64_t* it, int64_t n_steps,
                       float* sums, float* preds, int64_t* sizes, matcha_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Op-level entry points (the kernels behind matcha_forward/backward, exposed so that each one is
 * parity-tested on its own against the oracle).
 * ------------------------------------------------------------------------------------------ */
#define MATCHA_GEMM_NT 0 /* C[M,N] = A[M,K] . B[N,K]^T   (F.linear, Modules.py:111/:392/:481-483/:340)   */
#define MATCHA_GEMM_NN 1 /* C[M,N] = A[M,K] . B[K,N]     (input gradient of a Linear)                     */
#define MATCHA_GEMM_TN 2 /* C[M,N] = A[R,M]^T . B[R,N]   (weight gradient; reduction over R rows/tokens)  */

/* epilogue steps, applied in THIS order: */
#define MATCHA_EPI_BIAS 1        /* + bias[n]                                                         */
#define MATCHA_EPI_TANH 2        /* tanh()                                                            */
#define MATCHA_EPI_RESIDUAL 16   /* + residual[m,n]                                                   */
#define MATCHA_EPI_DROPOUT 4     /* * keep/(1-p); keep = rand_u32(key(seed,stream), m, n) >= p*2^32   */
#define MATCHA_EPI_ROWMASK 8     /* * (row_ids[m] != 0)                                               */
#define MATCHA_EPI_DTANH 32      /* * (1 - (aux[m,n]*aux_scale)^2)   aux = saved tanh output          */
#define MATCHA_EPI_ACCUM 64      /* C += result instead of C = result                                 */

typedef struct matcha_gemm_epilogue {
  int32_t flags;
  const float* bias;        /* [N] */
  const float* residual;    /* [M,N] */
  const float* aux;         /* [M,N] */
  const int64_t* row_ids;   /* [M] node ids (row mask) */
  const uint64_t* seed;     /* device */
  int32_t stream_id;        /* RNG stream (oracle/rng.py) */
  float p_drop;
  float aux_scale;          /* (1-p) when aux holds dropout(tanh(.)), else 1 */
} matcha_gemm_epilogue;

/* f32 MFMA GEMM. For TN, `ws` must hold matcha_gemm_tn_workspace_bytes(M,N,R) bytes; `colsum` (may be
 * NULL) additionally receives sum_r A[r,m] (bias gradient), accumulated like C. */
size_t matcha_gemm_tn_workspace_bytes(int64_t M, int64_t N, int64_t R);
int matcha_gemm(int32_t op, const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t K,
                const matcha_gemm_epilogue* epi, float* colsum, const int64_t* b_row_gather,
                void* ws, size_t ws_bytes, matcha_stream_t stream);

/* The ragged execution plan of a batch (csrc/ragged.hip): the real slots (x != 0, get_non_pad_mask Modules.py:12-14) of the
 * zero-padded x [B,L] (pad_sequence, main.py:435-437) compacted into a CSR token list + ONE shared padding token, and tiles of
 * whole hyperedges (<= 63 tokens) for the fused kernels.  matcha_forward builds it inside its workspace; this entry point
 * builds it alone so that it can be compared bit for bit with the C restatement oracle/c/ragged_plan.c.
 *   view->row_off  int32 [B+1]; tok_slot, tok_key, tok_pos int32 [B*L+1]; tok_id int64 [B*L+1]; count int32 [4] = {Tr+1, Tr,
 *   tiles, half tiles}; tile_meta int32 [tiles_cap][4] = {first token, tokens, first hyperedge, hyperedges}; half_meta: the same
 *   stream cut into half tiles of <= 31 tokens (one wavefront each in the fused forward) -- device pointers into `ws`. */
typedef struct matcha_ragged_view {
  const int32_t* row_off; const int32_t* tok_slot; const int64_t* tok_id; const int32_t* tok_key; const int32_t* tok_pos;
  const int32_t* count; const int32_t* tile_meta; int64_t tiles_cap;
  const int32_t* half_meta; int64_t halves_cap;   /* half tiles: whole hyperedges, <= 31 tokens, same four fields; count[3] of them */
  const int32_t* tok_tile;                        /* int32 [B*L+1]: (tile << 6) | row of each token inside tile_meta's tiling        */
} matcha_ragged_view;
size_t matcha_ragged_plan_bytes(int64_t B, int32_t L);
int matcha_ragged_plan(const int64_t* x, int64_t B, int32_t L, int32_t n_nodes, int32_t* status, void* ws, size_t ws_bytes,
                       matcha_ragged_view* view, matcha_stream_t stream);

/* K1 + K6: x0[t] = rows[t] + attr_table[x[t]] . attr_w^T + attr_b, rows[t] = table[x[t]] (table != NULL) or
 * dense[t] (Modules.py:34, :263-269).  Backward: scatter-add into dtable (row 0 skipped, padding_idx=0). */
int matcha_embed_fwd(const int64_t* x, int64_t T, int32_t d, const float* table, const float* dense,
                     const float* attr_table, int32_t n_attr, const float* attr_w, const float* attr_b,
                     float* x0, matcha_stream_t stream);
int matcha_embed_scatter_bwd(const int64_t* x, int64_t T, int32_t d, const float* dx0, float* dtable,
                             matcha_stream_t stream);

/* K8 (LayerNorm part): xhat-normalise X once, emit the three affine variants (Modules.py:519-521). */
int matcha_ln3_fwd(const float* X, int64_t T, int32_t d, const float* gq, const float* bq, const float* gk,
                   const float* bk, const float* gv, const float* bv, float* qin, float* kin, float* vin,
                   float* stats /* [T,2] mean,rstd */, matcha_stream_t stream);

/* K9: per (hyperedge, head) attention with the diagonal masked and pad slots attended (Modules.py:449-458;
 * SURVEY.md headline fact 7), on the ragged token layout: hyperedge b owns the compact token rows
 * [row_off[b], row_off[b+1]) (k_b = their number <= L real nodes), row row_off[B] is the ONE shared padding token
 * whose K/V stand for each of the L - k_b padding slots of the batch-wide [B, L] layout.
 * Q,K,V,O,dQ,dK,dV: [row_off[B] + 1, 8d]; P: [B, 8, L, L] (row i: real columns j < k_b, then the per-slot
 * padding probability in column k_b).  row_off: device int32 [B+1].  bwd also writes dK/dV (and dQ = 0) of the
 * padding token row; `ws` needs matcha_attn_bwd_workspace_bytes(B, d) bytes. */
size_t matcha_attn_bwd_workspace_bytes(int64_t B, int32_t d);
int matcha_attn_fwd(const float* Q, const float* K, const float* V, const int32_t* row_off, int64_t B, int32_t L,
                    int32_t d, float* O, float* P, matcha_stream_t stream);
int matcha_attn_bwd(const float* Q, const float* K, const float* V, const float* P, const float* dO,
                    const int32_t* row_off, int64_t B, int32_t L, int32_t d, float* dQ, float* dK, float* dV,
                    void* ws, size_t ws_bytes, matcha_stream_t stream);

/* ---- k-mer generation (Code/generate_kmers.py:8-69, one k-mer size per call) -------------------------
 * Every ascending k-subset of a cluster whose adjacent node gaps all exceed `min_dis` (generate_kmers.py:17, :24-32),
 * counted over all clusters (:34-37), kept when seen >= `min_freq` times (:40).
 *   ids, offsets   clusters in CSR form: cluster c = ids[offsets[c] .. offsets[c+1]), sorted unique node ids
 *                  (process.py:66-77), device int32 / int64 [n_clusters + 1]
 *   comb_off       device int64 [n_clusters + 1]: exclusive prefix sums of C(len_c, k) over the clusters with
 *                  k <= len_c <= max_cluster_size (<= 64; generate_kmers.py:88) and 0 for the others -- the caller has the
 *                  cluster lengths on the host; total_combos = comb_off[n_clusters] < 2^32 - 1
 *   out_kmers      int64 [cap, k] rows in lexicographic order (the reference's order depends on worker scheduling),
 *   out_freq       int64 [cap], n_out: device int64, the number of k-mers kept (if > cap only the first cap are written)
 *   ws             matcha_kmer_workspace_bytes(total_combos, k, n_nodes) bytes */
size_t matcha_kmer_workspace_bytes(int64_t total_combos, int32_t k, int32_t n_nodes);
int matcha_kmer_generate(const int32_t* ids, const int64_t* offsets, const int64_t* comb_off, int64_t n_clusters,
                         int64_t total_combos, int32_t k, int32_t n_nodes, int32_t min_dis, int32_t min_freq,
                         int64_t* out_kmers, int64_t* out_freq, int64_t cap, int64_t* n_out, void* ws, size_t ws_bytes,
                         matcha_stream_t stream);

/* ---- positive-weight preprocessing (Code/main.py:555, :653) -------------------------------------------
 * The uniform quantile transform of one float32 column of k-mer frequencies: what
 * sklearn.preprocessing.QuantileTransformer(n_quantiles, output_distribution='uniform').fit_transform computes when it fits
 * on every row (subsample=None; above 10 000 rows scikit-learn's default fits on a random subsample -- the reference's only
 * non-determinism here, deliberately not reproduced).  Bit-identical to oracle/positives.py.
 *   freq           device float32 [n], 1 <= n < 2^31, no NaNs (frequencies are counts)
 *   n_quantiles    1 .. 4096 (the reference uses 1000); clamped to n as scikit-learn does
 *   out            device float32 [n] in [0, 1]; must not alias freq
 *   quantiles_out  optional device float64 [min(n_quantiles, n)]: the fitted landmarks (QuantileTransformer.quantiles_)
 *   ws             matcha_quantile_workspace_bytes(n) bytes */
size_t matcha_quantile_workspace_bytes(int64_t n);
int matcha_quantile_uniform(const float* freq, int64_t n, int32_t n_quantiles, float* out, double* quantiles_out, void* ws,
                            size_t ws_bytes, matcha_stream_t stream);

/* ---- feature construction from contact maps ----------------------------------------------------------
 * matcha_pixels_to_adj   Code/process.py:144-172: cooler pixels -> adjacency.  For every pixel i whose two bins map to nodes
 *   (index2node[bin] >= 1; 0 = chromosome outside chrom_list) and whose count is not NaN: M[n1-1][n2-1] += count and
 *   M[n2-1][n1-1] += count, M = intra when node2chrom[n1] == node2chrom[n2], else inter.  Adds into the caller's matrices
 *   (zero them first; call repeatedly to stream a large pixel table).
 *     bin1, bin2 device int64 [n_pixels]; count device float64 [n_pixels]; index2node device int32 [n_index];
 *     node2chrom device int32 [n_nodes + 1] (entry 0 unused); intra, inter device float64 [n_nodes, n_nodes]
 * matcha_corrcoef_block  Code/main.py:571-575: np.corrcoef of one chromosome's intra block, NaN -> 0.
 *     adj device float32, the block's top-left element, row stride ld (elements); out device float32 [n, n];
 *     ws matcha_corrcoef_workspace_bytes(n) bytes.  float64 arithmetic; differs from numpy only by summation order.
 * matcha_zscore_rows     Code/Modules.py:146-152: every row of a float32 [rows, cols] matrix, in place: strictly positive
 *     entries -> (x - mean) / std (ddof 0) over those entries, then NaN -> 0. */
int matcha_pixels_to_adj(const int64_t* bin1, const int64_t* bin2, const double* count, int64_t n_pixels,
                         const int32_t* index2node, int64_t n_index, const int32_t* node2chrom, int32_t n_nodes,
                         double* intra, double* inter, matcha_stream_t stream);
size_t matcha_corrcoef_workspace_bytes(int32_t n);
int matcha_corrcoef_block(const float* adj, int64_t ld, int32_t n, float* out, void* ws, size_t ws_bytes,
                          matcha_stream_t stream);
int matcha_zscore_rows(float* matrix, int64_t rows, int64_t cols, matcha_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MATCHA_HIP_H */
