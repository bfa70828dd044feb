/*
 * newsrec_hip.h — C ABI of libnewsrec_hip.so: the MI355X (gfx950) kernels behind the two-tower
 * hot path of namespace-Pt/News-Recommendation.
 *
 * The reference has no FFI layer: its boundary is Python (torch modules).  Each entry point
 * below replaces the torch-op sequence at the cited reference lines (paths relative to the
 * reference's src/).  The reference-side binding is the ctypes stub shown in INTEGRATION.md;
 * the product's own binding is news-recommendation_amd/newsrec_hip/_lib.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name says host; fp32 = float, ids/masks/labels = int64
 *     (the dtypes the reference's batches carry, utils/dataset.py:95-103,134-150)
 *   - all matrices are dense row-major; outputs and workspaces are caller-allocated, nothing is
 *     allocated, freed or retained by the library (safe under the caller's caching allocator)
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); every call is
 *     asynchronous on it; the library keeps no global mutable state except the opt-in profiler
 *   - return value: 0 = NR_OK, otherwise an NR_ERR_* code; nr_last_error() gives the message
 *     (thread-local).  Kernels never run on a non-zero return.
 *   - dropout: keep-mask stream is philox4x32-10(key=seed, counter=(element_index>>2, site)),
 *     element kept iff u32[element_index & 3] >= p * 2^32, kept values scaled by 1/(1-p)
 *     (torch's own Philox stream depends on its launch geometry and cannot be reproduced; parity
 *     tests inject this mask into the oracle).  p = 0 selects eval mode.
 */
#ifndef NEWSREC_HIP_H
#define NEWSREC_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NR_OK 0
#define NR_ERR_ARG 1          /* bad pointer / size */
#define NR_ERR_UNSUPPORTED 2  /* shape outside what the kernels are built for */
#define NR_ERR_LAUNCH 3       /* HIP reported a launch error */
#define NR_ERR_WORKSPACE 4    /* workspace too small */

int nr_version(void);
const char* nr_last_error(void);

/* ---- profiling (bench.py roofline leg; no counterpart in the reference) --------------------
 * When enabled, every launch of a kernel family is bracketed by HIP events on its stream.
 * nr_prof_read synchronises those events, returns the family's summed device time (ms), launch
 * count and summed algorithmic work (flops for family 0, bytes otherwise) and clears them. */
enum { NR_FAM_GEMM = 0, NR_FAM_GATHER = 1, NR_FAM_ATTN = 2, NR_FAM_LN = 3, NR_FAM_POOL = 4, NR_FAM_SCORE = 5, NR_FAM_MISC = 6, NR_FAM_SCATTER = 7,
       NR_FAM_GEMM_SPLIT = 8 /* the split-operand GEMM launches of mode 2; NR_FAM_GEMM then holds the rest */,
       NR_FAM_HBM_SMALL = 9 /* launches of families 1, 3, 4, 5 that move under 16 MB (launch-bound: booked apart) */, NR_FAM_COUNT = 10 };
int nr_prof_enable(int on);
int nr_prof_read(int family, double* ms, int64_t* launches, double* work);

/* ---- matrix-core precision of the GEMM family -------------------------------------------
 * mode 2 (default): fp32 results on the bf16 matrix pipe.  Each fp32 operand is split on its way into LDS into three
 * bf16 pieces (a = a0 + a1 + a2, all 24 significand bits) and a*b is accumulated in fp32 from the six piece products of
 * weight >= 2^-16; the dropped ones are below fp32's own rounding of the product, so the result has fp32 accuracy
 * (the 1e-5 parity bar, same tests; tests/test_gpu_gemm_fp64.py holds it against float64 next to mode 0 on operands whose
 * rows and columns span 1e-6 .. 1e4) at 1.3-1.6x the rate of the fp32 instruction.  Products too small to fill the chip
 * with the kernel's 16 384-element tiles run as in mode 0.
 * PRECONDITION of mode 2: finite operands with |a| <= 3.38e38 (the largest finite bf16).  a = +-inf, or an |a| that
 * rounds up to inf in bf16, makes the residual piece inf - inf = NaN and the whole dot product NaN where mode 0 would
 * give +-inf or a finite value; NaN operands give NaN in both modes.  Operands below about 1e-33 in magnitude lose their
 * second / third piece to bf16's exponent range (absolute error <= 2^-8 |a| |b| for those terms, i.e. <= 4e-36 |b|):
 * invisible unless the other operand is astronomically large.  Use mode 0 for data outside these ranges.
 * mode 0: every product on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32, an exact fp32 fma chain).
 * mode 1: operands rounded to bf16 on their way into
 * LDS, f32 accumulate (v_mfma_f32_32x32x16_bf16) for every product with M, N, K >= 128 (conv views included); tensors in HBM
 * stay fp32.  This is the "bf16-MFMA build" of SURVEY §8 a7 (AllBertNewsEncoder, encoder.py:62-77; parity bar
 * 1e-2); the closest thing in the reference is running the model under torch.autocast.  Process-wide and sticky:
 * set it before the first step.  nr_set_gemm_compute returns NR_ERR_ARG for an unknown mode. */
int nr_set_gemm_compute(int mode);
/* (no counterpart in the reference) nr_tfm_layer_bwd_f32 / nr_tfm_layer_packed_bwd_f32 run their four weight-gradient products
 * on a second stream of the same GPU when the layer has >= 1024 rows (forked and joined by events inside the call: the caller
 * sees the same ordering on its own stream).  nr_set_side_stream(0) keeps everything on the caller's stream, e.g. to time
 * kernels one at a time; returns the previous setting.  NR_SIDE_STREAM=0 in the environment sets the initial value. */
int nr_set_side_stream(int on);
int nr_get_gemm_compute(void);

/* ---- (i) token-embedding gather — nn.Embedding lookup, models/modules/encoder.py:52,135 ------
 * out[r,:] = table[ids[r],:] (bit-exact).  *oob_flag (device int, may be NULL) is set to 1 if any
 * id is outside [0,V) (such rows read row 0 instead of faulting).
 * Also used for news_embeddings[his_idx] / [cdd_idx], models/BaseModel.py:214-215.
 * With oob_flag == NULL the library's own per-GPU index flag is raised instead; the *_from_table forms of
 * nr_query_pool_fwd_f32 / nr_score_fwd_f32 / nr_seq_reduce_f32 (idx != NULL) raise it too when they clamp an index.
 * nr_index_status copies that flag to *out_flag (0 / 1), clears it and synchronises the stream: call it where the
 * caller syncs anyway.  The reference's nn.Embedding / tensor indexing raise IndexError at this point
 * (encoder.py:52,135; BaseModel.py:214-215). */
int nr_index_status(int* out_flag, void* stream);
int nr_gather_rows_f32(const float* table, int64_t V, int64_t E, const int64_t* ids, int64_t n, float* out, int* oob_flag, void* stream);
/* embedding backward (autograd of encoder.py:52): grad_table[ids[r],:] += grad_rows[r,:], rows with
 * ids[r] == padding_idx skipped (BERT word_embeddings has padding_idx = 0).  grad_table is NOT zeroed.  Rows of a 128-row
 * window that share an id are summed in LDS first ([CLS] / [SEP] sit in every title), so E <= 1920 (NR_ERR_UNSUPPORTED above);
 * the order of the floating-point additions is not fixed from run to run (atomics), as in torch's index_add. */
int nr_scatter_add_rows_f32(const float* grad_rows, const int64_t* ids, int64_t n, int64_t E, int64_t V, int64_t padding_idx, float* grad_table, void* stream);

/* ---- dense layers on f32 MFMA — nn.Linear / nn.Conv1d call sites --------------------------------
 * act: 0 none, 1 relu, 2 gelu(erf), 3 tanh.
 * y[R,N] = act(x[R,K] . w[N,K]^T + b[N]); for act == 2, z (pre-activation, may be NULL) is also stored.
 * Replaces attention.py:82-84,110,142,155 and encoder.py:57,141,326 (newsProject/userProject). */
int nr_linear_fwd_f32(const float* x, const float* w, const float* b, int64_t R, int64_t K, int64_t N, int act, float* y, float* z, void* stream);
/* dx[R,K] = dy[R,N] . w[N,K], then   mode 0: as is | 1: + aux[R,K] | 2: * gelu'(aux[R,K]) |
 * 3: (+ aux) masked by aux2 > 0 | 4: masked by aux2 > 0. */
int nr_linear_bwd_input_f32(const float* dy, const float* w, int64_t R, int64_t N, int64_t K, int mode, const float* aux, const float* aux2, float* dx, void* stream);
/* dw[N,K] = dy[R,N]^T . x[R,K] ; db[N] = column sums of dy (db may be NULL).  ws: scratch of at
 * least nr_linear_bwd_weight_ws_bytes(R,N,K) bytes. */
size_t nr_linear_bwd_weight_ws_bytes(int64_t R, int64_t N, int64_t K);
int nr_linear_bwd_weight_f32(const float* dy, const float* x, int64_t R, int64_t N, int64_t K, float* dw, float* db, void* ws, size_t ws_bytes, void* stream);

/* Conv1d(E->D, kernel_size=3, padding=1) + bias + ReLU over titles, encoder.py:32-38,55-56, as an
 * implicit GEMM over x[M*T, E] (no im2col buffer).  w is the torch layout [D,E,3];
 * wpack: scratch of 2*D*3E floats filled by nr_conv1d_k3_pack (re-run after every weight update). */
int nr_conv1d_k3_pack(const float* w, int64_t D, int64_t E, float* wpack, void* stream);
int nr_conv1d_k3_relu_fwd_f32(const float* x, const float* wpack, const float* b, int64_t M, int64_t T, int64_t E, int64_t D, float* y, void* stream);
/* dy must already be masked by relu (y > 0).  dx (may be NULL) [M*T,E]; dw [D,E,3]; db [D]. */
size_t nr_conv1d_k3_bwd_ws_bytes(int64_t M, int64_t T, int64_t E, int64_t D);
int nr_conv1d_k3_bwd_f32(const float* dy, const float* x, const float* wpack, int64_t M, int64_t T, int64_t E, int64_t D, float* dx, float* dw, float* db, void* ws, size_t ws_bytes, void* stream);

/* ---- residual + dropout + LayerNorm — TFMSelfOutput / TFMOutput, attention.py:109-113,154-158 ----
 * y = LN(drop(h) + x; eps) * gamma + beta.  s_out (may be NULL, may alias h) receives drop(h)+x,
 * mean/rstd [R] (may be NULL) the row statistics needed by the backward.
 * x_period = 0: x is [R,D]; x_period = T > 0: x is [T,D] and row r adds x[r % T] — BERT's position (+ token
 * type) embeddings added to the gathered word rows before embeddings.LayerNorm (HF BertEmbeddings, reached
 * from encoder.py:74). */
int nr_add_layernorm_fwd_f32(const float* h, const float* x, int64_t x_period, const float* gamma, const float* beta, int64_t R, int64_t D, float eps, float p_drop, uint64_t seed, uint32_t site, float* y, float* s_out, float* mean, float* rstd, void* stream);
/* ds = dLN/ds (also the residual gradient); dh (may be NULL) = dropout-masked ds; dgamma/dbeta [D].
 * ws: nr_layernorm_bwd_ws_bytes(R,D) bytes. */
size_t nr_layernorm_bwd_ws_bytes(int64_t R, int64_t D);
int nr_add_layernorm_bwd_f32(const float* dy, const float* s, const float* mean, const float* rstd, const float* gamma, int64_t R, int64_t D, float p_drop, uint64_t seed, uint32_t site, float* ds, float* dh, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream);

/* The same backward for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.  It takes no workspace: the per-block dgamma/dbeta
 * partials live in a scratch array built into the library (nr_noalloc_scratch_bytes() = 8 MiB + 64 KiB per GPU, held
 * until the process ends), which covers every R at D <= 1024; larger problems return NR_ERR_WORKSPACE.  It calls no
 * hipMalloc / hipFree.  All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams,
 * two threads) race — serialise them or use nr_add_layernorm_bwd_f32 with separate workspaces.  Results are
 * bit-identical to nr_add_layernorm_bwd_f32 (same kernels, same reduction order). */
size_t nr_noalloc_scratch_bytes(void);
int nr_add_layernorm_bwd_noalloc_f32(const float* dy, const float* s, const float* mean, const float* rstd, const float* gamma, int64_t R, int64_t D, float p_drop, uint64_t seed, uint32_t site, float* ds, float* dh, float* dgamma, float* dbeta, void* stream);

/* ---- multi-head self-attention core — TFMSelfAttention, attention.py:87-97 ---------------------
 * qkv [N*S, 3*H*dh] (q | k | v blocks per row), mask [N,S] (1 = real token), ctx [N*S, H*dh].
 * row_max/row_sum [N,H,S] (may be NULL in eval) are the softmax statistics the backward reuses.
 * mask_mode 0: masked keys get an additive -1e5 in fp32 before the softmax (attention.py:44-49,90);
 * mask_mode 1: masked keys are excluded and a sequence without any valid key gets zero probabilities — what
 *   HF BertModel computes under torch SDPA for AllBertNewsEncoder (encoder.py:74; transformers 5.15 / torch 2.10). */
int nr_mhsa_fwd_f32(const float* qkv, const int64_t* mask, int64_t N, int64_t S, int64_t H, int64_t dh, float p_drop, uint64_t seed, uint32_t site, int mask_mode, float* ctx, float* row_max, float* row_sum, void* stream);
int nr_mhsa_bwd_f32(const float* qkv, const int64_t* mask, const float* ctx, const float* dctx, const float* row_max, const float* row_sum, int64_t N, int64_t S, int64_t H, int64_t dh, float p_drop, uint64_t seed, uint32_t site, int mask_mode, float* dqkv, void* stream);
/* The same core on bf16 tensors (raw 16-bit storage; S <= 32, head dim <= 64): what the bf16 build of AllBertNewsEncoder's
 * layers calls between its bf16 GEMMs (encoder.py:62-77; parity bar 1e-2).  Head dims that are multiples of 8 run all seven
 * products on v_mfma_f32_32x32x16_bf16 (probabilities and score gradients rounded to bf16 as operands, softmax in f32);
 * other head dims take f32 arithmetic on the bf16 data. */
int nr_mhsa_fwd_bf16(const void* qkv, const int64_t* mask, int64_t N, int64_t S, int64_t H, int64_t dh, float p_drop, uint64_t seed, uint32_t site, int mask_mode, void* ctx, float* row_max, float* row_sum, void* stream);
int nr_mhsa_bwd_bf16(const void* qkv, const int64_t* mask, const void* dctx, const float* row_max, const float* row_sum, int64_t N, int64_t S, int64_t H, int64_t dh, float p_drop, uint64_t seed, uint32_t site, int mask_mode, void* dqkv, void* stream);

/* ---- additive ("query") pooling — scaled_dp_attention, attention.py:6-34 -----------------------
 * out[n,:] = softmax_s(query . key[n,s,:] / sqrt(D) + (1-mask[n,s]) * -1e5) . value[n,s,:]
 * key/value [nseq,S,D]; if idx != NULL, key and value rows are table[idx[n,s],:] instead (eval:
 * AttnUserEncoder over cached news vectors).  mask may be NULL (encoder.py:298).  prob [nseq,S] may be NULL. */
int nr_query_pool_fwd_f32(const float* key, const float* value, const float* table, const int64_t* idx, int64_t table_rows, const float* query, const int64_t* mask, int64_t nseq, int64_t S, int64_t D, float* out, float* prob, void* stream);
/* tanh_key: key = tanh(pre) and dkeypre is the gradient w.r.t. pre (encoder.py:57,141,326);
 * key_is_value: key = value = same tensor (encoder.py:298) -> dvalue carries both paths.
 * dquery [D]; ws: nr_query_pool_bwd_ws_bytes(nseq,D). */
size_t nr_query_pool_bwd_ws_bytes(int64_t nseq, int64_t D);
int nr_query_pool_bwd_f32(const float* key, const float* value, const float* query, const float* prob, const float* dout, int64_t nseq, int64_t S, int64_t D, int tanh_key, int key_is_value, float* dkeypre, float* dvalue, float* dquery, void* ws, size_t ws_bytes, void* stream);

/* Sum / Avg user encoders, encoder.py:253,274 (mask ignored).  x [nseq,S,D] or table rows by idx. */
int nr_seq_reduce_f32(const float* x, const float* table, const int64_t* idx, int64_t table_rows, int64_t nseq, int64_t S, int64_t D, int mean, float* out, void* stream);
int nr_seq_reduce_bwd_f32(const float* dout, int64_t nseq, int64_t S, int64_t D, int mean, float* dx, void* stream);

/* ---- candidate scoring (+ softmax cross-entropy) — _compute_logits + CrossEntropyLoss,
 *      models/BaseModel.py:164-175,198-205; with idx: the cached `infer` path, BaseModel.py:212-218 ----
 * logits[b,c] = cdd[b,c,:] . user[b,:] / sqrt(D); cdd rows are table[idx[b,c],:] when idx != NULL.
 * If loss != NULL: loss[0] = mean_b(-log_softmax(logits[b])[label[b]]); loss_rows [B] scratch. */
int nr_score_fwd_f32(const float* cdd, const float* table, const int64_t* idx, int64_t table_rows, const float* user, const int64_t* label, int64_t B, int64_t C, int64_t D, float* logits, float* loss_rows, float* loss, void* stream);
/* gradients of the mean CE loss, scaled by the device scalar grad_out[0]. */
int nr_score_ce_bwd_f32(const float* cdd, const float* user, const int64_t* label, const float* logits, const float* grad_out, int64_t B, int64_t C, int64_t D, float* d_cdd, float* d_user, void* stream);

/* backward of the plain logits (models/BaseModel.py:174) for callers that apply their own loss:
 * d_cdd[b,c,:] = dlogits[b,c]/sqrt(D) * user[b,:] ; d_user[b,:] = sum_c dlogits[b,c]/sqrt(D) * cdd[b,c,:]. */
int nr_score_bwd_f32(const float* cdd, const float* user, const float* dlogits, int64_t B, int64_t C, int64_t D, float* d_cdd, float* d_user, void* stream);

/* ReLU backward from the saved output (self.Relu in encoder.py:56): out = y > 0 ? dy : 0. */
int nr_relu_bwd_f32(const float* dy, const float* y, int64_t n, float* out, void* stream);

/* erf-GELU backward from the saved pre-activation (TFMIntermediate used on its own, attention.py:135-144):
 * out = dy * gelu'(z).  The fused layer folds this into the FFN2 dX GEMM's epilogue instead. */
int nr_gelu_bwd_f32(const float* dy, const float* z, int64_t n, float* out, void* stream);

/* ---- one post-LN transformer layer — TFMLayer, attention.py:161-182 -----------------------------
 * Fused driver over the kernels above (one C call per layer keeps the host off the critical path).
 * Weights use the reference's tensors unchanged ([out,in] row-major). */
typedef struct {
    const float *wq, *bq, *wk, *bk, *wv, *bv;        /* attention.self.{query,key,value} */
    const float *wo, *bo, *ln1_g, *ln1_b;            /* attention.output.{dense,LayerNorm} */
    const float *w1, *b1;                            /* intermediate.dense [4D,D] */
    const float *w2, *b2, *ln2_g, *ln2_b;            /* output.{dense,LayerNorm}  [D,4D] */
} nr_tfm_weights;
typedef struct {
    float *dwq, *dbq, *dwk, *dbk, *dwv, *dbv, *dwo, *dbo, *dln1_g, *dln1_b, *dw1, *db1, *dw2, *db2, *dln2_g, *dln2_b;
} nr_tfm_grads;
/* activations kept for the backward (all caller-allocated; R = N*S rows) */
typedef struct {
    float *wqkv;      /* [3D, D] packed q|k|v weight   */
    float *bqkv;      /* [3D]                          */
    float *qkv;       /* [R, 3D]                       */
    float *ctx;       /* [R, D]                        */
    float *s1;        /* [R, D]  drop(dense(ctx)) + x  */
    float *y1;        /* [R, D]  attention output      */
    float *z;         /* [R, 4D] pre-GELU              */
    float *a;         /* [R, 4D] GELU output           */
    float *s2;        /* [R, D]  drop(dense(a)) + y1   */
    float *stats;     /* [4, R]  mean1, rstd1, mean2, rstd2 */
    float *row_max;   /* [N, H, S]                     */
    float *row_sum;   /* [N, H, S]                     */
} nr_tfm_saved;
/* y [R,D] = TFMLayer(x [N,S,D], mask [N,S]).  p_drop = 0 -> eval.  Dropout sites: site_base+0
 * (attention probs, attention.py:92), +1 (attention.py:111), +2 (attention.py:156).  mask_mode as in
 * nr_mhsa_fwd_f32: 0 for the reference's TFMLayer, 1 for the layers of HF BertModel (encoder.py:65,74),
 * which are this same layer (attention.py copies BERT's structure and parameter names). */
int nr_tfm_layer_fwd_f32(const float* x, const int64_t* mask, int64_t N, int64_t S, int64_t D, int64_t H, const nr_tfm_weights* w, float p_drop, uint64_t seed, uint32_t site_base, int mask_mode, nr_tfm_saved* saved, float* y, void* stream);
/* dx [R,D] and all parameter gradients (overwritten, not accumulated).
 * ws: nr_tfm_layer_bwd_ws_bytes(N,S,D) bytes of scratch. */
size_t nr_tfm_layer_bwd_ws_bytes(int64_t N, int64_t S, int64_t D);
int nr_tfm_layer_bwd_f32(const float* dy, const float* x, const int64_t* mask, int64_t N, int64_t S, int64_t D, int64_t H, const nr_tfm_weights* w, float p_drop, uint64_t seed, uint32_t site_base, int mask_mode, const nr_tfm_saved* saved, nr_tfm_grads* grads, float* dx, void* ws, size_t ws_bytes, void* stream);

/* ---- packed title tokens: the news tower over real token rows only ------------------------------------
 * The reference pads every title to T tokens (utils/dataset.py:72-100) and runs its encoders over all of them
 * (models/modules/encoder.py:130-143); at MIND-like title lengths about half the rows of a step are padding.  In the
 * reference's own arithmetic a padded position of a title with >= 1 real token contributes exact zeros everywhere that
 * is kept (masked key: exp(x - 1e5 - max) == 0.0f, attention.py:24,49,90; masked query: dropped by the pooling softmax
 * the same way; gradients carry those zeros), so running the layer on the real rows only gives the same news vectors
 * and parameter gradients.  Titles with NO real token (news 0, the padding of candidate lists and histories) keep all
 * T rows: with every key masked the shifted scores still rank.  Dropout keep-bits stay indexed by the ORIGINAL
 * position (row_map), so train-mode results equal the padded layout's.
 * nr_pack_plan: offs [N+1] int32 (exclusive scan of len'[n] = len[n] ? len[n] : T), info[0] = total rows R,
 *   info[1] = 1 iff every mask row is prefix-shaped (mask[n][j] == (j < len[n])) — the caller must use the padded
 *   entry points when it is 0.  title_keep (may be NULL) [N] 0/1: titles with 0 get NO rows (len' = 0) — for history
 *   slots the user encoder masks out (his_mask == 0 under TfmUserEncoder / RnnUserEncoder, encoder.py:216-233,314-327):
 *   their news vectors reach nothing that is kept and receive exactly zero gradient; pooling returns a zero vector
 *   for them.  nr_pack_ids: ids_packed [R], row_map [R] (packed row -> n*T + j).
 * The *_packed_* entry points take x / y / saved activations with R rows (allocate nr_tfm_saved for R rows, stats
 * [4,R]; row_max/row_sum stay [N,H,S]) and need S <= 32, head dim <= 64; workspace: nr_tfm_layer_bwd_ws_bytes(R, 1, D).
 * nr_tfm_layer_packed_fwd_f32 alone also takes 32 < S <= 64 (round 3: the user tower of cached evaluation over the clicked
 * history slots only, BaseModel.py:208-218 / encoder.py:314-327; the packed backward refuses it with NR_ERR_UNSUPPORTED). */
int nr_pack_plan(const int64_t* mask, const int64_t* title_keep, int64_t N, int64_t T, int32_t* offs, int32_t* info, void* stream);
int nr_pack_ids(const int64_t* ids, const int32_t* offs, int64_t N, int64_t T, int64_t* ids_packed, int32_t* row_map, void* stream);
int nr_tfm_layer_packed_fwd_f32(const float* x, const int64_t* mask, const int32_t* offs, const int32_t* row_map, int64_t N, int64_t S, int64_t R, int64_t D, int64_t H, const nr_tfm_weights* w, float p_drop, uint64_t seed, uint32_t site_base, int mask_mode, nr_tfm_saved* saved, float* y, void* stream);
int nr_tfm_layer_packed_bwd_f32(const float* dy, const float* x, const int64_t* mask, const int32_t* offs, const int32_t* row_map, int64_t N, int64_t S, int64_t R, int64_t D, int64_t H, const nr_tfm_weights* w, float p_drop, uint64_t seed, uint32_t site_base, int mask_mode, const nr_tfm_saved* saved, nr_tfm_grads* grads, float* dx, void* ws, size_t ws_bytes, void* stream);
/* scaled_dp_attention pooling (attention.py:6-34) over packed key/value [R,D]; prob [nseq,S] gets 0 at dropped positions.
 * Backward: dkeypre / dvalue [R,D]; ws: nr_query_pool_bwd_ws_bytes(nseq, D). */
int nr_query_pool_packed_fwd_f32(const float* key, const float* value, const int32_t* offs, int64_t R, const float* query, const int64_t* mask, int64_t nseq, int64_t S, int64_t D, float* out, float* prob, void* stream);
int nr_query_pool_packed_bwd_f32(const float* key, const float* value, const int32_t* offs, int64_t R, const float* query, const float* prob, const float* dout, int64_t nseq, int64_t S, int64_t D, int tanh_key, float* dkeypre, float* dvalue, float* dquery, void* ws, size_t ws_bytes, void* stream);

/* The same backward when the pooled vectors went to two consumers: sequences 0 .. split-1 take their gradient from dout0
 * [split, D], the rest from dout1 [nseq - split, D] (a train step's candidate titles -> scoring, history titles -> user
 * encoder, models/BaseModel.py:198-205): no concatenation of the two gradients is needed. */
int nr_query_pool_packed_bwd_split_f32(const float* key, const float* value, const int32_t* offs, int64_t R, const float* query, const float* prob, const float* dout0, const float* dout1, int64_t split, int64_t nseq, int64_t S, int64_t D, int tanh_key, float* dkeypre, float* dvalue, float* dquery, void* ws, size_t ws_bytes, void* stream);
/* Title tensors of one train step in one launch — replaces the reference's two host->device copies + two encoder calls
 * (models/BaseModel.py:178-195) and the torch concatenations this package used before: block-major ids / mask
 * [B*C + B*L, T] (all candidate titles, then all history titles) and keep [B*C + B*L] (1 for candidates; history slot
 * (b,l): keep_mode 0 -> 1, 1 -> his_mask[b,l] != 0 (TfmUserEncoder), 2 -> l < sum(his_mask[b]) (RnnUserEncoder)). */
int nr_assemble_titles(const int64_t* cdd_ids, const int64_t* his_ids, const int64_t* cdd_mask, const int64_t* his_attn_mask, const int64_t* his_mask,
                       int keep_mode, int64_t B, int64_t C, int64_t L, int64_t T, int64_t* ids, int64_t* mask, int64_t* keep, void* stream);

/* ---- bf16 build: building blocks (tensors in HBM as bf16 = raw uint16; no counterpart in the reference, whose closest
 * thing is torch.autocast over encoder.py:62-77) ---------------------------------------------------------
 * nr_cast_bf16: y = bf16(x), round to nearest even.
 * nr_transpose_bf16: dst [C, Rp] = bf16(src [R, C])^T, columns R..Rp-1 zero (Rp % 64 == 0): the transposed,
 *   reduction-padded operand copies that bring dX = dY.W and dW = dY^T.X to the one form the bf16 GEMM takes.
 * nr_gemm_nt_bf16: C[m,n] = act(sum_k A[m,k] B[n,k] + bias[n]); A [M,K], B [N,K] bf16, K % 64 == 0, f32 accumulate on
 *   v_mfma_f32_32x32x16_bf16; results as f32 (C, may be NULL) and / or bf16 (Ch, may be NULL); act as nr_linear_fwd_f32
 *   (2 = gelu also stores the pre-activation as bf16 in zh when given).  split_k != 0: f32 C only, no bias/act, slabs in ws. */
int nr_cast_bf16(const float* x, int64_t n, uint16_t* y, void* stream);
int nr_transpose_bf16(const void* src, int src_is_bf16, int64_t R, int64_t C, int64_t ld, uint16_t* dst, int64_t Rp, void* stream);
size_t nr_gemm_nt_bf16_ws_bytes(int64_t M, int64_t N, int64_t K, int split_k);
int nr_gemm_nt_bf16(const uint16_t* A, const uint16_t* B, const float* bias, int64_t M, int64_t N, int64_t K, int act, float* C, uint16_t* Ch, uint16_t* zh, int split_k, void* ws, size_t ws_bytes, void* stream);
/* nr_gemm_bf16 (round 3): the same product with either operand REDUCTION-MAJOR, as the backward finds them in memory
 * (no counterpart in the reference: torch.autocast transposes inside its BLAS calls):
 *   a_rm = 0: A(m,k) = A[m*lda + k]      a_rm = 1: A(m,k) = A[k*lda + m]   (dY of dW = dY^T X; needs M % 8 == 0)
 *   b_rm = 0: B(n,k) = B[n*ldb + k]      b_rm = 1: B(n,k) = B[k*ldb + n]   (X of dW, the weight of dX = dY W; N % 8 == 0)
 * (a_rm, b_rm) = (0,0) | (0,1) | (1,1); K % 64 == 0 unless both operands are reduction-major (their rows past K read
 * zeros: any K).  Tiles go to LDS as they lie in memory and the matrix-core fragments are taken with transposed LDS
 * reads (ds_read_b64_tr_b16), so the layer backward makes no transposed copies.  Pitches in elements, multiples of 8;
 * 16-byte aligned bases.  Outputs / act / split_k as nr_gemm_nt_bf16. */
int nr_gemm_bf16(const uint16_t* A, int a_rm, int64_t lda, const uint16_t* B, int b_rm, int64_t ldb, const float* bias, int64_t M, int64_t N, int64_t K, int act, float* C, uint16_t* Ch, uint16_t* zh, int split_k, void* ws, size_t ws_bytes, void* stream);

/* ---- the transformer layer of the bf16 build (AllBertNewsEncoder, encoder.py:62-77, parity bar 1e-2) ----
 * Same layer, same argument meaning as nr_tfm_layer_{fwd,bwd}_f32; every GEMM operand (x, qkv, ctx, LN1 output,
 * pre-GELU, GELU output and their gradients, plus bf16 copies of the weights made per call) and the attention tensors live
 * in HBM as bf16; the residual stream, LayerNorm inputs / statistics, softmax arithmetic, accumulation and all parameter
 * gradients are f32.  Needs D % 64 == 0, S <= 32, head dim <= 64, q|k|v weights / biases / gradients contiguous
 * (packed [3D, D]); otherwise NR_ERR_UNSUPPORTED / NR_ERR_ARG and nothing runs.  R = N*S. */
typedef struct {
    uint16_t *x_h, *qkv_h, *ctx_h, *y1_h, *z_h, *a_h;   /* bf16 [R,D] [R,3D] [R,D] [R,D] [R,4D] [R,4D] */
    float *s1, *y1, *s2;                                /* f32 [R,D]: drop(dense)+x, LN1 output (forward only), drop(dense)+y1 */
    float *stats, *row_max, *row_sum;                   /* [4,R], [N,H,S], [N,H,S] */
    uint16_t *w_h;                                      /* bf16 wqkv | wo | w1 | w2: 12 D^2 elements */
    uint16_t *wT_h;                                     /* bf16, 12 D^2 elements, may be NULL: wqkv^T [D,3D] | wo^T | w1^T [D,4D] | w2^T [4D,D],
                                                           written by the forward in the same pass as w_h and read by the backward's
                                                           dX products (NULL: the backward transposes the f32 weights itself).  Until
                                                           round 2 this slot held the transposed GELU output, which nothing needs any
                                                           more: the dW products read row-major tensors as reduction-major operands */
} nr_tfm_saved_bf16;
/* offs / row_map: NULL for the dense [N,S,D] layout (R = N*S), or the packed-title-token plan of nr_pack_plan / nr_pack_ids (R rows). */
int nr_tfm_layer_fwd_bf16(const float* x, const int64_t* mask, const int32_t* offs, const int32_t* row_map, int64_t N, int64_t S, int64_t R, int64_t D, int64_t H, const nr_tfm_weights* w, float p_drop, uint64_t seed, uint32_t site_base, int mask_mode, nr_tfm_saved_bf16* saved, float* y, void* stream);
size_t nr_tfm_layer_bwd_bf16_ws_bytes(int64_t R, int64_t D);
int nr_tfm_layer_bwd_bf16(const float* dy, const int64_t* mask, const int32_t* offs, const int32_t* row_map, int64_t N, int64_t S, int64_t R, int64_t D, int64_t H, const nr_tfm_weights* w, float p_drop, uint64_t seed, uint32_t site_base, int mask_mode, const nr_tfm_saved_bf16* saved, nr_tfm_grads* grads, float* dx, void* ws, size_t ws_bytes, void* stream);

/* packed-row variants of the BERT embedding stage pieces (HF BertEmbeddings behind AllBertNewsEncoder, encoder.py:65,74):
 * the LayerNorm pair with the Philox element index AND the position row (x_period) taken from the ORIGINAL row
 * row_map[r]; the embedding dropout likewise; nr_unpack_rows_f32 scatters packed rows back into a zeroed padded
 * [Rfull, D] tensor (the position-row gradient sums over titles per position). */
int nr_add_layernorm_packed_fwd_f32(const float* h, const float* x, int64_t x_period, const int32_t* row_map, const float* gamma, const float* beta, int64_t R, int64_t D, float eps, float p_drop, uint64_t seed, uint32_t site, float* y, float* s_out, float* mean, float* rstd, void* stream);
int nr_add_layernorm_packed_bwd_f32(const float* dy, const float* s, const float* mean, const float* rstd, const float* gamma, const int32_t* row_map, int64_t R, int64_t D, float p_drop, uint64_t seed, uint32_t site, float* ds, float* dh, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream);
int nr_dropout_packed_f32(const float* x, const int32_t* row_map, int64_t R, int64_t D, float p_drop, uint64_t seed, uint32_t site, float* out, void* stream);
int nr_unpack_rows_f32(const float* src, const int32_t* row_map, int64_t R, int64_t D, int64_t Rfull, float* dst, void* stream);

/* ---- GRU user encoder — RnnUserEncoder, models/modules/encoder.py:204-233 -----------------------------
 * One-layer nn.GRU(Din, H, batch_first) over x [B,L,Din], packed by len_b = sum_l mask[b,l] (>= 1: the
 * reference sets his_mask[b,0] = 1 for an empty history, dataset.py:140-141); the user vector is the state
 * after len_b steps = hall[L] (states freeze when a sequence ends).  torch layouts: w_ih [3H,Din], w_hh [3H,H],
 * gate order r|z|n.  Caller-allocated: gi [B*L,3H], gh [B,3H] scratch, hall [(L+1),B,H] (hall[0] is zeroed
 * here), gates [L,B,4H] (r,z,n,gh_n; NULL = eval, nothing saved), lens [B] int32.  No host sync
 * (the reference's lens.cpu() at encoder.py:227 is gone). */
size_t nr_gru_fwd_ws_bytes(int64_t B, int64_t H);   /* split-K scratch of the per-step h W_hh^T GEMM (32 sequences alone fill 1/8 of the CUs) */
int nr_gru_fwd_f32(const float* x, const int64_t* mask, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, int64_t B, int64_t L, int64_t Din, int64_t H, float* gi, float* gh, float* hall, float* gates, int32_t* lens, void* ws, size_t ws_bytes, void* stream);
/* dh_out [B,H] = gradient of hall[L]; dx [B*L,Din] may be NULL; dw_ih, dw_hh, db_ih, db_hh are overwritten. */
size_t nr_gru_bwd_ws_bytes(int64_t B, int64_t L, int64_t Din, int64_t H);
int nr_gru_bwd_f32(const float* dh_out, const float* x, const float* w_ih, const float* w_hh, int64_t B, int64_t L, int64_t Din, int64_t H, const float* hall, const float* gates, const int32_t* lens, float* dx, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, void* ws, size_t ws_bytes, void* stream);

/* The persistent recurrences synchronise their workgroups with a grid barrier whose spin is bounded (2^22 polls, seconds),
 * so a lost workgroup cannot hang the GPU.  A workgroup that gives up overwrites the outputs it owns with NaN (hall[L]
 * columns in the forward, dGI columns -> dw_ih / dx in the backward: the loss or the gradients turn NaN) and raises a flag
 * at the head of the workspace.  nr_gru_status synchronises `stream`, reads that flag from the `ws` of the LAST
 * nr_gru_fwd_f32 / nr_gru_bwd_f32 call that used it and returns NR_ERR_LAUNCH if the barrier timed out, NR_OK otherwise.
 * nr_gru_debug_set_spin_limit is a TEST HOOK (process-wide): polls before giving up; 0 restores the default. */
int nr_gru_status(const void* ws, void* stream);
int nr_gru_debug_set_spin_limit(uint32_t polls);

/* ---- per-impression ranking metrics — pack_results + compute_metrics, utils/util.py:37-57,126-158,197-250 ----
 * Result rows (<= Ce-candidate chunks of an impression, models/BaseModel.py:254-260) stay in HBM as flat tensors:
 * preds [R,Ce] f32, labels [R,Ce] int64 (-1 = padding), masks [R,Ce] uint8.  Impression g = rows
 * row_start[g] .. +row_count[g]; its candidates are the mask-selected entries in row order.
 * out [G, 2+nk] f64 = (auc, mrr, ndcg@ks[0], ...); ks_host: HOST array of nk <= 4 cut-offs.
 * flags [G]: bit 0 = an exact score tie (recompute MRR/nDCG of that impression with numpy's own argsort: its tie
 * order is not reproducible elsewhere; AUC is exact), 2 = more than 1024 candidates, 4 = a label outside {0,1},
 * 8 = one class only (the reference raises) — 2/4/8: not computed, fall back to the host functions. */
int nr_impression_metrics_f64(const float* preds, const int64_t* labels, const uint8_t* masks, int64_t Ce, const int64_t* row_start, const int32_t* row_count, int64_t G, int nk, const int* ks_host, double* out, int32_t* flags, void* stream);

/* ---- helpers of the BERT embedding stage (HF BertEmbeddings behind AllBertNewsEncoder, encoder.py:65,74) ----
 * out[i] = keep(i) ? x[i]/(1-p) : 0 with this library's Philox stream: embeddings.dropout, the one dropout
 * site that is not followed by a LayerNorm.  The same call on the gradient is its backward. */
int nr_dropout_f32(const float* x, int64_t n, float p_drop, uint64_t seed, uint32_t site, float* out, void* stream);
/* out[r,:] = x[r,:] + b[:]  (position rows + token_type row 0) */
int nr_add_bias_rows_f32(const float* x, const float* b, int64_t R, int64_t D, float* out, void* stream);
/* out[n] = sum_r x[r,n], deterministic two-stage sum (gradient of a row broadcast: position / token-type rows). */
size_t nr_colsum_ws_bytes(int64_t R, int64_t N);
int nr_colsum_f32(const float* x, int64_t R, int64_t N, float* out, void* ws, size_t ws_bytes, void* stream);

/* ---- dense Adam — optim.Adam(lr) in BaseModel.get_optimizer, models/BaseModel.py:45-46 ----------
 * torch defaults (betas 0.9/0.999, eps 1e-8, no weight decay, no amsgrad); step counts from 1.
 * Hyper-parameters are doubles: torch forms 1-beta, lr/(1-beta1^t), sqrt(1-beta2^t) in double before rounding to fp32. */
int nr_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1, double beta2, double eps, int64_t step, void* stream);
/* The same update for `count` tensors that share one step number, in one launch per 48 tensors (reference
 * BaseModel.py:46 builds ONE Adam over self.parameters(); manager.py:392 steps it once per batch).  The five
 * tables are HOST arrays of device pointers / element counts; they are copied into the kernel arguments. */
int nr_adam_multi_f32(int count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                      const int64_t* numel, double lr, double beta1, double beta2, double eps, int64_t step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NEWSREC_HIP_H */
