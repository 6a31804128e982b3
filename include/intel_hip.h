/*
 * intel_hip.h -- C ABI of the MI355X-native IntEL hot path (libintel_hip.so).
 *
 * The reference (JiayuLi-997/IntEL-SIGIR2023) is pure Python and has no FFI: its extension point
 * is class lookup by name (IntEL/src/main.py:127-130).  This header is therefore the boundary a
 * maintainer binds from Python with ctypes; every entry point cites the
 * reference code it replaces (paths relative to IntEL/src).
 *
 * Conventions
 *   - plain C: device pointers are passed as void* / typed pointers, sizes as int / long long,
 *     the HIP stream as void* (hipStream_t); no torch types.
 *   - all tensors are dense row-major; float = fp32, ids/lengths = int32.
 *   - every function returns 0 on success, a negative INTEL_E_* code or a positive hipError_t
 *     otherwise; intel_last_error() returns a message for the calling thread.
 *   - nothing here allocates device memory or synchronises: the caller provides a workspace of
 *     intel_workspace_bytes() bytes and owns all ordering through the stream (graph-capturable).
 */
#ifndef INTEL_HIP_H
#define INTEL_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define INTEL_ABI_VERSION 5

enum {
  INTEL_OK = 0,
  INTEL_E_ARG = -1,        /* bad argument / unsupported shape */
  INTEL_E_WORKSPACE = -2,  /* workspace too small */
  INTEL_E_STATE = -3       /* backward without a matching training forward */
};

enum { INTEL_ENC_BERT4REC = 0, INTEL_ENC_GRU4REC = 1 };

/* Model hyper-parameters: the flags of IntEL.parse_model_args (models/IntEL/IntEL.py:17-34),
 * GeneralSeq (models/GeneralSeq.py:15-17) and the corpus sizes read in IntEL.__init__ (:36-115). */
typedef struct IntelDesc {
  int model_num;        /* K  --model_num                     */
  int intent_num;       /* I  len(corpus.zero_int)            */
  int item_num;         /* rows of iid_embeddings             */
  int class_num;        /* rows of item_embeddings (itemfnum) */
  int user_num;         /* rows of uid_embeddings             */
  int ctx_num;          /* rows of context_embeddings         */
  int d_id, d_im, d_u, d_s, d_c, d_int; /* --i/im/u/s/context/intent_emb_size */
  int q_size;           /* --cross_attn_qsize                 */
  int heads, layers;    /* --num_heads / --num_layers (tied)  */
  int cross_attention;  /* --cross_attention                  */
  int encoder;          /* INTEL_ENC_*                        */
  int history_max;      /* --history_max                      */
  int enc_layers, enc_heads; /* BERT4Rec: hard-coded 2/2 (IntEL.py:108-109) */
  int gru_hidden;       /* GRU4Rec: hard-coded 128 (IntEL.py:105-106)       */
  /* ---- ABI version 2 ---- */
  int weight_norm;      /* softmax applications over the K fusion weights: 0 = none (IntEL.py:214, the parity default),
                         * 1 = one softmax (SURVEY.md 0.3 `weight_norm=softmax`), 2 = the double softmax of the
                         * aWELv_IntEL variant (models/supervise/aWELv_IntEL.py:199-200)                              */
  int pool_mean;        /* 1 = aWELv_IntEL feature: h * g(intent) mean-pooled over ALL L rows, one weight vector per
                         * session repeated over the list, nothing masked (aWELv_IntEL.py:188-201); needs
                         * cross_attention = 0 (the gate MLPs are the intent_{item,score}_embeddings slots)           */
  int dtype;            /* INTEL_DTYPE_F32 (parity mode) | INTEL_DTYPE_BF16 (bf16 storage / single bf16 product)      */
} IntelDesc;
#define INTEL_DTYPE_F32 0
#define INTEL_DTYPE_BF16 1

/* Parameter slots.  Each maps 1:1 to a reference state_dict key.  The arrays
 * passed as `params` / `grads` have INTEL_P_COUNT entries; unused slots are NULL. */
enum IntelParam {
  INTEL_P_IID_EMB = 0,   /* iid_embeddings.weight      [item_num, d_id]  */
  INTEL_P_ITEM_EMB,      /* item_embeddings.weight     [class_num, d_im] */
  INTEL_P_UID_EMB,       /* uid_embeddings.weight      [user_num, d_u]   */
  INTEL_P_CTX_EMB,       /* context_embeddings.weight  [ctx_num, d_c]    */
  INTEL_P_INTENT_W,      /* intent_embeddings.weight   [d_int, I]        */
  INTEL_P_INTENT_B,      /* intent_embeddings.bias     [d_int]           */
  INTEL_P_SCORE_W,       /* score_embeddings.weight    [d_s, K]          */
  INTEL_P_SCORE_B,
  /* item tower: i_attn_head.{q,k,v}_linear.weight, i_W1, i_W2, i_layer_norm */
  INTEL_P_I_WQ, INTEL_P_I_WK, INTEL_P_I_WV, INTEL_P_I_W1, INTEL_P_I_B1, INTEL_P_I_W2, INTEL_P_I_B2,
  INTEL_P_I_LNG, INTEL_P_I_LNB,
  /* score tower: s_* */
  INTEL_P_S_WQ, INTEL_P_S_WK, INTEL_P_S_WV, INTEL_P_S_W1, INTEL_P_S_B1, INTEL_P_S_W2, INTEL_P_S_B2,
  INTEL_P_S_LNG, INTEL_P_S_LNB,
  /* cross attention: intent_{item,score}_attention.{query,key,value}_layer.weight */
  INTEL_P_XI_WQ, INTEL_P_XI_WK, INTEL_P_XI_WV,
  INTEL_P_XS_WQ, INTEL_P_XS_WK, INTEL_P_XS_WV,
  /* --cross_attention 0: intent_{item,score}_embeddings.{0.weight,0.bias,2.weight} */
  INTEL_P_MI_W0, INTEL_P_MI_B0, INTEL_P_MI_W2,
  INTEL_P_MS_W0, INTEL_P_MS_B0, INTEL_P_MS_W2,
  INTEL_P_WE_W, INTEL_P_WE_B,     /* weight_embeddings */
  INTEL_P_PRED_W, INTEL_P_PRED_B, /* pred_layer        */
  /* sequence encoders: slot = INTEL_P_ENC0 + e*INTEL_ENC_STRIDE + offset, e = 0 ("encoder"),
   * 1 ("item_encoder").  BERT4Rec: POS then per block l: INTEL_ENC_BLOCK0 + l*INTEL_ENC_BLOCK_STRIDE
   * + {WQ,BQ,WK,BK,WV,BV,LN1G,LN1B,W1,B1,W2,B2,LN2G,LN2B}.  GRU4Rec: WIH,WHH,BIH,BHH,OUT. */
  INTEL_P_ENC0
};
enum {
  INTEL_ENC_POS = 0,
  INTEL_ENC_GRU_WIH = 1, INTEL_ENC_GRU_WHH, INTEL_ENC_GRU_BIH, INTEL_ENC_GRU_BHH, INTEL_ENC_GRU_OUT,
  INTEL_ENC_BLOCK0 = 6,
  INTEL_ENC_WQ = 0, INTEL_ENC_BQ, INTEL_ENC_WK, INTEL_ENC_BK, INTEL_ENC_WV, INTEL_ENC_BV,
  INTEL_ENC_LN1G, INTEL_ENC_LN1B, INTEL_ENC_W1, INTEL_ENC_B1, INTEL_ENC_W2, INTEL_ENC_B2,
  INTEL_ENC_LN2G, INTEL_ENC_LN2B,
  INTEL_ENC_BLOCK_STRIDE = 14,
  INTEL_ENC_MAX_BLOCKS = 4,
  INTEL_ENC_STRIDE = 6 + 14 * 4,
  INTEL_P_COUNT = INTEL_P_ENC0 + 2 * (6 + 14 * 4)
};

/* One collated batch: the tensors of BaseModel.Dataset.collate_batch (models/BaseModel.py:121-142)
 * that IntEL.forward reads (IntEL.py:126-217), ids narrowed to int32 and scores to fp32 by the host.
 * L / H / Hi are the PADDED lengths of this batch (pad rows are real rows: SURVEY.md §0.5). */
typedef struct IntelBatch {
  int B, L, H, Hi;
  const int* i_id_s;            /* [B,L]            */
  const int* i_class_c;         /* [B,L]            */
  const float* scores;          /* [B,L,K]          */
  const int* session_len;       /* [B]              */
  const int* u_id_c;            /* [B]              */
  const int* context_mh;        /* [B]              */
  const int* his_context_mh;    /* [B,H]            */
  const float* his_intents;     /* [B,H,I]          */
  const int* history_len;       /* [B]              */
  const int* his_item_id;       /* [B,Hi]           */
  const int* his_item_idx;      /* [B,Hi] intent index of each history item, -1 = all-zero row;
                                   NULL when his_item_int is given                       */
  const float* his_item_int;    /* [B,Hi,I] dense form (reference layout) or NULL        */
  const int* history_item_len;  /* [B]              */
  /* ---- ABI version 2: optional packing of the two histories (BERT4Rec encoders at any history length, GRU4Rec encoders).  The padded positions
   * t >= history_len[b] of a history never reach a valid row (their keys are masked, GeneralSeq.py:100; the block is row-wise
   * otherwise; the output is multiplied by `valid` and only row len-1 is used, :103-105), so the encoder may run on the valid
   * rows alone.  his_off / hisitem_off = exclusive prefix sums of history_len / history_item_len ([B] ints, device);
   * n_his_rows / n_hisitem_rows = their totals (HOST ints: they size the launches).  NULL / 0 = run the padded [B,H] rows.
   * PRECONDITION for packed rows: history_len[b] >= 1 and history_item_len[b] >= 1, as the reference's Dataset guarantees (a
   * session without history carries ONE all-zero event, GeneralSeq.py:48-52, IntEL.py:234-237; torch's pack_padded_sequence
   * rejects a zero length).  An isolated empty history is tolerated (its encoder output is the pruned block applied to a zero
   * row, no out-of-bounds access), but the fused BERT4Rec kernels take at most 64 session starts per 33..64-row tile window,
   * which only histories of >= 1 row guarantee: a batch that may hold empty histories should leave his_off NULL.  The Python
   * producers (data.collate_batch, feed, synth) omit the totals -- and with them the packing -- when a length is 0. */
  const int* his_off;
  const int* hisitem_off;
  int n_his_rows, n_hisitem_rows;
  /* optional: the batch's ids sorted ascending with their row indices (device int arrays; NULL = unsorted scatter).  The
   * dense embedding gradients are accumulated with float atomics; with the pairs sorted, runs of equal ids
   * are summed in registers first, so a popular id costs one atomic row per 16 hits instead of one per hit:
   *   iid_sort_*      keys i_id_s [B*L]            (row = b*L + l)
   *   cls_sort_*      keys i_class_c [B*L]
   *   hisitem_sort_*  keys his_item_id [B*Hi]      (row = b*Hi + t; padded positions are skipped through history_item_len) */
  const int* iid_sort_ids;     const int* iid_sort_rows;
  const int* cls_sort_ids;     const int* cls_sort_rows;
  const int* hisitem_sort_ids; const int* hisitem_sort_rows;
  /* optional (GRU4Rec encoders): the sessions ordered by history_len / history_item_len (device int [B], a permutation of
   * 0 .. B-1; NULL = batch order).  The one-kernel recurrence gives 16 consecutive sessions of this order to a workgroup, whose
   * time loop runs to the longest of them: with sessions of similar length together a workgroup stops at its own length instead
   * of (almost always) the maximum.  Results do not depend on the order. */
  const int* his_order; const int* hisitem_order;
} IntelBatch;

/* Outputs of IntEL.forward (IntEL.py:117-124). */
typedef struct IntelOut {
  float* weights;    /* [B,L,K] */
  float* ens_score;  /* [B,L]   */
  float* intents;    /* [B,I]   */
} IntelOut;

typedef struct IntelCtx IntelCtx;

const char* intel_last_error(void);
int intel_abi_version(void);
/* out4 = {sizeof(IntelDesc), sizeof(IntelBatch), sizeof(IntelOut), INTEL_P_COUNT}: lets a binding check its mirror */
void intel_abi_sizes(int* out4);

/* Context = desc + host-side plan.  Holds no device memory. */
IntelCtx* intel_create(const IntelDesc* desc);
void intel_destroy(IntelCtx* ctx);

/* The independent branches of a step (two towers, two sequence encoders) run on internal side streams that
 * fork from / join into the caller's stream (event-ordered, graph-capturable).  on = 0 keeps everything on
 * the caller's stream (bench.py does this while it prices single kernels).  Default: on. */
void intel_set_concurrency(IntelCtx* ctx, int on);

/* stream (may be NULL = off): a stream that intel_backward (the one-call form) makes wait for the completion of
 * grads[INTEL_P_IID_EMB] -- before its tail of shared-weight gradients and deferred reductions.  The caller enqueues the
 * table's optimizer sweep there (HBM-bound, it then runs underneath that tail) and joins the streams before the next forward.
 * Where the four-branch schedule is not taken (branch concurrency off) the stream is made to wait for the whole backward instead.
 * The two-call form (intel_backward_phase) does not use it: there phase 1 returns at that point. */
void intel_set_table_stream(IntelCtx* ctx, void* stream);

/* One of the context's three internal side streams (i = 0..2; NULL when branch concurrency is off).  The runtime multiplexes
 * streams onto four hardware queues by default, and a fifth ACTIVE stream shares a queue with one of the others: a caller that
 * wants work next to the context's branches (the table's optimizer sweep after intel_backward, which leaves all three idle)
 * should borrow one of these instead of creating its own.  Valid until intel_destroy. */
void* intel_side_stream(IntelCtx* ctx, int i);
/* ABI version 5.  One shot: the NEXT intel_forward makes the streams that READ the item-id table (the item tower's and the item-history encoder's
 * gathers) wait for `event` (a hipEvent_t the caller recorded behind its optimizer's sweep of that table) -- the other two branches, which do not
 * touch the table, start under the sweep.  NULL: off.  The caller orders every OTHER reader of the table itself.  No reference counterpart (the
 * reference's step is one stream: helpers/BaseRunner.py:279-290). */
void intel_set_table_wait_event(IntelCtx* ctx, void* event);

/* A promise for the following intel_forward calls: the parameter VALUES equal those of the previous intel_forward on this
 * context.  The forward then reuses the packed weight images that call left in the workspace, provided the workspace
 * pointer, the batch shape and `train` are unchanged too (evaluation loops over a frozen model, helpers/BaseRunner.py:328-343:
 * about 5 % of a forward).  Default off; every forward under on = 0 repacks.  The caller owns the promise: parameters updated
 * in place behind it are not noticed. */
void intel_set_params_unchanged(IntelCtx* ctx, int on);

/* nn.Dropout(--dropout) of the two tower layers (models/IntEL/IntEL.py:63,187,196) for the following
 * intel_forward(train=1) calls: p = 0 (default) disables it; evaluation never drops.  keep_flags (optional, device):
 * 0/1 floats -- item-tower layers [layers][B*L][d_i] then score-tower layers [layers][B*L][d_s] -- replace the
 * built-in counter-based generator (parity tests pass the reference's own draw).  Call it BEFORE
 * intel_workspace_bytes: the training workspace grows by one mask per tower layer. */
int intel_set_dropout(IntelCtx* ctx, float p, unsigned long long seed, const float* keep_flags);

/* row_flags[item_num] (device, may be NULL = off): intel_backward sets row_flags[i] = 1 for every row i of the
 * iid_embeddings.weight gradient it adds into (the nn.Embedding backward of IntEL.py:135,170).  The caller owns the
 * array and keeps "flag == 0 => gradient row == 0" true; intel_adam_step_rows consumes and resets the flags. */
int intel_set_iid_grad_row_flags(IntelCtx* ctx, unsigned char* row_flags);

/* Bytes of workspace intel_forward/intel_backward need for a batch of this shape.  `train` != 0
 * also reserves the activation stash the backward pass reads. */
size_t intel_workspace_bytes(const IntelCtx* ctx, int B, int L, int H, int Hi, int train);

/* IntEL.forward (IntEL.py:117-124) = predict_intent (:126-155) + predict_ensemble (:158-217). */
int intel_forward(IntelCtx* ctx, const void* const* params, const IntelBatch* batch, void* workspace,
                  size_t workspace_bytes, const IntelOut* out, int train, void* stream);

/* loss.backward() for the model part (helpers/BaseRunner.py:288): given d(loss)/d(out) computes
 * d(loss)/d(param) for every parameter.  `grads[slot]` must be zero-initialised for the embedding
 * tables (row grads are accumulated with atomics) -- all other slots are overwritten.  Must follow
 * an intel_forward(train=1) on the same ctx / batch / workspace. */
int intel_backward(IntelCtx* ctx, const void* const* params, const IntelBatch* batch, void* workspace,
                   size_t workspace_bytes, const float* d_weights, const float* d_ens_score,
                   const float* d_intents, void* const* grads, void* stream);

/* The same backward in two halves for data-parallel overlap: phase 1 computes everything the item-id table
 * gradient (the one large all-reduce) depends on, phase 2 the rest (score-tower layers, session-history
 * encoder); the caller starts the all-reduce of grads[INTEL_P_IID_EMB] between the two calls.  phase 0 = both.
 * Same arguments and preconditions as intel_backward; phase 1 must precede phase 2. */
int intel_backward_phase(IntelCtx* ctx, const void* const* params, const IntelBatch* batch, void* workspace,
                         size_t workspace_bytes, const float* d_weights, const float* d_ens_score,
                         const float* d_intents, void* const* grads, int phase, void* stream);

/* ---- losses ------------------------------------------------------------------------------- */
/* BPRloss.forward (loss/BPRloss.py:37-56) incl. bpr_loss (:20-34) and diversity (:12-18).
 *   noise [B,L,L] replaces torch.rand_like (BPRloss.py:26); scores_f64 may be NULL (then scores_f32
 *   is used for the diversity term).  Outputs: loss[1] (fp32), select[B,L] (int32).  When d_ens /
 *   d_weights are non-NULL the gradients of `loss * grad_scale` are written too. */
int intel_bpr_loss(int B, int L, int K, const float* ens_score, const int* ranking, const int* session_len,
                   const float* noise, const double* scores_f64, const float* scores_f32,
                   const float* weights, int cal_diversity, double alpha, float grad_scale,
                   float* loss, int* select, float* d_ens, float* d_weights, void* workspace,
                   size_t workspace_bytes, void* stream);
/* The same with the tie-breaking noise of BPRloss.py:26 drawn inside the kernel (counter-based generator keyed by
 * seed and the GLOBAL element index ((session0 + b) * L + i) * L + j) instead of read from a [B,L,L] tensor: the fused
 * training step uses this form.  session0 = global index of this launch's first session: data-parallel rank r passes
 * r * B with the step's common seed, so the N shards draw exactly what one process draws for the whole batch and no two
 * sessions of a global batch share their tie-breaks. */
int intel_bpr_loss_seeded(int B, int L, int K, const float* ens_score, const int* ranking, const int* session_len,
                          unsigned long long seed, unsigned long long session0, const double* scores_f64, const float* scores_f32,
                          const float* weights, int cal_diversity, double alpha, float grad_scale,
                          float* loss, int* select, float* d_ens, float* d_weights, void* workspace,
                          size_t workspace_bytes, void* stream);
/* Listloss.forward (loss/Listloss.py:25-43) incl. list_loss (:12-15) and diversity (:17-23). */
int intel_list_loss(int B, int L, int K, const float* ens_score, const int* ranking, const int* session_len,
                    const double* scores_f64, const float* scores_f32, const float* weights,
                    int cal_diversity, double alpha, float grad_scale, float* loss, float* d_ens,
                    float* d_weights, void* workspace, size_t workspace_bytes, void* stream);
/* MSEloss.forward (loss/MSEloss.py:22-30) incl. diversity (:14-20): per-list mean of (ens - max(label, 0))^2
 * over the valid slots; same conventions as the two losses above. */
int intel_mse_loss(int B, int L, int K, const float* ens_score, const int* ranking, const int* session_len,
                   const double* scores_f64, const float* scores_f32, const float* weights,
                   int cal_diversity, double alpha, float grad_scale, float* loss, float* d_ens,
                   float* d_weights, void* workspace, size_t workspace_bytes, void* stream);
/* BaseIntloss.get_intloss (loss/BaseIntloss.py:57-67): out3 = {intent_loss, ce, kl} (fp64);
 * d_pred (optional) = grad_scale * d(intent_loss)/d(pred). */
int intel_intent_loss(int B, int I, const float* pred, const double* label, double kl_weight, double kl_temp,
                      float grad_scale, double* out3, float* d_pred, void* workspace, size_t workspace_bytes,
                      void* stream);
size_t intel_loss_workspace_bytes(int B, int L, int K);
/* The three values IntBPRloss / IntListloss / IntMSEloss.forward return (loss/IntBPRloss.py:15-20):
 * out3 = {ensemble_loss * ensemble_weight + intent_loss * intent_weight (float64), ensemble_loss, intent_loss};
 * intent_out3 = the out3 of intel_intent_loss, or NULL for the losses without the intent term (all three = ensemble loss). */
int intel_loss_total(const float* ensemble_loss, const double* intent_out3, double ensemble_weight, double intent_weight,
                     double* out3, void* stream);

/* ---- optimizer ---------------------------------------------------------------------------- */
/* torch.optim.Adam.step as configured by BaseRunner._build_optimizer (helpers/BaseRunner.py:182-188)
 * with the groups of BaseModel.customize_parameters (models/BaseModel.py:53-62): coupled L2
 * (g += wd*p), bias-corrected moments, dense over all n elements.  zero_grad != 0 also clears g. */
int intel_adam_step(float* p, float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int step, float grad_scale, int zero_grad, void* stream);

/* ABI version 4: the same update for TWO parameter groups in one launch (BaseModel.customize_parameters makes exactly two: names
 * without 'bias' -> weight_decay = --l2, names with 'bias' -> 0; models/BaseModel.py:53-62).  p / g / m / v / n / weight_decay are
 * HOST arrays of two entries (device pointers, element counts, decays); a group with n <= 0 is skipped.  Bit-identical to two
 * intel_adam_step calls. */
int intel_adam_step_pair(float* const* p, float* const* g, float* const* m, float* const* v, const long long* n,
                         const float* weight_decay, float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                         int zero_grad, void* stream);

/* The same update over a [rows, d] embedding table whose gradient g is zero outside the rows flagged in row_flags
 * (one byte per row; see intel_set_iid_grad_row_flags): g is read, cleared and the flag reset only in flagged rows, every
 * other row is updated with g = 0 -- bit-identical to intel_adam_step(..., zero_grad = 1) at 6 instead of 8 memory
 * streams.  d in {16, 32, 64, 128, 256}. */
int intel_adam_step_rows(float* p, float* g, float* m, float* v, long long rows, int d, unsigned char* row_flags,
                         float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                         float grad_scale, void* stream);

/* ---- lazy form of the item-id table's dense Adam -------------------------------------------------------------
 * torch.optim.Adam updates EVERY row of iid_embeddings.weight every step (helpers/BaseRunner.py:182-188, 288-290: dense
 * gradient, coupled L2), although a step's gradient is non-zero only in the rows of its batch.  The update of a row with
 * g = 0 reads nothing but the row's own (p, m, v) and the step's two scalars, so it can be REPLAYED later, step by step,
 * with exactly the arithmetic of intel_adam_step_rows: last[r] = the step row r has been updated through;
 * sched[2*(s - base - 1) + {0, 1}] = (lr / (1 - beta1^s), 1 / sqrt(1 - beta2^s)) of step s, base < s <= base + cap (written
 * by intel_adam_lazy_step).  Whatever a reader then observes -- the rows a forward pass gathers (intel_set_lazy_table),
 * the rows named to intel_adam_lazy_catchup, the whole table after intel_adam_lazy_flush -- is bit-identical to
 * the dense sweep; the table's traffic per step drops from 6 passes over all rows to the touched rows.  d as for
 * intel_adam_step_rows.  The caller owns every array; all hyper-parameters except lr are fixed while rows are pending. */
typedef struct IntelLazyTable {
  float* p; float* m; float* v;        /* [rows, d] parameter and Adam moments */
  int* last;                           /* [rows] */
  float* sched;                        /* [cap, 2] */
  long long rows;
  int d, base, cap;
  float beta1, beta2, eps, weight_decay;
} IntelLazyTable;
int intel_lazy_table_sizeof(void);
/* Step `step` (1-based, base < step <= base + cap) for the rows flagged in row_flags (see intel_set_iid_grad_row_flags): the
 * steps a row missed are replayed, then its gradient row of g is applied, cleared and the flag reset; last[r] = step. */
int intel_adam_lazy_step(const IntelLazyTable* t, float* g, unsigned char* row_flags, float lr, int step, void* stream);
/* Rows ids_a[0..n_a) and ids_b[0..n_b) (ids may repeat, ids outside [0, rows) are skipped, ids_b may be NULL) brought up to
 * step `upto`. */
int intel_adam_lazy_catchup(const IntelLazyTable* t, const int* ids_a, long long n_a, const int* ids_b, long long n_b, int upto,
                            void* stream);
/* Every row brought up to step `upto`. */
int intel_adam_lazy_flush(const IntelLazyTable* t, int upto, void* stream);
/* t != NULL: the item-id gathers of every following intel_forward (i_id_s, his_item_id) deliver the rows AS OF step `upto`: a
 * row that is behind is replayed in registers from its stored (p, m, v); nothing is written back, so a reader that asks
 * for the same stale rows again and again (an evaluation pass) should call intel_adam_lazy_flush once first.  The struct
 * is copied.  t == NULL: off (default). */
int intel_set_lazy_table(IntelCtx* ctx, const IntelLazyTable* t, int upto);

/* ---- data-parallel gradient exchange of a table's touched rows ------------------------------ */
/* out[i,:] = table[idx[i],:] (zeros for idx[i] < 0); zero_rows != 0 also clears those table rows.  idx: a rank's
 * unique touched rows, padded with -1.  No reference counterpart (the reference is single-GPU). */
int intel_rows_take(float* table, int d, const int* idx, int n, float* out, int zero_rows, void* stream);
/* table[idx[i],:] += rows[i,:] for idx[i] >= 0; idx must not repeat within one call (no atomics: called once per
 * source rank, in rank order, so that every replica sums in the same order). */
int intel_rows_add(float* table, int d, const int* idx, int n, const float* rows, void* stream);
/* ABI version 5.  The unique touched rows of a table from the row marks the backward leaves (intel_set_iid_grad_row_flags): idx[0 .. count) = the
 * marked rows in ascending order, idx[count .. cap) = -1 -- a static shape (cap = the batch's id count, equal on every rank) without a sort and
 * without a host-sized result, so the touched-rows exchange is enqueued like any other step.  scratch: intel_rows_compact_scratch_ints(rows) ints.
 * intel_rows_mark sets flags[idx[i]] = 1 (idx[i] >= 0): the rows the OTHER ranks' buffers add into must be visited by the table's Adam sweep too.
 * No reference counterpart (the reference is single-GPU: helpers/BaseRunner.py:279-290). */
long long intel_rows_compact_scratch_ints(long long rows);
int intel_rows_compact(const unsigned char* flags, long long rows, int* idx, int cap, int* scratch, void* stream);
int intel_rows_mark(unsigned char* flags, const int* idx, int n, void* stream);

/* ---- evaluation --------------------------------------------------------------------------- */
/* Overall NDCG@k of BaseRunner.evaluate_method (helpers/BaseRunner.py:117-126) for a padded batch:
 * ndcg[b] per session (linear gains, pads scored 0 / labelled 0, width = max(L, k)). */
int intel_ndcg(int B, int L, int k, const float* ens_score, const int* ranking, const int* session_len,
               float* ndcg, void* stream);
/* Every key of BaseRunner.evaluate_method (helpers/BaseRunner.py:56-131) per session, for n_topk cutoffs (topk: HOST array,
 * n_topk <= 8, cutoffs <= 64):  out[b] = { [behaviour pay, fav, click][cutoff][HR, NDCG], [cutoff] overall NDCG } (7 * n_topk
 * doubles per session), valid[b][3] = that behaviour has positives in session b (the reference averages a behaviour's keys
 * over those sessions only, :96-98; the overall NDCG over all sessions, :117-126).
 *   width      the reference's max_len = max(longest list of the WHOLE evaluation set, largest cutoff) (:66): a session owns
 *              width - session_len pad slots of score 0 that outrank items with a negative score (0: max(L, largest cutoff));
 *   pos_nums   [B,3] pay / fav / click counts of the corpus (:59-63,88-94), or NULL: counted from the labels 3 / 2 / 1;
 *   label_pos  [B,L] slot of item l in the reference's label-descending pre-sort (:78-81), or NULL: the stable form of that
 *              sort (among equal labels the later list position first).  The order among EQUAL labels is numpy's and decides
 *              which items count as "fav positives" when a list holds pay and fav items in different numbers; it depends on
 *              the labels only, so the host computes it once per evaluation set with the reference's own call
 *              (runner.label_positions) and every evaluation afterwards runs on the device.
 * Equal predictions rank the later slot of the label order first (a stable ascending argsort read from its end, :86,117). */
int intel_eval_metrics(int B, int L, int width, int n_topk, const int* topk, const float* ens_score, const int* ranking,
                       const int* session_len, const int* pos_nums, const int* label_pos, double* out, unsigned char* valid,
                       void* stream);

/* ---- input feed ------------------------------------------------------------------------------ */
/* Device-side batch assembly: the work of the reference's per-sample Dataset._get_feed_dict chain
 * (models/BaseModel.py:158-197, models/GeneralSeq.py:35-54, models/IntEL/IntEL.py:220-239) and of
 * collate_batch (models/BaseModel.py:121-142), done by one kernel over a columnar corpus that stays in HBM.
 * The store is the corpus of helpers/BaseReader.py + helpers/SeqReader.py flattened to arrays (CSR for the
 * ragged parts); intel_sigir2023_amd/feed.py builds it.  All pointers are device pointers. */
typedef struct IntelFeedStore {
  int n_sessions, n_users, n_scores, intent_num, max_his, n_intent_rows;
  /* per session [n_sessions] */
  const int* u_id;            /* u_id_c */
  const int* context_mh;      /* combined context feature index (BaseModel.py:163-165) */
  const int* n_pay;           /* c_paynum_i, c_favnum_i, c_clicknum_i, c_trueneg_i: label counts in list order */
  const int* n_fav;
  const int* n_click;
  const int* n_trueneg;
  const int* position;        /* number of earlier sessions of the user (SeqReader.py:43) */
  const int* item_position;   /* number of earlier positive items of the user (SeqReader.py:44) */
  const int* intent_row;      /* row of intent_rows holding the session's intent label (0 = the zero row) */
  const long long* list_off;  /* [n_sessions + 1] offsets into the per-candidate arrays (lists already cut at max_session_len) */
  /* per candidate */
  const int* item_id;         /* i_id_s */
  const int* item_class;      /* i_class_c of the item */
  const double* scores;       /* [n_candidates, n_scores] raw base-ranker scores (min-max normalised per list on the fly) */
  /* intents */
  const float* intent_rows;   /* [n_intent_rows, intent_num]; row 0 is all zeros */
  /* per user, chronological (CSR over u_id) */
  const long long* uhis_off;  /* [n_users + 1] */
  const int* uhis_context_mh; /* context index of each historical session */
  const int* uhis_intent_row; /* its intent row */
  const long long* uitem_off; /* [n_users + 1] */
  const int* uitem_id;        /* positive items in order */
  const int* uitem_intent_idx;/* int(behaviour * I / K + class) of each (IntEL.py:226) */
} IntelFeedStore;

/* Caller-allocated outputs in the layout of IntelBatch (+ the labels the losses read).  L / H / Hi must be at
 * least the batch maxima of session_len / history_len / history_item_len (feed.py computes them on the host). */
typedef struct IntelFeedOut {
  int B, L, H, Hi;
  int* i_id_s;           /* [B, L]    pad 0 */
  int* i_class_c;        /* [B, L]    pad 0 */
  float* scores;         /* [B, L, K] pad 0 */
  int* ranking;          /* [B, L]    3/2/1/0 labels, -1 beyond the labelled prefix, pad 0 (as pad_sequence does) */
  int* session_len;      /* [B] */
  int* u_id_c;           /* [B] */
  int* context_mh;       /* [B] */
  float* intents;        /* [B, I] */
  int* his_context_mh;   /* [B, H]    pad 0 */
  float* his_intents;    /* [B, H, I] pad 0 */
  int* history_len;      /* [B]       1 when the user has no history (one all-zero entry, GeneralSeq.py:49-51) */
  int* his_item_id;      /* [B, Hi]   pad 0 */
  int* his_item_idx;     /* [B, Hi]   intent index of the one-hot row, -1 = all-zero row / pad */
  int* history_item_len; /* [B] */
} IntelFeedOut;

/* shuffle = 0: candidates keep their stored order; 1: per-list permutation from a counter-based RNG keyed by
 * (seed, session index) -- GeneralShuffleModel's per-access shuffle (BaseModel.py:194-196) without the host;
 * 2: perm[b*L + i] gives the stored position that lands at slot i (the reference's np.random.choice draw made
 * on the host, for bit-exact parity).  sess_idx: [B] session indices into the store. */
/* sizeof(IntelFeedStore), sizeof(IntelFeedOut): lets a binding verify its mirror of the two structs. */
void intel_feed_abi_sizes(int* out2);
int intel_feed_collate(const IntelFeedStore* store, const int* sess_idx, int shuffle, const int* perm,
                       unsigned long long seed, const IntelFeedOut* out, void* stream);

/* ---- measurement ---------------------------------------------------------------------------- */
/* Per-kernel HIP-event timing on the launch stream (used by bench.py for the `roofline` block; the
 * reference only has wall-clock _check_time, helpers/BaseRunner.py:174-180).  intel_prof_collect()
 * synchronises the device and returns a JSON object {"kernel": {"launches","ms","flops","bytes"}}. */
void intel_prof_enable(int on);
const char* intel_prof_collect(void);
/* The same records as a timeline instead of per-kernel totals: a JSON array [{"name", "stream", "t0", "t1"}, ...] (ms from the
 * first record, launch order).  With concurrency left on this is the step as its branches really overlap.  Synchronises and
 * clears the records (call it INSTEAD of intel_prof_collect). */
const char* intel_prof_timeline(void);

/* ---- building blocks (exported for unit tests; see tests/test_ops_gpu.py) ------------------ */
/* y[M,N] = x[M,K] @ w[N,K]^T (+bias) (relu) -- torch.nn.Linear. */
int intel_op_linear(const float* x, int M, int K, const float* w, int N, const float* bias, int relu,
                    float* y, void* workspace, size_t workspace_bytes, void* stream);
/* dx[M,K] = dy[M,N] @ w[N,K]. */
int intel_op_linear_dgrad(const float* dy, int M, int N, const float* w, int K, float* dx, void* workspace,
                          size_t workspace_bytes, void* stream);
/* dw[N,K] = dy^T x, db[N] = colsum(dy) (db may be NULL). */
int intel_op_linear_wgrad(const float* dy, const float* x, int M, int N, int K, float* dw, float* db,
                          void* workspace, size_t workspace_bytes, void* stream);
/* The two products of a [d -> d] nn.Linear's backward in ONE pass over the rows (d = 64 / 128, fp32 mode; csrc/pair.hip):
 * dx[M,d] = (dy @ w) [* (x > 0) when relu_mask: x is the relu output that fed the linear], dw[d,d] = dy^T x, db[d] = colsum(dy)
 * -- what torch autograd computes for `y = linear(x)` (models/IntEL/IntEL.py:186-187,195-196) as two separate kernels.
 * workspace: intel_op_linear_bwd_workspace_bytes(M, d) bytes. */
size_t intel_op_linear_bwd_workspace_bytes(int M, int d);
int intel_op_linear_bwd(const float* dy, const float* x, int M, int d, const float* w, int relu_mask, float* dx, float* dw,
                        float* db, void* workspace, size_t workspace_bytes, void* stream);
/* The backward of a fused q/k/v projection (nb = 3; nb = 2: k/v only) in ONE pass over the rows (d = 64 / 128 with nb = 3, d = 128 with nb = 2; fp32 mode;
 * csrc/pair.hip): dy [M, nb*d] = [dq | dk | dv], w [nb*d, d] = the nb weights stacked (torch layout [out, in] each), x [M, d] the projection's input,
 * res [M, d] an optional residual gradient:  dx = dy @ w (+ res),  dw [nb*d, d] = dy^T x,  db [nb*d] = colsum(dy) (db may be NULL)
 * -- torch autograd of q, k, v = Linear(x) x 3 (modules/layers.py:44-48 under loss.backward()).
 * workspace: intel_op_linear_bwd_qkv_workspace_bytes(M, d, nb) bytes. */
size_t intel_op_linear_bwd_qkv_workspace_bytes(int M, int d, int nb);
int intel_op_linear_bwd_qkv(const float* dy, const float* x, const float* res, int M, int d, int nb, const float* w, float* dx, float* dw,
                            float* db, void* workspace, size_t workspace_bytes, void* stream);
/* softmax(QK^T/sqrt(dk)) V per (session, head) on a packed [B*T, 3*d] QKV buffer
 * (modules/layers.py:50-60); key_len NULL = all T rows are keys. */
int intel_op_attention(const float* qkv, int B, int T, int d, int heads, const int* key_len, float* out,
                       float* lse, void* stream);
/* workspace: intel_op_attention_bwd_workspace_bytes(B, T, d, heads) bytes of device memory. */
size_t intel_op_attention_bwd_workspace_bytes(int B, int T, int d, int heads);
int intel_op_attention_bwd(const float* qkv, const float* out, const float* d_out, const float* lse, int B,
                           int T, int d, int heads, const int* key_len, float* d_qkv, float* workspace,
                           void* stream);
/* y = LayerNorm(x + r) (r may be NULL), eps 1e-5; xhat/rstd optional stash. */
int intel_op_add_layernorm(const float* x, const float* r, int M, int N, const float* gamma, const float* beta,
                           float* y, float* xhat, float* rstd, void* stream);
size_t intel_op_workspace_bytes(int M, int N, int K);

#ifdef __cplusplus
}
#endif
#endif /* INTEL_HIP_H */
