// Internal launcher API shared by the kernel files, the model plan (model.cpp) and the C ABI.
// Every launcher enqueues on `st`, never synchronises, returns 0 or an error code.
#pragma once
#include "common.h"

// ---- fp32 MFMA GEMM family (gemm.hip) -------------------------------------------------------
// Packed B operand: tiles of 16 k x 16 n in the register order of v_mfma_f32_16x16x4_f32,
// P[((nt*KG + g)*64 + lane)*4 + s] = B[g*16 + 4*(lane>>4) + s][nt*16 + (lane&15)], zero padded.
static inline size_t packed_floats(int Kd, int Nd) { return (size_t)rup(Kd, 16) * rup(Nd, 16); }
// trans=0: B[k][n] = W[n*ldw + k]  (y = x W^T, W is [Nd,Kd]);  trans=1: B[k][n] = W[k*ldw + n].
// Writes tiles nt_off .. nt_off+ceil(Nd/16)-1 of a packed matrix whose k extent is Kd.
// g_off / KG_total: place this matrix's k groups at offset g_off of a packed matrix with KG_total
// k groups (stacking several weights along k; Kd of each must then be a multiple of 16).
int launch_pack_b(const float* W, int ldw, int Kd, int Nd, int trans, float* P, int nt_off, hipStream_t st,
                  int g_off = 0, int KG_total = 0);

// batch mode: between begin/end launch_pack_b only records jobs; end launches them together
void pack_batch_begin();
int pack_batch_end(hipStream_t st);

struct GemmEpilogue {
  const float* bias = nullptr;    // [N]
  int relu = 0;                   // max(.,0) after bias
  const float* mask = nullptr;    // multiply by (mask[m][n] > 0) after bias/relu (relu backward)
  int ldmask = 0;
  const float* res = nullptr;     // + res[m][n]
  int ldres = 0;
  const float* gamma = nullptr;   // LayerNorm over the N columns (N <= 128) when non-null
  const float* beta = nullptr;
  float* xhat = nullptr;          // optional stash for LayerNorm backward
  int ldxhat = 0;
  float* rstd = nullptr;          // [M]
  int accumulate = 0;             // C += value instead of C = value
  int no_out = 0;                 // LayerNorm epilogue only: keep the x-hat / rstd stash, do not store C
  const void* b3 = nullptr;       // optional pre-split bf16 image of B (launch_pack_b3): used when K > 128
  // bf16 mode (gemm_set_planes(1)), bf16-pipe kernels only: operands stored as bf16 arrays of the same shape and leading dimension
  int a_bf16 = 0;                 // A [M, lda]  (K = 64 / 128, or K > 128 with b3)
  int c_bf16 = 0;                 // C [M, ldc]  (K = 64 / 128)
  int mask_bf16 = 0;              // mask [M, ldmask]  (K = 64 / 128)
};
// arithmetic mode of the matrix-pipe products: 3 = fp32 accuracy (three bf16 planes, six products), 1 = bf16 (one product)
void gemm_set_planes(int planes);
int gemm_planes();
// bf16 three-plane image of a packed fp32 B (any launch_pack_b result) for the K > 128 GEMM on the bf16 pipe
size_t packed_b3_bytes(int Kd, int Nd);
int launch_pack_b3(const float* Pf32, int Kd, int Nd, void* Pb3, hipStream_t st);   // records; pack_b3_flush launches
int pack_b3_flush(hipStream_t st);
// dst[0:n] = src[0:n], recorded; vec_copy_flush launches all recorded copies as one kernel
int launch_vec_copy(const float* src, float* dst, int n, hipStream_t st);
int vec_copy_flush(hipStream_t st);
void pack_jobs_reset();
// C[M,N] = epilogue(A[M,K] @ B) with B packed by launch_pack_b (k extent K, n extent N).
int launch_gemm_rows(const float* A, int lda, int M, int K, const float* Bp, int N, float* C, int ldc,
                     const GemmEpilogue& ep, hipStream_t st);
// dW[N,K] (+)= dY[M,N]^T X[M,K];  db[N] (+)= colsum(dY) when db != null.  `slabs` needs
// wgrad_slab_floats(M,N,K) floats.
size_t wgrad_slab_floats(int M, int N, int K);

// Deferred slab reductions.  A backward pass produces ~50 small [N,K] weight gradients, each as a few
// hundred per-workgroup partial slabs.  Reducing each one right after its producer costs two tiny
// launches per gradient; with a queue the producers only record a job (their slabs live in an arena
// until the flush) and one batched launch per dependency round sums everything in a fixed order.
// Jobs whose destinations overlap (tied layer weights, the shared intent embedding) land in
// successive rounds in push order, so the result does not depend on scheduling.
struct ReduceQueue;
ReduceQueue* redq_create();
void redq_destroy(ReduceQueue* q);
// drop queued jobs and hand the queue a fresh arena
void redq_reset(ReduceQueue* q, float* arena, size_t arena_floats);
// carve `floats` from the arena (256-byte aligned); nullptr when the arena is exhausted
float* redq_alloc(ReduceQueue* q, size_t floats);
// out[r*ldo + c] (+)= sum_s slabs[s*stride + r*cols + c]
void redq_push(ReduceQueue* q, const float* slabs, size_t stride, int S, int rows, int cols, float* out, int ldo,
               int accumulate);
int redq_flush(ReduceQueue* q, hipStream_t st);
// branch-local reductions: jobs pushed after redq_set_tag(q, t > 0) can be reduced ahead of the rest by redq_flush_tag(q, t, stream)
// (their slabs stay allocated until redq_flush); destinations shared between branches must be pushed under tag 0
void redq_set_tag(ReduceQueue* q, int tag);
int redq_flush_tag(ReduceQueue* q, int tag, hipStream_t st);

// with q == nullptr the reduction is launched immediately and `slabs` is used; with a queue the
// partials go to the queue's arena and dW / db are valid only after redq_flush
// split: the N columns of dY are the stacked outputs of several weights that share the input X (fused q/k/v): one
// product, one reduction job per weight (dW[p] is [N/n, K]; db[p] all null or all set).  Needs the queue.
struct WgradSplit { int n; float* dW[4]; float* db[4]; int acc[4]; int dy_bf16 = 0; };   // dy_bf16 (bf16 mode): dY is a bf16 array [M, lddy]
int launch_wgrad(const float* dY, int lddy, const float* X, int ldx, int M, int N, int K, float* dW, int lddw,
                 float* db, int accumulate, float* slabs, hipStream_t st, ReduceQueue* q = nullptr, const WgradSplit* split = nullptr,
                 int io16 = 0);      // bf16 mode, bf16-pipe kernel only: bit 0 dY, bit 1 X stored as bf16 arrays (WgradSplit.dy_bf16 likewise)

// ---- tiny input width (smallk.hip): K <= 32, N <= 128, raw (unpacked) weights W[N, K] ---------------------
bool smallk_supported(int N, int K);
int launch_linear_smallk(const float* X, int ldx, int M, int K, const float* W, const float* bias, int N, float* Y, int ldy,
                         int relu, hipStream_t st, const float* pos = nullptr, const int* row_t = nullptr);      // pos: Y[m, n] += pos[row_t[m] * ldy + n]
int smallk_wgrad_slabs(int M);
// batch scope for SMALL weight-gradient products (rows <= 32 768, 32-wide blocks): between begin and flush launch_wgrad records them, flush issues one
// launch per arithmetic mode.  The operands must stay untouched until the flush; the reduce-queue jobs are pushed at record time as usual.
void wgrad_batch_begin();
int wgrad_batch_flush(hipStream_t st);
void wgrad_batch_reset();      // close the scope and drop what it recorded (entry of every backward: an earlier error exit may have left it open)
int launch_wgrad_smallk(const float* dY, int lddy, const float* X, int ldx, int M, int N, int K, float* dW, int lddw, float* db,
                        int accumulate, float* slabs, hipStream_t st, ReduceQueue* q = nullptr);

// ---- attention (attn.hip) -------------------------------------------------------------------
// qkv: [B*T, 3*d] rows = [q | k | v]; out [B*T, d]; lse [B*heads*T]; key_len optional [B].
// row_off (optional, [B]): PACKED rows -- session b owns rows row_off[b] .. row_off[b] + key_len[b] - 1 of qkv / out / dout /
// dqkv instead of rows b*T .. b*T + T - 1 (no padding rows in memory); needs key_len
bool attn_seq_packed_supported(int T, int dk);
// packed rows (row_off) through launch_attn_fwd / launch_attn_bwd: the general kernels take them at any length, the whole-sequence ones in their fused form
bool attn_packed_supported(int T, int dk);
int launch_attn_fwd(const float* qkv, int B, int T, int d, int heads, const int* key_len, float* out, float* lse,
                    hipStream_t st, const int* row_off = nullptr);
// the general kernels with every product on the bf16 matrix pipe at fp32 accuracy (attn_p3.hip: three-plane operands split once at staging): fp32 mode,
// T > 64, head dim 64 / 128; INTEL_ATTN_P3=0 keeps the exact-fp32 MFMA kernels of attn.hip
bool attn_p3_supported(int T, int dk);
int launch_attn_p3_fwd(const float* qkv, int B, int T, int d, int heads, const int* key_len, float* out, float* lse, hipStream_t st, const int* row_off);
int launch_attn_p3_bwd(const float* qkv, const float* dout, const float* lse, const float* dsum, int B, int T, int d, int heads, const int* key_len,
                       float* dqkv, float* dS, int ldS, hipStream_t st, const int* row_off);
// whole-sequence kernels (attn_seq.hip): T <= 64, head dim 64 / 128
bool attn_seq_supported(int T, int dk);
int launch_attn_seq_fwd(const float* qkv, int B, int T, int d, int heads, const int* key_len, float* out, float* lse,
                        hipStream_t st, const int* row_off = nullptr);
size_t attn_seq_bwd_scratch_floats(int B, int T, int heads);
// h16 (bf16 mode, fused backward only: attn_seq_h16_supported): qkv and dout are read and dqkv written as bf16 arrays of the same shape
bool attn_seq_h16_supported(int T, int dk);
int launch_attn_seq_bwd(const float* qkv, const float* out, const float* dout, const float* lse, int B, int T, int d,
                        int heads, const int* key_len, float* dqkv, float* dS, hipStream_t st, const int* row_off = nullptr, int h16 = 0);
// scratch: attn_bwd_scratch_floats(B, T, d, heads) floats (row sums of dO*O; dS tiles of the whole-sequence path)
size_t attn_bwd_scratch_floats(int B, int T, int d, int heads);
int launch_attn_bwd(const float* qkv, const float* out, const float* dout, const float* lse, int B, int T, int d,
                    int heads, const int* key_len, float* dqkv, float* scratch, hipStream_t st, const int* row_off = nullptr, int h16 = 0);

// ---- one tower layer in one kernel (tower.hip) -------------------------------------------------
// L <= 64, d in {64, 128}, head dim in {32, 64, 128}; INTEL_FUSE_TOWER=0 turns the fused path off
bool tower_fused_supported(int L, int d, int heads);
// the policy (INTEL_FUSE_TOWER unset): inference always, training in bf16 mode only (tower.hip)
bool tower_fused_wanted(int train, int d);
// W*_b3: bf16 three-plane images (launch_pack_b3) of the packed [d -> 3d] / [d -> d] / [d -> d] weights.  out may be NULL
// (x-hat / rstd only); train = 0: none of the stash pointers is written.
// in (inference, X == NULL): the first layer's input built inside the kernel -- two row gathers side by side (tab0[idx0] | tab1[idx1]: the item tower's
// id / class embeddings, IntEL.py:170-173); the [B*L, d] input tensor then never exists in HBM
struct TowerInput {
  const float* tab0 = nullptr; const int* idx0 = nullptr; int d0 = 0;      // columns 0 .. d0-1 (idx < 0: a zero row)
  const float* tab1 = nullptr; const int* idx1 = nullptr;                  // columns d0 .. d-1 (unused when d0 == d)
};
int launch_tower_fwd_fused(const float* X, int B, int L, int d, int heads, const void* Wqkv_b3, const void* W1_b3, const void* W2_b3,
                           const float* b1, const float* b2, const float* gamma, const float* beta, float* out, int train,
                           float* QKV, float* A, float* LSE, float* R1, float* XH, float* RSTD, hipStream_t st, int qkv16 = 0, const TowerInput* in = nullptr);

// ---- the middle of one tower layer's backward in one kernel (tower_bwd.hip): dZ -> dQKV, with dW1 / db1 / dW2 / db2 accumulated in the workgroup ----
// L <= 64, d in {64, 128}, head dim in {64, 128}; needs the forward's A (attention output) and LSE stashes and the layer input X; the q/k/v weight
// gradient and dX = dQKV Wqkv + dZ stay on the GEMM kernels.  INTEL_FUSE_TOWER_BWD=0 turns the path off, =1 forces it wherever supported.
bool tower_bwd_fused_supported(int L, int d, int heads);
bool tower_bwd_fused_wanted(int d);      // the policy (INTEL_FUSE_TOWER_BWD unset): bf16 mode only (tower_bwd.hip)
size_t tower_bwd_slab_floats(int B, int d);      // arena floats one launch takes from the reduce queue
// What one launch covers beyond dQKV (by width and arithmetic mode): bit 0 = dX = dQKV Wqkv + dZ too, bit 1 = the q/k/v weight gradients too (dQKV is
// then not written at all).  64-wide: 3; 128-wide: 1 in bf16 mode, 0 in fp32 (six-plane operands + 64 accumulator registers leave no room).
int tower_bwd_fused_scope(int d);
// W*_b3: three-plane images (launch_pack_b3) of the packed forward [d -> 3d] / [d -> d] weights, of the TRANSPOSED feed-forward weights and (scope
// bit 0) of the stacked transposed q/k/v weights [3d -> d]; grads / accumulate [7]: dW2, db2, dW1, db1, dWq, dWk, dWv (valid after the queue's flush;
// NULL = not wanted; the last three only with scope bit 1); a16 / dqkv16 (bf16 mode): A read / dQKV written as bf16 arrays
int launch_tower_bwd_fused(const float* X, const float* A, const float* LSE, const float* dZ, int B, int L, int d, int heads, const void* Wqkv_b3,
                           const void* W1_b3, const void* W2T_b3, const void* W1T_b3, const void* WqkvT_b3, const float* b1, float* dQKV, float* dX,
                           float* const* grads, const int* accumulate, ReduceQueue* q, hipStream_t st, int a16 = 0, int dqkv16 = 0);

// ---- one nn.Linear's backward in one pass over its rows (pair.hip): data gradient + weight gradient from ONE staging of dY ----
// d x d linears of the towers (d = 64 / 128), fp32 mode (six plane products):  dXout[M, d] = (dY WT) [* (X > 0)],  dW[d, d] (+)= dY^T X,
// db[d] (+)= colsum(dY).  WT_b3: the three-plane image (launch_pack_b3) of the packed TRANSPOSED weight (launch_pack_b(.., trans = 1)).  dW / db are valid
// after the queue's flush (NULL = not wanted).  INTEL_PAIR_BWD=0 turns the path off.
bool linear_bwd_pair_supported(int M, int d);
size_t linear_bwd_pair_slab_floats(int M, int d);      // arena floats one launch takes from the reduce queue
int launch_linear_bwd_pair(const float* dY, int lddy, const float* X, int ldx, int M, int d, const void* WT_b3, int relu_mask, float* dXout, int ldo,
                           float* dW, float* db, int acc_w, int acc_b, ReduceQueue* q, hipStream_t st);

// the fused q/k/v projection's backward in one pass (pair.hip): dXout[M, d] = dY[M, nb*d] WT (+ res), dW[c][d, d] (+)= dY[:, c*d:(c+1)*d]^T X, db[c] (+)= colsum
// (nb = 3: [dQ | dK | dV]; nb = 2 at d = 128: the pruned last encoder block's [dK | dV]).  WT_b3: the three-plane image of the stacked transposed weights
// (k extent nb*d, n extent d: what gemm_rows_b3k takes).  dW / db: nb pointers each (NULL entries = not wanted; db may be NULL); valid after the queue's flush
bool linear_bwd_qkv_supported(int M, int d, int nb);
size_t linear_bwd_qkv_slab_floats(int M, int d, int nb);
int launch_linear_bwd_qkv(const float* dY, int lddy, const float* X, int ldx, const float* res, int ldr, int M, int d, int nb, const void* WT_b3, float* dXout, int ldo,
                          float* const* dW, float* const* db, const int* acc, ReduceQueue* q, hipStream_t st);

// ---- the whole tied tower at the reference's own 32-wide shapes, one kernel per direction (tower32.hip) ---------------------------
// d = 32, 1-2 heads, L <= 128, any number of tied layers; raw (unpacked) reference weights W [32, 32], vectors [32].  No activation stash:
// the backward recomputes the forward from the tower input.  INTEL_TOWER32=0 turns the path off.
bool tower32_supported(int L, int d, int heads, int layers, int train);
size_t tower32_slab_floats(int B);      // arena floats one launch_tower32_bwd takes from the reduce queue
// training-mode nn.Dropout of the tower layers (IntEL.py:187,196): the draw of launch_dropout_mask (same seed, stream id stream0 + layer) or
// external keep flags [layers, B*L, 32]
struct Tower32Dropout { float p; unsigned long long seed; unsigned stream0; const float* ext; };
int launch_tower32_fwd(const float* X, int B, int L, int heads, int layers, const float* Wq, const float* Wk, const float* Wv, const float* W1,
                       const float* b1, const float* W2, const float* b2, const float* gamma, const float* beta, float* out, hipStream_t st,
                       const Tower32Dropout* drop = nullptr);
// grads / accumulate: dWq, dWk, dWv, dW1, db1, dW2, db2, dgamma, dbeta (NULL = not wanted); valid after the queue's flush
int launch_tower32_bwd(const float* X, const float* dout, int B, int L, int heads, int layers, const float* Wq, const float* Wk, const float* Wv,
                       const float* W1, const float* b1, const float* W2, const float* b2, const float* gamma, const float* beta, float* dX,
                       float* const* grads, const int* accumulate, ReduceQueue* q, hipStream_t st, const Tower32Dropout* drop = nullptr, int share8 = 8);      // share8: eighths of the CUs a small batch's grid may take

// BERT4Rec at the reference's default widths (dm = 32, <= 2 blocks, 1-2 heads, histories of <= 32 events): the whole encoder of a session
// history as one kernel per direction (tower32.hip: enc32_*).  X: [rows, 32] input rows incl. the position embedding, packed (off) or
// padded [B, T]; out[b * ldo + c]: row len-1 of the last block.  INTEL_ENC32=0 turns the path off.
struct Enc32Block { const float *Wq, *bq, *Wk, *bk, *Wv, *bv, *g1, *be1, *W1, *b1, *W2, *b2, *g2, *be2; };
bool enc32_supported(int T, int dm, int heads, int layers, int train);
bool enc32_batch_ok(int B, int train);      // training: one slab per session and block, batches of <= 1024 sessions
size_t enc32_slab_floats(int B, int layers);
int launch_enc32_fwd(const float* X, const int* off, const int* len, int B, int T, int heads, int layers, const Enc32Block* blk, float* out, int ldo,
                     hipStream_t st);
int launch_enc32_bwd(const float* X, const int* off, const int* len, int B, int T, int heads, int layers, const Enc32Block* blk, const float* dout,
                     int ldd, float* dX, float* const (*grads)[14], const int (*accumulate)[14], ReduceQueue* q, hipStream_t st);

// ---- row / session kernels (rowops.hip) -----------------------------------------------------
// dst[m, col0:col0+d] = table[idx[m], :]  (idx<0 -> zeros); optional relu
int launch_gather_rows(const float* table, int d, const int* idx, int M, float* dst, int ldd, int col0, int relu,
                       hipStream_t st, const float* pos = nullptr, const int* row_t = nullptr);
// dst[b*T+t, col0:col0+d] = src[b, :]  broadcast of a per-session vector
int launch_bcast_rows(const float* src, int lds, int d, int B, int T, float* dst, int ldd, int col0, hipStream_t st);
// grad_table[idx[m], :] += src[m, col0:col0+d] (* (relu_src>0) if relu_src given) ; atomics
int launch_scatter_add_rows(const float* src, int lds, int col0, int d, const int* idx, int M, float* grad_table,
                            const float* relu_out, int ldr, int rcol0, hipStream_t st, unsigned char* row_flags = nullptr);
// the same with (id, source row) pairs sorted by id: runs of equal ids are summed in registers before the atomics
// (row_off / len / T: the source rows are packed history rows, pair row b*T + t -> source row row_off[b] + t)
int launch_scatter_add_sorted(const float* src, int lds, int col0, int d, const int* sorted_ids, const int* sorted_rows, int n,
                              float* grad_table, hipStream_t st, unsigned char* row_flags = nullptr, const int* row_off = nullptr,
                              const int* len = nullptr, int T = 0);
// y = LN(x + r) rows
int launch_add_layernorm(const float* x, int ldx, const float* r, int ldr, int M, int N, const float* gamma,
                         const float* beta, float* y, int ldy, float* xhat, int ldxh, float* rstd, hipStream_t st,
                         const float* xscale = nullptr);   // xscale: optional elementwise factor on x (dropout mask)
// dropout keep mask scaled by 1/(1-p); ext = optional 0/1 keep flags, else a counter-based generator
int launch_dropout_mask(float* mask, long long n, float p, unsigned long long seed, unsigned stream_id, const float* ext, hipStream_t st);
int launch_mul2(const float* a, const float* b, long long n, float* y, hipStream_t st);
// dz = LN backward; dgamma/dbeta (+)= column sums (via slabs: needs ln_bwd_slab_floats(M,N))
size_t ln_bwd_slab_floats(int M, int N);
int launch_layernorm_bwd(const float* dy, int lddy, const float* xhat, int ldxh, const float* rstd, int M, int N,
                         const float* gamma, float* dz, int lddz, float* dgamma, float* dbeta, int accumulate,
                         float* slabs, hipStream_t st, ReduceQueue* q = nullptr);
// touched-row exchange of a gradient table (data parallel): see rowops.hip
int launch_rows_take(float* table, int d, const int* idx, int n, float* out, int zero_rows, hipStream_t st);
int launch_rows_add(float* table, int d, const int* idx, int n, const float* rows, hipStream_t st);
// idx[0 .. count) = the rows marked in flags (ascending), the rest of idx[0 .. cap) = -1; scratch: rows_compact_scratch_ints(rows) ints
size_t rows_compact_scratch_ints(long long rows);
int launch_rows_compact(const unsigned char* flags, long long rows, int* idx, int cap, int* scratch, hipStream_t st);
int launch_rows_mark(unsigned char* flags, const int* idx, int n, hipStream_t st);
// softmax over rows of length N (in place allowed)
int launch_softmax_rows(const float* x, int M, int N, float* y, hipStream_t st);
// dx = y * (dy - sum(dy*y))
int launch_softmax_rows_bwd(const float* y, const float* dy, int M, int N, float* dx, hipStream_t st);
// generic elementwise: y = a (+ b)
int launch_add2(const float* a, const float* b, long long n, float* y, hipStream_t st);
int launch_fill(float* p, long long n, float v, hipStream_t st);
int launch_adam_pair(float* const* p, float* const* g, float* const* m, float* const* v, const long long* n, const float* wd, float lr, float beta1,
                     float beta2, float eps, int step, float grad_scale, int zero_grad, hipStream_t st);
