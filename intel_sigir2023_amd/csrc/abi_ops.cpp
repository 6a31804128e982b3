// C ABI: error reporting + the building-block entry points of include/intel_hip.h.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/intel_hip.h"
#include "kernels.h"

static thread_local char g_err[512] = "";

void intel_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* intel_last_error(void) { return g_err; }
extern "C" int intel_abi_version(void) { return INTEL_ABI_VERSION; }
// sizeof of the public structs, so that a binding can verify its mirror of the header
extern "C" void intel_abi_sizes(int* out4) {
  out4[0] = (int)sizeof(IntelDesc);
  out4[1] = (int)sizeof(IntelBatch);
  out4[2] = (int)sizeof(IntelOut);
  out4[3] = (int)INTEL_P_COUNT;
}

// workspace: packed weight + wgrad slabs
extern "C" size_t intel_op_workspace_bytes(int M, int N, int K) {
  size_t f = packed_floats(K > N ? K : N, K > N ? K : N) + packed_floats(K, N) + wgrad_slab_floats(M, N, K) + 64;
  return f * sizeof(float);
}

extern "C" int intel_op_linear(const float* x, int M, int K, const float* w, int N, const float* bias, int relu,
                               float* y, void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  INTEL_CHECK_ARG(workspace_bytes >= packed_floats(K, N) * sizeof(float), "op_linear: workspace too small");
  float* Bp = (float*)workspace;
  int rc = launch_pack_b(w, K, K, N, 0, Bp, 0, st);
  if (rc) return rc;
  GemmEpilogue ep;
  ep.bias = bias;
  ep.relu = relu;
  return launch_gemm_rows(x, K, M, K, Bp, N, y, N, ep, st);
}

extern "C" int intel_op_linear_dgrad(const float* dy, int M, int N, const float* w, int K, float* dx, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  INTEL_CHECK_ARG(workspace_bytes >= packed_floats(N, K) * sizeof(float), "op_linear_dgrad: workspace too small");
  float* Bp = (float*)workspace;
  // dx[m][k] = sum_n dy[m][n] w[n][k]: reduction dim = N, B[n][k] = w[n*K + k] (trans form)
  int rc = launch_pack_b(w, K, N, K, 1, Bp, 0, st);
  if (rc) return rc;
  GemmEpilogue ep;
  return launch_gemm_rows(dy, N, M, N, Bp, K, dx, K, ep, st);
}

extern "C" int intel_op_linear_wgrad(const float* dy, const float* x, int M, int N, int K, float* dw, float* db,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  INTEL_CHECK_ARG(workspace_bytes >= wgrad_slab_floats(M, N, K) * sizeof(float), "op_linear_wgrad: workspace too small");
  return launch_wgrad(dy, N, x, K, M, N, K, dw, K, db, 0, (float*)workspace, st);
}

// workspace: fp32 packed transposed weight | its three-plane image | reduce arena
static size_t op_linear_bwd_parts(int M, int d, size_t* pk, size_t* img) {
  *pk = rup_sz(packed_floats(d, d) * sizeof(float), 256);
  *img = rup_sz(packed_b3_bytes(d, d), 256);
  return *pk + *img + (linear_bwd_pair_slab_floats(M, d) + 64) * sizeof(float);
}
extern "C" size_t intel_op_linear_bwd_workspace_bytes(int M, int d) {
  size_t pk, img;
  return op_linear_bwd_parts(M, d, &pk, &img);
}
extern "C" int intel_op_linear_bwd(const float* dy, const float* x, int M, int d, const float* w, int relu_mask, float* dx, float* dw,
                                   float* db, void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  INTEL_CHECK_ARG(dy && x && w && dx && workspace, "op_linear_bwd: null argument");
  INTEL_CHECK_ARG(linear_bwd_pair_supported(M, d), "op_linear_bwd: unsupported shape M=%d d=%d (d = 64 / 128, fp32 mode)", M, d);
  size_t pk, img;
  INTEL_CHECK_ARG(workspace_bytes >= op_linear_bwd_parts(M, d, &pk, &img), "op_linear_bwd: workspace too small");
  float* Bp = (float*)workspace;
  void* B3 = (char*)workspace + pk;
  float* arena = (float*)((char*)workspace + pk + img);
  int rc = launch_pack_b(w, d, d, d, 1, Bp, 0, st);      // dx[m][k] = sum_n dy[m][n] w[n][k]
  if (rc) return rc;
  rc = launch_pack_b3(Bp, d, d, B3, st);
  if (rc) return rc;
  rc = pack_b3_flush(st);
  if (rc) return rc;
  ReduceQueue* q = redq_create();
  INTEL_CHECK_ARG(q != nullptr, "op_linear_bwd: out of memory");
  redq_reset(q, arena, linear_bwd_pair_slab_floats(M, d) + 64);
  rc = launch_linear_bwd_pair(dy, d, x, d, M, d, B3, relu_mask, dx, d, dw, db, 0, 0, q, st);
  if (rc == 0) rc = redq_flush(q, st);
  redq_destroy(q);
  return rc;
}

// workspace: fp32 packed stacked transposed weights | their three-plane image | reduce arena
static size_t op_linear_bwd_qkv_parts(int M, int d, int nb, size_t* pk, size_t* img) {
  *pk = rup_sz(packed_floats(nb * rup(d, 16), d) * sizeof(float), 256);
  *img = rup_sz(packed_b3_bytes(nb * rup(d, 16), d), 256);
  return *pk + *img + (linear_bwd_qkv_slab_floats(M, d, nb) + 64) * sizeof(float);
}
extern "C" size_t intel_op_linear_bwd_qkv_workspace_bytes(int M, int d, int nb) {
  size_t pk, img;
  return op_linear_bwd_qkv_parts(M, d, nb, &pk, &img);
}
extern "C" int intel_op_linear_bwd_qkv(const float* dy, const float* x, const float* res, int M, int d, int nb, const float* w, float* dx, float* dw,
                                       float* db, void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  INTEL_CHECK_ARG(dy && x && w && dx && workspace, "op_linear_bwd_qkv: null argument");
  INTEL_CHECK_ARG(linear_bwd_qkv_supported(M, d, nb), "op_linear_bwd_qkv: unsupported shape M=%d d=%d nb=%d (d = 64 / 128 with 3 blocks, 128 with 2; fp32 mode)", M, d, nb);
  size_t pk, img;
  INTEL_CHECK_ARG(workspace_bytes >= op_linear_bwd_qkv_parts(M, d, nb, &pk, &img), "op_linear_bwd_qkv: workspace too small");
  float* Bp = (float*)workspace;
  void* B3 = (char*)workspace + pk;
  float* arena = (float*)((char*)workspace + pk + img);
  const int nt = rup(d, 16) / 16;
  int rc = 0;
  for (int c = 0; c < nb && !rc; ++c)      // dx[m][k] = sum_c sum_n dy[m][c*d + n] w[c*d + n][k]: the nb weights stacked along the reduction index
    rc = launch_pack_b(w + (size_t)c * d * d, d, d, d, 1, Bp, 0, st, c * nt, nb * nt);
  if (rc) return rc;
  rc = launch_pack_b3(Bp, nb * rup(d, 16), d, B3, st);
  if (rc) return rc;
  rc = pack_b3_flush(st);
  if (rc) return rc;
  ReduceQueue* q = redq_create();
  INTEL_CHECK_ARG(q != nullptr, "op_linear_bwd_qkv: out of memory");
  redq_reset(q, arena, linear_bwd_qkv_slab_floats(M, d, nb) + 64);
  float *pw[3], *pb[3];
  int acc[3] = {0, 0, 0};
  for (int c = 0; c < nb; ++c) {
    pw[c] = dw ? dw + (size_t)c * d * d : nullptr;
    pb[c] = db ? db + (size_t)c * d : nullptr;
  }
  rc = launch_linear_bwd_qkv(dy, nb * d, x, d, res, d, M, d, nb, B3, dx, d, pw, db ? pb : nullptr, acc, q, st);
  if (rc == 0) rc = redq_flush(q, st);
  redq_destroy(q);
  return rc;
}

extern "C" int intel_op_attention(const float* qkv, int B, int T, int d, int heads, const int* key_len, float* out,
                                  float* lse, void* stream) {
  return launch_attn_fwd(qkv, B, T, d, heads, key_len, out, lse, (hipStream_t)stream);
}

extern "C" size_t intel_op_attention_bwd_workspace_bytes(int B, int T, int d, int heads) {
  if (B <= 0 || T <= 0 || d <= 0 || heads <= 0) return 0;
  return attn_bwd_scratch_floats(B, T, d, heads) * sizeof(float);
}

extern "C" int intel_op_attention_bwd(const float* qkv, const float* out, const float* d_out, const float* lse, int B,
                                      int T, int d, int heads, const int* key_len, float* d_qkv, float* workspace,
                                      void* stream) {
  return launch_attn_bwd(qkv, out, d_out, lse, B, T, d, heads, key_len, d_qkv, workspace, (hipStream_t)stream);
}

extern "C" int intel_rows_take(float* table, int d, const int* idx, int n, float* out, int zero_rows, void* stream) {
  INTEL_CHECK_ARG(table && idx && out && d > 0 && n >= 0, "intel_rows_take: bad argument");
  return launch_rows_take(table, d, idx, n, out, zero_rows, (hipStream_t)stream);
}

extern "C" int intel_rows_add(float* table, int d, const int* idx, int n, const float* rows, void* stream) {
  INTEL_CHECK_ARG(table && idx && rows && d > 0 && n >= 0, "intel_rows_add: bad argument");
  return launch_rows_add(table, d, idx, n, rows, (hipStream_t)stream);
}
extern "C" long long intel_rows_compact_scratch_ints(long long rows) { return (long long)rows_compact_scratch_ints(rows); }
extern "C" int intel_rows_compact(const unsigned char* flags, long long rows, int* idx, int cap, int* scratch, void* stream) {
  INTEL_CHECK_ARG(flags && idx && scratch && rows > 0 && cap >= 0, "intel_rows_compact: bad argument");
  return launch_rows_compact(flags, rows, idx, cap, scratch, (hipStream_t)stream);
}
extern "C" int intel_rows_mark(unsigned char* flags, const int* idx, int n, void* stream) {
  INTEL_CHECK_ARG(flags && idx && n >= 0, "intel_rows_mark: bad argument");
  return launch_rows_mark(flags, idx, n, (hipStream_t)stream);
}

extern "C" int intel_op_add_layernorm(const float* x, const float* r, int M, int N, const float* gamma,
                                      const float* beta, float* y, float* xhat, float* rstd, void* stream) {
  return launch_add_layernorm(x, N, r, N, M, N, gamma, beta, y, N, xhat, N, rstd, (hipStream_t)stream);
}

// ---- losses / optimizer / evaluation ------------------------------------------------------------
#include "session.h"

extern "C" size_t intel_loss_workspace_bytes(int B, int L, int K) {
  (void)L; (void)K;
  size_t a = loss_ws_bytes(B), b = intent_ws_bytes(B);
  return (a > b ? a : b) + 256;
}

extern "C" int intel_bpr_loss(int B, int L, int K, const float* ens_score, const int* ranking, const int* session_len,
                              const float* noise, const double* scores_f64, const float* scores_f32,
                              const float* weights, int cal_diversity, double alpha, float grad_scale, float* loss,
                              int* select, float* d_ens, float* d_weights, void* workspace, size_t workspace_bytes,
                              void* stream) {
  INTEL_CHECK_ARG(B > 0 && L > 0 && K > 0 && ens_score && ranking && session_len && loss && workspace, "bpr loss: bad argument");
  return launch_bpr_loss(B, L, K, ens_score, ranking, session_len, noise, scores_f64, scores_f32, weights, cal_diversity, alpha,
                         grad_scale, loss, select, d_ens, d_weights, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int intel_bpr_loss_seeded(int B, int L, int K, const float* ens_score, const int* ranking, const int* session_len,
                                     unsigned long long seed, unsigned long long session0, const double* scores_f64,
                                     const float* scores_f32, const float* weights, int cal_diversity, double alpha, float grad_scale, float* loss,
                                     int* select, float* d_ens, float* d_weights, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  INTEL_CHECK_ARG(B > 0 && L > 0 && K > 0 && ens_score && ranking && session_len && loss && workspace, "bpr loss: bad argument");
  return launch_bpr_loss(B, L, K, ens_score, ranking, session_len, nullptr, scores_f64, scores_f32, weights, cal_diversity, alpha,
                         grad_scale, loss, select, d_ens, d_weights, workspace, workspace_bytes, (hipStream_t)stream, seed, 1, session0);
}

extern "C" int intel_list_loss(int B, int L, int K, const float* ens_score, const int* ranking, const int* session_len,
                               const double* scores_f64, const float* scores_f32, const float* weights,
                               int cal_diversity, double alpha, float grad_scale, float* loss, float* d_ens,
                               float* d_weights, void* workspace, size_t workspace_bytes, void* stream) {
  INTEL_CHECK_ARG(B > 0 && L > 0 && K > 0 && ens_score && ranking && session_len && loss && workspace, "list loss: bad argument");
  return launch_list_loss(B, L, K, ens_score, ranking, session_len, scores_f64, scores_f32, weights, cal_diversity, alpha,
                          grad_scale, loss, d_ens, d_weights, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int intel_mse_loss(int B, int L, int K, const float* ens_score, const int* ranking, const int* session_len,
                              const double* scores_f64, const float* scores_f32, const float* weights,
                              int cal_diversity, double alpha, float grad_scale, float* loss, float* d_ens,
                              float* d_weights, void* workspace, size_t workspace_bytes, void* stream) {
  INTEL_CHECK_ARG(B > 0 && L > 0 && K > 0 && ens_score && ranking && session_len && loss && workspace, "mse loss: bad argument");
  return launch_mse_loss(B, L, K, ens_score, ranking, session_len, scores_f64, scores_f32, weights, cal_diversity, alpha,
                         grad_scale, loss, d_ens, d_weights, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int intel_intent_loss(int B, int I, const float* pred, const double* label, double kl_weight, double kl_temp,
                                 float grad_scale, double* out3, float* d_pred, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  INTEL_CHECK_ARG(B > 0 && I > 0 && pred && label && out3 && workspace, "intent loss: bad argument");
  return launch_intent_loss(B, I, pred, label, kl_weight, kl_temp, grad_scale, out3, d_pred, workspace, workspace_bytes,
                            (hipStream_t)stream);
}

extern "C" int intel_adam_step(float* p, float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                               float eps, float weight_decay, int step, float grad_scale, int zero_grad, void* stream) {
  INTEL_CHECK_ARG(p && g && m && v, "adam: null tensor");
  return launch_adam(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, zero_grad, (hipStream_t)stream);
}

extern "C" int intel_adam_step_pair(float* const* p, float* const* g, float* const* m, float* const* v, const long long* n, const float* weight_decay,
                                    float lr, float beta1, float beta2, float eps, int step, float grad_scale, int zero_grad, void* stream) {
  INTEL_CHECK_ARG(p && g && m && v && n && weight_decay, "adam_pair: null argument");
  for (int k = 0; k < 2; ++k) INTEL_CHECK_ARG(n[k] <= 0 || (p[k] && g[k] && m[k] && v[k]), "adam_pair: null tensor in group %d", k);
  return launch_adam_pair(p, g, m, v, n, weight_decay, lr, beta1, beta2, eps, step, grad_scale, zero_grad, (hipStream_t)stream);
}

extern "C" int intel_lazy_table_sizeof(void) { return (int)sizeof(IntelLazyTable); }

extern "C" int intel_adam_lazy_step(const IntelLazyTable* t, float* g, unsigned char* row_flags, float lr, int step, void* stream) {
  INTEL_CHECK_ARG(t, "adam_lazy: null table");
  return launch_adam_lazy_step(*t, g, row_flags, lr, step, (hipStream_t)stream);
}

extern "C" int intel_adam_lazy_catchup(const IntelLazyTable* t, const int* ids_a, long long n_a, const int* ids_b, long long n_b,
                                       int upto, void* stream) {
  INTEL_CHECK_ARG(t && (ids_a || n_a == 0) && (ids_b || n_b == 0), "adam_lazy_catchup: null table / ids");
  return launch_adam_lazy_ids(*t, ids_a, n_a, ids_b, n_b, upto, (hipStream_t)stream);
}

extern "C" int intel_adam_lazy_flush(const IntelLazyTable* t, int upto, void* stream) {
  INTEL_CHECK_ARG(t, "adam_lazy: null table");
  return launch_adam_lazy_flush(*t, upto, (hipStream_t)stream);
}

extern "C" int intel_adam_step_rows(float* p, float* g, float* m, float* v, long long rows, int d, unsigned char* row_flags,
                                    float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                    float grad_scale, void* stream) {
  INTEL_CHECK_ARG(p && g && m && v && row_flags, "adam_rows: null tensor");
  return launch_adam_rows(p, g, m, v, rows, d, row_flags, lr, beta1, beta2, eps, weight_decay, step, grad_scale, (hipStream_t)stream);
}

extern "C" int intel_ndcg(int B, int L, int k, const float* ens_score, const int* ranking, const int* session_len,
                          float* ndcg, void* stream) {
  INTEL_CHECK_ARG(ens_score && ranking && session_len && ndcg, "ndcg: null tensor");
  return launch_ndcg(B, L, k, ens_score, ranking, session_len, ndcg, (hipStream_t)stream);
}

extern "C" int intel_eval_metrics(int B, int L, int width, int n_topk, const int* topk, const float* ens_score, const int* ranking,
                                  const int* session_len, const int* pos_nums, const int* label_pos, double* out,
                                  unsigned char* valid, void* stream) {
  INTEL_CHECK_ARG(topk && ens_score && ranking && session_len && out && valid, "eval_metrics: null tensor");
  return launch_eval_metrics(B, L, width, n_topk, topk, ens_score, ranking, session_len, pos_nums, label_pos, out, valid, (hipStream_t)stream);
}

extern "C" int intel_loss_total(const float* ensemble_loss, const double* intent_out3, double ensemble_weight, double intent_weight,
                                double* out3, void* stream) {
  INTEL_CHECK_ARG(ensemble_loss && out3, "loss_total: null tensor");
  return launch_loss_total(ensemble_loss, intent_out3, ensemble_weight, intent_weight, out3, (hipStream_t)stream);
}
