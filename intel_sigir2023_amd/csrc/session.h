// Launchers of session.hip / loss.hip / optim.hip (internal).
#pragma once
#include "common.h"

struct ReduceQueue;

int launch_xatt_pool_fwd(const float* X, int B, int L, int d, const float* qk, const int* slen, float scale,
                         float* xbar, float* attw, hipStream_t st, const float* gamma = nullptr,
                         const float* beta = nullptr);
// the tower's last LayerNorm folded into the pooling: X given as its x-hat stash (forward: gamma/beta above; backward below)
bool xatt_ln_fused_supported(int L, int d);
size_t xatt_ln_bwd_slab_floats(int B, int d);
int launch_xatt_pool_ln_bwd(const float* XH, const float* rstd, const float* gamma, const float* beta, int B, int L, int d,
                            const float* qk, const float* attw, const float* dxbar, int ldxb, float scale, float* dZ, float* dqk,
                            float* dgamma, float* dbeta, int accumulate, hipStream_t st, ReduceQueue* q);
int launch_xatt_pool_bwd(const float* X, int B, int L, int d, const float* qk, const float* attw, const float* dxbar,
                         int ldxb, float scale, float* dX, float* dqk, hipStream_t st);
int launch_ens_fwd(const float* wv, const float* wpad, const float* scores, const int* slen, int B, int L, int K,
                   int per_item, float* weights, float* ens, hipStream_t st);
int launch_ens_bwd(const float* d_weights, const float* d_ens, const float* scores, const int* slen, int B, int L, int K,
                   int per_item, float* dwv, float* dwpad, float* dwt, hipStream_t st);
int launch_gate_fwd(const float* x, int d, const float* vec, int B, int L, float* dst, int ldd, int col0, hipStream_t st);
int launch_gate_bwd(const float* dfeat, int ldf, int col0, const float* x, int d, const float* vec, int B, int L, float* dx,
                    float* dvec, hipStream_t st);
int launch_gate_mean_fwd(const float* x, int d, const float* vec, int B, int L, float* xbar, float* feat, int ldf, int col0, hipStream_t st);
int launch_gate_mean_bwd(const float* dfeat, int ldf, int col0, const float* xbar, int d, const float* vec, int B, int L, float* dx,
                         float* dvec, hipStream_t st);
int launch_session_colsum(const float* src, int lds, int col0, int d, int B, int L, float* out, int ldo, int ocol0,
                          int accumulate, hipStream_t st);
int launch_add_pos(float* E, int dm, const float* pos, const int* len, int B, int T, hipStream_t st);
int launch_onehot_linear(const float* W, const float* bias, int d_int, int I, const int* idx, int M, float* E, int lde,
                         int col0, hipStream_t st, const float* pos = nullptr, const int* row_t = nullptr);
int launch_make_onehot(const int* idx, const int* len, int T, int M, int R, float* oh, hipStream_t st);
// row_off (optional, [B]): packed history rows (kernels.h)
int launch_attn_lastq_fwd(const float* kv, const float* q, const int* len, int B, int T, int dm, int heads, float* out,
                          float* P, hipStream_t st, const int* row_off = nullptr);
int launch_attn_lastq_bwd(const float* kv, const float* q, const float* P, const float* d_out, const int* len, int B, int T,
                          int dm, int heads, float* dq, float* dkv, hipStream_t st, const int* row_off = nullptr);
int launch_add_at_last(const float* src, int lds, int dm, const int* len, int B, int T, float* dX, hipStream_t st,
                       const int* row_off = nullptr);
int launch_select_last(const float* E, int dm, const int* len, int B, int T, float* out, int ldo, int col0, hipStream_t st,
                       const int* row_off = nullptr);
int launch_add_pos_rows(float* E, int dm, const float* pos, const int* row_t, int rows, hipStream_t st);
bool pos_grad_supported(int T, int dm);
int pos_grad_slabs(int rows);
int launch_pos_grad(const float* dE, int dm, const int* row_t, const int* len, int T, int rows, float* dpos, hipStream_t st, ReduceQueue* q,
                    const int* off = nullptr, int B = 0);      // off (packed rows: first row of every session): the atomic-free form
int launch_his_pack(const int* len, const int* off, int B, int T, const int* ids, int* ids_out, const int* idx2, int* idx2_out,
                    const float* vec, int w, float* vec_out, int* row_t, hipStream_t st);
int launch_copy_cols(const float* src, int lds, int scol0, int d, long long M, float* dst, int ldd, int dcol0,
                     const float* relu_out, int ldr, int rcol0, int accumulate, hipStream_t st);
int launch_slab_reduce(const float* slabs, size_t stride, int S, int rows, int cols, float* out, int ldo,
                       int accumulate, hipStream_t st);

// loss.hip
size_t loss_ws_bytes(int B);
size_t intent_ws_bytes(int B);
int launch_bpr_loss(int B, int L, int K, const float* ens, const int* ranking, const int* slen, const float* noise,
                    const double* sc64, const float* sc32, const float* weights, int cal_div, double alpha,
                    float grad_scale, float* loss, int* select, float* d_ens, float* d_weights, void* ws, size_t ws_bytes,
                    hipStream_t st, unsigned long long seed = 0, int use_seed = 0, unsigned long long session0 = 0);
int launch_list_loss(int B, int L, int K, const float* ens, const int* ranking, const int* slen, const double* sc64,
                     const float* sc32, const float* weights, int cal_div, double alpha, float grad_scale, float* loss,
                     float* d_ens, float* d_weights, void* ws, size_t ws_bytes, hipStream_t st);
int launch_mse_loss(int B, int L, int K, const float* ens, const int* ranking, const int* slen, const double* sc64,
                    const float* sc32, const float* weights, int cal_div, double alpha, float grad_scale, float* loss,
                    float* d_ens, float* d_weights, void* ws, size_t ws_bytes, hipStream_t st);
int launch_loss_total(const float* loss_e, const double* out3_int, double w_e, double w_i, double* out, hipStream_t st);
int launch_intent_loss(int B, int I, const float* pred, const double* label, double kl_weight, double kl_temp,
                       float grad_scale, double* out3, float* d_pred, void* ws, size_t ws_bytes, hipStream_t st);
// optim.hip
int launch_adam(float* p, float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                float wd, int step, float grad_scale, int zero_grad, hipStream_t st);
int launch_adam_rows(float* p, float* g, float* m, float* v, long long rows, int d, unsigned char* row_flags, float lr,
                     float beta1, float beta2, float eps, float wd, int step, float grad_scale, hipStream_t st);
// lazy form of the table's dense Adam (IntelLazyTable: include/intel_hip.h)
struct IntelLazyTable;
int launch_adam_lazy_step(const IntelLazyTable& t, float* g, unsigned char* row_flags, float lr, int step, hipStream_t st);
int launch_adam_lazy_ids(const IntelLazyTable& t, const int* ids_a, long long n_a, const int* ids_b, long long n_b, int upto, hipStream_t st);
int launch_adam_lazy_flush(const IntelLazyTable& t, int upto, hipStream_t st);
int launch_gather_rows_lazy(const IntelLazyTable& t, int upto, const int* idx, int M, float* dst, int ldd, int col0, hipStream_t st,
                            const float* pos, const int* row_t);
int launch_ndcg(int B, int L, int k, const float* ens, const int* ranking, const int* slen, float* out, hipStream_t st);
int launch_eval_metrics(int B, int L, int width, int nk, const int* topk, const float* ens, const int* ranking, const int* slen,
                        const int* pos_nums, const int* label_pos, double* out, unsigned char* valid, hipStream_t st);
