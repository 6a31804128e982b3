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

extern "C" int intel_op_attention(const float* qkv, int B, int T, int d, int heads, const int* key_len, float* out,
                                  float* lse, void* stream) {
  return launch_attn_fwd(qkv, B, T, d, heads, key_len, out, lse, (hipStream_t)stream);
}

extern "C" int intel_op_attention_bwd(const float* qkv, const float* out, const float* d_out, const float* lse, int B,
                                      int T, int d, int heads, const int* key_len, float* d_qkv, float* dsum_ws,
                                      void* stream) {
  return launch_attn_bwd(qkv, out, d_out, lse, B, T, d, heads, key_len, d_qkv, dsum_ws, (hipStream_t)stream);
}

extern "C" int intel_op_add_layernorm(const float* x, const float* r, int M, int N, const float* gamma,
                                      const float* beta, float* y, float* xhat, float* rstd, void* stream) {
  return launch_add_layernorm(x, N, r, N, M, N, gamma, beta, y, N, xhat, N, rstd, (hipStream_t)stream);
}
