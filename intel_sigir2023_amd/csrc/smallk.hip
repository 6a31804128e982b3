// Linear layers whose input width is tiny (K <= 32): score_embeddings (K = model_num, IntEL.py:190) and
// intent_embeddings over the history rows (K = intent_num, IntEL.py:141-147).  A 16-wide MFMA k-step is mostly
// padding there and the rows are not 16-byte aligned (K = 3, 30), so these run on the VALU: lane = output column,
// the row's K inputs are read once by lanes 0..K-1 and broadcast with v_readlane, the weights sit in LDS.
// Both are HBM-bound (the [M, N] side).
#include "kernels.h"
#include "session.h"

#define SK_MAXK 32
#define SK_MAXN 128
#define SK_ROWS 4          // rows per wave trip

namespace {

// y[m, col0 + n] = act(b[n] + sum_k x[m, k] W[n, k])
__global__ __launch_bounds__(256) void linear_smallk_kernel(const float* __restrict__ X, int ldx, int M, int K, const float* __restrict__ W,
                                                            const float* __restrict__ bias, int N, float* __restrict__ Y, int ldy, int relu,
                                                            const float* __restrict__ pos, const int* __restrict__ row_t) {
  __shared__ float wT[SK_MAXK * SK_MAXN];        // [k][n]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < N * K; i += 256) {
    const int n = i / K, k = i - n * K;
    wT[k * N + n] = W[i];
  }
  __syncthreads();
  const int n0 = lane, n1 = lane + 64;
  const float b0 = (bias && n0 < N) ? bias[n0] : 0.f, b1 = (bias && n1 < N) ? bias[n1] : 0.f;
  // a wave takes SK_ROWS rows per trip: their input rows are in flight together and share each weight read
  const int nrg = (M + SK_ROWS - 1) / SK_ROWS;
  for (int rg = blockIdx.x * 4 + wave; rg < nrg; rg += gridDim.x * 4) {
    const int m0 = rg * SK_ROWS;
    float xr[SK_ROWS], a0[SK_ROWS], a1[SK_ROWS];
#pragma unroll
    for (int i = 0; i < SK_ROWS; ++i) {
      xr[i] = (lane < K && m0 + i < M) ? X[(size_t)(m0 + i) * ldx + lane] : 0.f;
      a0[i] = b0;
      a1[i] = b1;
    }
    for (int k = 0; k < K; ++k) {
      const float w0 = wT[k * N + (n0 < N ? n0 : 0)], w1 = wT[k * N + (n1 < N ? n1 : 0)];
#pragma unroll
      for (int i = 0; i < SK_ROWS; ++i) {
        const float xs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xr[i]), k));
        a0[i] = __builtin_fmaf(xs, w0, a0[i]);
        a1[i] = __builtin_fmaf(xs, w1, a1[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < SK_ROWS; ++i) {
      if (m0 + i >= M) break;
      float v0 = a0[i], v1 = a1[i];
      if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
      if (pos) {             // position embedding of a packed history row (pos is offset like Y)
        const float* pr = pos + (size_t)row_t[m0 + i] * ldy;
        if (n0 < N) v0 += pr[n0];
        if (n1 < N) v1 += pr[n1];
      }
      if (n0 < N) Y[(size_t)(m0 + i) * ldy + n0] = v0;
      if (n1 < N) Y[(size_t)(m0 + i) * ldy + n1] = v1;
    }
  }
}

// per-workgroup partials of dW[n, k] = sum_m dY[m, n] x[m, k] and db[n] = sum_m dY[m, n]; slab layout [N*K | N]
template <bool WIDE>
__global__ __launch_bounds__(256) void wgrad_smallk_kernel(const float* __restrict__ dY, int lddy, const float* __restrict__ X, int ldx,
                                                           int M, int N, int K, float* __restrict__ slabs, int want_db) {
  __shared__ float red[3][64 * (SK_MAXK + 1)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float acc0[SK_MAXK], acc1[WIDE ? SK_MAXK : 1];
#pragma unroll
  for (int k = 0; k < SK_MAXK; ++k) acc0[k] = 0.f;
  if (WIDE) {
#pragma unroll
    for (int k = 0; k < SK_MAXK; ++k) acc1[k] = 0.f;
  }
  float db0 = 0.f, db1 = 0.f;
  const int n0 = lane, n1 = lane + 64;
  const int nrg = (M + SK_ROWS - 1) / SK_ROWS;
  for (int rg = blockIdx.x * 4 + wave; rg < nrg; rg += gridDim.x * 4) {
    const int m0 = rg * SK_ROWS;
    float xr[SK_ROWS], d0[SK_ROWS], d1[SK_ROWS];
#pragma unroll
    for (int i = 0; i < SK_ROWS; ++i) {
      const bool ok = m0 + i < M;
      xr[i] = (ok && lane < K) ? X[(size_t)(m0 + i) * ldx + lane] : 0.f;
      d0[i] = (ok && n0 < N) ? dY[(size_t)(m0 + i) * lddy + n0] : 0.f;
      d1[i] = (WIDE && ok && n1 < N) ? dY[(size_t)(m0 + i) * lddy + n1] : 0.f;
      db0 += d0[i];
      db1 += d1[i];
    }
#pragma unroll
    for (int k = 0; k < SK_MAXK; ++k) {
      if (k < K) {               // uniform; a `break` would keep the loop rolled and the accumulators in scratch
#pragma unroll
        for (int i = 0; i < SK_ROWS; ++i) {
          const float xs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xr[i]), k));
          acc0[k] = __builtin_fmaf(d0[i], xs, acc0[k]);
          if (WIDE) acc1[k] = __builtin_fmaf(d1[i], xs, acc1[k]);
        }
      }
    }
  }
  // sum the four waves (fixed order), one column half at a time
  float* slab = slabs + (size_t)blockIdx.x * ((size_t)N * K + N);
#pragma unroll
  for (int half = 0; half < (WIDE ? 2 : 1); ++half) {
    __syncthreads();
    if (wave > 0) {
#pragma unroll
      for (int k = 0; k < SK_MAXK; ++k) red[wave - 1][lane * (SK_MAXK + 1) + k] = half ? acc1[WIDE ? k : 0] : acc0[k];
      red[wave - 1][lane * (SK_MAXK + 1) + SK_MAXK] = half ? db1 : db0;
    }
    __syncthreads();
    if (wave == 0) {
      const int n = half ? n1 : n0;
      if (n < N) {
#pragma unroll
        for (int k = 0; k < SK_MAXK; ++k) {
          if (k < K) {
            float v = half ? acc1[WIDE ? k : 0] : acc0[k];
            v += red[0][lane * (SK_MAXK + 1) + k];
            v += red[1][lane * (SK_MAXK + 1) + k];
            v += red[2][lane * (SK_MAXK + 1) + k];
            slab[(size_t)n * K + k] = v;
          }
        }
        if (want_db) {
          float v = half ? db1 : db0;
          v += red[0][lane * (SK_MAXK + 1) + SK_MAXK];
          v += red[1][lane * (SK_MAXK + 1) + SK_MAXK];
          v += red[2][lane * (SK_MAXK + 1) + SK_MAXK];
          slab[(size_t)N * K + n] = v;
        }
      }
    }
  }
}

}  // namespace

bool smallk_supported(int N, int K) { return K >= 1 && K <= SK_MAXK && N >= 1 && N <= SK_MAXN; }

int launch_linear_smallk(const float* X, int ldx, int M, int K, const float* W, const float* bias, int N, float* Y, int ldy,
                         int relu, hipStream_t st, const float* pos, const int* row_t) {
  if (!row_t) pos = nullptr;
  if (M <= 0) return 0;
  INTEL_CHECK_ARG(smallk_supported(N, K), "linear_smallk: N=%d K=%d unsupported", N, K);
  const int grid = min(cdiv(M, 4), 8 * num_cus());
  LAUNCH_W(2.0 * M * N * K, 4.0 * ((double)M * K + (double)M * N), linear_smallk_kernel, dim3(grid), dim3(256), 0, st, X, ldx, M, K, W, bias, N, Y, ldy,
           relu, pos, row_t);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// one slab per workgroup; a workgroup takes at least 64 rows (16 per wave): with cdiv(M, 4) workgroups a 5 000-row product left 512 slabs
// of a 960-element gradient -- 2 MB of partials whose reduction (one uncoalesced 4-byte read per slab and output) took longer than the product
int smallk_wgrad_slabs(int M) { return max(1, min(cdiv(M, 64), 2 * num_cus())); }

// dW[N,K] (+)= dY^T X, db[N] (+)= colsum(dY) through the reduce queue (or immediately when q == nullptr, using `slabs`)
int launch_wgrad_smallk(const float* dY, int lddy, const float* X, int ldx, int M, int N, int K, float* dW, int lddw, float* db,
                        int accumulate, float* slabs, hipStream_t st, ReduceQueue* q) {
  if (M <= 0) return 0;
  INTEL_CHECK_ARG(smallk_supported(N, K), "wgrad_smallk: N=%d K=%d unsupported", N, K);
  const int S = smallk_wgrad_slabs(M);
  const size_t stride = (size_t)N * K + N;
  if (q) {
    slabs = redq_alloc(q, (size_t)S * stride);
    if (!slabs) {
      intel_set_error("wgrad_smallk: reduction arena exhausted");
      return -2;
    }
  }
  if (N > 64)
    LAUNCH_W(2.0 * M * N * K, 4.0 * ((double)M * K + (double)M * N), wgrad_smallk_kernel<true>, dim3(S), dim3(256), 0, st, dY, lddy, X, ldx, M, N, K, slabs, db != nullptr);
  else
    LAUNCH_W(2.0 * M * N * K, 4.0 * ((double)M * K + (double)M * N), wgrad_smallk_kernel<false>, dim3(S), dim3(256), 0, st, dY, lddy, X, ldx, M, N, K, slabs, db != nullptr);
  INTEL_CHECK_LAUNCH();
  if (q) {
    redq_push(q, slabs, stride, S, N, K, dW, lddw, accumulate);
    if (db) redq_push(q, slabs + (size_t)N * K, stride, S, 1, N, db, N, accumulate);
    return 0;
  }
  int rc = launch_slab_reduce(slabs, stride, S, N, K, dW, lddw, accumulate, st);
  if (rc) return rc;
  if (db) rc = launch_slab_reduce(slabs + (size_t)N * K, stride, S, 1, N, db, N, accumulate, st);
  return rc;
}
