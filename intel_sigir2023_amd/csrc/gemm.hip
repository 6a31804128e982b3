// fp32 MFMA GEMM family for the IntEL path (gfx950, v_mfma_f32_16x16x4_f32: exact fp32 = fmaf chain).
//
//  pack_b       weights -> MFMA-fragment order (one 16-byte load per lane feeds 4 MFMAs)
//  gemm_rows    C[M,N] = epilogue(A[M,K] @ B): torch.nn.Linear forward and its data gradient,
//               with bias / relu / relu-mask / residual / LayerNorm fused in the epilogue
//  wgrad        dW[N,K] = dY^T X, db = colsum(dY): split over row slabs + deterministic reduce
//
// Tiling (64-lane waves): a 256-thread workgroup owns 64 rows of A staged in LDS (row stride
// K+4 floats so that the 16-byte fragment reads of a 16-lane group hit distinct banks) and up to
// 128 output columns; wave w owns column tiles {w, w+4} x all four 16-row tiles (8 accumulators).
// The k index inside a 16-wide k group is permuted (lane group j holds k = 4j..4j+3) identically
// in A fragments and in the packed B, which is what lets both sides use 16-byte loads.
#include "kernels.h"

// ------------------------------------------------------------------------------------------
// pack
// ------------------------------------------------------------------------------------------
__global__ void pack_b_kernel(const float* __restrict__ W, int ldw, int Kd, int Nd, int trans,
                              float* __restrict__ P, int nt_off, int KG, int NT, int g_off, int KG_total) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (nt, g, lane)
  int total = NT * KG * 64;
  if (idx >= total) return;
  int lane = idx & 63;
  int g = (idx >> 6) % KG;
  int nt = (idx >> 6) / KG;
  int n = nt * 16 + (lane & 15);
  f32x4 v;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    int k = g * 16 + 4 * (lane >> 4) + s;
    float x = 0.f;
    if (k < Kd && n < Nd) x = trans ? W[(size_t)k * ldw + n] : W[(size_t)n * ldw + k];
    v[s] = x;
  }
  *reinterpret_cast<f32x4*>(P + ((size_t)((nt_off + nt) * KG_total + g_off + g) * 64 + lane) * 4) = v;
}

int launch_pack_b(const float* W, int ldw, int Kd, int Nd, int trans, float* P, int nt_off, hipStream_t st, int g_off,
                  int KG_total) {
  int KG = rup(Kd, 16) / 16, NT = rup(Nd, 16) / 16;
  int total = NT * KG * 64;
  if (KG_total <= 0) KG_total = KG;
  hipLaunchKernelGGL(pack_b_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, W, ldw, Kd, Nd, trans, P, nt_off, KG, NT, g_off, KG_total);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------
// gemm_rows
// ------------------------------------------------------------------------------------------
#define GR_BM 64
#define GR_KC 128
#define GR_LDA (GR_KC + 4)
#define GR_LDE (128 + 4)

struct GemmRowsArgs {
  const float* A; int lda; int M; int K;
  const float* Bp; int N;
  float* C; int ldc;
  GemmEpilogue ep;
};

__global__ __launch_bounds__(256) void gemm_rows_kernel(GemmRowsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                       // [64][GR_LDA]
  float* Es = smem + GR_BM * GR_LDA;      // [64][GR_LDE]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * GR_BM;
  const int Kp = (a.K + 15) & ~15, KG = Kp >> 4;
  const int NT = (a.N + 15) >> 4;
  const bool vecA = ((a.lda & 3) == 0) && ((a.K & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.A) & 15) == 0);
  const GemmEpilogue& ep = a.ep;

  for (int nc = 0; nc < NT; nc += 8) {
    const int ntc = min(8, NT - nc);
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kc = 0; kc < Kp; kc += GR_KC) {
      const int kcl = min(GR_KC, Kp - kc);
      if (!(Kp <= GR_KC && nc > 0)) {
        __syncthreads();
        if (vecA) {
          const int c4n = kcl >> 2;   // float4 per row
          for (int i = tid; i < GR_BM * c4n; i += 256) {
            int r = i / c4n, c4 = i - r * c4n;
            int row = m0 + r, col = kc + c4 * 4;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < a.M && col < a.K) v = *reinterpret_cast<const f32x4*>(a.A + (size_t)row * a.lda + col);
            *reinterpret_cast<f32x4*>(As + r * GR_LDA + c4 * 4) = v;
          }
        } else {
          for (int i = tid; i < GR_BM * kcl; i += 256) {
            int r = i / kcl, c = i - r * kcl;
            int row = m0 + r, col = kc + c;
            float v = 0.f;
            if (row < a.M && col < a.K) v = a.A[(size_t)row * a.lda + col];
            As[r * GR_LDA + c] = v;
          }
        }
        __syncthreads();
      }
      const int ng = kcl >> 4;
      const int gbase = kc >> 4;
      for (int g = 0; g < ng; ++g) {
        f32x4 af[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
          af[rt] = *reinterpret_cast<const f32x4*>(As + (rt * 16 + (lane & 15)) * GR_LDA + g * 16 + 4 * (lane >> 4));
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int ct = wave + 4 * c;
          if (ct < ntc) {
            const f32x4 bf = *reinterpret_cast<const f32x4*>(a.Bp + ((size_t)((nc + ct) * KG + gbase + g) * 64 + lane) * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int rt = 0; rt < 4; ++rt) acc[rt][c] = mfma16(af[rt][s], bf[s], acc[rt][c]);
          }
        }
      }
    }
    // ---- epilogue: accumulators -> LDS -> row-wise processing, coalesced stores
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int ct = wave + 4 * c;
      if (ct < ntc) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) Es[(rt * 16 + 4 * (lane >> 4) + r) * GR_LDE + ct * 16 + (lane & 15)] = acc[rt][c][r];
      }
    }
    __syncthreads();
    const int n0 = nc * 16;
    const int ncols = min(128, a.N - n0);
    for (int rr = 0; rr < 16; ++rr) {
      const int r = wave * 16 + rr;
      const int row = m0 + r;
      if (row >= a.M) break;
      float v[2];
      bool ok[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int cl = lane + 64 * h;
        ok[h] = cl < ncols;
        float x = 0.f;
        if (ok[h]) {
          const int col = n0 + cl;
          x = Es[r * GR_LDE + cl];
          if (ep.bias) x += ep.bias[col];
          if (ep.relu) x = fmaxf(x, 0.f);
          if (ep.mask) x = (ep.mask[(size_t)row * ep.ldmask + col] > 0.f) ? x : 0.f;
          if (ep.res) x += ep.res[(size_t)row * ep.ldres + col];
        }
        v[h] = x;
      }
      if (ep.gamma) {   // LayerNorm over the row (host guarantees N <= 128 -> single chunk)
        const float inv_n = 1.f / (float)a.N;
        const float mean = wave_sum(v[0] + v[1]) * inv_n;
        const float d0 = ok[0] ? v[0] - mean : 0.f, d1 = ok[1] ? v[1] - mean : 0.f;
        const float var = wave_sum(d0 * d0 + d1 * d1) * inv_n;
        const float rs = 1.f / sqrtf(var + 1e-5f);
        if (ep.rstd && lane == 0) ep.rstd[row] = rs;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (ok[h]) {
            const int col = n0 + lane + 64 * h;
            const float xh = (h ? d1 : d0) * rs;
            if (ep.xhat) ep.xhat[(size_t)row * ep.ldxhat + col] = xh;
            v[h] = xh * ep.gamma[col] + ep.beta[col];
          }
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (ok[h]) {
          float* dst = a.C + (size_t)row * a.ldc + n0 + lane + 64 * h;
          *dst = ep.accumulate ? (*dst + v[h]) : v[h];
        }
      }
    }
    __syncthreads();
  }
}

int launch_gemm_rows(const float* A, int lda, int M, int K, const float* Bp, int N, float* C, int ldc,
                     const GemmEpilogue& ep, hipStream_t st) {
  if (M <= 0 || N <= 0) return 0;
  INTEL_CHECK_ARG(K > 0, "gemm_rows: K must be positive");
  INTEL_CHECK_ARG(!(ep.gamma && N > 128), "gemm_rows: fused LayerNorm needs N <= 128 (got %d)", N);
  GemmRowsArgs a;
  a.A = A; a.lda = lda; a.M = M; a.K = K; a.Bp = Bp; a.N = N; a.C = C; a.ldc = ldc; a.ep = ep;
  size_t smem = (size_t)(GR_BM * GR_LDA + GR_BM * GR_LDE) * sizeof(float);
  allow_lds(gemm_rows_kernel, smem);
  hipLaunchKernelGGL(gemm_rows_kernel, dim3(cdiv(M, GR_BM)), dim3(256), smem, st, a);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------
// wgrad: dW[N,K] = sum_m dY[m,n] X[m,k]; db[n] = sum_m dY[m,n]
// ------------------------------------------------------------------------------------------
#define WG_RT 32          // rows per LDS tile
#define WG_MAXS 256       // max slabs

static inline int wgrad_num_slabs(int M) {
  int s = cdiv(M, 4 * WG_RT);
  return s < 1 ? 1 : (s > WG_MAXS ? WG_MAXS : s);
}
size_t wgrad_slab_floats(int M, int N, int K) { return (size_t)wgrad_num_slabs(M) * ((size_t)N * K + N); }

struct WgradArgs {
  const float* dY; int lddy; const float* X; int ldx; int M, N, K;
  float* slabs; int S; int want_db;
};

// grid: (S, ceil(N/128), ceil(K/128)); each block: rows tiles s, s+S, ... ; wave (wn,wk) owns a
// 4x4 block of 16x16 output tiles.
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.y * 128, k0 = blockIdx.z * 128;
  const int nb = min(128, a.N - n0), kb = min(128, a.K - k0);
  const int NTb = (nb + 15) >> 4, KTb = (kb + 15) >> 4;
  const int ldy = ((NTb * 16 + 31) & ~31) + 16, ldx = ((KTb * 16 + 31) & ~31) + 16;
  float* Ys = smem;                 // [WG_RT][ldy]
  float* Xs = smem + WG_RT * ldy;   // [WG_RT][ldx]
  const int wn = wave >> 1, wk = wave & 1;
  const int ntw = (NTb + 1) >> 1, ktw = (KTb + 1) >> 1;     // tiles per wave along n / k (<= 4)
  const int nt0 = wn * ntw, kt0 = wk * ktw;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbacc = 0.f;
  const int ntiles = (a.M + WG_RT - 1) / WG_RT;
  for (int t = blockIdx.x; t < ntiles; t += a.S) {
    const int m0 = t * WG_RT;
    __syncthreads();
    for (int i = tid; i < WG_RT * NTb * 16; i += 256) {
      int r = i / (NTb * 16), c = i - r * (NTb * 16);
      int row = m0 + r;
      float v = 0.f;
      if (row < a.M && c < nb) v = a.dY[(size_t)row * a.lddy + n0 + c];
      Ys[r * ldy + c] = v;
    }
    for (int i = tid; i < WG_RT * KTb * 16; i += 256) {
      int r = i / (KTb * 16), c = i - r * (KTb * 16);
      int row = m0 + r;
      float v = 0.f;
      if (row < a.M && c < kb) v = a.X[(size_t)row * a.ldx + k0 + c];
      Xs[r * ldx + c] = v;
    }
    __syncthreads();
    if (a.want_db && blockIdx.z == 0 && tid < nb) {
      float s = 0.f;
#pragma unroll 8
      for (int r = 0; r < WG_RT; ++r) s += Ys[r * ldy + tid];
      dbacc += s;
    }
#pragma unroll 2
    for (int ms = 0; ms < WG_RT / 4; ++ms) {
      const int rr = ms * 4 + (lane >> 4);
      float af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = (nt0 + i < NTb && i < ntw) ? Ys[rr * ldy + (nt0 + i) * 16 + (lane & 15)] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = (kt0 + j < KTb && j < ktw) ? Xs[rr * ldx + (kt0 + j) * 16 + (lane & 15)] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (nt0 + i < NTb && i < ntw) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (kt0 + j < KTb && j < ktw) acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
        }
      }
    }
  }
  float* slab = a.slabs + (size_t)blockIdx.x * ((size_t)a.N * a.K + a.N);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (!(nt0 + i < NTb && i < ntw)) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!(kt0 + j < KTb && j < ktw)) continue;
      const int k = k0 + (kt0 + j) * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + (nt0 + i) * 16 + 4 * (lane >> 4) + r;
        if (n < a.N && k < a.K) slab[(size_t)n * a.K + k] = acc[i][j][r];
      }
    }
  }
  if (a.want_db && blockIdx.z == 0 && tid < nb) slab[(size_t)a.N * a.K + n0 + tid] = dbacc;
}

// out[i] (+)= sum_s slabs[s][i]; i < n.  Fixed summation order -> bitwise reproducible.
__global__ void slab_reduce_kernel(const float* __restrict__ slabs, size_t stride, int S, int n, int rows, int cols,
                                   float* __restrict__ out, int ldo, int accumulate) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int s = 0;
  for (; s + 3 < S; s += 4) {
    s0 += slabs[(size_t)s * stride + i];
    s1 += slabs[(size_t)(s + 1) * stride + i];
    s2 += slabs[(size_t)(s + 2) * stride + i];
    s3 += slabs[(size_t)(s + 3) * stride + i];
  }
  for (; s < S; ++s) s0 += slabs[(size_t)s * stride + i];
  float v = (s0 + s1) + (s2 + s3);
  int r = i / cols, c = i - r * cols;
  (void)rows;
  float* dst = out + (size_t)r * ldo + c;
  *dst = accumulate ? (*dst + v) : v;
}

int launch_slab_reduce(const float* slabs, size_t stride, int S, int rows, int cols, float* out, int ldo,
                       int accumulate, hipStream_t st) {
  int n = rows * cols;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, slabs, stride, S, n, rows, cols, out, ldo, accumulate);
  INTEL_CHECK_LAUNCH();
  return 0;
}

int launch_wgrad(const float* dY, int lddy, const float* X, int ldx, int M, int N, int K, float* dW, int lddw,
                 float* db, int accumulate, float* slabs, hipStream_t st) {
  if (N <= 0 || K <= 0) return 0;
  WgradArgs a;
  a.dY = dY; a.lddy = lddy; a.X = X; a.ldx = ldx; a.M = M; a.N = N; a.K = K; a.slabs = slabs;
  a.S = wgrad_num_slabs(M); a.want_db = db != nullptr;
  const int nbm = min(128, N), kbm = min(128, K);
  const int ldy = (rup(rup(nbm, 16), 32)) + 16, ldxs = (rup(rup(kbm, 16), 32)) + 16;
  size_t smem = (size_t)WG_RT * (ldy + ldxs) * sizeof(float);
  hipLaunchKernelGGL(wgrad_kernel, dim3(a.S, cdiv(N, 128), cdiv(K, 128)), dim3(256), smem, st, a);
  INTEL_CHECK_LAUNCH();
  size_t stride = (size_t)N * K + N;
  int rc = launch_slab_reduce(slabs, stride, a.S, N, K, dW, lddw, accumulate, st);
  if (rc) return rc;
  if (db) rc = launch_slab_reduce(slabs + (size_t)N * K, stride, a.S, 1, N, db, N, accumulate, st);
  return rc;
}
