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
#include <stdlib.h>
#include <new>
#include <vector>

#include "kernels.h"

// ------------------------------------------------------------------------------------------
// pack: weights -> MFMA fragment order.  A model step re-packs ~80 small matrices (the parameters
// change every step); they are batched into job tables passed by value so that a whole model is
// packed in two launches instead of eighty.
// ------------------------------------------------------------------------------------------
struct PackJob {
  const float* W; float* P;
  int ldw, Kd, Nd, trans, nt_off, g_off, KG_total, KG, NT, block0;
};
#define PACK_MAX_JOBS 56
struct PackJobs { int n; PackJob j[PACK_MAX_JOBS]; };

__global__ void pack_b_kernel(PackJobs jobs) {
  int ji = 0;
  while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.j[ji + 1].block0) ++ji;
  const PackJob& jb = jobs.j[ji];
  const int idx = ((int)blockIdx.x - jb.block0) * blockDim.x + threadIdx.x;   // one thread per (nt, g, lane)
  const int total = jb.NT * jb.KG * 64;
  if (idx >= total) return;
  const int lane = idx & 63;
  const int g = (idx >> 6) % jb.KG;
  const int nt = (idx >> 6) / jb.KG;
  const int n = nt * 16 + (lane & 15);
  f32x4 v;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k = g * 16 + 4 * (lane >> 4) + s;
    float x = 0.f;
    if (k < jb.Kd && n < jb.Nd) x = jb.trans ? jb.W[(size_t)k * jb.ldw + n] : jb.W[(size_t)n * jb.ldw + k];
    v[s] = x;
  }
  *reinterpret_cast<f32x4*>(jb.P + ((size_t)((jb.nt_off + nt) * jb.KG_total + jb.g_off + g) * 64 + lane) * 4) = v;
}

static thread_local PackJobs* g_pack_batch = nullptr;
static thread_local int g_pack_blocks = 0;

static int pack_flush(PackJobs& jobs, int blocks, hipStream_t st) {
  if (jobs.n == 0) return 0;
  LAUNCH(pack_b_kernel, dim3(blocks), dim3(256), 0, st, jobs);
  INTEL_CHECK_LAUNCH();
  jobs.n = 0;
  return 0;
}

// Between pack_batch_begin / pack_batch_end launch_pack_b only records jobs.
static thread_local PackJobs g_pack_storage;
void pack_batch_begin() {
  g_pack_storage.n = 0;
  g_pack_blocks = 0;
  g_pack_batch = &g_pack_storage;
}
int pack_batch_end(hipStream_t st) {
  int rc = g_pack_batch ? pack_flush(*g_pack_batch, g_pack_blocks, st) : 0;
  g_pack_batch = nullptr;
  g_pack_blocks = 0;
  return rc;
}

int launch_pack_b(const float* W, int ldw, int Kd, int Nd, int trans, float* P, int nt_off, hipStream_t st, int g_off,
                  int KG_total) {
  PackJob jb;
  jb.W = W; jb.P = P; jb.ldw = ldw; jb.Kd = Kd; jb.Nd = Nd; jb.trans = trans; jb.nt_off = nt_off; jb.g_off = g_off;
  jb.KG = rup(Kd, 16) / 16; jb.NT = rup(Nd, 16) / 16;
  jb.KG_total = KG_total > 0 ? KG_total : jb.KG;
  const int blocks = cdiv(jb.NT * jb.KG * 64, 256);
  if (g_pack_batch) {
    if (g_pack_batch->n == PACK_MAX_JOBS) {
      int rc = pack_flush(*g_pack_batch, g_pack_blocks, st);
      if (rc) return rc;
      g_pack_blocks = 0;
    }
    jb.block0 = g_pack_blocks;
    g_pack_batch->j[g_pack_batch->n++] = jb;
    g_pack_blocks += blocks;
    return 0;
  }
  PackJobs one;
  one.n = 1;
  jb.block0 = 0;
  one.j[0] = jb;
  int tmp = blocks;
  return pack_flush(one, tmp, st);
}

// ------------------------------------------------------------------------------------------
// gemm_rows
// ------------------------------------------------------------------------------------------
// Arithmetic mode of the matrix-pipe products (set per call by the model plan from IntelDesc.dtype):
//   3 = fp32 accuracy (hi + mid + lo bf16 planes, six plane products) -- the parity mode;
//   1 = bf16 mode: operands rounded to bf16, ONE product, fp32 accumulate.
static thread_local int g_planes = 3;
void gemm_set_planes(int planes) { g_planes = planes == 1 ? 1 : 3; }
int gemm_planes() { return g_planes; }

#define GR_BM 64
#define GR_KC 128
#define GR_LDA (GR_KC + 4)
#define GR_LDE (128 + 4)

struct GemmRowsArgs {
  int vec_ep;   // epilogue operands are 16-byte aligned with leading dimensions % 4 == 0
  int dbg;   // ablation bits (INTEL_DEBUG_GEMM): 1 no epilogue, 2 no MFMA, 4 no LDS staging, 8 no A prefetch
  const float* A; int lda; int M; int K;
  const float* Bp; int N;
  float* C; int ldc;
  GemmEpilogue ep;
};

// bytes a launch must move: A, packed B, C, plus the epilogue operands (residual / relu mask / accumulate input,
// LayerNorm x-hat stash)
static double gemm_algorithmic_bytes(const GemmRowsArgs& a) {
  double b = 4.0 * ((double)a.M * a.K + (double)a.K * a.N + (double)a.M * a.N);
  if (a.ep.res || a.ep.mask || a.ep.accumulate) b += 4.0 * (double)a.M * a.N;
  if (a.ep.xhat) b += 4.0 * (double)a.M * a.N + 4.0 * (double)a.M;
  if (a.ep.no_out) b -= 4.0 * (double)a.M * a.N;
  if (a.ep.a_bf16) b -= 2.0 * (double)a.M * a.K;
  if (a.ep.c_bf16) b -= 2.0 * (double)a.M * a.N;
  if (a.ep.mask && a.ep.mask_bf16) b -= 2.0 * (double)a.M * a.N;
  return b;
}


// Epilogues.  The MFMAs are issued with the WEIGHT fragment as the A operand and the activation
// fragment as the B operand, so an accumulator register quad holds 4 CONSECUTIVE OUTPUT COLUMNS of
// one row (row = lane&15 of the tile): every epilogue load/store is a 16-byte access.
//   acc[rt][c][r]  <->  row m0 + rt*16 + (lane&15),  column n0 + ct*16 + 4*(lane>>4) + r
//
// vmcnt counts loads AND stores in issue order, so a load issued after a store cannot be waited for
// without also draining that store.  Both epilogues therefore issue EVERY global load (bias and the
// one auxiliary operand: residual, relu mask or the accumulate destination) before the first store.
__device__ __forceinline__ void gr_epilogue_direct(const GemmRowsArgs& a, const f32x4 (&acc)[4][2], int m0, int n0, int ntc,
                                                   int wave, int lane) {
  const GemmEpilogue& ep = a.ep;
  // one auxiliary operand per call (the plan never combines them): 1 mask, 2 residual, 3 accumulate
  const float* auxp = ep.mask ? ep.mask : (ep.res ? ep.res : (ep.accumulate ? a.C : nullptr));
  const int auxld = ep.mask ? ep.ldmask : (ep.res ? ep.ldres : a.ldc);
  const int mode = ep.mask ? 1 : (ep.res ? 2 : (ep.accumulate ? 3 : 0));
  const bool late_acc = ep.accumulate && mode != 3;      // accumulate combined with mask/res (unused by the plan)
  if (a.vec_ep) {
    f32x4 bias[2], aux[4][2];
    bool okc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int ct = wave + 4 * c;
      const int col = n0 + ct * 16 + 4 * (lane >> 4);
      okc[c] = ct < ntc && col + 3 < a.N;
      bias[c] = (ep.bias && okc[c]) ? *reinterpret_cast<const f32x4*>(ep.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const int row = m0 + rt * 16 + (lane & 15);
        aux[rt][c] = (mode && okc[c] && row < a.M) ? *reinterpret_cast<const f32x4*>(auxp + (size_t)row * auxld + col)
                                                   : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int ct = wave + 4 * c;
      const int col = n0 + ct * 16 + 4 * (lane >> 4);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const int row = m0 + rt * 16 + (lane & 15);
        f32x4 x = acc[rt][c] + bias[c];
        if (ep.relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] = fmaxf(x[r], 0.f);
        }
        if (mode == 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] = aux[rt][c][r] > 0.f ? x[r] : 0.f;
        } else {
          x += aux[rt][c];                      // residual / accumulate (zeros when unused)
        }
        if (okc[c] && row < a.M) {
          f32x4* dst = reinterpret_cast<f32x4*>(a.C + (size_t)row * a.ldc + col);
          if (late_acc) x += *dst;
          *dst = x;
        }
      }
    }
    // ragged right edge (N % 4 != 0 never happens with vec_ep unless N % 16 != 0): scalar tail
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int ct = wave + 4 * c;
      const int col = n0 + ct * 16 + 4 * (lane >> 4);
      if (ct >= ntc || okc[c] || col >= a.N) continue;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const int row = m0 + rt * 16 + (lane & 15);
        if (row >= a.M) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cc = col + r;
          if (cc >= a.N) continue;
          float x = acc[rt][c][r] + (ep.bias ? ep.bias[cc] : 0.f);
          if (ep.relu) x = fmaxf(x, 0.f);
          if (ep.mask) x = (ep.mask[(size_t)row * ep.ldmask + cc] > 0.f) ? x : 0.f;
          if (ep.res) x += ep.res[(size_t)row * ep.ldres + cc];
          float* dst = a.C + (size_t)row * a.ldc + cc;
          *dst = ep.accumulate ? (*dst + x) : x;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int ct = wave + 4 * c;
    if (ct >= ntc) continue;
    const int col = n0 + ct * 16 + 4 * (lane >> 4);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const int row = m0 + rt * 16 + (lane & 15);
      if (row >= a.M) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cc = col + r;
        if (cc >= a.N) continue;
        float x = acc[rt][c][r] + (ep.bias ? ep.bias[cc] : 0.f);
        if (ep.relu) x = fmaxf(x, 0.f);
        if (ep.mask) x = (ep.mask[(size_t)row * ep.ldmask + cc] > 0.f) ? x : 0.f;
        if (ep.res) x += ep.res[(size_t)row * ep.ldres + cc];
        float* dst = a.C + (size_t)row * a.ldc + cc;
        *dst = ep.accumulate ? (*dst + x) : x;
      }
    }
  }
}

// LayerNorm epilogue through an LDS tile Es[64][GR_LDE] (caller syncs before; N <= 128).  One wave
// normalises 16 rows; the residual rows are all loaded before the first store (see above).
__device__ __forceinline__ void gr_epilogue_ln(const GemmRowsArgs& a, const f32x4 (&acc)[4][2], float* Es, int m0, int ntc,
                                               int wave, int lane) {
  const GemmEpilogue& ep = a.ep;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int ct = wave + 4 * c;
    if (ct < ntc) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
        *reinterpret_cast<f32x4*>(Es + (rt * 16 + (lane & 15)) * GR_LDE + ct * 16 + 4 * (lane >> 4)) = acc[rt][c];
    }
  }
  const int ncols = min(128, a.N);
  const bool ok0 = lane < ncols, ok1 = lane + 64 < ncols;
  float res0[16], res1[16];
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    const int row = m0 + wave * 16 + rr;
    const bool rok = row < a.M && ep.res != nullptr;
    res0[rr] = (rok && ok0) ? ep.res[(size_t)row * ep.ldres + lane] : 0.f;
    res1[rr] = (rok && ok1) ? ep.res[(size_t)row * ep.ldres + lane + 64] : 0.f;
  }
  const float bias0 = (ep.bias && ok0) ? ep.bias[lane] : 0.f, bias1 = (ep.bias && ok1) ? ep.bias[lane + 64] : 0.f;
  const float g0 = ok0 ? ep.gamma[lane] : 0.f, g1 = ok1 ? ep.gamma[lane + 64] : 0.f;
  const float be0 = ok0 ? ep.beta[lane] : 0.f, be1 = ok1 ? ep.beta[lane + 64] : 0.f;
  __syncthreads();
  const float inv_n = 1.f / (float)a.N;
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    const int r = wave * 16 + rr;
    const int row = m0 + r;
    float v0 = 0.f, v1 = 0.f;
    if (ok0) { v0 = Es[r * GR_LDE + lane] + bias0; if (ep.relu) v0 = fmaxf(v0, 0.f); v0 += res0[rr]; }
    if (ok1) { v1 = Es[r * GR_LDE + lane + 64] + bias1; if (ep.relu) v1 = fmaxf(v1, 0.f); v1 += res1[rr]; }
    const float mean = wave_sum(v0 + v1) * inv_n;
    const float d0 = ok0 ? v0 - mean : 0.f, d1 = ok1 ? v1 - mean : 0.f;
    const float var = wave_sum(d0 * d0 + d1 * d1) * inv_n;
    const float rs = 1.f / sqrtf(var + 1e-5f);
    if (row < a.M) {
      if (ep.rstd && lane == 0) ep.rstd[row] = rs;
      if (ok0) {
        const float xh = d0 * rs;
        if (ep.xhat) ep.xhat[(size_t)row * ep.ldxhat + lane] = xh;
        if (!ep.no_out) a.C[(size_t)row * a.ldc + lane] = xh * g0 + be0;
      }
      if (ok1) {
        const float xh = d1 * rs;
        if (ep.xhat) ep.xhat[(size_t)row * ep.ldxhat + lane + 64] = xh;
        if (!ep.no_out) a.C[(size_t)row * a.ldc + lane + 64] = xh * g1 + be1;
      }
    }
  }
}

__global__ __launch_bounds__(256) void gemm_rows_kernel(GemmRowsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                       // [64][GR_LDA]; re-used as the LayerNorm epilogue tile
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
    const bool has0 = wave < ntc, has1 = wave + 4 < ntc;

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
      // B fragments come straight from L2 in MFMA order; prefetch one k group ahead
      const f32x4* bp0 = reinterpret_cast<const f32x4*>(a.Bp) + ((size_t)(nc + wave) * KG + gbase) * 64 + lane;
      const f32x4* bp1 = reinterpret_cast<const f32x4*>(a.Bp) + ((size_t)(nc + wave + 4) * KG + gbase) * 64 + lane;
      f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f}, b1 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (has0) b0 = bp0[0];
      if (has1) b1 = bp1[0];
      for (int g = 0; g < ng; ++g) {
        f32x4 n0v = b0, n1v = b1;
        if (g + 1 < ng) {
          if (has0) n0v = bp0[(size_t)(g + 1) * 64];
          if (has1) n1v = bp1[(size_t)(g + 1) * 64];
        }
        f32x4 af[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
          af[rt] = *reinterpret_cast<const f32x4*>(As + (rt * 16 + (lane & 15)) * GR_LDA + g * 16 + 4 * (lane >> 4));
        if (has0) {
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) acc[rt][0] = mfma16(b0[s], af[rt][s], acc[rt][0]);
        }
        if (has1) {
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) acc[rt][1] = mfma16(b1[s], af[rt][s], acc[rt][1]);
        }
        b0 = n0v;
        b1 = n1v;
      }
    }
    if (!ep.gamma) {
      gr_epilogue_direct(a, acc, m0, nc * 16, ntc, wave, lane);
      continue;
    }
    // LayerNorm epilogue (N <= 128: single chunk): the A tile is dead, re-use it
    __syncthreads();
    gr_epilogue_ln(a, acc, As, m0, ntc, wave, lane);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// gemm_rows, the odd products of the B-row chains (intent logits: N = I = 30; fusion weights: N = K = 3; their
// transposes with a 30- or 3-wide reduction): few rows, and an N or a K that is no multiple of four.  The 64-row
// workgroups of the generic kernel leave 3/4 of the chip idle at 4096 rows and (for N <= 32) two of their four waves
// without a column tile; here a workgroup owns 16 rows (256 workgroups at 4096 rows) and
//   THIN (N <= 32): the four waves split the reduction (k group g goes to wave g % 4), read their A fragments straight
//         from global memory (every element is used by one wave only) and sum their partial tiles through LDS;
//   wide (K <= 128): the 16 x K tile is staged once, every wave keeps its A fragments in registers and sweeps the
//         column tiles ct = wave, wave + 4, ...
// Exact fp32 MFMA (16x16x4) like the generic kernel.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void small_tile_epilogue(const GemmRowsArgs& a, const f32x4& acc, int row, int col) {
  if (row >= a.M || col >= a.N) return;
  const GemmEpilogue& ep = a.ep;
  if (a.vec_ep && col + 3 < a.N) {
    f32x4 x = acc;
    if (ep.bias) x += *reinterpret_cast<const f32x4*>(ep.bias + col);
    if (ep.relu) {
#pragma unroll
      for (int r = 0; r < 4; ++r) x[r] = fmaxf(x[r], 0.f);
    }
    if (ep.mask) {
      const f32x4 m = *reinterpret_cast<const f32x4*>(ep.mask + (size_t)row * ep.ldmask + col);
#pragma unroll
      for (int r = 0; r < 4; ++r) x[r] = m[r] > 0.f ? x[r] : 0.f;
    }
    if (ep.res) x += *reinterpret_cast<const f32x4*>(ep.res + (size_t)row * ep.ldres + col);
    f32x4* dst = reinterpret_cast<f32x4*>(a.C + (size_t)row * a.ldc + col);
    if (ep.accumulate) x += *dst;
    *dst = x;
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int cc = col + r;
    if (cc >= a.N) continue;
    float x = acc[r] + (ep.bias ? ep.bias[cc] : 0.f);
    if (ep.relu) x = fmaxf(x, 0.f);
    if (ep.mask) x = (ep.mask[(size_t)row * ep.ldmask + cc] > 0.f) ? x : 0.f;
    if (ep.res) x += ep.res[(size_t)row * ep.ldres + cc];
    float* dst = a.C + (size_t)row * a.ldc + cc;
    *dst = ep.accumulate ? (*dst + x) : x;
  }
}

template <bool THIN>
__global__ __launch_bounds__(256) void gemm_rows_small_kernel(GemmRowsArgs a) {
  __shared__ __attribute__((aligned(16))) float sm[THIN ? 4 * 2 * 64 * 4 : 16 * GR_LDA];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * 16;
  const int Kp = (a.K + 15) & ~15, KG = Kp >> 4;
  const int NT = (a.N + 15) >> 4;
  const int row = m0 + (lane & 15), kq = 4 * (lane >> 4);
  const f32x4* bp = reinterpret_cast<const f32x4*>(a.Bp);
  if (THIN) {
    const bool vecA = ((a.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.A) & 15) == 0);
    const float* ap = a.A + (size_t)min(row, a.M - 1) * a.lda + kq;
    const bool two = NT > 1;
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    for (int g = wave; g < KG; g += 4) {
      const int k0 = g * 16 + kq;
      f32x4 av = f32x4{0.f, 0.f, 0.f, 0.f};
      if (vecA && k0 + 3 < a.K) {
        av = *reinterpret_cast<const f32x4*>(ap + g * 16);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (k0 + r < a.K) av[r] = ap[g * 16 + r];
      }
      const f32x4 b0 = bp[(size_t)g * 64 + lane];
      f32x4 b1 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (two) b1 = bp[((size_t)KG + g) * 64 + lane];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc0 = mfma16(b0[q], av[q], acc0);
      if (two) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc1 = mfma16(b1[q], av[q], acc1);
      }
    }
    f32x4* red = reinterpret_cast<f32x4*>(sm);
    red[(wave * 2 + 0) * 64 + lane] = acc0;
    red[(wave * 2 + 1) * 64 + lane] = acc1;
    __syncthreads();
    if (wave < NT) {
      f32x4 x = red[(0 * 2 + wave) * 64 + lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) x += red[(w * 2 + wave) * 64 + lane];
      small_tile_epilogue(a, x, row, wave * 16 + kq);
    }
    return;
  }
  // wide: K <= 128
  if (((a.lda & 3) == 0) && ((a.K & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.A) & 15) == 0)) {
    const int c4n = Kp >> 2;
    for (int i = tid; i < 16 * c4n; i += 256) {
      const int r = i / c4n, c = (i - r * c4n) * 4;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (m0 + r < a.M && c < a.K) v = *reinterpret_cast<const f32x4*>(a.A + (size_t)(m0 + r) * a.lda + c);
      *reinterpret_cast<f32x4*>(sm + r * GR_LDA + c) = v;
    }
  } else {
    for (int i = tid; i < 16 * Kp; i += 256) {
      const int r = i / Kp, c = i - r * Kp;
      float v = 0.f;
      if (m0 + r < a.M && c < a.K) v = a.A[(size_t)(m0 + r) * a.lda + c];
      sm[r * GR_LDA + c] = v;
    }
  }
  __syncthreads();
  f32x4 af[8];
#pragma unroll
  for (int g = 0; g < 8; ++g)
    af[g] = g < KG ? *reinterpret_cast<const f32x4*>(sm + (lane & 15) * GR_LDA + g * 16 + kq) : f32x4{0.f, 0.f, 0.f, 0.f};
  for (int ct = wave; ct < NT; ct += 4) {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (g < KG) {
        const f32x4 b = bp[((size_t)ct * KG + g) * 64 + lane];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = mfma16(b[q], af[g][q], acc);
      }
    }
    small_tile_epilogue(a, acc, row, ct * 16 + kq);
  }
}

// ------------------------------------------------------------------------------------------
// gemm_rows, B-stationary persistent form (K <= 128, 16-byte aligned A).
//
// 512-thread workgroup = 8 waves; wave (ct, rg) owns ONE 16-column tile and RT of the four 16-row
// tiles of a 64-row A tile, and keeps its B fragments for the whole K extent in registers (32
// VGPRs) while the workgroup sweeps row tiles: weights are read from L2 once per workgroup.  The
// next A tile is prefetched global -> registers under the MFMAs of the current one (two LDS
// buffers, one barrier per tile).  ~100 VGPRs -> 2 workgroups (16 waves) per CU.
//   RT = 4: 128-column chunks;  RT = 2: 64-column chunks;  RT = 1: 32-column chunks.
// Per-iteration order: MFMA(t) -> LDS write of tile t+S -> epilogue stores of tile t -> barrier ->
// issue loads of tile t+2S; the only wait for loads sits before the stores are issued, so stores
// drain under the next tile's MFMAs (vmcnt counts loads and stores in order).
// ------------------------------------------------------------------------------------------
template <int RT, bool LN>
__global__ __launch_bounds__(512, 4) void gemm_rows_w8_kernel(GemmRowsArgs a) {
  constexpr int RG = 4 / RT;            // row groups
  constexpr int CT = 8 / RG;            // column tiles per chunk
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave / RG, rg = wave % RG;
  const int Kp = (a.K + 15) & ~15, KG = Kp >> 4;      // KG <= 8
  const int NT = (a.N + 15) >> 4;
  const int nc = blockIdx.y * CT;
  const int ntc = min(CT, NT - nc);
  const bool active = ct < ntc;
  const GemmEpilogue& ep = a.ep;
  f32x4 bfr[8];
#pragma unroll
  for (int g = 0; g < 8; ++g)
    bfr[g] = (g < KG && active) ? reinterpret_cast<const f32x4*>(a.Bp)[((size_t)(nc + ct) * KG + g) * 64 + lane] : f32x4{0.f, 0.f, 0.f, 0.f};
  const int c4n = Kp >> 2;                    // float4 per tile row
  const int tot4 = GR_BM * c4n;               // float4 per tile
  f32x4 pre[4];
  // this thread's (up to) four 16-byte slots of an A tile, computed once.  Loads are UNCONDITIONAL
  // (rows past M are clamped to the last row; their products are never stored) so that the compiler
  // emits plain loads without zero-fill moves and waits; K-padding slots are zeroed in LDS once.
  int trow[4], tcol[4], loff[4];
  bool slot[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = tid + 512 * j;
    const int r = i / c4n, c = (i - r * c4n) * 4;
    slot[j] = i < tot4 && c < a.K;
    trow[j] = r;
    tcol[j] = c;
    loff[j] = r * GR_LDA + c;
    if (i < tot4 && c >= a.K) {          // k padding (K % 16 != 0): stays zero in both buffers
      *reinterpret_cast<f32x4*>(smem + loff[j]) = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(smem + GR_BM * GR_LDA + loff[j]) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const int ntiles = (a.M + GR_BM - 1) / GR_BM;
  int t = blockIdx.x;
  if (t >= ntiles) return;
  auto load_tile = [&](int tt) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (slot[j]) {
        const int row = min(tt * GR_BM + trow[j], a.M - 1);
        pre[j] = *reinterpret_cast<const f32x4*>(a.A + (size_t)row * a.lda + tcol[j]);
      }
    }
  };
  auto store_tile = [&](float* As) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (slot[j]) *reinterpret_cast<f32x4*>(As + loff[j]) = pre[j];
  };
  // auxiliary epilogue operand (the plan never combines them): 1 relu mask, 2 residual, 3 accumulate
  const float* auxp = ep.mask ? ep.mask : (ep.res ? ep.res : (ep.accumulate ? a.C : nullptr));
  const int auxld = ep.mask ? ep.ldmask : (ep.res ? ep.ldres : a.ldc);
  const int mode = ep.mask ? 1 : (ep.res ? 2 : (ep.accumulate ? 3 : 0));
  const int col = (nc + ct) * 16 + 4 * (lane >> 4);
  const bool colok = active && col < a.N;
  f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
  if (!LN && ep.bias && colok) bias = *reinterpret_cast<const f32x4*>(ep.bias + col);

  load_tile(t);
  store_tile(smem);
  // B fragments and the bias are loop invariants loaded once: retire those loads HERE so that no wait
  // inside the loop (vmcnt is in-order and also counts the epilogue stores) has to cover them
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
  __syncthreads();
  if (t + (int)gridDim.x < ntiles) load_tile(t + gridDim.x);
  int buf = 0;
  for (; t < ntiles; t += gridDim.x) {
    float* As = smem + buf * (GR_BM * GR_LDA);
    f32x4 acc[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (active) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if (g < KG) {
          f32x4 af[RT];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            af[rt] = *reinterpret_cast<const f32x4*>(As + ((rg * RT + rt) * 16 + (lane & 15)) * GR_LDA + g * 16 + 4 * (lane >> 4));
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = mfma16(bfr[g][s], af[rt][s], acc[rt]);
        }
      }
    }
    if (t + (int)gridDim.x < ntiles) store_tile(smem + (buf ^ 1) * (GR_BM * GR_LDA));
    const int m0 = t * GR_BM;
    if (!LN) {
      if (colok) {       // host guarantees 16-byte aligned epilogue operands and N % 4 == 0
        f32x4 aux[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int row = m0 + (rg * RT + rt) * 16 + (lane & 15);
          aux[rt] = (mode && row < a.M) ? *reinterpret_cast<const f32x4*>(auxp + (size_t)row * auxld + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int row = m0 + (rg * RT + rt) * 16 + (lane & 15);
          f32x4 x = acc[rt] + bias;
          if (ep.relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = fmaxf(x[r], 0.f);
          }
          if (mode == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = aux[rt][r] > 0.f ? x[r] : 0.f;
          } else {
            x += aux[rt];
          }
          if (row < a.M) *reinterpret_cast<f32x4*>(a.C + (size_t)row * a.ldc + col) = x;
        }
      }
    } else {
      // LayerNorm epilogue (one chunk covers all N <= 128 columns): accumulators -> LDS tile ->
      // one wave per 8 rows; every residual row is loaded before the first store
      __syncthreads();                 // all waves are done reading this A buffer
      float* Es = As;
      if (active) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          *reinterpret_cast<f32x4*>(Es + ((rg * RT + rt) * 16 + (lane & 15)) * GR_LDE + ct * 16 + 4 * (lane >> 4)) = acc[rt];
      }
      const int ncols = a.N;
      const bool ok0 = lane < ncols, ok1 = lane + 64 < ncols;
      float res0[8], res1[8];
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int row = m0 + wave * 8 + rr;
        const bool rok = row < a.M && ep.res != nullptr;
        res0[rr] = (rok && ok0) ? ep.res[(size_t)row * ep.ldres + lane] : 0.f;
        res1[rr] = (rok && ok1) ? ep.res[(size_t)row * ep.ldres + lane + 64] : 0.f;
      }
      const float bias0 = (ep.bias && ok0) ? ep.bias[lane] : 0.f, bias1 = (ep.bias && ok1) ? ep.bias[lane + 64] : 0.f;
      const float g0 = ok0 ? ep.gamma[lane] : 0.f, g1 = ok1 ? ep.gamma[lane + 64] : 0.f;
      const float be0 = ok0 ? ep.beta[lane] : 0.f, be1 = ok1 ? ep.beta[lane + 64] : 0.f;
      __syncthreads();
      const float inv_n = 1.f / (float)a.N;
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int r = wave * 8 + rr;
        const int row = m0 + r;
        float v0 = 0.f, v1 = 0.f;
        if (ok0) { v0 = Es[r * GR_LDE + lane] + bias0; if (ep.relu) v0 = fmaxf(v0, 0.f); v0 += res0[rr]; }
        if (ok1) { v1 = Es[r * GR_LDE + lane + 64] + bias1; if (ep.relu) v1 = fmaxf(v1, 0.f); v1 += res1[rr]; }
        const float mean = wave_sum(v0 + v1) * inv_n;
        const float d0 = ok0 ? v0 - mean : 0.f, d1 = ok1 ? v1 - mean : 0.f;
        const float var = wave_sum(d0 * d0 + d1 * d1) * inv_n;
        const float rs = 1.f / sqrtf(var + 1e-5f);
        if (row < a.M) {
          if (ep.rstd && lane == 0) ep.rstd[row] = rs;
          if (ok0) {
            const float xh = d0 * rs;
            if (ep.xhat) ep.xhat[(size_t)row * ep.ldxhat + lane] = xh;
            if (!ep.no_out) a.C[(size_t)row * a.ldc + lane] = xh * g0 + be0;
          }
          if (ok1) {
            const float xh = d1 * rs;
            if (ep.xhat) ep.xhat[(size_t)row * ep.ldxhat + lane + 64] = xh;
            if (!ep.no_out) a.C[(size_t)row * a.ldc + lane + 64] = xh * g1 + be1;
          }
        }
      }
    }
    __syncthreads();
    if (t + 2 * (int)gridDim.x < ntiles) load_tile(t + 2 * gridDim.x);
    buf ^= 1;
  }
}

// ------------------------------------------------------------------------------------------
// gemm_rows_b3: the same product on the bf16 matrix pipe at fp32 accuracy.  Every fp32 operand value is split
// into three bf16 planes x = hi + mid + lo (24 significant bits: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi
// - mid); the two subtractions are exact) and a product is the sum of the six plane products whose weight is
// >= 2^-16: hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid, each exact in the fp32 accumulator.  Measured on
// random data the result is closer to the exact dot product than the fp32 fmaf chain (3e-8 vs 1e-7 relative to
// sum |terms|).  v_mfma_f32_16x16x32_bf16 does 16x16x32 in 16 cycles, v_mfma_f32_16x16x4_f32 16x16x4 in 32: six
// plane products cost 96 cycles per 32-deep block against 256 -- the GEMM leaves the MFMA roof for the HBM roof.
//   * the A tile (64 rows x K) is split ONCE by the 512 threads on its way global -> registers -> LDS (three
//     bf16 planes, 288-byte rows: b128 fragment reads are conflict-free), not by every consuming wave;
//   * the weight fragments come from the same fp32 packed buffer as the fp32 kernels and are split once per
//     workgroup into registers (3 planes x K/32 blocks x 4 VGPRs);
//   * operand order and accumulator layout equal the fp32 kernels': the epilogues are shared verbatim.
// K in {64, 128} (template KBN = K / 32).  One LDS buffer, the next tile is prefetched into registers under the MFMAs.
// ------------------------------------------------------------------------------------------
#define B3_LDP 144          // bf16 elements per LDS row: 128 + 16
#ifndef B3_ABLATE           // tools/b3_ablate.sh: 1 no C stores, 2 no MFMAs, 8 only the first A tile is loaded, 16 no split, 32 one LDS fragment address
#define B3_ABLATE 0
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void b3_split4(const f32x4& x, bf16x4& h, bf16x4& m, bf16x4& l) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const __bf16 hh = (__bf16)x[i];
    const float r1 = x[i] - (float)hh;
    const __bf16 mm = (__bf16)r1;
    const float r2 = r1 - (float)mm;
    h[i] = hh;
    m[i] = mm;
    l[i] = (__bf16)r2;
  }
}
__device__ __forceinline__ bf16x8 b3_cat(const bf16x4& a, const bf16x4& b) {
  return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

template <int RT, bool LN, int KBN, int NP = 3>
__global__ __launch_bounds__(512, 4) void gemm_rows_b3_kernel(GemmRowsArgs a) {
  constexpr int RG = 4 / RT;
  constexpr int CT = 8 / RG;
  constexpr int KW = KBN * 32;                  // K
  constexpr int C4N = KW / 4;                   // float4 per tile row
  constexpr int NJ = GR_BM * C4N / 512;         // float4 per thread per tile: 4 (K = 128) or 2 (K = 64)
  constexpr int PLANE = GR_BM * B3_LDP;         // bf16 elements per plane
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* planes = reinterpret_cast<__bf16*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave / RG, rg = wave % RG;
  const int KG = KW >> 4;
  const int NT = (a.N + 15) >> 4;
  const int nc = blockIdx.y * CT;
  const int ntc = min(CT, NT - nc);
  const bool active = ct < ntc;
  const GemmEpilogue& ep = a.ep;
  // weight fragments: lane (n = lane & 15, j = lane >> 4) needs k = kb*32 + 8j .. +7 of column n; in the fp32
  // packed buffer those are two float4 of k-group g = 2 kb + (j >> 1), lanes (2 (j & 1)) * 16 + n and + 16
  bf16x8 bh[KBN], bm[KBN], bl[KBN];
  {
    const int n = lane & 15, j = lane >> 4;
    const int ls = (2 * (j & 1)) * 16 + n;
#pragma unroll
    for (int kb = 0; kb < KBN; ++kb) {
      const int g = 2 * kb + (j >> 1);
      f32x4 w0 = f32x4{0.f, 0.f, 0.f, 0.f}, w1 = w0;
      if (active) {
        const f32x4* P4 = reinterpret_cast<const f32x4*>(a.Bp) + ((size_t)(nc + ct) * KG + g) * 64;
        w0 = P4[ls];
        w1 = P4[ls + 16];
      }
      bf16x4 h0, m0, l0, h1, m1, l1;
      b3_split4(w0, h0, m0, l0);
      b3_split4(w1, h1, m1, l1);
      bh[kb] = b3_cat(h0, h1);
      bm[kb] = b3_cat(m0, m1);
      bl[kb] = b3_cat(l0, l1);
    }
  }
  f32x4 pre[NJ];
  int trow[NJ], tcol[NJ];
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int i = tid + 512 * jj;
    trow[jj] = i / C4N;
    tcol[jj] = (i - trow[jj] * C4N) * 4;
  }
  const int ntiles = (a.M + GR_BM - 1) / GR_BM;
  int t = blockIdx.x;
  if (t >= ntiles) return;
  auto load_tile = [&](int tt) {
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      const int row = min(tt * GR_BM + trow[jj], a.M - 1);
      if (NP == 1 && !LN && ep.a_bf16) {       // bf16 mode: A stored as bf16 (dF1)
        const bf16x4 hv = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(a.A) + (size_t)row * a.lda + tcol[jj]);
        pre[jj] = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
      } else {
        pre[jj] = *reinterpret_cast<const f32x4*>(a.A + (size_t)row * a.lda + tcol[jj]);
      }
    }
  };
  auto store_tile = [&]() {            // split into the three planes on the way into LDS
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      bf16x4 h, m, l;
      if (B3_ABLATE & 16) {
        h = m = l = bf16x4{(__bf16)pre[jj][0], (__bf16)pre[jj][1], (__bf16)pre[jj][2], (__bf16)pre[jj][3]};
      } else {
        b3_split4(pre[jj], h, m, l);
      }
      const int off = trow[jj] * B3_LDP + tcol[jj];
      *reinterpret_cast<bf16x4*>(planes + off) = h;
      if (NP == 3) {
        *reinterpret_cast<bf16x4*>(planes + PLANE + off) = m;
        *reinterpret_cast<bf16x4*>(planes + 2 * PLANE + off) = l;
      }
    }
  };
  const float* auxp = ep.mask ? ep.mask : (ep.res ? ep.res : (ep.accumulate ? a.C : nullptr));
  const int auxld = ep.mask ? ep.ldmask : (ep.res ? ep.ldres : a.ldc);
  const int mode = ep.mask ? 1 : (ep.res ? 2 : (ep.accumulate ? 3 : 0));
  const int col = (nc + ct) * 16 + 4 * (lane >> 4);
  const bool colok = active && col < a.N;
  f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
  if (!LN && ep.bias && colok) bias = *reinterpret_cast<const f32x4*>(ep.bias + col);
  load_tile(t);
  __builtin_amdgcn_s_waitcnt(0x0F70);      // weight fragments, bias and the first tile
  float* As = smem;                        // the LayerNorm epilogue stages its tile over the planes
  for (; t < ntiles; t += gridDim.x) {
    store_tile();
    __syncthreads();
    if (!(B3_ABLATE & 8) && t + (int)gridDim.x < ntiles) load_tile(t + gridDim.x);     // in flight during the MFMAs and the epilogue
    f32x4 acc[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      const __bf16* frag = planes + ((rg * RT) * 16 + (lane & 15)) * B3_LDP + 8 * (lane >> 4);
#pragma unroll
      for (int kb = 0; kb < KBN; ++kb) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const __bf16* fp = (B3_ABLATE & 32) ? frag : frag + rt * 16 * B3_LDP + kb * 32;
          const bf16x8 ah = *reinterpret_cast<const bf16x8*>(fp);
          if (NP == 1) {            // bf16 mode: one product of the bf16-rounded operands
            acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[kb], ah, acc[rt], 0, 0, 0);
            continue;
          }
          const bf16x8 am = *reinterpret_cast<const bf16x8*>(fp + PLANE);
          const bf16x8 al = *reinterpret_cast<const bf16x8*>(fp + 2 * PLANE);
          f32x4 c = acc[rt];
          if (B3_ABLATE & 2) {
            acc[rt] = c + f32x4{(float)ah[0], (float)am[1], (float)al[2], 0.f};
            continue;
          }
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm[kb], am, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[kb], al, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[kb], ah, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[kb], am, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm[kb], ah, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[kb], ah, c, 0, 0, 0);
          acc[rt] = c;
        }
      }
    }
    const int m0 = t * GR_BM;
    if (!LN) {
      if (colok) {       // host guarantees 16-byte aligned epilogue operands and N % 4 == 0
        f32x4 aux[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int row = m0 + (rg * RT + rt) * 16 + (lane & 15);
          if (NP == 1 && mode == 1 && ep.mask_bf16) {      // bf16 mode: the relu stash is a bf16 array
            const bf16x4 hv = row < a.M ? *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(auxp) + (size_t)row * auxld + col)
                                        : bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
            aux[rt] = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
          } else {
            aux[rt] = (mode && row < a.M) ? *reinterpret_cast<const f32x4*>(auxp + (size_t)row * auxld + col) : f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int row = m0 + (rg * RT + rt) * 16 + (lane & 15);
          f32x4 x = acc[rt] + bias;
          if (ep.relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = fmaxf(x[r], 0.f);
          }
          if (mode == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = aux[rt][r] > 0.f ? x[r] : 0.f;
          } else {
            x += aux[rt];
          }
          if (row < a.M && !((B3_ABLATE & 1) && x[0] != 12345.678f)) {
            if (NP == 1 && ep.c_bf16)
              *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(a.C) + (size_t)row * a.ldc + col) = bf16x4{(__bf16)x[0], (__bf16)x[1], (__bf16)x[2], (__bf16)x[3]};
            else
              *reinterpret_cast<f32x4*>(a.C + (size_t)row * a.ldc + col) = x;
          }
        }
      }
    } else {
      // LayerNorm epilogue (one chunk covers all N <= 128 columns): accumulators -> LDS tile ->
      // one wave per 8 rows; every residual row is loaded before the first store
      __syncthreads();                 // all waves are done reading this A buffer
      float* Es = As;
      if (active) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          *reinterpret_cast<f32x4*>(Es + ((rg * RT + rt) * 16 + (lane & 15)) * GR_LDE + ct * 16 + 4 * (lane >> 4)) = acc[rt];
      }
      const int ncols = a.N;
      const bool ok0 = lane < ncols, ok1 = lane + 64 < ncols;
      float res0[8], res1[8];
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int row = m0 + wave * 8 + rr;
        const bool rok = row < a.M && ep.res != nullptr;
        res0[rr] = (rok && ok0) ? ep.res[(size_t)row * ep.ldres + lane] : 0.f;
        res1[rr] = (rok && ok1) ? ep.res[(size_t)row * ep.ldres + lane + 64] : 0.f;
      }
      const float bias0 = (ep.bias && ok0) ? ep.bias[lane] : 0.f, bias1 = (ep.bias && ok1) ? ep.bias[lane + 64] : 0.f;
      const float g0 = ok0 ? ep.gamma[lane] : 0.f, g1 = ok1 ? ep.gamma[lane + 64] : 0.f;
      const float be0 = ok0 ? ep.beta[lane] : 0.f, be1 = ok1 ? ep.beta[lane + 64] : 0.f;
      __syncthreads();
      const float inv_n = 1.f / (float)a.N;
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int r = wave * 8 + rr;
        const int row = m0 + r;
        float v0 = 0.f, v1 = 0.f;
        if (ok0) { v0 = Es[r * GR_LDE + lane] + bias0; if (ep.relu) v0 = fmaxf(v0, 0.f); v0 += res0[rr]; }
        if (ok1) { v1 = Es[r * GR_LDE + lane + 64] + bias1; if (ep.relu) v1 = fmaxf(v1, 0.f); v1 += res1[rr]; }
        const float mean = wave_sum(v0 + v1) * inv_n;
        const float d0 = ok0 ? v0 - mean : 0.f, d1 = ok1 ? v1 - mean : 0.f;
        const float var = wave_sum(d0 * d0 + d1 * d1) * inv_n;
        const float rs = 1.f / sqrtf(var + 1e-5f);
        if (row < a.M) {
          if (ep.rstd && lane == 0) ep.rstd[row] = rs;
          if (ok0) {
            const float xh = d0 * rs;
            if (ep.xhat) ep.xhat[(size_t)row * ep.ldxhat + lane] = xh;
            if (!ep.no_out) a.C[(size_t)row * a.ldc + lane] = xh * g0 + be0;
          }
          if (ok1) {
            const float xh = d1 * rs;
            if (ep.xhat) ep.xhat[(size_t)row * ep.ldxhat + lane + 64] = xh;
            if (!ep.no_out) a.C[(size_t)row * a.ldc + lane + 64] = xh * g1 + be1;
          }
        }
      }
    }
    __syncthreads();                   // every wave is done with the planes (fragments / LayerNorm tile)
  }
}

// K > 128 on the bf16 pipe (the data gradients through the fused q/k/v weights, K = 3d or 2d): the K extent is
// swept in 128-wide chunks, one (row tile, chunk) per iteration with the accumulators carried across the chunks of a
// tile.  The weight fragments cannot stay in registers (3 planes x K/32 blocks), so they come pre-split from a
// bf16 image made once per step by pack_b3_kernel: 12 x 16 bytes per lane per chunk from L2, no VALU work.
// K is padded to a multiple of 128 with zero planes on both sides, so the chunk loop has no conditional MFMAs.
// image layout: uint4 index ((nt * KBT + kb) * 3 + plane) * 64 + lane, KBT = 4 * ceil(K / 128).
// small contiguous vector copies (bias concatenation for the fused q/k/v projections), recorded and issued together
#define VCOPY_MAX_JOBS 32
struct VCopyJob { const float* src; float* dst; int n, pad; };
struct VCopyJobs { int n; int pad; VCopyJob j[VCOPY_MAX_JOBS]; };
__global__ void vec_copy_batch_kernel(VCopyJobs jobs) {
  const VCopyJob jb = jobs.j[blockIdx.x];
  for (int i = threadIdx.x; i < jb.n; i += blockDim.x) jb.dst[i] = jb.src[i];
}
static thread_local VCopyJobs g_vcopy;
int vec_copy_flush(hipStream_t st) {
  if (g_vcopy.n == 0) return 0;
  LAUNCH(vec_copy_batch_kernel, dim3(g_vcopy.n), dim3(256), 0, st, g_vcopy);
  g_vcopy.n = 0;
  INTEL_CHECK_LAUNCH();
  return 0;
}
int launch_vec_copy(const float* src, float* dst, int n, hipStream_t st) {
  if (n <= 0) return 0;
  if (g_vcopy.n == VCOPY_MAX_JOBS) {
    int rc = vec_copy_flush(st);
    if (rc) return rc;
  }
  VCopyJob& jb = g_vcopy.j[g_vcopy.n++];
  jb.src = src; jb.dst = dst; jb.n = n; jb.pad = 0;
  return 0;
}

#define PACK3_MAX_JOBS 16
struct Pack3Job { const float* Pf; uint4* Pb; int KG, KBT, blk0, pad; };
struct Pack3Jobs { int n; int pad; Pack3Job j[PACK3_MAX_JOBS]; };
__global__ void pack_b3_kernel(Pack3Jobs jobs) {
  int ji = 0;
  while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.j[ji + 1].blk0) ++ji;
  const float* __restrict__ Pf = jobs.j[ji].Pf;
  uint4* __restrict__ Pb = jobs.j[ji].Pb;
  const int KG = jobs.j[ji].KG, KBT = jobs.j[ji].KBT;
  const int blk = blockIdx.x - jobs.j[ji].blk0;
  const int nt = blk / KBT, kb = blk - nt * KBT;
  const int lane = threadIdx.x, n = lane & 15, j = lane >> 4;
  const int g = 2 * kb + (j >> 1);
  f32x4 w0 = f32x4{0.f, 0.f, 0.f, 0.f}, w1 = w0;
  if (g < KG) {
    const f32x4* P4 = reinterpret_cast<const f32x4*>(Pf) + ((size_t)nt * KG + g) * 64;
    const int ls = (2 * (j & 1)) * 16 + n;
    w0 = P4[ls];
    w1 = P4[ls + 16];
  }
  bf16x4 h0, m0, l0, h1, m1, l1;
  b3_split4(w0, h0, m0, l0);
  b3_split4(w1, h1, m1, l1);
  const bf16x8 H = b3_cat(h0, h1), Mi = b3_cat(m0, m1), Lo = b3_cat(l0, l1);
  uint4* dst = Pb + ((size_t)(nt * KBT + kb) * 3) * 64 + lane;
  dst[0] = __builtin_bit_cast(uint4, H);
  dst[64] = __builtin_bit_cast(uint4, Mi);
  dst[128] = __builtin_bit_cast(uint4, Lo);
}
size_t packed_b3_bytes(int Kd, int Nd) { return (size_t)(rup(Nd, 16) / 16) * (4 * cdiv(Kd, 128)) * 3 * 64 * 16; }
// conversions are recorded and issued together by pack_b3_flush (one launch per 16 weights)
static thread_local Pack3Jobs g_pack3;
static thread_local int g_pack3_blocks = 0;
// drop anything a failed earlier call left recorded
void pack_jobs_reset() {
  g_pack3.n = 0;
  g_pack3_blocks = 0;
  g_vcopy.n = 0;
}
int pack_b3_flush(hipStream_t st) {
  if (g_pack3.n == 0) return 0;
  LAUNCH(pack_b3_kernel, dim3(g_pack3_blocks), dim3(64), 0, st, g_pack3);
  g_pack3.n = 0;
  g_pack3_blocks = 0;
  INTEL_CHECK_LAUNCH();
  return 0;
}
int launch_pack_b3(const float* Pf32, int Kd, int Nd, void* Pb3, hipStream_t st) {
  const int NT = rup(Nd, 16) / 16, KG = rup(Kd, 16) / 16, KBT = 4 * cdiv(Kd, 128);
  if (g_pack3.n == PACK3_MAX_JOBS) {
    int rc = pack_b3_flush(st);
    if (rc) return rc;
  }
  Pack3Job& jb = g_pack3.j[g_pack3.n++];
  jb.Pf = Pf32; jb.Pb = reinterpret_cast<uint4*>(Pb3); jb.KG = KG; jb.KBT = KBT; jb.blk0 = g_pack3_blocks; jb.pad = 0;
  g_pack3_blocks += NT * KBT;
  return 0;
}

// A16 (bf16 mode only): the A operand is a bf16 array [M, lda] (the q/k/v gradients written by the attention backward)
template <int RT, int NP = 3, bool A16 = false>
__global__ __launch_bounds__(512, 4) void gemm_rows_b3k_kernel(GemmRowsArgs a) {
  constexpr int RG = 4 / RT;
  constexpr int CT = 8 / RG;
  constexpr int NJ = GR_BM * 32 / 512;          // float4 per thread per (tile, chunk): 4
  constexpr int PLANE = GR_BM * B3_LDP;
  constexpr bool LN = false;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* planes = reinterpret_cast<__bf16*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave / RG, rg = wave % RG;
  const int nch = (a.K + 127) >> 7, KBT = 4 * nch;
  const int NT = (a.N + 15) >> 4;
  const int nc = blockIdx.y * CT;
  const int ntc = min(CT, NT - nc);
  const bool active = ct < ntc;
  const GemmEpilogue& ep = a.ep;
  const uint4* Bimg = reinterpret_cast<const uint4*>(ep.b3) + ((size_t)(nc + (active ? ct : 0)) * KBT * 3) * 64 + lane;
  f32x4 pre[NJ];
  int trow[NJ], tcol[NJ];
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int i = tid + 512 * jj;
    trow[jj] = i >> 5;
    tcol[jj] = (i & 31) * 4;
  }
  const int ntiles = (a.M + GR_BM - 1) / GR_BM;
  auto load_iter = [&](int tt, int c) {
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      const int row = min(tt * GR_BM + trow[jj], a.M - 1);
      const int col = c * 128 + tcol[jj];
      f32x4 v;
      if (A16) {
        const bf16x4 hv = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(a.A) + (size_t)row * a.lda + min(col, a.K - 4));
        v = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
      } else {
        v = *reinterpret_cast<const f32x4*>(a.A + (size_t)row * a.lda + min(col, a.K - 4));
      }
      pre[jj] = col < a.K ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_iter = [&]() {
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      bf16x4 h, m, l;
      b3_split4(pre[jj], h, m, l);
      const int off = trow[jj] * B3_LDP + tcol[jj];
      *reinterpret_cast<bf16x4*>(planes + off) = h;
      if (NP == 3) {
        *reinterpret_cast<bf16x4*>(planes + PLANE + off) = m;
        *reinterpret_cast<bf16x4*>(planes + 2 * PLANE + off) = l;
      }
    }
  };
  const float* auxp = ep.mask ? ep.mask : (ep.res ? ep.res : (ep.accumulate ? a.C : nullptr));
  const int auxld = ep.mask ? ep.ldmask : (ep.res ? ep.ldres : a.ldc);
  const int mode = ep.mask ? 1 : (ep.res ? 2 : (ep.accumulate ? 3 : 0));
  const int col = (nc + ct) * 16 + 4 * (lane >> 4);
  const bool colok = active && col < a.N;
  f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
  if (ep.bias && colok) bias = *reinterpret_cast<const f32x4*>(ep.bias + col);
  int t = blockIdx.x;
  if (t >= ntiles) return;
  int c = 0;
  load_iter(t, 0);
  f32x4 acc[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  while (t < ntiles) {
    // weight fragments of this chunk: in flight while the A chunk is split and stored
    uint4 bw[4][3];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) bw[kb][pl] = Bimg[((size_t)(c * 4 + kb) * 3 + pl) * 64];
    store_iter();
    __syncthreads();
    {   // prefetch the next (tile, chunk)
      int tn = t, cn = c + 1;
      if (cn == nch) { cn = 0; tn += gridDim.x; }
      if (tn < ntiles) load_iter(tn, cn);
    }
    {
      const __bf16* frag = planes + ((rg * RT) * 16 + (lane & 15)) * B3_LDP + 8 * (lane >> 4);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, bw[kb][0]);
        if (NP == 1) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, *reinterpret_cast<const bf16x8*>(frag + rt * 16 * B3_LDP + kb * 32), acc[rt], 0, 0, 0);
          continue;
        }
        const bf16x8 bm = __builtin_bit_cast(bf16x8, bw[kb][1]), bl = __builtin_bit_cast(bf16x8, bw[kb][2]);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const __bf16* fp = frag + rt * 16 * B3_LDP + kb * 32;
          const bf16x8 ah = *reinterpret_cast<const bf16x8*>(fp);
          const bf16x8 am = *reinterpret_cast<const bf16x8*>(fp + PLANE);
          const bf16x8 al = *reinterpret_cast<const bf16x8*>(fp + 2 * PLANE);
          f32x4 cc = acc[rt];
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, am, cc, 0, 0, 0);
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al, cc, 0, 0, 0);
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah, cc, 0, 0, 0);
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, am, cc, 0, 0, 0);
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, ah, cc, 0, 0, 0);
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, cc, 0, 0, 0);
          acc[rt] = cc;
        }
      }
    }
    if (c == nch - 1) {
      const int m0 = t * GR_BM;
      {
        if (colok) {       // host guarantees 16-byte aligned epilogue operands and N % 4 == 0
          f32x4 aux[RT];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const int row = m0 + (rg * RT + rt) * 16 + (lane & 15);
            aux[rt] = (mode && row < a.M) ? *reinterpret_cast<const f32x4*>(auxp + (size_t)row * auxld + col) : f32x4{0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const int row = m0 + (rg * RT + rt) * 16 + (lane & 15);
            f32x4 x = acc[rt] + bias;
            if (ep.relu) {
#pragma unroll
              for (int r = 0; r < 4; ++r) x[r] = fmaxf(x[r], 0.f);
            }
            if (mode == 1) {
#pragma unroll
              for (int r = 0; r < 4; ++r) x[r] = aux[rt][r] > 0.f ? x[r] : 0.f;
            } else {
              x += aux[rt];
            }
            if (row < a.M) *reinterpret_cast<f32x4*>(a.C + (size_t)row * a.ldc + col) = x;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < RT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    if (++c == nch) { c = 0; t += gridDim.x; }
  }
  (void)LN;
}

template <int RT>
static int launch_b3k(const GemmRowsArgs& a, hipStream_t st) {
  constexpr int CT = 8 / (4 / RT);
  const int ntiles = cdiv(a.M, GR_BM), nchunks = cdiv(rup(a.N, 16) / 16, CT);
  int gx = ntiles < 512 ? ntiles : 512;
  if (gx * nchunks > 512) gx = cdiv(512, nchunks) < ntiles ? cdiv(512, nchunks) : ntiles;
  const size_t smem = (size_t)3 * GR_BM * B3_LDP * sizeof(__bf16);
  if (a.ep.a_bf16) {
    INTEL_CHECK_ARG(g_planes == 1, "gemm_rows: a bf16-stored A operand needs the bf16 mode");
    allow_lds((gemm_rows_b3k_kernel<RT, 1, true>), smem);
    LAUNCH_S(a.M, a.N, a.K, 2.0 * a.M * a.N * a.K, gemm_algorithmic_bytes(a),
             (gemm_rows_b3k_kernel<RT, 1, true>), dim3(gx, nchunks), dim3(512), smem, st, a);
    INTEL_CHECK_LAUNCH();
    return 0;
  }
  if (g_planes == 1) {
    allow_lds((gemm_rows_b3k_kernel<RT, 1>), smem);
    LAUNCH_S(a.M, a.N, a.K, 2.0 * a.M * a.N * a.K, gemm_algorithmic_bytes(a),
             (gemm_rows_b3k_kernel<RT, 1>), dim3(gx, nchunks), dim3(512), smem, st, a);
    INTEL_CHECK_LAUNCH();
    return 0;
  }
  allow_lds(gemm_rows_b3k_kernel<RT>, smem);
  LAUNCH_S(a.M, a.N, a.K, 2.0 * a.M * a.N * a.K, gemm_algorithmic_bytes(a),
           gemm_rows_b3k_kernel<RT>, dim3(gx, nchunks), dim3(512), smem, st, a);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------
// gemm_rows for K > 128 (the data gradients through the fused q/k/v weights: K = 3d), same 8-wave
// persistent structure; the K extent is swept in 128-wide chunks: one (row tile, k chunk) per
// iteration, accumulators carried across the chunks of a tile, B fragments of the chunk re-read from
// L2 at the top of the iteration (they cannot stay resident: 3 x 32 VGPRs), epilogue after the last
// chunk.  No LayerNorm epilogue (never needed with K > 128).
// ------------------------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(512, 4) void gemm_rows_w8k_kernel(GemmRowsArgs a) {
  constexpr int RG = 4 / RT;
  constexpr int CT = 8 / RG;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave / RG, rg = wave % RG;
  const int Kp = (a.K + 15) & ~15, KG = Kp >> 4;
  const int nch = (KG + 7) >> 3;                      // 128-wide k chunks
  const int NT = (a.N + 15) >> 4;
  const int nc = blockIdx.y * CT;
  const int ntc = min(CT, NT - nc);
  const bool active = ct < ntc;
  const GemmEpilogue& ep = a.ep;
  f32x4 pre[4];
  int trow[4], tcol[4], loff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = tid + 512 * j;          // 64 rows x 32 float4 = 2048 slots
    trow[j] = i >> 5;
    tcol[j] = (i & 31) * 4;
    loff[j] = trow[j] * GR_LDA + tcol[j];
  }
  const int ntiles = (a.M + GR_BM - 1) / GR_BM;
  const int niter = ntiles * nch;
  // iteration it = (tile, chunk); this workgroup owns tiles blockIdx.x, +gridDim.x, ...
  auto load_iter = [&](int tt, int c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = min(tt * GR_BM + trow[j], a.M - 1);
      const int col = c * GR_KC + tcol[j];
      pre[j] = (col < a.K) ? *reinterpret_cast<const f32x4*>(a.A + (size_t)row * a.lda + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_iter = [&](float* As) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(As + loff[j]) = pre[j];
  };
  const float* auxp = ep.mask ? ep.mask : (ep.res ? ep.res : (ep.accumulate ? a.C : nullptr));
  const int auxld = ep.mask ? ep.ldmask : (ep.res ? ep.ldres : a.ldc);
  const int mode = ep.mask ? 1 : (ep.res ? 2 : (ep.accumulate ? 3 : 0));
  const int col = (nc + ct) * 16 + 4 * (lane >> 4);
  const bool colok = active && col < a.N;
  f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
  if (ep.bias && colok) bias = *reinterpret_cast<const f32x4*>(ep.bias + col);
  (void)niter;
  int t = blockIdx.x;
  if (t >= ntiles) return;
  int c = 0;
  load_iter(t, 0);
  store_iter(smem);
  __builtin_amdgcn_s_waitcnt(0x0F70);      // retire bias / first tile loads before the loop
  __syncthreads();
  {   // prefetch the second iteration
    int tn = t, cn = 1;
    if (cn == nch) { cn = 0; tn += gridDim.x; }
    if (tn < ntiles) load_iter(tn, cn);
  }
  int buf = 0;
  f32x4 acc[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  while (t < ntiles) {
    float* As = smem + buf * (GR_BM * GR_LDA);
    const int kg0 = c * 8, kgn = min(8, KG - kg0);
    if (active) {
      f32x4 bfr[8];
#pragma unroll
      for (int g = 0; g < 8; ++g)
        if (g < kgn) bfr[g] = reinterpret_cast<const f32x4*>(a.Bp)[((size_t)(nc + ct) * KG + kg0 + g) * 64 + lane];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if (g < kgn) {
          f32x4 af[RT];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            af[rt] = *reinterpret_cast<const f32x4*>(As + ((rg * RT + rt) * 16 + (lane & 15)) * GR_LDA + g * 16 + 4 * (lane >> 4));
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = mfma16(bfr[g][s], af[rt][s], acc[rt]);
        }
      }
    }
    // next iteration's coordinates
    int tn = t, cn = c + 1;
    if (cn == nch) { cn = 0; tn += gridDim.x; }
    if (tn < ntiles) store_iter(smem + (buf ^ 1) * (GR_BM * GR_LDA));
    if (c == nch - 1) {          // tile finished: epilogue (all loads before the stores)
      const int m0 = t * GR_BM;
      if (colok) {
        f32x4 aux[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int row = m0 + (rg * RT + rt) * 16 + (lane & 15);
          aux[rt] = (mode && row < a.M) ? *reinterpret_cast<const f32x4*>(auxp + (size_t)row * auxld + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int row = m0 + (rg * RT + rt) * 16 + (lane & 15);
          f32x4 x = acc[rt] + bias;
          if (ep.relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = fmaxf(x[r], 0.f);
          }
          if (mode == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = aux[rt][r] > 0.f ? x[r] : 0.f;
          } else {
            x += aux[rt];
          }
          if (row < a.M) *reinterpret_cast<f32x4*>(a.C + (size_t)row * a.ldc + col) = x;
          acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    __syncthreads();
    {   // prefetch the iteration after next
      int t2 = tn, c2 = cn + 1;
      if (c2 == nch) { c2 = 0; t2 += gridDim.x; }
      if (tn < ntiles && t2 < ntiles) load_iter(t2, c2);
    }
    t = tn;
    c = cn;
    buf ^= 1;
  }
}

template <int RT>
static int launch_w8k(const GemmRowsArgs& a, hipStream_t st) {
  constexpr int CT = 8 / (4 / RT);
  const int ntiles = cdiv(a.M, GR_BM), nchunks = cdiv(rup(a.N, 16) / 16, CT);
  int gx = ntiles < 512 ? ntiles : 512;
  if (gx * nchunks > 512) gx = cdiv(512, nchunks) < ntiles ? cdiv(512, nchunks) : ntiles;
  size_t smem = (size_t)(2 * GR_BM * GR_LDA) * sizeof(float);
  allow_lds(gemm_rows_w8k_kernel<RT>, smem);
  LAUNCH_S(a.M, a.N, a.K, 2.0 * a.M * a.N * a.K, gemm_algorithmic_bytes(a),
           gemm_rows_w8k_kernel<RT>, dim3(gx, nchunks), dim3(512), smem, st, a);
  INTEL_CHECK_LAUNCH();
  return 0;
}

template <int RT, bool LN>
static int launch_b3(const GemmRowsArgs& a, hipStream_t st) {
  constexpr int CT = 8 / (4 / RT);
  const int ntiles = cdiv(a.M, GR_BM), nchunks = cdiv(rup(a.N, 16) / 16, CT);
  int gx = ntiles < 512 ? ntiles : 512;
  if (gx * nchunks > 512) gx = cdiv(512, nchunks) < ntiles ? cdiv(512, nchunks) : ntiles;
  // several column chunks per row tile (N = 3d: the fused q/k/v projection): workgroup (x, y) has linear id x + y * gx and lands
  // on XCD id % 8, so with gx a multiple of 8 the chunks of one row tile -- which walk the same row tiles in step -- share an
  // XCD and its L2: the A tile comes from HBM once instead of once per chunk
  if (nchunks > 1 && gx >= 16) gx &= ~7;
  const size_t smem = (size_t)3 * GR_BM * B3_LDP * sizeof(__bf16);      // >= the LayerNorm staging tile (64 x 132 floats)
  if (g_planes == 1) {          // bf16 mode (the LayerNorm epilogue tile still needs the full staging area)
    if (a.K == 128) {
      allow_lds((gemm_rows_b3_kernel<RT, LN, 4, 1>), smem);
      LAUNCH_S(a.M, a.N, a.K, 2.0 * a.M * a.N * a.K, gemm_algorithmic_bytes(a),
               (gemm_rows_b3_kernel<RT, LN, 4, 1>), dim3(gx, nchunks), dim3(512), smem, st, a);
    } else {
      allow_lds((gemm_rows_b3_kernel<RT, LN, 2, 1>), smem);
      LAUNCH_S(a.M, a.N, a.K, 2.0 * a.M * a.N * a.K, gemm_algorithmic_bytes(a),
               (gemm_rows_b3_kernel<RT, LN, 2, 1>), dim3(gx, nchunks), dim3(512), smem, st, a);
    }
    INTEL_CHECK_LAUNCH();
    return 0;
  }
  if (a.K == 128) {
    allow_lds((gemm_rows_b3_kernel<RT, LN, 4>), smem);
    LAUNCH_S(a.M, a.N, a.K, 2.0 * a.M * a.N * a.K, gemm_algorithmic_bytes(a),
             (gemm_rows_b3_kernel<RT, LN, 4>), dim3(gx, nchunks), dim3(512), smem, st, a);
  } else {
    allow_lds((gemm_rows_b3_kernel<RT, LN, 2>), smem);
    LAUNCH_S(a.M, a.N, a.K, 2.0 * a.M * a.N * a.K, gemm_algorithmic_bytes(a),
             (gemm_rows_b3_kernel<RT, LN, 2>), dim3(gx, nchunks), dim3(512), smem, st, a);
  }
  INTEL_CHECK_LAUNCH();
  return 0;
}

template <int RT, bool LN>
static int launch_w8(const GemmRowsArgs& a, hipStream_t st) {
  constexpr int CT = 8 / (4 / RT);
  const int ntiles = cdiv(a.M, GR_BM), nchunks = cdiv(rup(a.N, 16) / 16, CT);
  int gx = ntiles < 512 ? ntiles : 512;
  if (gx * nchunks > 512) gx = cdiv(512, nchunks) < ntiles ? cdiv(512, nchunks) : ntiles;
  size_t smem = (size_t)(2 * GR_BM * GR_LDA) * sizeof(float);
  allow_lds((gemm_rows_w8_kernel<RT, LN>), smem);
  LAUNCH_S(a.M, a.N, a.K, 2.0 * a.M * a.N * a.K, gemm_algorithmic_bytes(a),
           (gemm_rows_w8_kernel<RT, LN>), dim3(gx, nchunks), dim3(512), smem, st, a);
  INTEL_CHECK_LAUNCH();
  return 0;
}

int launch_gemm_rows(const float* A, int lda, int M, int K, const float* Bp, int N, float* C, int ldc,
                     const GemmEpilogue& ep, hipStream_t st) {
  if (M <= 0 || N <= 0) return 0;
  INTEL_CHECK_ARG(K > 0, "gemm_rows: K must be positive");
  INTEL_CHECK_ARG(!(ep.gamma && N > 128), "gemm_rows: fused LayerNorm needs N <= 128 (got %d)", N);
  GemmRowsArgs a;
  static const int dbg = INTEL_DEBUG_ENV("INTEL_DEBUG_GEMM", 0);      // ablation bits: debug builds only (common.h)
  a.dbg = dbg;
  {
    uintptr_t bits = reinterpret_cast<uintptr_t>(C) | (uintptr_t)(ldc & 3) << 60;
    bool ok = ((reinterpret_cast<uintptr_t>(C) & 15) == 0) && ((ldc & 3) == 0);
    if (ep.bias) ok = ok && ((reinterpret_cast<uintptr_t>(ep.bias) & 15) == 0);
    if (ep.mask) ok = ok && ((reinterpret_cast<uintptr_t>(ep.mask) & 15) == 0) && ((ep.ldmask & 3) == 0);
    if (ep.res) ok = ok && ((reinterpret_cast<uintptr_t>(ep.res) & 15) == 0) && ((ep.ldres & 3) == 0);
    (void)bits;
    a.vec_ep = ok ? 1 : 0;
  }
  a.A = A; a.lda = lda; a.M = M; a.K = K; a.Bp = Bp; a.N = N; a.C = C; a.ldc = ldc; a.ep = ep;
  const bool vecA = ((lda & 3) == 0) && ((K & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  INTEL_CHECK_ARG(!(ep.gamma && (ep.mask || ep.accumulate)), "gemm_rows: LayerNorm epilogue cannot be combined with mask/accumulate");
  if (rup(K, 16) <= GR_KC && vecA && a.vec_ep && (N & 3) == 0 && !((ep.mask || ep.res) && ep.accumulate) && !(ep.mask && ep.res)) {
    static int use_b3 = -1;
    if (use_b3 < 0) { const char* e = getenv("INTEL_GEMM_B3"); use_b3 = (e && e[0] == '0') ? 0 : 1; }
    if (use_b3 && (K == 128 || K == 64)) {         // bf16 matrix pipe, three-plane split (fp32 accuracy)
      if (ep.gamma) {
        if (N > 64) return launch_b3<4, true>(a, st);
        if (N > 32) return launch_b3<2, true>(a, st);
        return launch_b3<1, true>(a, st);
      }
      // few row tiles (the B-row products of the pooling / fusion chains): 32-column workgroups, one tile pair per wave --
      // four times the workgroups and a quarter of the MFMA chain per wave (the A tile is re-read from L2 by the column chunks)
      static const int small_m = INTEL_DEBUG_ENV("INTEL_GEMM_SMALLM", 8192);
      if (M <= small_m && N > 32) return launch_b3<1, false>(a, st);
      if (N > 64) return launch_b3<4, false>(a, st);
      if (N > 32) return launch_b3<2, false>(a, st);
      return launch_b3<1, false>(a, st);
    }
    if (ep.gamma) {
      if (N > 64) return launch_w8<4, true>(a, st);
      if (N > 32) return launch_w8<2, true>(a, st);
      return launch_w8<1, true>(a, st);
    }
    if (N > 64) return launch_w8<4, false>(a, st);
    if (N > 32) return launch_w8<2, false>(a, st);
    return launch_w8<1, false>(a, st);
  }
  if (ep.a_bf16 || ep.c_bf16 || ep.mask_bf16) {
    static int use_b3x = -1;
    if (use_b3x < 0) { const char* e = getenv("INTEL_GEMM_B3"); use_b3x = (e && e[0] == '0') ? 0 : 1; }
    const bool common = g_planes == 1 && use_b3x && vecA && a.vec_ep && (N & 3) == 0 && !ep.gamma && !ep.accumulate && !(ep.mask && ep.res);
    const bool big_k = rup(K, 16) > GR_KC && ep.b3 && (K & 3) == 0 && !ep.c_bf16 && !ep.mask_bf16;      // gemm_rows_b3k: A only
    const bool small_k = (K == 128 || K == 64);                                                          // gemm_rows_b3: A, C, mask
    INTEL_CHECK_ARG(common && (big_k || small_k), "gemm_rows: bf16-stored operands need the bf16 mode and a bf16-pipe kernel (K = 64, 128, or K > 128 with a weight image)");
  }
  if (rup(K, 16) > GR_KC && vecA && a.vec_ep && (N & 3) == 0 && !ep.gamma && !((ep.mask || ep.res) && ep.accumulate) &&
      !(ep.mask && ep.res)) {
    static int use_b3k = -1;
    if (use_b3k < 0) { const char* e = getenv("INTEL_GEMM_B3"); use_b3k = (e && e[0] == '0') ? 0 : 1; }
    if (use_b3k && ep.b3 && (K & 3) == 0) {      // pre-split weight image available: bf16 pipe
      if (N > 64) return launch_b3k<4>(a, st);
      if (N > 32) return launch_b3k<2>(a, st);
      return launch_b3k<1>(a, st);
    }
    if (N > 64) return launch_w8k<4>(a, st);
    if (N > 32) return launch_w8k<2>(a, st);
    return launch_w8k<1>(a, st);
  }
  if (!ep.gamma && M <= 16384) {      // B-row chains: 16-row workgroups (see gemm_rows_small_kernel)
    if (N <= 32) {
      LAUNCH_S(M, N, K, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)K * N + (double)M * N), gemm_rows_small_kernel<true>, dim3(cdiv(M, 16)), dim3(256), 0, st, a);
      INTEL_CHECK_LAUNCH();
      return 0;
    }
    if (rup(K, 16) <= GR_KC) {
      LAUNCH_S(M, N, K, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)K * N + (double)M * N), gemm_rows_small_kernel<false>, dim3(cdiv(M, 16)), dim3(256), 0, st, a);
      INTEL_CHECK_LAUNCH();
      return 0;
    }
  }
  size_t smem = (size_t)(GR_BM * GR_LDA) * sizeof(float);
  allow_lds(gemm_rows_kernel, smem);
  LAUNCH_S(M, N, K, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)K * N + (double)M * N), gemm_rows_kernel, dim3(cdiv(M, GR_BM)), dim3(256), smem, st, a);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------
// wgrad: dW[N,K] = sum_m dY[m,n] X[m,k]; db[n] = sum_m dY[m,n]
//
// The reduction runs over the B*L rows: every workgroup sweeps a strided set of 32-row tiles and
// keeps a full (<=128 x <=128) partial product in accumulators (wave (wn,wk) owns a 4x4 block of
// 16x16 tiles).  Per 32 rows a CU moves 32 KB from HBM and issues 1 MFLOP: balanced against both
// roofs, so the grid is sized for 4 workgroups per CU (1024 slabs) to overlap loads and MFMAs.
// Partials go to per-workgroup slabs and are summed in a fixed order (bitwise reproducible).
// ------------------------------------------------------------------------------------------
#define WG_RT 32          // rows per LDS tile
#define WG_MAXS 512       // max slabs: 2 resident workgroups per CU (72 KB LDS each)

static inline int wgrad_num_slabs(int M, int N = 128, int K = 128) {
  static const int maxs = [] { const char* e = getenv("INTEL_WGRAD_SLABS"); int v = e ? atoi(e) : 128; return v < 1 ? 1 : (v > WG_MAXS ? WG_MAXS : v); }();
  // default 128 = one workgroup on every second CU: with the products on the bf16 pipe the STEP is HBM-bound, and the slab traffic (written
  // here, read by the batched reduction: 0.6 GB of 12.7 GB per headline step at 256 slabs) matters more than this kernel's own time --
  // measured same-box at the headline (round 4, 100 steps, twice): 64 / 96 / 128 / 160 / 192 / 256 slabs = 1.166 / 1.191 / 1.204 / 1.198 /
  // 1.175 / 1.170 M sessions/s although the kernel itself takes 0.90 instead of 0.67 ms per step at 128; bf16 mode +1 %, published
  // hyper-parameters +1.5 %, LifeData / stress unchanged.
  // Narrow products (N + K <= 128: 30 KB of LDS, 16 KB slabs) are latency-bound instead: two workgroups per CU.
  int cap = (N + K <= 128 && N % 32 == 0 && K % 32 == 0) ? 2 * maxs : maxs;
  if (cap > WG_MAXS) cap = WG_MAXS;
  // several 128-column blocks of dY against a 128-wide X (the fused q/k/v weight gradient, N = 3d): all S x N/128 workgroups
  // should be resident at once (two per CU) -- no half-empty second wave, and the column blocks of one slab, which share an
  // XCD when S is a multiple of 8, read their common X tile through the same L2
  if (N > 128 && K > 64) {
    const int ny = cdiv(N, 128), fit = ((2 * 256) / ny) & ~7;
    if (cap > fit) cap = fit;
  }
  int s = cdiv(M, WG_RT);
  return s < 1 ? 1 : (s > cap ? cap : s);
}
size_t wgrad_slab_floats(int M, int N, int K) { return (size_t)wgrad_num_slabs(M, N, K) * ((size_t)N * K + N); }

struct WgradArgs {
  const float* dY; int lddy; const float* X; int ldx; int M, N, K;
  float* slabs; int S; int want_db;
  int dy_bf16, x_bf16;      // bf16 mode: dY / X is a bf16 array (wgrad_b3_kernel with NP = 1 only)
};

// the next 32-row tile is prefetched global -> registers (16-byte loads when the operands are aligned)
// while the MFMAs of the current tile run from LDS; two LDS buffers, one barrier per tile.
__global__ __launch_bounds__(256) void wgrad_pipe_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.y * 128, k0 = blockIdx.z * 128;
  const int nb = min(128, a.N - n0), kb = min(128, a.K - k0);
  const int NTb = (nb + 15) >> 4, KTb = (kb + 15) >> 4;
  const int ldy = ((NTb * 16 + 31) & ~31) + 16, ldx = ((KTb * 16 + 31) & ~31) + 16;
  const int bufsz = WG_RT * (ldy + ldx);
  const int wn = wave >> 1, wk = wave & 1;
  const int ntw = (NTb + 1) >> 1, ktw = (KTb + 1) >> 1;
  const int nt0 = wn * ntw, kt0 = wk * ktw;
  const bool vecY = ((a.lddy & 3) == 0) && ((nb & 3) == 0) && ((n0 & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.dY) & 15) == 0);
  const bool vecX = ((a.ldx & 3) == 0) && ((kb & 3) == 0) && ((k0 & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.X) & 15) == 0);
  const int cy = NTb * 4, cx = KTb * 4;                  // float4 per row
  const int njy = (WG_RT * cy + 255) >> 8, njx = (WG_RT * cx + 255) >> 8;   // float4 per thread (<= 4 each)
  f32x4 py[4], px[4];
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbacc = 0.f;
  const int ntiles = (a.M + WG_RT - 1) / WG_RT;
  int t = blockIdx.x;
  auto load_tile = [&](int tt) {
    const int m0 = tt * WG_RT;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      py[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (j < njy) {
        const int i = tid + 256 * j;
        const int r = i / cy, c = (i - r * cy) * 4;
        const int row = m0 + r;
        if (r < WG_RT && row < a.M && c < nb) {
          if (vecY) py[j] = *reinterpret_cast<const f32x4*>(a.dY + (size_t)row * a.lddy + n0 + c);
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) py[j][e] = (c + e < nb) ? a.dY[(size_t)row * a.lddy + n0 + c + e] : 0.f;
          }
        }
      }
      px[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (j < njx) {
        const int i = tid + 256 * j;
        const int r = i / cx, c = (i - r * cx) * 4;
        const int row = m0 + r;
        if (r < WG_RT && row < a.M && c < kb) {
          if (vecX) px[j] = *reinterpret_cast<const f32x4*>(a.X + (size_t)row * a.ldx + k0 + c);
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) px[j][e] = (c + e < kb) ? a.X[(size_t)row * a.ldx + k0 + c + e] : 0.f;
          }
        }
      }
    }
  };
  if (t < ntiles) load_tile(t);
  int buf = 0;
  for (; t < ntiles; t += a.S) {
    float* Ys = smem + buf * bufsz;
    float* Xs = Ys + WG_RT * ldy;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < njy) {
        const int i = tid + 256 * j;
        const int r = i / cy, c = (i - r * cy) * 4;
        if (r < WG_RT) *reinterpret_cast<f32x4*>(Ys + r * ldy + c) = py[j];
      }
      if (j < njx) {
        const int i = tid + 256 * j;
        const int r = i / cx, c = (i - r * cx) * 4;
        if (r < WG_RT) *reinterpret_cast<f32x4*>(Xs + r * ldx + c) = px[j];
      }
    }
    __syncthreads();
    if (t + a.S < ntiles) load_tile(t + a.S);
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
    buf ^= 1;
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

// ------------------------------------------------------------------------------------------
// wgrad_b3_kernel<NTW,KTW>: the slab product on the bf16 matrix pipe with the three-plane split of gemm_rows_b3
// (fp32 accuracy).  The reduction runs over ROWS, so one v_mfma_f32_16x16x32_bf16 consumes a whole 32-row tile and
// both operands are needed "k-contiguous": 8 consecutive rows of one column per lane.  The tile is therefore
// staged TRANSPOSED: a thread loads a 4-row x 4-column block (four 16-byte loads), splits its 16 values and writes,
// per column, the four rows as one 8-byte store into planesT[plane][column][row] (40-element = 80-byte column
// pitch, see WB_LDT).  Single LDS stage (61 KB at 128+128 columns -> two workgroups per
// CU), the next tile is prefetched into registers under the MFMAs.  db = column sums of dY is accumulated by the
// staging threads in registers (a thread always owns the same four columns).
// ------------------------------------------------------------------------------------------
#ifndef WB_ABLATE          // tools/wgrad_ablate.sh: 1 no MFMAs, 2 no LDS staging stores (the loaded registers are only waited for), 4 only the first tile is loaded
#define WB_ABLATE 0
#endif
#define WB_LDT 40          // 80-byte column pitch: four columns = 2.5 bank rows -> the 8-byte stores of a 16-lane group (two column
                           // blocks x eight row blocks) fall into disjoint bank halves; the b128 fragment reads are 2-way on 3 of 16 slots
// TAIL: M is not a multiple of 32 (zero-padded last tile); Y16 / X16 (bf16 mode, NP = 1): dY / X is stored as a bf16 array
// (compile-time: a run-time choice in the tile loader costs this kernel 80 % of its speed)
template <int NTW, int KTW, int NP = 3, bool TAIL = false, bool Y16 = false, bool X16 = false>
__device__ __forceinline__ void wgrad_b3_body(const WgradArgs& a, const int bx, const int by, const int bz, const int S) {
  constexpr int NB = 32 * NTW, KB = 32 * KTW;
  constexpr int YBL = 8 * (NB / 4), XBL = 8 * (KB / 4);          // 4x4 blocks per tile of each operand
  constexpr int YPT = (YBL + 255) / 256, XPT = (XBL + 255) / 256;  // blocks per thread (1 at 128 columns)
  constexpr int PLANE = (NB + KB) * WB_LDT;                        // bf16 elements per plane
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* planes = reinterpret_cast<__bf16*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  const int n0 = by * NB, k0 = bz * KB;
  const int ntiles = (a.M + WG_RT - 1) / WG_RT;      // a ragged last tile is padded with zero rows
  // staging blocks of this thread: block b -> rows 4*(b % 8) .., columns 4*(b / 8) ..  (row block fastest: the stores of
  // eight adjacent lanes fill one column's 64 bytes)
  f32x4 py[YPT][4], px[XPT][4];
  float dbacc[YPT][4];
#pragma unroll
  for (int u = 0; u < YPT; ++u)
#pragma unroll
    for (int c = 0; c < 4; ++c) dbacc[u][c] = 0.f;
  auto ldy = [&](size_t row, int col) -> f32x4 {
    if (Y16) {
      const bf16x4 hv = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(a.dY) + row * a.lddy + col);
      return f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
    }
    return *reinterpret_cast<const f32x4*>(a.dY + row * a.lddy + col);
  };
  auto ldxv = [&](size_t row, int col) -> f32x4 {
    if (X16) {
      const bf16x4 hv = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(a.X) + row * a.ldx + col);
      return f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
    }
    return *reinterpret_cast<const f32x4*>(a.X + row * a.ldx + col);
  };
  auto load_tile = [&](int tt) {
    const size_t m0 = (size_t)tt * WG_RT;
    const bool full = !TAIL || (tt + 1) * WG_RT <= a.M;  // workgroup-uniform: only a ragged last tile takes the guarded loads
#pragma unroll
    for (int u = 0; u < YPT; ++u) {
      const int b = min(tid + 256 * u, YBL - 1), rb = b & 7, cb = b >> 3;
      if (full) {
#pragma unroll
        for (int r = 0; r < 4; ++r) py[u][r] = ldy(m0 + 4 * rb + r, n0 + 4 * cb);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const size_t row = m0 + 4 * rb + r;
          const f32x4 v = ldy(min(row, (size_t)a.M - 1), n0 + 4 * cb);
          py[u][r] = row < (size_t)a.M ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
#pragma unroll
    for (int u = 0; u < XPT; ++u) {
      const int b = min(tid + 256 * u, XBL - 1), rb = b & 7, cb = b >> 3;
      if (full) {
#pragma unroll
        for (int r = 0; r < 4; ++r) px[u][r] = ldxv(m0 + 4 * rb + r, k0 + 4 * cb);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const size_t row = m0 + 4 * rb + r;
          const f32x4 v = ldxv(min(row, (size_t)a.M - 1), k0 + 4 * cb);
          px[u][r] = row < (size_t)a.M ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
  };
  auto store_block = [&](const f32x4 (&v)[4], int colbase, int rb, int cb) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 colv = f32x4{v[0][c], v[1][c], v[2][c], v[3][c]};       // four consecutive rows of one column
      bf16x4 h, m, l;
      b3_split4(colv, h, m, l);
      const int off = (colbase + 4 * cb + c) * WB_LDT + 4 * rb;
      *reinterpret_cast<bf16x4*>(planes + off) = h;
      if (NP == 3) {
        *reinterpret_cast<bf16x4*>(planes + PLANE + off) = m;
        *reinterpret_cast<bf16x4*>(planes + 2 * PLANE + off) = l;
      }
    }
  };
  auto store_tile = [&]() {
    if (WB_ABLATE & 2) {
#pragma unroll
      for (int u = 0; u < YPT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" ::"v"(py[u][r]));
#pragma unroll
      for (int u = 0; u < XPT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" ::"v"(px[u][r]));
      return;
    }
#pragma unroll
    for (int u = 0; u < YPT; ++u) {
      const int b = tid + 256 * u;
      if (b < YBL) {
        const int rb = b & 7, cb = b >> 3;
        store_block(py[u], 0, rb, cb);
#pragma unroll
        for (int c = 0; c < 4; ++c) dbacc[u][c] += (py[u][0][c] + py[u][1][c]) + (py[u][2][c] + py[u][3][c]);
      }
    }
#pragma unroll
    for (int u = 0; u < XPT; ++u) {
      const int b = tid + 256 * u;
      if (b < XBL) {
        const int rb = b & 7, cb = b >> 3;
        store_block(px[u], NB, rb, cb);
      }
    }
  };
  f32x4 acc[NTW][KTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int j = 0; j < KTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int t = bx;
  if (t < ntiles) load_tile(t);
  // fragment addresses: column (wn*16*NTW + i*16 + p) of dY, (NB + wk*16*KTW + j*16 + p) of X, rows 8g .. 8g+7
  const __bf16* fy = planes + (wn * 16 * NTW + p) * WB_LDT + 8 * g;
  const __bf16* fx = planes + (NB + wk * 16 * KTW + p) * WB_LDT + 8 * g;
  for (; t < ntiles; t += S) {
    store_tile();
    __syncthreads();
    if (!(WB_ABLATE & 4) && t + S < ntiles) load_tile(t + S);
    if (WB_ABLATE & 1) { __syncthreads(); continue; }
    bf16x8 xh[KTW], xm[KTW], xl[KTW];
#pragma unroll
    for (int j = 0; j < KTW; ++j) {
      const __bf16* q = fx + j * 16 * WB_LDT;
      xh[j] = *reinterpret_cast<const bf16x8*>(q);
      if (NP == 3) {
        xm[j] = *reinterpret_cast<const bf16x8*>(q + PLANE);
        xl[j] = *reinterpret_cast<const bf16x8*>(q + 2 * PLANE);
      }
    }
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      const __bf16* q = fy + i * 16 * WB_LDT;
      const bf16x8 yh = *reinterpret_cast<const bf16x8*>(q);
      if (NP == 1) {            // bf16 mode: one product of the bf16-rounded operands
#pragma unroll
        for (int j = 0; j < KTW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yh, xh[j], acc[i][j], 0, 0, 0);
        continue;
      }
      const bf16x8 ym = *reinterpret_cast<const bf16x8*>(q + PLANE);
      const bf16x8 yl = *reinterpret_cast<const bf16x8*>(q + 2 * PLANE);
#pragma unroll
      for (int j = 0; j < KTW; ++j) {
        f32x4 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ym, xm[j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yh, xl[j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yl, xh[j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yh, xm[j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ym, xh[j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yh, xh[j], c, 0, 0, 0);
        acc[i][j] = c;
      }
    }
    __syncthreads();
  }
  float* slab = a.slabs + (size_t)bx * ((size_t)a.N * a.K + a.N);
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int j = 0; j < KTW; ++j) {
      const int k = k0 + wk * 16 * KTW + j * 16 + p;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 16 * NTW + i * 16 + 4 * g + r;
        slab[(size_t)n * a.K + k] = acc[i][j][r];
      }
    }
  if (a.want_db && bz == 0) {
    // column sums: the 8 row-blocks of a column block live in threads cb, cb + NB/4, ... ; reduce through LDS
    float* red = smem;                          // [8][NB] after the last barrier of the loop
#pragma unroll
    for (int u = 0; u < YPT; ++u) {
      const int b = tid + 256 * u;
      if (b < YBL) {
        const int rb = b & 7, cb = b >> 3;
#pragma unroll
        for (int c = 0; c < 4; ++c) red[rb * NB + 4 * cb + c] = dbacc[u][c];
      }
    }
    __syncthreads();
    if (tid < NB) {
      float sdb = 0.f;
#pragma unroll
      for (int rb = 0; rb < 8; ++rb) sdb += red[rb * NB + tid];
      slab[(size_t)a.N * a.K + n0 + tid] = sdb;
    }
  }
}

template <int NTW, int KTW, int NP = 3, bool TAIL = false, bool Y16 = false, bool X16 = false>
__global__ __launch_bounds__(256, 2) void wgrad_b3_kernel(WgradArgs a) {
  wgrad_b3_body<NTW, KTW, NP, TAIL, Y16, X16>(a, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x);
}
// Several SMALL weight-gradient products in one launch (rows <= 32 768: a launch of its own costs such a product more than its work; the
// published hyper-parameters' 32-wide tower layers issue five per layer on their critical chain).  Job table in the kernel arguments; a
// workgroup finds its job and its (slab, column block, k block) in it.  launch_wgrad records, wgrad_batch_flush launches.
#define WGRAD_BATCH_MAX 8
struct WgradBatch { int n; int blk0[WGRAD_BATCH_MAX + 1]; WgradArgs j[WGRAD_BATCH_MAX]; };
template <int NTW, int KTW, int NP, bool TAIL>
__global__ __launch_bounds__(256, 2) void wgrad_b3_batch_kernel(WgradBatch jb) {
  int ji = 0;
  while (ji + 1 < jb.n && (int)blockIdx.x >= jb.blk0[ji + 1]) ++ji;
  const WgradArgs a = jb.j[ji];
  const int ny = a.N / (32 * NTW);
  const int local = blockIdx.x - jb.blk0[ji];
  const int bx = local % a.S, rest = local / a.S;
  wgrad_b3_body<NTW, KTW, NP, TAIL, false, false>(a, bx, rest % ny, rest / ny, a.S);
}

// ------------------------------------------------------------------------------------------
// wgrad_tr_kernel: the bf16 mode's slab product for 128 x 128 output blocks.  With ONE bf16 plane the kernel above is bound by its
// transposed staging (4 x 4 register transposes, 8-byte LDS stores), not by HBM.  gfx950 can transpose on the way OUT of LDS instead:
// the 32-row tile is staged row-major as bf16 (one 16-byte store per 8 elements, chunks XOR-swizzled) and every lane gets its
// k-contiguous operand -- 8 consecutive rows of one column -- from two ds_read_b64_tr_b16 (per 16-lane group a 4-row x 16-column block,
// delivered column-major).  Two LDS buffers of 16 KB (one barrier per tile, three workgroups per CU by registers), the next tile
// prefetched into registers as 8 packed dwords per thread.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int wtr_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }      // byte offset of 16-byte chunk ch of row `row`
typedef short wtr_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 wtr_frag(const unsigned char* img, int off_lo, int off_hi) {
  const wtr_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wtr_s16x4*)(img + off_lo));
  const wtr_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wtr_s16x4*)(img + off_hi));
  typedef short s16x8_t __attribute__((ext_vector_type(8)));
  return __builtin_bit_cast(bf16x8, s16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}
template <bool TAIL, bool Y16, bool X16>
__global__ __launch_bounds__(256, 3) void wgrad_tr_kernel(WgradArgs a) {
  constexpr int NB = 128, KB = 128, IMG = 32 * 256;               // bytes of one operand's image
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* lds = reinterpret_cast<unsigned char*>(smem);   // [2][dY image | X image]
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  const int n0 = blockIdx.y * NB, k0 = blockIdx.z * KB;
  const int ntiles = (a.M + WG_RT - 1) / WG_RT, S = gridDim.x;
  // staging: thread = chunk (tid & 15) of rows (tid >> 4) and (tid >> 4) + 16, of both operands
  const int ch = tid & 15, rowa = tid >> 4;
  uint4 ry[2], rx[2];
  float dbacc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) dbacc[e] = 0.f;
  auto pack8 = [](const f32x4& lo, const f32x4& hi) -> uint4 {
    typedef __bf16 b8 __attribute__((ext_vector_type(8)));
    return __builtin_bit_cast(uint4, b8{(__bf16)lo[0], (__bf16)lo[1], (__bf16)lo[2], (__bf16)lo[3], (__bf16)hi[0], (__bf16)hi[1], (__bf16)hi[2], (__bf16)hi[3]});
  };
  auto load_tile = [&](int tt) {
    const size_t m0 = (size_t)tt * WG_RT;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const size_t row = m0 + rowa + 16 * u;
      const bool ok = !TAIL || row < (size_t)a.M;
      const size_t rr = ok ? row : 0;
      if (Y16) {
        uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(a.dY) + rr * a.lddy + n0 + 8 * ch);
        if (TAIL && !ok) v = uint4{0u, 0u, 0u, 0u};
        ry[u] = v;
        if (a.want_db) {
          typedef __bf16 b8 __attribute__((ext_vector_type(8)));
          const b8 h = __builtin_bit_cast(b8, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) dbacc[e] += (float)h[e];
        }
      } else {
        f32x4 lo = *reinterpret_cast<const f32x4*>(a.dY + rr * a.lddy + n0 + 8 * ch);
        f32x4 hi = *reinterpret_cast<const f32x4*>(a.dY + rr * a.lddy + n0 + 8 * ch + 4);
        if (TAIL && !ok) lo = hi = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dbacc[e] += lo[e];
          dbacc[4 + e] += hi[e];
        }
        ry[u] = pack8(lo, hi);
      }
      if (X16) {
        uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(a.X) + rr * a.ldx + k0 + 8 * ch);
        if (TAIL && !ok) v = uint4{0u, 0u, 0u, 0u};
        rx[u] = v;
      } else {
        f32x4 lo = *reinterpret_cast<const f32x4*>(a.X + rr * a.ldx + k0 + 8 * ch);
        f32x4 hi = *reinterpret_cast<const f32x4*>(a.X + rr * a.ldx + k0 + 8 * ch + 4);
        if (TAIL && !ok) lo = hi = f32x4{0.f, 0.f, 0.f, 0.f};
        rx[u] = pack8(lo, hi);
      }
    }
  };
  const int so0 = wtr_off(rowa, ch), so1 = wtr_off(rowa + 16, ch);
  auto store_tile = [&](int buf) {
    unsigned char* b = lds + buf * 2 * IMG;
    *reinterpret_cast<uint4*>(b + so0) = ry[0];
    *reinterpret_cast<uint4*>(b + so1) = ry[1];
    *reinterpret_cast<uint4*>(b + IMG + so0) = rx[0];
    *reinterpret_cast<uint4*>(b + IMG + so1) = rx[1];
  };
  // transposed-read addresses: lane 4q + pp of a 16-lane group names row q, columns 4pp .. 4pp+3 of the group's 4 x 16 block;
  // group g takes rows 8g .. 8g+3 (lo) and 8g+4 .. 8g+7 (hi) of the 16 columns of tile `tl` (chunks 2 tl, 2 tl + 1)
  const int q = p >> 2, pp = p & 3;
  auto fo = [&](int tl, int half) { return wtr_off(8 * g + 4 * half + q, 2 * tl + (pp >> 1)) + 8 * (pp & 1); };
  int ylo[4], yhi[4], xlo[4], xhi[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ylo[i] = fo(wn * 4 + i, 0); yhi[i] = fo(wn * 4 + i, 1);
    xlo[i] = IMG + fo(wk * 4 + i, 0); xhi[i] = IMG + fo(wk * 4 + i, 1);
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int t = blockIdx.x, buf = 0;
  if (t < ntiles) load_tile(t);
  for (; t < ntiles; t += S) {
    store_tile(buf);
    __syncthreads();
    if (t + S < ntiles) load_tile(t + S);
    const unsigned char* b = lds + buf * 2 * IMG;
    bf16x8 xf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xf[j] = wtr_frag(b, xlo[j], xhi[j]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16x8 yf = wtr_frag(b, ylo[i], yhi[i]);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf, xf[j], acc[i][j], 0, 0, 0);
    }
    buf ^= 1;
  }
  float* slab = a.slabs + (size_t)blockIdx.x * ((size_t)a.N * a.K + a.N);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + wk * 64 + j * 16 + p;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + i * 16 + 4 * g + r;
        slab[(size_t)n * a.K + k] = acc[i][j][r];
      }
    }
  if (a.want_db && blockIdx.z == 0) {
    __syncthreads();
    float* red = smem;                          // [16][NB]
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rowa * NB + 8 * ch + e] = dbacc[e];
    __syncthreads();
    if (tid < NB) {
      float sdb = 0.f;
#pragma unroll
      for (int rb = 0; rb < 16; ++rb) sdb += red[rb * NB + tid];
      slab[(size_t)a.N * a.K + n0 + tid] = sdb;
    }
  }
}

// out[i] (+)= sum_s slabs[s][i]; i < n.  16 slab lanes x 16 output groups per workgroup; each lane
// sums its strided slabs with 4 independent accumulators, then a fixed-order LDS tree: reproducible.
template <int VEC>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, size_t stride, int S, int n, int cols,
                                                          float* __restrict__ out, int ldo, int accumulate) {
  __shared__ float red[16][16 * VEC + 1];
  const int o = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int i0 = (blockIdx.x * 16 + o) * VEC;
  float s0[VEC], s1[VEC], s2[VEC], s3[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) s0[v] = s1[v] = s2[v] = s3[v] = 0.f;
  if (i0 < n) {
    int s = q;
    if (VEC == 4) {
      for (; s + 48 < S; s += 64) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(slabs + (size_t)s * stride + i0);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(slabs + (size_t)(s + 16) * stride + i0);
        const f32x4 a2 = *reinterpret_cast<const f32x4*>(slabs + (size_t)(s + 32) * stride + i0);
        const f32x4 a3 = *reinterpret_cast<const f32x4*>(slabs + (size_t)(s + 48) * stride + i0);
#pragma unroll
        for (int v = 0; v < VEC; ++v) { s0[v] += a0[v]; s1[v] += a1[v]; s2[v] += a2[v]; s3[v] += a3[v]; }
      }
      for (; s < S; s += 16) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(slabs + (size_t)s * stride + i0);
#pragma unroll
        for (int v = 0; v < VEC; ++v) s0[v] += a0[v];
      }
    } else {
      for (; s + 48 < S; s += 64) {
        s0[0] += slabs[(size_t)s * stride + i0];
        s1[0] += slabs[(size_t)(s + 16) * stride + i0];
        s2[0] += slabs[(size_t)(s + 32) * stride + i0];
        s3[0] += slabs[(size_t)(s + 48) * stride + i0];
      }
      for (; s < S; s += 16) s0[0] += slabs[(size_t)s * stride + i0];
    }
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) red[q][o * VEC + v] = (s0[v] + s1[v]) + (s2[v] + s3[v]);
  __syncthreads();
  if (q == 0 && i0 < n) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const int i = i0 + v;
      if (i >= n) break;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) acc += red[k][o * VEC + v];
      const int r = i / cols, c = i - r * cols;
      float* dst = out + (size_t)r * ldo + c;
      *dst = accumulate ? (*dst + acc) : acc;
    }
  }
}

// few outputs, many slabs (bias / LayerNorm parameter gradients): 4 outputs x 64 slab lanes per workgroup
__global__ __launch_bounds__(256) void slab_reduce_small_kernel(const float* __restrict__ slabs, size_t stride, int S, int n, int cols,
                                                                float* __restrict__ out, int ldo, int accumulate) {
  __shared__ float red[64][5];
  const int o = threadIdx.x & 3, q = threadIdx.x >> 2;
  const int i = blockIdx.x * 4 + o;
  float s0 = 0.f, s1 = 0.f;
  if (i < n) {
    int s = q;
    for (; s + 64 < S; s += 128) {
      s0 += slabs[(size_t)s * stride + i];
      s1 += slabs[(size_t)(s + 64) * stride + i];
    }
    for (; s < S; s += 64) s0 += slabs[(size_t)s * stride + i];
  }
  red[q][o] = s0 + s1;
  __syncthreads();
  if (q == 0 && i < n) {
    float acc = 0.f;
    for (int k = 0; k < 64; ++k) acc += red[k][o];
    const int r = i / cols, c = i - r * cols;
    float* dst = out + (size_t)r * ldo + c;
    *dst = accumulate ? (*dst + acc) : acc;
  }
}

int launch_slab_reduce(const float* slabs, size_t stride, int S, int rows, int cols, float* out, int ldo,
                       int accumulate, hipStream_t st) {
  int n = rows * cols;
  if (n <= 0) return 0;
  if (n < 1024 && S > 32) {
    LAUNCH_W(0.0, 4.0 * (double)S * n, slab_reduce_small_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, slabs, stride, S, n, cols, out, ldo, accumulate);
    INTEL_CHECK_LAUNCH();
    return 0;
  }
  const bool vec = ((stride & 3) == 0) && ((n & 3) == 0) && ((reinterpret_cast<uintptr_t>(slabs) & 15) == 0) && n >= 1024;
  if (vec)
    LAUNCH_W(0.0, 4.0 * (double)S * n, slab_reduce_kernel<4>, dim3(cdiv(n, 64)), dim3(256), 0, st, slabs, stride, S, n, cols, out, ldo, accumulate);
  else
    LAUNCH_W(0.0, 4.0 * (double)S * n, slab_reduce_kernel<1>, dim3(cdiv(n, 16)), dim3(256), 0, st, slabs, stride, S, n, cols, out, ldo, accumulate);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// dW and db of one wgrad in a single launch: outputs [0, N*K) -> dW (row stride lddw), [N*K, N*K+N) -> db
__global__ __launch_bounds__(256) void slab_reduce_wb_kernel(const float* __restrict__ slabs, size_t stride, int S, int N, int K,
                                                             float* __restrict__ dW, int lddw, float* __restrict__ db, int accumulate) {
  __shared__ float red[16][17];
  const int o = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int n = N * K + N;
  const int i = blockIdx.x * 16 + o;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n) {
    int s = q;
    for (; s + 48 < S; s += 64) {
      s0 += slabs[(size_t)s * stride + i];
      s1 += slabs[(size_t)(s + 16) * stride + i];
      s2 += slabs[(size_t)(s + 32) * stride + i];
      s3 += slabs[(size_t)(s + 48) * stride + i];
    }
    for (; s < S; s += 16) s0 += slabs[(size_t)s * stride + i];
  }
  red[q][o] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (q == 0 && i < n) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += red[k][o];
    float* dst;
    if (i < N * K) { const int r = i / K, c = i - r * K; dst = dW + (size_t)r * lddw + c; }
    else dst = db + (i - N * K);
    *dst = accumulate ? (*dst + v) : v;
  }
}

// ---- deferred reductions: job queue + one batched kernel per dependency round ----------------
#define RED_MAX_JOBS 56
struct RedJob {
  const float* slabs; float* out; unsigned long long stride;
  int S, n, cols, ldo;
  int accumulate, mode, blk0, round;
  int next, pad_;      // next: the job (index in this launch's table, -1 = none) that adds ITS slabs into the same destination right after this one
};
struct RedJobs { int n; int pad; RedJob j[RED_MAX_JOBS]; };      // 3.6 KB of kernel arguments

// mode 0: 16 output lanes x float4, 16 slab lanes; mode 1: the same, scalar; mode 2: 4 outputs x 64 slab lanes
__global__ __launch_bounds__(256) void slab_reduce_batch_kernel(RedJobs jobs) {
  __shared__ float red[16 * 65];
  int ji = 0;
  while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.j[ji + 1].blk0) ++ji;
  const int blk = blockIdx.x - jobs.j[ji].blk0;
  // a destination shared by several producers (tied layers, the shared intent embedding): their jobs have the same shape and are chained --
  // the same thread adds them one after the other, in push order, instead of one launch per producer
  for (int first = 1; ji >= 0; ji = jobs.j[ji].next, first = 0) {
  if (!first) __syncthreads();                    // `red` is reused
  const float* __restrict__ slabs = jobs.j[ji].slabs;
  float* __restrict__ out = jobs.j[ji].out;
  const size_t stride = jobs.j[ji].stride;
  const int S = jobs.j[ji].S, n = jobs.j[ji].n, cols = jobs.j[ji].cols, ldo = jobs.j[ji].ldo;
  const int accumulate = first ? jobs.j[ji].accumulate : 1, mode = jobs.j[ji].mode;
  if (mode == 0) {
    const int o = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int i0 = (blk * 16 + o) * 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (i0 < n) {
      int s = q;
      for (; s + 48 < S; s += 64) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(slabs + (size_t)s * stride + i0);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(slabs + (size_t)(s + 16) * stride + i0);
        const f32x4 a2 = *reinterpret_cast<const f32x4*>(slabs + (size_t)(s + 32) * stride + i0);
        const f32x4 a3 = *reinterpret_cast<const f32x4*>(slabs + (size_t)(s + 48) * stride + i0);
        s0 += a0; s1 += a1; s2 += a2; s3 += a3;
      }
      for (; s < S; s += 16) s0 += *reinterpret_cast<const f32x4*>(slabs + (size_t)s * stride + i0);
    }
    const f32x4 t = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int v = 0; v < 4; ++v) red[q * 65 + o * 4 + v] = t[v];
    __syncthreads();
    if (q < 4) {                       // 64 outputs of this workgroup, one per thread
      const int i = blk * 64 + o * 4 + q;
      if (i < n) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += red[k * 65 + o * 4 + q];
        const int r = i / cols, c = i - r * cols;
        float* dst = out + (size_t)r * ldo + c;
        *dst = accumulate ? (*dst + acc) : acc;
      }
    }
  } else if (mode == 1) {
    const int o = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int i = blk * 16 + o;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
      int s = q;
      for (; s + 48 < S; s += 64) {
        s0 += slabs[(size_t)s * stride + i];
        s1 += slabs[(size_t)(s + 16) * stride + i];
        s2 += slabs[(size_t)(s + 32) * stride + i];
        s3 += slabs[(size_t)(s + 48) * stride + i];
      }
      for (; s < S; s += 16) s0 += slabs[(size_t)s * stride + i];
    }
    red[q * 17 + o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0 && i < n) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) acc += red[k * 17 + o];
      const int r = i / cols, c = i - r * cols;
      float* dst = out + (size_t)r * ldo + c;
      *dst = accumulate ? (*dst + acc) : acc;
    }
  } else {
    const int o = threadIdx.x & 3, q = threadIdx.x >> 2;
    const int i = blk * 4 + o;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
      int s = q;
      for (; s + 192 < S; s += 256) {
        s0 += slabs[(size_t)s * stride + i];
        s1 += slabs[(size_t)(s + 64) * stride + i];
        s2 += slabs[(size_t)(s + 128) * stride + i];
        s3 += slabs[(size_t)(s + 192) * stride + i];
      }
      for (; s < S; s += 64) s0 += slabs[(size_t)s * stride + i];
    }
    red[q * 5 + o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0 && i < n) {
      float acc = 0.f;
      for (int k = 0; k < 64; ++k) acc += red[k * 5 + o];
      const int r = i / cols, c = i - r * cols;
      float* dst = out + (size_t)r * ldo + c;
      *dst = accumulate ? (*dst + acc) : acc;
    }
  }
  }
}

struct ReduceQueue {
  std::vector<RedJob> jobs;
  std::vector<int> tags;          // per job: the branch that pushed it (redq_set_tag); 0 = untagged
  int cur_tag = 0;
  float* arena = nullptr;
  size_t cap = 0, used = 0;
  int max_round = -1;
};
ReduceQueue* redq_create() { return new (std::nothrow) ReduceQueue(); }
void redq_destroy(ReduceQueue* q) { delete q; }
void redq_reset(ReduceQueue* q, float* arena, size_t arena_floats) {
  q->jobs.clear();
  q->tags.clear();
  q->cur_tag = 0;
  q->arena = arena; q->cap = arena_floats; q->used = 0; q->max_round = -1;
}
void redq_set_tag(ReduceQueue* q, int tag) { q->cur_tag = tag; }
float* redq_alloc(ReduceQueue* q, size_t floats) {
  const size_t need = (floats + 63) & ~(size_t)63;
  if (!q->arena || q->used + need > q->cap) return nullptr;
  float* p = q->arena + q->used;
  q->used += need;
  return p;
}
void redq_push(ReduceQueue* q, const float* slabs, size_t stride, int S, int rows, int cols, float* out, int ldo,
               int accumulate) {
  if (!out || rows <= 0 || cols <= 0) return;
  RedJob j;
  j.slabs = slabs; j.out = out; j.stride = stride; j.S = S; j.n = rows * cols; j.cols = cols; j.ldo = ldo;
  j.accumulate = accumulate; j.blk0 = 0;
  const bool vec = ((stride & 3) == 0) && ((j.n & 3) == 0) && ((reinterpret_cast<uintptr_t>(slabs) & 15) == 0) && j.n >= 1024;
  j.mode = vec ? 0 : ((j.n < 1024 && S > 32) ? 2 : 1);
  // destinations that overlap an earlier job must be reduced after it
  const float* lo = out;
  const float* hi = out + (size_t)(rows - 1) * ldo + cols;
  int round = 0;
  for (const RedJob& e : q->jobs) {
    const float* elo = e.out;
    const float* ehi = e.out + (size_t)(e.n / e.cols - 1) * e.ldo + e.cols;
    if (lo < ehi && elo < hi && e.round + 1 > round) round = e.round + 1;
  }
  j.round = round;
  if (round > q->max_round) q->max_round = round;
  q->jobs.push_back(j);
  q->tags.push_back(q->cur_tag);
}
// tag < 0: every job, then the arena is free again; tag >= 0: the jobs pushed under that tag only (a branch reduces its own weight
// gradients on its own stream as soon as it has produced them; destinations shared between branches must stay untagged)
static int redq_flush_impl(ReduceQueue* q, int tag, hipStream_t st) {
  // the jobs of this flush in push order; their rounds are recomputed among themselves (a destination that overlaps an earlier job
  // is reduced after it) -- except that jobs of IDENTICAL shape on the same destination are chained behind the first one: one
  // launch, the same thread adds them in push order
  std::vector<int> sel;
  for (size_t ji = 0; ji < q->jobs.size(); ++ji)
    if (tag < 0 || q->tags[ji] == tag) sel.push_back((int)ji);
  const int ns = (int)sel.size();
  std::vector<int> eff(ns, 0), head(ns, -1), nxt(ns, -1);
  int max_eff = -1;
  auto span_hi = [](const RedJob& e) { return e.out + (size_t)(e.n / e.cols - 1) * e.ldo + e.cols; };
  for (int a = 0; a < ns; ++a) {
    const RedJob& f = q->jobs[sel[a]];
    bool all_same = true;
    int first = -1, rmax = -1;
    for (int b = 0; b < a; ++b) {
      const RedJob& e = q->jobs[sel[b]];
      if (!(f.out < span_hi(e) && e.out < span_hi(f))) continue;
      if (!(e.out == f.out && e.n == f.n && e.cols == f.cols && e.ldo == f.ldo)) all_same = false;
      if (first < 0) first = b;
      if (eff[b] > rmax) rmax = eff[b];
    }
    if (first >= 0 && all_same && f.accumulate) {      // (a later job that OVERWRITES its destination is not an addend of the chain: it keeps a round of its own)
      const int h = head[first] >= 0 ? head[first] : first;
      head[a] = h;
      eff[a] = eff[h];
      int t = h;
      while (nxt[t] >= 0) t = nxt[t];
      nxt[t] = a;
    } else {
      eff[a] = rmax + 1;
    }
    if (eff[a] > max_eff) max_eff = eff[a];
  }
  for (int rnd = 0; rnd <= max_eff; ++rnd) {
    // chains are kept whole within one launch: heads first (they own the blocks), chained jobs behind them in the table
    std::vector<int> heads;
    for (int a = 0; a < ns; ++a)
      if (eff[a] == rnd && head[a] < 0) heads.push_back(a);
    size_t hi = 0;
    while (hi < heads.size()) {
      RedJobs jb;
      jb.n = 0; jb.pad = 0;
      int blocks = 0;
      double bytes = 0.0;
      std::vector<int> members;      // sel indices in table order
      size_t h2 = hi;
      int used = 0;
      for (; h2 < heads.size(); ++h2) {
        int len = 0;
        for (int t = heads[h2]; t >= 0; t = nxt[t]) ++len;
        if (used + len > RED_MAX_JOBS) break;
        used += len;
      }
      if (h2 == hi) { intel_set_error("slab reduction: a chain of more than %d jobs on one destination", RED_MAX_JOBS); return -1; }
      const int nh = (int)(h2 - hi);
      // table: [heads ... | chained jobs ...]
      std::vector<int> pos(ns, -1);
      int at = 0;
      for (size_t k = hi; k < h2; ++k) pos[heads[k]] = at++;
      for (size_t k = hi; k < h2; ++k)
        for (int t = nxt[heads[k]]; t >= 0; t = nxt[t]) pos[t] = at++;
      for (size_t k = hi; k < h2; ++k) {
        for (int t = heads[k]; t >= 0; t = nxt[t]) {
          RedJob j = q->jobs[sel[t]];
          // the members of a chain must map threads to outputs identically: where their strategies differ, all take the scalar one
          for (int u = heads[k]; u >= 0; u = nxt[u])
            if (q->jobs[sel[u]].mode != q->jobs[sel[heads[k]]].mode) j.mode = j.n < 1024 ? 2 : 1;      // (2: 64 slab lanes per output -- the members with hundreds of slabs set the pace)
          j.next = nxt[t] >= 0 ? pos[nxt[t]] : -1;
          j.pad_ = 0;
          if (t == heads[k]) {
            j.blk0 = blocks;
            blocks += cdiv(j.n, j.mode == 0 ? 64 : (j.mode == 1 ? 16 : 4));
          } else {
            j.blk0 = 0x7fffffff;
          }
          bytes += 4.0 * (double)j.S * j.n;
          jb.j[pos[t]] = j;
        }
      }
      jb.n = nh;      // the kernel searches the heads only; chained entries are reached through `next`
      (void)used;
      LAUNCH_W(0.0, bytes, slab_reduce_batch_kernel, dim3(blocks), dim3(256), 0, st, jb);
      INTEL_CHECK_LAUNCH();
      hi = h2;
    }
  }
  if (tag < 0) {
    q->jobs.clear();
    q->tags.clear();
    q->used = 0; q->max_round = -1;
    return 0;
  }
  size_t w = 0;
  for (size_t ji = 0; ji < q->jobs.size(); ++ji)
    if (q->tags[ji] != tag) { q->jobs[w] = q->jobs[ji]; q->tags[w] = q->tags[ji]; ++w; }
  q->jobs.resize(w);
  q->tags.resize(w);
  return 0;
}
int redq_flush(ReduceQueue* q, hipStream_t st) { return redq_flush_impl(q, -1, st); }
int redq_flush_tag(ReduceQueue* q, int tag, hipStream_t st) { return redq_flush_impl(q, tag, st); }

struct WgradPending { int planes, tail; WgradArgs a; };
static thread_local std::vector<WgradPending> g_wq;
static thread_local bool g_wq_on = false;
void wgrad_batch_reset() {      // an error exit between begin and flush must not leave the scope open for the next backward
  g_wq.clear();
  g_wq_on = false;
}
void wgrad_batch_begin() {
  g_wq.clear();        // (anything a failed earlier call left recorded is dropped)
  g_wq_on = true;
}
int wgrad_batch_flush(hipStream_t st) {
  g_wq_on = false;
  for (int planes = 1; planes <= 3; planes += 2)
    for (int tail = 0; tail < 2; ++tail) {
      WgradBatch jb;
      jb.n = 0;
      int blocks = 0;
      double flops = 0.0, bytes = 0.0;
      auto fire = [&]() -> int {
        if (jb.n == 0) return 0;
        jb.blk0[jb.n] = blocks;
        const size_t smem = (size_t)3 * 32 * 2 * WB_LDT * sizeof(__bf16);
#define WBB_LAUNCH(P_, T_)                                                                                        \
  do {                                                                                                            \
    allow_lds((wgrad_b3_batch_kernel<1, 1, P_, T_>), smem);                                                       \
    LAUNCH_W(flops, bytes, (wgrad_b3_batch_kernel<1, 1, P_, T_>), dim3(blocks), dim3(256), smem, st, jb);       \
  } while (0)
        if (planes == 3 && !tail) WBB_LAUNCH(3, false);
        else if (planes == 3) WBB_LAUNCH(3, true);
        else if (!tail) WBB_LAUNCH(1, false);
        else WBB_LAUNCH(1, true);
#undef WBB_LAUNCH
        INTEL_CHECK_LAUNCH();
        jb.n = 0; blocks = 0; flops = bytes = 0.0;
        return 0;
      };
      for (const WgradPending& w : g_wq) {
        if (w.planes != planes || w.tail != tail) continue;
        if (jb.n == WGRAD_BATCH_MAX) { int rc = fire(); if (rc) { g_wq.clear(); return rc; } }
        jb.blk0[jb.n] = blocks;
        jb.j[jb.n++] = w.a;
        blocks += w.a.S * (w.a.N / 32) * (w.a.K / 32);
        flops += 2.0 * w.a.M * w.a.N * w.a.K;
        bytes += 4.0 * ((double)w.a.M * w.a.K + (double)w.a.M * w.a.N + (double)w.a.K * w.a.N);
      }
      int rc = fire();
      if (rc) { g_wq.clear(); return rc; }
    }
  g_wq.clear();
  return 0;
}

int launch_wgrad(const float* dY, int lddy, const float* X, int ldx, int M, int N, int K, float* dW, int lddw,
                 float* db, int accumulate, float* slabs, hipStream_t st, ReduceQueue* q, const WgradSplit* split, int io16) {
  if (N <= 0 || K <= 0) return 0;
  if (split) {
    INTEL_CHECK_ARG(q && split->n >= 1 && split->n <= 4 && N % split->n == 0, "wgrad: a split product needs the reduce queue and N divisible by the number of parts");
    db = split->db[0];
  }
  if (q) {
    slabs = redq_alloc(q, wgrad_slab_floats(M, N, K));
    if (!slabs) {
      intel_set_error("wgrad: reduction arena exhausted");
      return -2;   // INTEL_E_WORKSPACE
    }
  }
  WgradArgs a;
  a.dY = dY; a.lddy = lddy; a.X = X; a.ldx = ldx; a.M = M; a.N = N; a.K = K; a.slabs = slabs;
  a.S = wgrad_num_slabs(M, N, K); a.want_db = db != nullptr;
  a.dy_bf16 = split ? split->dy_bf16 : (io16 & 1);
  a.x_bf16 = (io16 >> 1) & 1;
  {
    // fast path: exact 32-row tiles, widths in 32-float steps, 16-byte aligned operands
    const bool aligned = ((lddy & 3) == 0) && ((ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(dY) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(X) & 15) == 0) && ((((size_t)N * K + N) & 3) == 0) &&
                         ((reinterpret_cast<uintptr_t>(slabs) & 15) == 0);
    static const int use_wb3 = [] { const char* e = getenv("INTEL_WGRAD_B3"); return (e && e[0] == '0') ? 0 : 1; }();
    INTEL_CHECK_ARG((!a.dy_bf16 && !a.x_bf16) || g_planes == 1, "wgrad: bf16-stored operands need the bf16 mode");
    if (use_wb3 && aligned && M >= 1 && N % 32 == 0 && K % 32 == 0) {      // any row count: the last tile is zero-padded in the kernel
      const int ntw = N % 128 == 0 ? 4 : (N % 64 == 0 ? 2 : 1), ktw = K % 128 == 0 ? 4 : (K % 64 == 0 ? 2 : 1);
      const dim3 grid(a.S, N / (32 * ntw), K / (32 * ktw));
      size_t smem = (size_t)3 * 32 * (ntw + ktw) * WB_LDT * sizeof(__bf16);
      if (smem < (size_t)8 * 32 * ntw * sizeof(float)) smem = (size_t)8 * 32 * ntw * sizeof(float);
#define WB_LAUNCH(A_, B_, P_, T_)                                                                                   \
  do {                                                                                                              \
    allow_lds((wgrad_b3_kernel<A_, B_, P_, T_>), smem);                                                             \
    LAUNCH_S(M, N, K, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)M * N + (double)K * N), (wgrad_b3_kernel<A_, B_, P_, T_>), grid, \
             dim3(256), smem, st, a);                                                                               \
  } while (0)
#define WB_LAUNCH16(A_, B_, T_, Y_, X_)                                                                             \
  do {                                                                                                              \
    allow_lds((wgrad_b3_kernel<A_, B_, 1, T_, Y_, X_>), smem);                                                      \
    LAUNCH_S(M, N, K, 2.0 * M * N * K, (X_ ? 2.0 : 4.0) * (double)M * K + (Y_ ? 2.0 : 4.0) * (double)M * N + 4.0 * (double)K * N, \
             (wgrad_b3_kernel<A_, B_, 1, T_, Y_, X_>), grid, dim3(256), smem, st, a);                               \
  } while (0)
#define WB_CASE16(A_, B_)                                                                                           \
  if (ntw == A_ && ktw == B_) {                                                                                     \
    const bool tail = M % WG_RT != 0;                                                                               \
    done16 = true;                                                                                                  \
    if (a.dy_bf16 && a.x_bf16) { if (tail) WB_LAUNCH16(A_, B_, true, true, true); else WB_LAUNCH16(A_, B_, false, true, true); }        \
    else if (a.dy_bf16) { if (tail) WB_LAUNCH16(A_, B_, true, true, false); else WB_LAUNCH16(A_, B_, false, true, false); }             \
    else { if (tail) WB_LAUNCH16(A_, B_, true, false, true); else WB_LAUNCH16(A_, B_, false, false, true); }                            \
  }
      static const int use_tr = [] { const char* e = getenv("INTEL_WGRAD_TR"); return (e && e[0] == '0') ? 0 : 1; }();
      if (use_tr && g_planes == 1 && ntw == 4 && ktw == 4) {      // bf16 mode, 128 x 128 blocks: row-major staging + transposing LDS reads
        const size_t smem_tr = (size_t)2 * 2 * 32 * 256;
#define WTR_LAUNCH(T_, Y_, X_)                                                                                        \
  do {                                                                                                                \
    allow_lds((wgrad_tr_kernel<T_, Y_, X_>), smem_tr);                                                                \
    LAUNCH_S(M, N, K, 2.0 * M * N * K, (X_ ? 2.0 : 4.0) * (double)M * K + (Y_ ? 2.0 : 4.0) * (double)M * N + 4.0 * (double)K * N, \
             (wgrad_tr_kernel<T_, Y_, X_>), grid, dim3(256), smem_tr, st, a);                                         \
  } while (0)
        const bool tail = M % WG_RT != 0;
        const int sel = (tail ? 4 : 0) | (a.dy_bf16 ? 2 : 0) | (a.x_bf16 ? 1 : 0);
        switch (sel) {
          case 0: WTR_LAUNCH(false, false, false); break;
          case 1: WTR_LAUNCH(false, false, true); break;
          case 2: WTR_LAUNCH(false, true, false); break;
          case 3: WTR_LAUNCH(false, true, true); break;
          case 4: WTR_LAUNCH(true, false, false); break;
          case 5: WTR_LAUNCH(true, false, true); break;
          case 6: WTR_LAUNCH(true, true, false); break;
          default: WTR_LAUNCH(true, true, true); break;
        }
#undef WTR_LAUNCH
        INTEL_CHECK_LAUNCH();
        goto reduce;
      }
      if (a.dy_bf16 || a.x_bf16) {          // bf16-stored operands: the square tower shapes only (d = 128, d = 64; N = d or 3d)
        bool done16 = false;
        WB_CASE16(4, 4) WB_CASE16(2, 2)
        INTEL_CHECK_ARG(done16, "wgrad: bf16-stored operands are supported for 64- and 128-wide products only");
        INTEL_CHECK_LAUNCH();
        goto reduce;
      }
      if (g_wq_on && q && ntw == 1 && ktw == 1 && M <= 32768) {          // small product inside a batch scope: recorded, launched by wgrad_batch_flush
        WgradPending w;
        w.planes = g_planes; w.tail = (M % WG_RT != 0) ? 1 : 0; w.a = a;
        g_wq.push_back(w);
        goto reduce;
      }
#define WB_CASE(A_, B_)                                                                                             \
  if (ntw == A_ && ktw == B_) {                                                                                     \
    const bool tail = M % WG_RT != 0;                                                                               \
    if (g_planes == 3 && !tail) WB_LAUNCH(A_, B_, 3, false);                                                        \
    else if (g_planes == 3) WB_LAUNCH(A_, B_, 3, true);                                                             \
    else if (!tail) WB_LAUNCH(A_, B_, 1, false);                                                                    \
    else WB_LAUNCH(A_, B_, 1, true);                                                                                \
  }
      WB_CASE(4, 4) WB_CASE(4, 2) WB_CASE(4, 1) WB_CASE(2, 4) WB_CASE(2, 2) WB_CASE(2, 1) WB_CASE(1, 4) WB_CASE(1, 2) WB_CASE(1, 1)
#undef WB_CASE
#undef WB_CASE16
#undef WB_LAUNCH
#undef WB_LAUNCH16
      INTEL_CHECK_LAUNCH();
      goto reduce;
    }
    INTEL_CHECK_ARG(!a.dy_bf16 && !a.x_bf16, "wgrad: bf16-stored operands need the bf16-pipe kernel (aligned operands, N and K multiples of 32)");
  }
  {
  const int nbm = min(128, N), kbm = min(128, K);
  const int ldy = (rup(rup(nbm, 16), 32)) + 16, ldxs = (rup(rup(kbm, 16), 32)) + 16;
  size_t smem = (size_t)WG_RT * (ldy + ldxs) * sizeof(float);
  {
    allow_lds(wgrad_pipe_kernel, 2 * smem);
    LAUNCH_S(M, N, K, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)M * N + (double)K * N), wgrad_pipe_kernel, dim3(a.S, cdiv(N, 128), cdiv(K, 128)), dim3(256), 2 * smem, st, a);
  }
  INTEL_CHECK_LAUNCH();
  }
reduce:
  size_t stride = (size_t)N * K + N;
  if (q && split) {       // one product over the stacked columns, one reduction job per weight
    const int Ns = N / split->n;
    for (int p = 0; p < split->n; ++p) {
      redq_push(q, slabs + (size_t)p * Ns * K, stride, a.S, Ns, K, split->dW[p], K, split->acc[p]);
      if (split->db[p]) redq_push(q, slabs + (size_t)N * K + (size_t)p * Ns, stride, a.S, 1, Ns, split->db[p], Ns, split->acc[p]);
    }
    return 0;
  }
  if (q) {
    redq_push(q, slabs, stride, a.S, N, K, dW, lddw, accumulate);
    if (db) redq_push(q, slabs + (size_t)N * K, stride, a.S, 1, N, db, N, accumulate);
    return 0;
  }
  if (db && (size_t)N * K < 4096) {     // small weight: one launch reduces dW and db together
    LAUNCH_W(0.0, 4.0 * (double)a.S * (N * K + N), slab_reduce_wb_kernel, dim3(cdiv(N * K + N, 16)), dim3(256), 0, st, slabs, stride, a.S, N, K, dW, lddw, db, accumulate);
    INTEL_CHECK_LAUNCH();
    return 0;
  }
  int rc = launch_slab_reduce(slabs, stride, a.S, N, K, dW, lddw, accumulate, st);
  if (rc) return rc;
  if (db) rc = launch_slab_reduce(slabs + (size_t)N * K, stride, a.S, 1, N, db, N, accumulate, st);
  return rc;
}
