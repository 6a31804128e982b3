// B-row chains: the session head of IntEL (models/IntEL/IntEL.py:147-153 intent prediction, :201-215 intent-conditioned pooling
// queries, fusion weights and score aggregation) and its backward are chains of SMALL dependent products over one row per session --
// pred_layer, softmax, intent_embeddings, the cross-attention query / key / value projections, weight_embeddings.  As one launch per
// link (13 forward, ~20 backward) their cost is the launch latency of the chain, which at the reference's batch of 512 is most of the
// step.  Here a chain is ONE launch: a workgroup owns 16 sessions, every intermediate is a [16, width] tile in LDS, and the links are
// executed level by level (workgroup barrier between dependency levels) from a small op table in the kernel arguments:
//
//   LOAD        global rows (optionally gathered through an id, optionally relu'd) -> tile
//   LIN         tile x packed weight (launch_pack_b fragment order: one coalesced 16-byte load per lane and 16 x 16 block) on exact
//               fp32 MFMAs, 16 sessions = the 16 rows of the MFMA tile; bias / relu / accumulate epilogue; tile (+ global copy).
//               CH_BF16 (bf16 mode, links of 64 / 128 input features): both operands rounded to bf16 first, as on the bf16 pipe
//   MASKCOPY    tile * (other tile > 0)  (relu backward)
//   SOFTMAX(+BWD), ENS_FWD / ENS_BWD (IntEL.py:214-215: per-session weight vectors broadcast over the list, weighted score sum)
//   WGRAD       the workgroup's share of a weight gradient dW = dY^T X (+ bias gradient) from two tiles: the 16 sessions are the MFMA's
//               k index; one partial per workgroup, summed by the batched slab reduction (kernels.h: ReduceQueue)
//
// The column tiles of a level's products are dealt to the eight waves.  Results that later launches read (the training stash, the
// operands of the deferred weight-gradient products) are written to the same global buffers the kernel-per-op path uses, so either
// path can run the other's backward.  model.cpp builds the four chains (forward a / b, backward a / b); the single-query pooling
// between a and b stays its own kernel (session.hip).
#include <stdio.h>
#include <stdlib.h>

#include "kernels.h"
#include "chain.h"

namespace {

// (all the global loads of a lane are issued before the first one is consumed: a chain kernel is a handful of workgroups whose time
// is the sum of its exposed memory latencies)
__device__ __forceinline__ void op_load(const ChainOp& op, float* lds, int B, int b0, int t, int lane) {
  const int n = 4 * op.NP;                       // elements of this unit: rows 4 t .. 4 t + 3
  for (int e0 = lane; e0 < n; e0 += 64 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 64 * u;
      const int rr = e / op.NP, c = e - rr * op.NP, b = b0 + 4 * t + rr;
      v[u] = 0.f;
      if (e < n && b < B && c < op.N) {
        const size_t srow = (size_t)((op.flags & CH_GATHER) ? op.idx[b] : b);
        v[u] = op.P[srow * op.src_ld + op.src_col + c];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 64 * u;
      if (e >= n) break;
      const int rr = e / op.NP, c = e - rr * op.NP, row = 4 * t + rr, b = b0 + row;
      float x = v[u];
      if (op.flags & CH_RELU) x = fmaxf(x, 0.f);
      if (op.gout && b < B && c < op.N) op.gout[(size_t)b * op.gld + op.gcol + c] = x;
      lds[op.out_off + row * op.out_ld + c] = x;
    }
  }
}

// round-to-nearest-even to bf16 and back: a product of two such values is exact in fp32, so the exact-fp32 MFMA of rounded operands IS the
// bf16 pipe's product with fp32 accumulation (bf16 mode, CH_BF16 links)
__device__ __forceinline__ float bfr(float x) { return (float)(__bf16)x; }

__device__ __forceinline__ void op_lin(const ChainOp& op, float* lds, int B, int b0, int mt, int lane) {
  const int i = lane & 15, j = lane >> 4;
  const bool bf = (op.flags & CH_BF16) != 0;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const f32x4* P = reinterpret_cast<const f32x4*>(op.P) + (size_t)mt * op.KG * 64 + lane;
  const float* xin = lds + op.in_off + i * op.in_ld + 4 * j;
  int g = 0;
  for (; g + 8 <= op.KG; g += 8) {               // eight weight fragments in flight
    f32x4 w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = P[(size_t)(g + u) * 64];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(xin + 16 * (g + u));
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = bf ? mfma16(bfr(w[u][s]), bfr(x[s]), acc) : mfma16(w[u][s], x[s], acc);
    }
  }
  {
    f32x4 w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = g + u < op.KG ? P[(size_t)(g + u) * 64] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (g + u >= op.KG) break;
      const f32x4 x = *reinterpret_cast<const f32x4*>(xin + 16 * (g + u));
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = bf ? mfma16(bfr(w[u][s]), bfr(x[s]), acc) : mfma16(w[u][s], x[s], acc);
    }
  }
  const int col = 16 * mt + 4 * j;
  float* orow = lds + op.out_off + i * op.out_ld + col;
  const int b = b0 + i;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = col + r;
    float v = acc[r];
    if (op.bias && c < op.N) v += op.bias[c];
    if (op.flags & CH_RELU) v = fmaxf(v, 0.f);
    if (op.flags & CH_ACCUM) v += orow[r];
    acc[r] = v;
    if (op.gout && b < B && c < op.N) op.gout[(size_t)b * op.gld + op.gcol + c] = v;
  }
  *reinterpret_cast<f32x4*>(orow) = acc;
}

__device__ __forceinline__ void op_maskcopy(const ChainOp& op, float* lds, int B, int b0, int t, int lane) {
  for (int rr = 0; rr < 4; ++rr) {
    const int row = 4 * t + rr, b = b0 + row;
    for (int c = lane; c < op.NP; c += 64) {
      float v = 0.f;
      if (c < op.N) {
        v = lds[op.in_off + row * op.in_ld + c];
        if (!(lds[op.aux_off + row * op.aux_ld + c] > 0.f)) v = 0.f;
        if (op.gout && b < B) op.gout[(size_t)b * op.gld + op.gcol + c] = v;
      }
      lds[op.out_off + row * op.out_ld + c] = v;
    }
  }
}

// rows = sessions; the four lanes (i, 0..3) of a row stride over its columns
__device__ __forceinline__ float row4_sum(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ float row4_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}

__device__ __forceinline__ void op_softmax(const ChainOp& op, float* lds, int B, int b0, int lane) {
  const int i = lane & 15, j = lane >> 4, b = b0 + i;
  const float* xr = lds + op.in_off + i * op.in_ld;
  float mx = -INFINITY;
  for (int c = j; c < op.N; c += 4) mx = fmaxf(mx, xr[c]);
  mx = row4_max(mx);
  float s = 0.f;
  for (int c = j; c < op.N; c += 4) s += expf(xr[c] - mx);
  s = row4_sum(s);
  const float inv = 1.f / s;
  for (int c = j; c < op.NP; c += 4) {
    const float v = c < op.N ? expf(xr[c] - mx) * inv : 0.f;
    lds[op.out_off + i * op.out_ld + c] = v;
    if (c < op.N && b < B) {
      if (op.gout) op.gout[(size_t)b * op.gld + op.gcol + c] = v;
      if (op.gout2) op.gout2[(size_t)b * op.N + c] = v;
    }
  }
}

// dx = y * (dy - sum(dy * y)),  dy = in (+ in2) (+ in3) (+ gadd[b])
__device__ __forceinline__ void op_softmax_bwd(const ChainOp& op, float* lds, int B, int b0, int lane) {
  const int i = lane & 15, j = lane >> 4, b = b0 + i;
  const float* y = lds + op.aux_off + i * op.aux_ld;
  float s = 0.f;
  for (int c = j; c < op.N; c += 4) {
    float d = lds[op.in_off + i * op.in_ld + c];
    if (op.in2_off >= 0) d += lds[op.in2_off + i * op.in_ld + c];
    if (op.in3_off >= 0) d += lds[op.in3_off + i * op.in_ld + c];
    if (op.gadd && b < B) d += op.gadd[(size_t)b * op.N + c];
    lds[op.in_off + i * op.in_ld + c] = d;          // (this lane's own columns)
    s += d * y[c];
  }
  s = row4_sum(s);
  for (int c = j; c < op.NP; c += 4) {
    const float v = c < op.N ? y[c] * (lds[op.in_off + i * op.in_ld + c] - s) : 0.f;
    lds[op.out_off + i * op.out_ld + c] = v;
    if (c < op.N && b < B && op.gout) op.gout[(size_t)b * op.gld + op.gcol + c] = v;
  }
}

// IntEL.py:214-215 with session-level weights: weights[b, l, :] = valid row ? wv[b] : wpad[b]; ens = sum_k weights * scores
__device__ __forceinline__ void op_ens_fwd(const ChainOp& op, const ChainEns& e, const float* lds, int B, int b0, int t, int lane) {
  for (int rr = 0; rr < 4; ++rr) {
    const int row = 4 * t + rr, b = b0 + row;
    if (b >= B) break;
    const int len = e.slen[b];
    const float* wv = lds + op.in_off + row * op.in_ld;
    const float* wp = lds + op.aux_off + row * op.aux_ld;
    for (int l = lane; l < e.L; l += 64) {
      const float* src = l < len ? wv : wp;
      const size_t m = (size_t)b * e.L + l;
      float acc = 0.f;
      for (int k = 0; k < e.K; ++k) {
        const float w = src[k];
        e.weights[m * e.K + k] = w;
        acc += w * e.scores[m * e.K + k];
      }
      e.ens[m] = acc;
    }
  }
}

// dwv[b, k] = sum over valid rows of d_weights + d_ens * scores; dwpad likewise over the padded rows
__device__ __forceinline__ void op_ens_bwd(const ChainOp& op, const ChainEns& e, float* lds, int B, int b0, int t, int lane) {
  for (int rr = 0; rr < 4; ++rr) {
    const int row = 4 * t + rr, b = b0 + row;
    float* dv = lds + op.out_off + row * op.out_ld;
    float* dp = lds + op.aux_off + row * op.aux_ld;
    if (b >= B) {
      if (lane < 16) { dv[lane] = 0.f; dp[lane] = 0.f; }
      continue;
    }
    const int len = e.slen[b];
    for (int k = 0; k < 16; ++k) {
      float sv = 0.f, sp = 0.f;
      if (k < e.K) {
        for (int l = lane; l < e.L; l += 64) {
          const size_t m = (size_t)b * e.L + l;
          float v = 0.f;
          if (e.d_weights) v += e.d_weights[m * e.K + k];
          if (e.d_ens) v += e.d_ens[m] * e.scores[m * e.K + k];
          if (l < len) sv += v; else sp += v;
        }
        sv = wave_sum(sv);
        sp = wave_sum(sp);
      }
      if (lane == 0) {
        dv[k] = sv;
        dp[k] = sp;
        if (k < e.K) {
          e.dwv[(size_t)b * e.K + k] = sv;
          e.dwpad[(size_t)b * e.K + k] = sp;
        }
      }
    }
  }
}

// the workgroup's 16 sessions are the k index of the product: both operands are read column-wise from their row-major tiles
__device__ __forceinline__ void op_wgrad(const ChainOp& op, const float* lds, int t, int lane) {
  const int i = lane & 15, j = lane >> 4;
  const int mt = t / op.KG, nt = t - mt * op.KG;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float cs = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float ya = lds[op.in_off + (4 * j + s) * op.in_ld + 16 * mt + i];
    const float xb = lds[op.aux_off + (4 * j + s) * op.aux_ld + 16 * nt + i];
    acc = (op.flags & CH_BF16) ? mfma16(bfr(ya), bfr(xb), acc) : mfma16(ya, xb, acc);
    cs += ya;      // (the bias gradient sums the unrounded rows in either mode)
  }
  float* slab = op.gout + (size_t)blockIdx.x * op.gstride;
  const int k = 16 * nt + i;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = 16 * mt + 4 * j + r;
    if (n < op.N && k < op.K) {
      float* p = slab + (size_t)n * op.gld + op.gcol + k;
      *p = (op.flags & CH_ACCUM) ? *p + acc[r] : acc[r];
    }
  }
  if (op.gout2 && nt == 0) {
    cs = row4_sum(cs);
    const int n = 16 * mt + i;
    if (j == 0 && n < op.N) {
      float* p = op.gout2 + (size_t)blockIdx.x * op.gstride + n;
      *p = (op.flags & CH_ACCUM) ? *p + cs : cs;
    }
  }
}

// The op table travels in the kernel arguments, and the kernel-argument segment lives in host-visible memory: every scalar load from it
// is a trip over the host link (~2 us).  An interpreter that reads its table field by field would spend its time there, so the whole
// table is copied into LDS with ONE round of vector loads first and decoded from there.
__global__ __launch_bounds__(512) void chain_kernel(ChainArgs ka) {
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  __shared__ __attribute__((aligned(16))) int tab[(sizeof(ChainArgs) + 3) / 4];
  const int tid = threadIdx.x, lane = tid & 63;
  {
    const int* src = reinterpret_cast<const int*>(&ka);
    constexpr int NW = (int)((sizeof(ChainArgs) + 3) / 4);
#pragma unroll
    for (int k = 0; k < (NW + 511) / 512; ++k)
      if (tid + 512 * k < NW) tab[tid + 512 * k] = src[tid + 512 * k];
  }
  __syncthreads();
  const ChainArgs& a = *reinterpret_cast<const ChainArgs*>(tab);
  float* lds = lds_all;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b0 = blockIdx.x * 16;
  int o0 = 0;
  for (int lev = 0; lev < a.nlevels; ++lev) {
    int u = 0;
    int o = o0;
    for (; o < a.nops && a.ops[o].level == lev; ++o) {
      const ChainOp& op = a.ops[o];
      const int nu = op.kind == CH_LIN ? op.NT : (op.kind == CH_WGRAD ? op.NT * op.KG : ((op.kind == CH_SOFTMAX || op.kind == CH_SOFTMAX_BWD) ? 1 : 4));
      for (int t = 0; t < nu; ++t, ++u) {
        if ((u & 7) != wave) continue;
        switch (op.kind) {
          case CH_LOAD: op_load(op, lds, a.B, b0, t, lane); break;
          case CH_LIN: op_lin(op, lds, a.B, b0, t, lane); break;
          case CH_MASKCOPY: op_maskcopy(op, lds, a.B, b0, t, lane); break;
          case CH_SOFTMAX: op_softmax(op, lds, a.B, b0, lane); break;
          case CH_SOFTMAX_BWD: op_softmax_bwd(op, lds, a.B, b0, lane); break;
          case CH_ENS_FWD: op_ens_fwd(op, a.ens, lds, a.B, b0, t, lane); break;
          case CH_ENS_BWD: op_ens_bwd(op, a.ens, lds, a.B, b0, t, lane); break;
          case CH_WGRAD: op_wgrad(op, lds, t, lane); break;
          default: break;
        }
      }
    }
    o0 = o;
    __syncthreads();
  }
}

}  // namespace

int launch_chain(const ChainArgs& a, size_t lds_floats, hipStream_t st) {
  if (a.B <= 0 || a.nops <= 0) return 0;
  const size_t smem = lds_floats * sizeof(float);
  INTEL_CHECK_ARG(smem <= 160 * 1024 - 256, "chain: %zu bytes of LDS tiles", smem);
  allow_lds(chain_kernel, smem);
  LAUNCH_S(a.B, a.nops, a.nlevels, 0.0, 0.0, chain_kernel, dim3(cdiv(a.B, 16)), dim3(512), smem, st, a);
  INTEL_CHECK_LAUNCH();
  return 0;
}
