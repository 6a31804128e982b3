// Backward of the fused BERT4Rec encoder kernels (enc.hip): the DATA-gradient chain of a block as one kernel, the matrix-shaped
// leftovers (weight gradients = reductions over all rows, the two K > 128 data-gradient products) stay on the GEMM kernels.
//
//   enc_last_bwd_kernel    the pruned last block, 16 sessions per workgroup: LayerNorm2 backward -> d(relu) through W2 -> through W1
//                          (+ residual) -> LayerNorm1 backward -> the one-row attention's backward (dq, and dK' / dV' for all rows of
//                          the session) -> d(x_last) = dq Wq + residual.  Leaves the B-row gradients the weight-gradient products
//                          read (dz2, df1, dq), d(x_last) and the [rows, 2D] key / value gradient.
//   enc_block_bwd_kernel   one full block for a tile of whole packed sessions (the tiling of enc_block_fwd_kernel):
//                          dE (+ d(x_last) at the sessions' last rows) -> LayerNorm2 backward -> dF1 = (dZ2 W2) * [F1 > 0] ->
//                          dC = dF1 W1 + dZ2 -> LayerNorm1 backward -> attention backward per session and head in exact fp32 MFMA
//                          (softmax recomputed from the stashed q / k / v; the transposed score tiles are recomputed rather than
//                          transposed: 7 tile products).  Leaves dZ2, dF1, dZ1 and dQKV = [dQ | dK | dV] for the weight-gradient
//                          products and the final dX = dQKV Wqkv + dZ1.
// LayerNorm parameter gradients: per-workgroup partial sums -> slabs -> the batched slab reduction (ReduceQueue).
// Template parameter NP = 3: fp32 accuracy (six plane products, exact fp32 attention); NP = 1: the bf16 mode (hi planes, single bf16 MFMAs in the
// attention backward, rounding points as the kernel-per-op path of that mode).
#include <stdio.h>
#include <stdlib.h>

#include "kernels.h"
#include "enc.h"
#include "planes.h"

namespace {

using namespace planes;

// self-contained plane product (as enc.hip: gemm_planes)
// `tail_loads` runs right behind the product's last fragment loads (vmcnt retires in order: global loads for a LATER phase issued
// there delay nothing of this product)
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// NP = 3: six plane products (fp32 accuracy); NP = 1 (bf16 mode): the hi planes alone -- the images keep their three-plane layout
template <int D, int CT, int RT, int ROWS, int NP, typename Hook = NoHook>
__device__ __forceinline__ void gemm_planes(const __bf16* frag, const uint4* img, int ct0, f32x4 (&acc)[CT][RT], Hook tail_loads = Hook()) {
  constexpr int KB = D / 32, KBT = 4, LDP = D + 8, PLANE = ROWS * LDP;
  uint4 bw[2][CT][NP];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) bw[0][c][pl] = img[((size_t)((ct0 + c) * KBT) * 3 + pl) * 64];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    if (kb + 1 < KB) {
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) bw[(kb + 1) & 1][c][pl] = img[((size_t)((ct0 + c) * KBT + kb + 1) * 3 + pl) * 64];
    }
    if (kb == (KB >= 2 ? KB - 2 : 0)) tail_loads();
    __builtin_amdgcn_sched_barrier(0);      // (pins the requests here: the scheduler otherwise sinks them behind this block's MFMAs -- no lookahead at all)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const __bf16* fp = frag + rt * 16 * LDP + kb * 32;
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(fp);
      bf16x8 am = ah, al = ah;
      if constexpr (NP == 3) {
        am = *reinterpret_cast<const bf16x8*>(fp + PLANE);
        al = *reinterpret_cast<const bf16x8*>(fp + 2 * PLANE);
      }
#pragma unroll
      for (int c = 0; c < CT; ++c)
        acc[c][rt] = mma<NP>(__builtin_bit_cast(bf16x8, bw[kb & 1][c][0]), __builtin_bit_cast(bf16x8, bw[kb & 1][c][NP == 3 ? 1 : 0]),
                             __builtin_bit_cast(bf16x8, bw[kb & 1][c][NP == 3 ? 2 : 0]), ah, am, al, acc[c][rt]);
    }
  }
}
// the attention backward's tile products: exact fp32 MFMAs, or (bf16 mode) one bf16 MFMA on the rounded operands
template <bool BF>
__device__ __forceinline__ f32x4 amma(const f32x4& x, const f32x4& y, f32x4 c) {
  if constexpr (BF) return mma4_bf16(to_bf16x4(x), to_bf16x4(y), c);
  else return mma4(x, y, c);
}

// ======================================================================================================================
// block backward
// ======================================================================================================================
struct EncBlockBwdArgs {
  const float* dE;           // [rows, D] gradient w.r.t. the block output
  const float* dxl;          // [B, D] extra gradient of row len-1 of every session (the pruned last block's d(x_last)), or NULL
  int rows, B, T, ntiles;
  const int* off; const int* tile_s;
  const uint4* W2T; const uint4* W1T;      // images of the transposed weights: dX = dY W
  const float* g1; const float* g2;
  const float* XH2; const float* RSTD2; const float* F1; const float* XH1; const float* RSTD1; const float* QKV;      // forward stash
  float* DZ2; float* DF1; float* DZ1;      // [rows, D]
  float* DQKV;                             // [rows, 3D]
  float* slab;                             // [gridDim.x][4][D]: partial d(gamma2), d(beta2), d(gamma1), d(beta1)
  unsigned long long* dbg;                 // INTEL_ENC_DBG=1: per-phase shader-clock totals of workgroup 0's thread 0
};

template <int D>
struct EncBwdCfg {
  static constexpr int NW = D / 16, NT = NW * 64, LDP = D + 8, PLANE = 64 * LDP, LQ = D + 4;
  static constexpr size_t ES_BYTES = (size_t)64 * LQ * 4;                 // dE -> dZ2 -> dC -> dZ1 rows (fp32)
  static constexpr size_t P_BYTES = (size_t)3 * PLANE * 2;                // one set of planes
  static constexpr size_t QKV_BYTES = (size_t)3 * 64 * LQ * 4;
  static constexpr size_t U_BYTES = 2 * P_BYTES > QKV_BYTES ? 2 * P_BYTES : QKV_BYTES;     // dZ2 planes + dF1 planes, later the q / k / v rows
  static constexpr size_t SMEM = ES_BYTES + U_BYTES;
};

// Attention backward of one (session, head) as TWO independent work items (different waves take them):
//   pass Q   transposed score tiles (keys on the accumulator rows, the query on the lane): softmax statistics by lane-group
//            shuffles, dS^T straight from the accumulators into dQ = dS K;
//   pass KV  straight tiles (operands swapped: queries on the accumulator rows, the key on the lane): the same statistics by 16-lane
//            DPP reductions, P and dS from the accumulators into dV = P^T dO and dK = dS^T Q.
// Recomputing the score tiles in both layouts (7 tile products instead of 5) needs no transposition through LDS and no hand-over
// between the two halves.  KT = tiles of 16 rows the session spans.  The tile loops are not unrolled for KT = 2 (registers).
template <int CTRL>
__device__ __forceinline__ float dpp_max_step(float v) {
  return fmaxf(v, dpp_mov<CTRL>(v));
}
__device__ __forceinline__ float row16_max(float v) {
  v = dpp_max_step<0xB1>(v);
  v = dpp_max_step<0x4E>(v);
  v = dpp_max_step<0x141>(v);
  v = dpp_max_step<0x140>(v);
  return v;
}

template <int D, int DK, int KT, bool BF>
__device__ __forceinline__ void attn_bwd_q(const EncBlockBwdArgs& a, const float* Qs, const float* Es, int base, int len, int h, int r0, int lane) {
  constexpr int LQ = D + 4;
  static_assert(DK == 64, "head dim 64");
  const int j = lane >> 4, p = lane & 15;
  const float* Ks = Qs + 64 * LQ;
  const float* Vs = Ks + 64 * LQ;
  const float scale = 1.0f / sqrtf((float)DK);
  const float c2 = scale * 1.4426950408889634f;
  const int hc = h * DK;
  // tile rows by lane index p / by 4j + r, clamped into the tile (masked where they lie past the session)
  auto rowp = [&](int t) { return min(base + t * 16 + p, 63); };
  auto row4 = [&](int t, int r) { return min(base + t * 16 + 4 * j + r, 63); };
  // register r of key tile kb at lane (p, j) = key 16 kb + 4j + r, query 16 qa + p
#pragma unroll 1
  for (int qa = 0; qa < KT; ++qa) {
    f32x4 st[KT], dpt[KT];
#pragma unroll
    for (int kb = 0; kb < KT; ++kb) st[kb] = dpt[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int qrow = rowp(qa);
#pragma unroll
    for (int g = 0; g < DK / 16; ++g) {
      const f32x4 qf = *reinterpret_cast<const f32x4*>(Qs + qrow * LQ + hc + 4 * j + 16 * g);
      const f32x4 of = *reinterpret_cast<const f32x4*>(Es + qrow * LQ + hc + 4 * j + 16 * g);
#pragma unroll
      for (int kb = 0; kb < KT; ++kb) {
        const int krow = rowp(kb);
        const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + krow * LQ + hc + 4 * j + 16 * g);
        const f32x4 vf = *reinterpret_cast<const f32x4*>(Vs + krow * LQ + hc + 4 * j + 16 * g);
        st[kb] = amma<BF>(kf, qf, st[kb]);
        dpt[kb] = amma<BF>(vf, of, dpt[kb]);
      }
      if (KT > 1) __builtin_amdgcn_sched_barrier(0);      // keep the fragment loads of the later k groups behind these products (registers)
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < KT; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = (kb * 16 + 4 * j + r) < len ? st[kb][r] : -INFINITY;
        st[kb][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = gmax16(mx);
    const float moff = -mx * c2;
    float ps = 0.f;
#pragma unroll
    for (int kb = 0; kb < KT; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kb][r], c2, moff));      // 0 for the masked keys
        st[kb][r] = e;
        ps += e;
      }
    ps = gsum16(ps);
    const float inv = 1.f / ps;
    float dl = 0.f;
#pragma unroll
    for (int kb = 0; kb < KT; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st[kb][r] *= inv;                                          // P^T
        dl += st[kb][r] * dpt[kb][r];
      }
    dl = gsum16(dl);
#pragma unroll
    for (int kb = 0; kb < KT; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) st[kb][r] = st[kb][r] * (dpt[kb][r] - dl) * scale;      // dS^T
    // dQ[query][dim] = sum_key dS[query][key] K[key][dim]: the dS^T registers are the first operand (row = query p, k = key 4j + s);
    // K rows are read as one b128 along the head dim (lane p: dims 4p .. 4p+3) feeding four products whose column p means dim 4p + t
    f32x4 dq[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) dq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KT; ++kb) {
      f32x4 kv[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) kv[s] = *reinterpret_cast<const f32x4*>(Ks + row4(kb, s) * LQ + hc + 4 * p);
#pragma unroll
      for (int t = 0; t < 4; ++t) dq[t] = amma<BF>(st[kb], f32x4{kv[0][t], kv[1][t], kv[2][t], kv[3][t]}, dq[t]);
    }
    // register r of product t at lane (p, j) = query 16 qa + 4j + r, dim 4p + t
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = qa * 16 + 4 * j + r;
      if (q < len) *reinterpret_cast<f32x4*>(a.DQKV + ((size_t)r0 + base + q) * (3 * D) + hc + 4 * p) = f32x4{dq[0][r], dq[1][r], dq[2][r], dq[3][r]};
    }
  }
}

template <int D, int DK, int KT, bool BF>
__device__ __forceinline__ void attn_bwd_kv(const EncBlockBwdArgs& a, const float* Qs, const float* Es, int base, int len, int h, int r0, int lane) {
  constexpr int LQ = D + 4;
  const int j = lane >> 4, p = lane & 15;
  const float* Ks = Qs + 64 * LQ;
  const float* Vs = Ks + 64 * LQ;
  const float scale = 1.0f / sqrtf((float)DK);
  const float c2 = scale * 1.4426950408889634f;
  const int hc = h * DK;
  auto rowp = [&](int t) { return min(base + t * 16 + p, 63); };
  auto row4 = [&](int t, int r) { return min(base + t * 16 + 4 * j + r, 63); };
  f32x4 dk[KT][4], dv[KT][4];
#pragma unroll
  for (int kb = 0; kb < KT; ++kb)
#pragma unroll
    for (int t = 0; t < 4; ++t) dk[kb][t] = dv[kb][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // register r of key tile kb at lane (p, j) = query 16 qa + 4j + r, key 16 kb + p
#pragma unroll 1
  for (int qa = 0; qa < KT; ++qa) {
    f32x4 s2[KT], dp2[KT];
#pragma unroll
    for (int kb = 0; kb < KT; ++kb) s2[kb] = dp2[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int qrow = rowp(qa);
#pragma unroll
    for (int g = 0; g < DK / 16; ++g) {
      const f32x4 qf = *reinterpret_cast<const f32x4*>(Qs + qrow * LQ + hc + 4 * j + 16 * g);
      const f32x4 of = *reinterpret_cast<const f32x4*>(Es + qrow * LQ + hc + 4 * j + 16 * g);
#pragma unroll
      for (int kb = 0; kb < KT; ++kb) {
        const int krow = rowp(kb);
        const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + krow * LQ + hc + 4 * j + 16 * g);
        const f32x4 vf = *reinterpret_cast<const f32x4*>(Vs + krow * LQ + hc + 4 * j + 16 * g);
        s2[kb] = amma<BF>(qf, kf, s2[kb]);
        dp2[kb] = amma<BF>(of, vf, dp2[kb]);
      }
      if (KT > 1) __builtin_amdgcn_sched_barrier(0);
    }
    // per query (accumulator row r of this lane group): max / sum / rowsum(P dP) over the keys = over the 16 lanes and the key tiles
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool qok = (qa * 16 + 4 * j + r) < len;
      float mx = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < KT; ++kb) {
        const float v = (kb * 16 + p) < len ? s2[kb][r] : -INFINITY;
        s2[kb][r] = v;
        mx = fmaxf(mx, v);
      }
      mx = row16_max(mx);                      // key 0 is live for every query
      const float moff = -mx * c2;
      float ps = 0.f;
#pragma unroll
      for (int kb = 0; kb < KT; ++kb) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s2[kb][r], c2, moff));
        s2[kb][r] = e;
        ps += e;
      }
      const float inv = qok ? 1.f / row16_sum(ps) : 0.f;      // rows past the session are other sessions' rows: no contribution
      float dl = 0.f;
#pragma unroll
      for (int kb = 0; kb < KT; ++kb) {
        s2[kb][r] *= inv;                                      // P
        dl += s2[kb][r] * dp2[kb][r];
      }
      dl = row16_sum(dl);
#pragma unroll
      for (int kb = 0; kb < KT; ++kb) dp2[kb][r] = s2[kb][r] * (dp2[kb][r] - dl) * scale;      // dS
    }
    // dK[key][dim] += sum_query dS[query][key] Q[query][dim], dV[key][dim] += sum_query P[query][key] dO[query][dim]
    f32x4 qv[4], ov[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qv[s] = *reinterpret_cast<const f32x4*>(Qs + row4(qa, s) * LQ + hc + 4 * p);
      ov[s] = *reinterpret_cast<const f32x4*>(Es + row4(qa, s) * LQ + hc + 4 * p);
    }
#pragma unroll
    for (int kb = 0; kb < KT; ++kb)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        dk[kb][t] = amma<BF>(dp2[kb], f32x4{qv[0][t], qv[1][t], qv[2][t], qv[3][t]}, dk[kb][t]);
        dv[kb][t] = amma<BF>(s2[kb], f32x4{ov[0][t], ov[1][t], ov[2][t], ov[3][t]}, dv[kb][t]);
      }
  }
#pragma unroll
  for (int kb = 0; kb < KT; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = kb * 16 + 4 * j + r;
      if (k < len) {
        float* dst = a.DQKV + ((size_t)r0 + base + k) * (3 * D) + hc + 4 * p;
        *reinterpret_cast<f32x4*>(dst + D) = f32x4{dk[kb][0][r], dk[kb][1][r], dk[kb][2][r], dk[kb][3][r]};
        *reinterpret_cast<f32x4*>(dst + 2 * D) = f32x4{dv[kb][0][r], dv[kb][1][r], dv[kb][2][r], dv[kb][3][r]};
      }
    }
}

template <int D, int DK, int NP>
__global__ __launch_bounds__((EncBwdCfg<D>::NT), 2) void enc_block_bwd_kernel(EncBlockBwdArgs a) {
  using C = EncBwdCfg<D>;
  constexpr int NW = C::NW, NT = C::NT, LDP = C::LDP, PLANE = C::PLANE, LQ = C::LQ, HEADS = D / DK;
  constexpr int CPL = D / 64, ROUNDS = 64 / (NW * 4);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ int s_start[65];
  __shared__ int s_rowlast[64];
  __shared__ __attribute__((aligned(16))) float s_g[2 * D];                  // gamma2 | gamma1
  float* Es = reinterpret_cast<float*>(smem_raw);
  __bf16* zplanes = reinterpret_cast<__bf16*>(smem_raw + C::ES_BYTES);                  // dZ2 planes
  __bf16* fplanes = reinterpret_cast<__bf16*>(smem_raw + C::ES_BYTES + C::P_BYTES);     // dF1 planes
  float* Qs = reinterpret_cast<float*>(smem_raw + C::ES_BYTES);                          // later: q | k | v rows
  const int tid = threadIdx.x, lane = tid & 63, j = lane >> 4, p = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2 * D; i += NT) s_g[i] = i < D ? a.g2[i] : a.g1[i - D];
  f32x4 pg2[CPL], pb2[CPL], pg1[CPL], pb1[CPL];      // this lane's partial d(gamma) / d(beta) of its columns (D/16) p + 4 cc ..
#pragma unroll
  for (int cc = 0; cc < CPL; ++cc) pg2[cc] = pb2[cc] = pg1[cc] = pb1[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned long long tstamp = 0;
  const bool probe = a.dbg != nullptr && blockIdx.x == 0 && tid == 0;
  auto mark = [&](int ph) {
    if (probe) {
      const unsigned long long now = clock64();
      if (ph >= 0) a.dbg[ph] += now - tstamp;
      tstamp = now;
    }
  };
  mark(-1);
  for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
    const int s_lo = a.tile_s[t], s_hi = a.tile_s[t + 1];
    const int ns = s_hi - s_lo;
    if (ns <= 0) break;                                      // only the last window can be without a session start
    const int r0 = a.off[s_lo];
    const int nrows = (s_hi < a.B ? a.off[s_hi] : a.rows) - r0;
    if (wave == 0) {
      s_rowlast[lane] = -1;
      if (lane < ns) {
        const int start = a.off[s_lo + lane] - r0;
        const int en = (s_lo + lane + 1 < a.B ? a.off[s_lo + lane + 1] : a.rows) - r0;
        s_start[lane] = start;
        if (en > start) s_rowlast[en - 1] = s_lo + lane;       // an empty history has no last row
      }
      if (lane == 0) s_start[ns] = nrows;
    }
    __syncthreads();      // bookkeeping visible; also: everybody is done with the previous tile's LDS
    mark(0);
    // ---- dE (+ d(x_last)) -> LayerNorm2 backward -> dZ2: fp32 rows, planes, HBM.  16 lanes per row, four rows per wave at a time
    const float inv_n = 1.f / (float)D;
#pragma unroll
    for (int rnd = 0; rnd < ROUNDS; ++rnd) {
      const int row = (rnd * NW + wave) * 4 + j;
      const bool rok = row < nrows;
      const size_t grow = (size_t)r0 + (rok ? row : 0);
      const int lastof = (rok && a.dxl) ? s_rowlast[row] : -1;
      const float rs = a.RSTD2[grow];
      f32x4 de[CPL], xh[CPL], g[CPL];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int cc = 0; cc < CPL; ++cc) {
        const int col = (D / 16) * p + 4 * cc;
        de[cc] = *reinterpret_cast<const f32x4*>(a.dE + grow * D + col);
        if (lastof >= 0) de[cc] += *reinterpret_cast<const f32x4*>(a.dxl + (size_t)lastof * D + col);
        xh[cc] = *reinterpret_cast<const f32x4*>(a.XH2 + grow * D + col);
        g[cc] = de[cc] * *reinterpret_cast<const f32x4*>(s_g + col);
        s1 += (g[cc][0] + g[cc][1]) + (g[cc][2] + g[cc][3]);
        s2 += (g[cc][0] * xh[cc][0] + g[cc][1] * xh[cc][1]) + (g[cc][2] * xh[cc][2] + g[cc][3] * xh[cc][3]);
      }
      const float m1 = row16_sum(s1) * inv_n, m2 = row16_sum(s2) * inv_n;
#pragma unroll
      for (int cc = 0; cc < CPL; ++cc) {
        const int col = (D / 16) * p + 4 * cc;
        f32x4 dz = (g[cc] - m1 - xh[cc] * m2) * rs;
        if (!rok) dz = f32x4{0.f, 0.f, 0.f, 0.f};
        else {
          pg2[cc] += de[cc] * xh[cc];
          pb2[cc] += de[cc];
          *reinterpret_cast<f32x4*>(a.DZ2 + grow * D + col) = dz;
        }
        *reinterpret_cast<f32x4*>(Es + row * LQ + col) = dz;
        store4<NP, PLANE>(zplanes + row * LDP + col, dz);
      }
    }
    mark(1);
    lds_barrier();
    mark(2);
    float warm[2];
    // ---- dF1 = (dZ2 W2) * [F1 > 0]; wave = one column tile, four row tiles
    {
      f32x4 acc[1][4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[0][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int col = wave * 16 + 4 * j;
      f32x4 mk[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) mk[rt] = *reinterpret_cast<const f32x4*>(a.F1 + ((size_t)r0 + min(rt * 16 + p, nrows - 1)) * D + col);
      // the stashed q / k / v and LayerNorm1 x-hat rows of this tile have not been touched since the forward pass: one 4-byte read per
      // 128-byte line now pulls them from HBM into L2 while the two products run (they are staged / read two phases from here)
      auto touch = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int i = tid + NT * u;                       // 64 rows x (12 q/k/v + 4 x-hat) lines
          const int row = min(i >> 4, nrows - 1), seg = i & 15;
          const float* src = seg < 12 ? a.QKV + ((size_t)r0 + row) * (3 * D) + seg * 32 : a.XH1 + ((size_t)r0 + row) * D + (seg - 12) * 32;
          warm[u] = *src;
        }
      };
      gemm_planes<D, 1, 4, 64, NP>(zplanes + p * LDP + 8 * j, launder(a.W2T) + lane, wave, acc, touch);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const int row = rt * 16 + p;
        f32x4 x = acc[0][rt];
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = mk[rt][r] > 0.f ? x[r] : 0.f;
        store4<NP, PLANE>(fplanes + row * LDP + col, x);
        if (row < nrows) *reinterpret_cast<f32x4*>(a.DF1 + ((size_t)r0 + row) * D + col) = x;
      }
    }
    mark(3);
    lds_barrier();
    mark(4);
    // ---- dC = dF1 W1 + dZ2, in place over the dZ2 rows (every element is read and rewritten by the same lane)
    {
      f32x4 acc[1][4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[0][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      gemm_planes<D, 1, 4, 64, NP>(fplanes + p * LDP + 8 * j, launder(a.W1T) + lane, wave, acc);
      const int col = wave * 16 + 4 * j;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        float* e = Es + (rt * 16 + p) * LQ + col;
        *reinterpret_cast<f32x4*>(e) = *reinterpret_cast<const f32x4*>(e) + acc[0][rt];
      }
    }
    mark(5);
    lds_barrier();
    mark(6);
    // ---- LayerNorm1 backward -> dZ1 (= the attention output's gradient and the residual into dX), in place; the stashed q / k / v
    // rows of the tile come in over the dead planes
    asm volatile("" ::"v"(warm[0]), "v"(warm[1]));         // (the L2 warm-up reads end here)
    for (int i = tid; i < 64 * (3 * D / 4); i += NT) {
      const int row = i / (3 * D / 4), c4 = i - row * (3 * D / 4);
      const int which = c4 / (D / 4), col = (c4 - which * (D / 4)) * 4;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (row < nrows) v = *reinterpret_cast<const f32x4*>(a.QKV + ((size_t)r0 + row) * (3 * D) + c4 * 4);
      *reinterpret_cast<f32x4*>(Qs + which * 64 * LQ + row * LQ + col) = v;
    }
#pragma unroll
    for (int rnd = 0; rnd < ROUNDS; ++rnd) {
      const int row = (rnd * NW + wave) * 4 + j;
      const bool rok = row < nrows;
      const size_t grow = (size_t)r0 + (rok ? row : 0);
      const float rs = a.RSTD1[grow];
      f32x4 dc[CPL], xh[CPL], g[CPL];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int cc = 0; cc < CPL; ++cc) {
        const int col = (D / 16) * p + 4 * cc;
        dc[cc] = *reinterpret_cast<const f32x4*>(Es + row * LQ + col);
        xh[cc] = *reinterpret_cast<const f32x4*>(a.XH1 + grow * D + col);
        g[cc] = dc[cc] * *reinterpret_cast<const f32x4*>(s_g + D + col);
        s1 += (g[cc][0] + g[cc][1]) + (g[cc][2] + g[cc][3]);
        s2 += (g[cc][0] * xh[cc][0] + g[cc][1] * xh[cc][1]) + (g[cc][2] * xh[cc][2] + g[cc][3] * xh[cc][3]);
      }
      const float m1 = row16_sum(s1) * inv_n, m2 = row16_sum(s2) * inv_n;
#pragma unroll
      for (int cc = 0; cc < CPL; ++cc) {
        const int col = (D / 16) * p + 4 * cc;
        f32x4 dz = (g[cc] - m1 - xh[cc] * m2) * rs;
        if (!rok) dz = f32x4{0.f, 0.f, 0.f, 0.f};
        else {
          pg1[cc] += dc[cc] * xh[cc];
          pb1[cc] += dc[cc];
          *reinterpret_cast<f32x4*>(a.DZ1 + grow * D + col) = dz;
        }
        *reinterpret_cast<f32x4*>(Es + row * LQ + col) = dz;
      }
    }
    mark(7);
    lds_barrier();
    mark(8);
    // ---- attention backward: (session, head, pass) work items over the waves
    for (int it = wave; it < ns * HEADS * 2; it += NW) {
      const int s = it / (HEADS * 2), hp = it - s * (HEADS * 2), h = hp >> 1;
      const int base = s_start[s], len = s_start[s + 1] - base;
      if (hp & 1) {
        if (len > 16) attn_bwd_kv<D, DK, 2, NP == 1>(a, Qs, Es, base, len, h, r0, lane);
        else attn_bwd_kv<D, DK, 1, NP == 1>(a, Qs, Es, base, len, h, r0, lane);
      } else {
        if (len > 16) attn_bwd_q<D, DK, 2, NP == 1>(a, Qs, Es, base, len, h, r0, lane);
        else attn_bwd_q<D, DK, 1, NP == 1>(a, Qs, Es, base, len, h, r0, lane);
      }
    }
    mark(9);
    lds_barrier();        // the next tile's bookkeeping and dZ2 rows go over what the attention reads
    mark(10);
  }
  // ---- LayerNorm parameter gradients of this workgroup: lanes of one DPP row group share their columns across j; waves through LDS
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem_raw);            // [NW][4][D]
#pragma unroll
  for (int cc = 0; cc < CPL; ++cc) {
    f32x4 v[4] = {pg2[cc], pb2[cc], pg1[cc], pb1[cc]};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[k][e] = gsum16(v[k][e]);
      if (j == 0) *reinterpret_cast<f32x4*>(red + (wave * 4 + k) * D + (D / 16) * p + 4 * cc) = v[k];
    }
  }
  __syncthreads();
  for (int i = tid; i < 4 * D; i += NT) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w * 4 * D + i];
    a.slab[(size_t)blockIdx.x * 4 * D + i] = s;
  }
}

// ======================================================================================================================
// pruned last block backward: 16 sessions per workgroup
// ======================================================================================================================
struct EncLastBwdArgs {
  const float* dvec; int ldv;            // [B, ldv] gradient of the encoder's output vector
  const float* KV; const int* off; const int* len;
  int B, T;
  const uint4* W2T; const uint4* W1T; const uint4* WqT;
  const float* g1; const float* g2;
  const float* XH2; const float* RSTD2; const float* F1; const float* XH1; const float* RSTD1; const float* PL; const float* QL;
  float* DZ2; float* DF1; float* DQ; float* DXL;      // [B, D]
  float* DKV;                                           // [rows, 2D]
  float* slab;                                          // [gridDim.x][4][D]
};

template <int D>
struct EncLastBwdCfg {
  static constexpr int NW = D / 16, NT = NW * 64, LDP = D + 8, PLANE = 16 * LDP, LQ = D + 4;
  static constexpr size_t PL_BYTES = (size_t)3 * PLANE * 2, T_BYTES = (size_t)16 * LQ * 4;
  static constexpr size_t RED_BYTES = (size_t)NW * 4 * D * 4;
  static constexpr size_t SMEM = 3 * PL_BYTES + 2 * T_BYTES + RED_BYTES;      // dz2 / df1 / dq planes; dz2 rows, working rows; reduction
};

template <int D, int DK, int NP>
__global__ __launch_bounds__((EncLastBwdCfg<D>::NT)) void enc_last_bwd_kernel(EncLastBwdArgs a) {
  using C = EncLastBwdCfg<D>;
  constexpr int LDP = C::LDP, PLANE = C::PLANE, LQ = C::LQ, HEADS = D / DK, NW = C::NW, SPB = ENC_LAST_SPB, SLOTS = 16 / C::NW;
  static_assert(HEADS == 2 && DK == 64, "one-row attention: lane = (head, key), 32 keys per head");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* zpl = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* fpl = reinterpret_cast<__bf16*>(smem_raw + C::PL_BYTES);
  __bf16* qpl = reinterpret_cast<__bf16*>(smem_raw + 2 * C::PL_BYTES);
  float* DZs = reinterpret_cast<float*>(smem_raw + 3 * C::PL_BYTES);      // dz2 rows
  float* Ws = DZs + 16 * LQ;                                              // dcl rows -> dz1 rows
  float* red = Ws + 16 * LQ;
  const int tid = threadIdx.x, lane = tid & 63, j = lane >> 4, p = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b0 = blockIdx.x * SPB;
  const float scale = 1.0f / sqrtf((float)DK);
  const float inv_n = 1.f / (float)D;
  float pg2[2] = {0.f, 0.f}, pb2[2] = {0.f, 0.f}, pg1[2] = {0.f, 0.f}, pb1[2] = {0.f, 0.f};      // columns lane, 64 + lane
  // ---- LayerNorm2 backward: a wave owns whole sessions, lane = columns lane and 64 + lane
#pragma unroll 1
  for (int ss = 0; ss < SLOTS; ++ss) {
    const int s = ss * NW + wave, b = b0 + s;          // tile row s holds session b0 + s when s < SPB
    float z0 = 0.f, z1 = 0.f;
    if (s < SPB && b < a.B) {
      const float d0 = a.dvec[(size_t)b * a.ldv + lane], d1 = a.dvec[(size_t)b * a.ldv + 64 + lane];
      const float x0 = a.XH2[(size_t)b * D + lane], x1 = a.XH2[(size_t)b * D + 64 + lane];
      const float g0 = d0 * a.g2[lane], g1v = d1 * a.g2[64 + lane];
      const float m1 = wave_sum(g0 + g1v) * inv_n, m2 = wave_sum(g0 * x0 + g1v * x1) * inv_n;
      const float rs = a.RSTD2[b];
      z0 = (g0 - m1 - x0 * m2) * rs;
      z1 = (g1v - m1 - x1 * m2) * rs;
      pg2[0] += d0 * x0; pg2[1] += d1 * x1; pb2[0] += d0; pb2[1] += d1;
      a.DZ2[(size_t)b * D + lane] = z0;
      a.DZ2[(size_t)b * D + 64 + lane] = z1;
    }
    DZs[s * LQ + lane] = z0;
    DZs[s * LQ + 64 + lane] = z1;
    __bf16 hh, mm, ll;
    split1(z0, hh, mm, ll);
    zpl[s * LDP + lane] = hh; zpl[PLANE + s * LDP + lane] = mm; zpl[2 * PLANE + s * LDP + lane] = ll;
    split1(z1, hh, mm, ll);
    zpl[s * LDP + 64 + lane] = hh; zpl[PLANE + s * LDP + 64 + lane] = mm; zpl[2 * PLANE + s * LDP + 64 + lane] = ll;
  }
  lds_barrier();
  const int col = wave * 16 + 4 * j;
  const bool rowok = p < SPB && b0 + p < a.B;
  const size_t growp = (size_t)(rowok ? b0 + p : 0);
  // ---- df1 = (dz2 W2) * [f1 > 0]
  {
    f32x4 acc[1][1];
    acc[0][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 mk = *reinterpret_cast<const f32x4*>(a.F1 + growp * D + col);
    gemm_planes<D, 1, 1, 16, NP>(zpl + p * LDP + 8 * j, launder(a.W2T) + lane, wave, acc);
    f32x4 x = acc[0][0];
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = (rowok && mk[r] > 0.f) ? x[r] : 0.f;
    store4<NP, PLANE>(fpl + p * LDP + col, x);
    if (rowok) *reinterpret_cast<f32x4*>(a.DF1 + growp * D + col) = x;
  }
  lds_barrier();
  // ---- dcl = df1 W1 + dz2
  {
    f32x4 acc[1][1];
    acc[0][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm_planes<D, 1, 1, 16, NP>(fpl + p * LDP + 8 * j, launder(a.W1T) + lane, wave, acc);
    *reinterpret_cast<f32x4*>(Ws + p * LQ + col) = acc[0][0] + *reinterpret_cast<const f32x4*>(DZs + p * LQ + col);
  }
  lds_barrier();
  // ---- LayerNorm1 backward, then the one-row attention's backward; a wave owns whole sessions
#pragma unroll 1
  for (int ss = 0; ss < SLOTS; ++ss) {
    const int s = ss * NW + wave, b = b0 + s;
    float q0 = 0.f, q1 = 0.f;                      // dq of this session, columns lane and 64 + lane
    if (s < SPB && b < a.B) {                      // wave-uniform
      const float c0 = Ws[s * LQ + lane], c1 = Ws[s * LQ + 64 + lane];
      const float x0 = a.XH1[(size_t)b * D + lane], x1 = a.XH1[(size_t)b * D + 64 + lane];
      const float g0 = c0 * a.g1[lane], g1v = c1 * a.g1[64 + lane];
      const float m1 = wave_sum(g0 + g1v) * inv_n, m2 = wave_sum(g0 * x0 + g1v * x1) * inv_n;
      const float rs = a.RSTD1[b];
      const float o0 = (g0 - m1 - x0 * m2) * rs, o1 = (g1v - m1 - x1 * m2) * rs;      // dz1 = d(attention output) = residual gradient
      pg1[0] += c0 * x0; pg1[1] += c1 * x1; pb1[0] += c0; pb1[1] += c1;
      Ws[s * LQ + lane] = o0;                       // (row s is this wave's: read above, rewritten here)
      Ws[s * LQ + 64 + lane] = o1;
      __builtin_amdgcn_wave_barrier();
      const int n = a.len[b];
      const size_t rb = (size_t)a.off[b];
      const int h = lane >> 5, t = lane & 31;
      float dp = 0.f;
      if (t < n) {                                  // dP_t = <dO_h, V_t>
        const float* vr = a.KV + (rb + t) * (2 * D) + D + h * DK;
        const float* orow = Ws + s * LQ + h * DK;
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int c = 0; c < DK; c += 8) {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(vr + c), v1 = *reinterpret_cast<const f32x4*>(vr + c + 4);
          const f32x4 g0v = *reinterpret_cast<const f32x4*>(orow + c), g1w = *reinterpret_cast<const f32x4*>(orow + c + 4);
          d0 += (v0[0] * g0v[0] + v0[1] * g0v[1]) + (v0[2] * g0v[2] + v0[3] * g0v[3]);
          d1 += (v1[0] * g1w[0] + v1[1] * g1w[1]) + (v1[2] * g1w[2] + v1[3] * g1w[3]);
        }
        dp = d0 + d1;
      }
      const float pw = (t < n && t < a.T) ? a.PL[((size_t)b * HEADS + h) * a.T + t] : 0.f;
      float dl = pw * dp;
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) dl += __shfl_xor(dl, o);
      const float ds = pw * (dp - dl) * scale;
      const float ql0 = a.QL[(size_t)b * D + lane], ql1 = a.QL[(size_t)b * D + 64 + lane];
      const float* kr = a.KV + rb * (2 * D);
      float* dkr = a.DKV + rb * (2 * D);
      for (int tt = 0; tt < n; ++tt) {
        const float s0 = __shfl(ds, tt), s1 = __shfl(ds, 32 + tt);
        const float p0 = __shfl(pw, tt), p1 = __shfl(pw, 32 + tt);
        q0 = __builtin_fmaf(s0, kr[(size_t)tt * (2 * D) + lane], q0);
        q1 = __builtin_fmaf(s1, kr[(size_t)tt * (2 * D) + 64 + lane], q1);
        dkr[(size_t)tt * (2 * D) + lane] = s0 * ql0;
        dkr[(size_t)tt * (2 * D) + 64 + lane] = s1 * ql1;
        dkr[(size_t)tt * (2 * D) + D + lane] = p0 * o0;
        dkr[(size_t)tt * (2 * D) + D + 64 + lane] = p1 * o1;
      }
      a.DQ[(size_t)b * D + lane] = q0;
      a.DQ[(size_t)b * D + 64 + lane] = q1;
    } else {
      Ws[s * LQ + lane] = 0.f;
      Ws[s * LQ + 64 + lane] = 0.f;
    }
    __bf16 hh, mm, ll;
    split1(q0, hh, mm, ll);
    qpl[s * LDP + lane] = hh; qpl[PLANE + s * LDP + lane] = mm; qpl[2 * PLANE + s * LDP + lane] = ll;
    split1(q1, hh, mm, ll);
    qpl[s * LDP + 64 + lane] = hh; qpl[PLANE + s * LDP + 64 + lane] = mm; qpl[2 * PLANE + s * LDP + 64 + lane] = ll;
  }
  lds_barrier();
  // ---- d(x_last) = dq Wq + dz1
  {
    f32x4 acc[1][1];
    acc[0][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm_planes<D, 1, 1, 16, NP>(qpl + p * LDP + 8 * j, launder(a.WqT) + lane, wave, acc);
    if (rowok) *reinterpret_cast<f32x4*>(a.DXL + growp * D + col) = acc[0][0] + *reinterpret_cast<const f32x4*>(Ws + p * LQ + col);
  }
  // ---- LayerNorm parameter gradients of this workgroup
  red[(wave * 4 + 0) * D + lane] = pg2[0]; red[(wave * 4 + 0) * D + 64 + lane] = pg2[1];
  red[(wave * 4 + 1) * D + lane] = pb2[0]; red[(wave * 4 + 1) * D + 64 + lane] = pb2[1];
  red[(wave * 4 + 2) * D + lane] = pg1[0]; red[(wave * 4 + 2) * D + 64 + lane] = pg1[1];
  red[(wave * 4 + 3) * D + lane] = pb1[0]; red[(wave * 4 + 3) * D + 64 + lane] = pb1[1];
  lds_barrier();
  for (int i = tid; i < 4 * D; i += C::NT) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w * 4 * D + i];
    a.slab[(size_t)blockIdx.x * 4 * D + i] = s;
  }
}

}  // namespace

size_t enc_bwd_slab_floats(int rows, int B, int T, int dm) {
  const int ntiles = cdiv(rows, enc_tile_rows(T));
  int g = num_cus();
  if (g > ntiles) g = ntiles;
  const int gl = cdiv(B, ENC_LAST_SPB);
  return (size_t)(g > gl ? g : gl) * 4 * dm + 64;
}

int launch_enc_block_bwd(const EncBlockBwd& f, hipStream_t st, ReduceQueue* q) {
  if (f.B <= 0 || f.rows <= 0) return 0;
  INTEL_CHECK_ARG(enc_fused_supported(f.T, f.dm, f.heads), "enc_block_bwd: unsupported shape T=%d dm=%d heads=%d", f.T, f.dm, f.heads);
  INTEL_CHECK_ARG(q, "enc_block_bwd: needs the reduce queue");
  constexpr int D = 128;
  using C = EncBwdCfg<D>;
  EncBlockBwdArgs a;
  a.dE = f.dE; a.dxl = f.dxl; a.rows = f.rows; a.B = f.B; a.T = f.T; a.ntiles = cdiv(f.rows, enc_tile_rows(f.T));
  a.off = f.off; a.tile_s = f.tile_s;
  a.W2T = reinterpret_cast<const uint4*>(f.W2T); a.W1T = reinterpret_cast<const uint4*>(f.W1T);
  a.g1 = f.g1; a.g2 = f.g2;
  a.XH2 = f.XH2; a.RSTD2 = f.RSTD2; a.F1 = f.F1; a.XH1 = f.XH1; a.RSTD1 = f.RSTD1; a.QKV = f.QKV;
  a.DZ2 = f.DZ2; a.DF1 = f.DF1; a.DZ1 = f.DZ1; a.DQKV = f.DQKV;
  int grid = num_cus();
  if (grid > a.ntiles) grid = a.ntiles;
  float* slab = redq_alloc(q, (size_t)grid * 4 * D);
  INTEL_CHECK_ARG(slab, "enc_block_bwd: reduction arena exhausted");
  a.slab = slab;
  const bool bf = gemm_planes() == 1;
  if (bf) allow_lds((enc_block_bwd_kernel<D, 64, 1>), C::SMEM);
  else allow_lds((enc_block_bwd_kernel<D, 64, 3>), C::SMEM);
  const double M = (double)f.rows;
  static const int dbg_on = INTEL_DEBUG_ENV("INTEL_ENC_DBG", 0);      // phase clocks: debug builds only (common.h)
  static unsigned long long* dbg_buf = nullptr;
  a.dbg = nullptr;
  if (dbg_on) {
    if (!dbg_buf) (void)hipMalloc(&dbg_buf, 16 * sizeof(unsigned long long));
    (void)hipMemsetAsync(dbg_buf, 0, 16 * sizeof(unsigned long long), st);
    a.dbg = dbg_buf;
  }
  if (bf) LAUNCH_S(f.rows, D, 64, 2.0 * M * D * D * 2 + 10.0 * M * f.T * D * 0.5, 4.0 * M * D * 12, (enc_block_bwd_kernel<D, 64, 1>), dim3(grid), dim3(C::NT), C::SMEM, st, a);
  else LAUNCH_S(f.rows, D, 64, 2.0 * M * D * D * 2 + 10.0 * M * f.T * D * 0.5, 4.0 * M * D * 12, (enc_block_bwd_kernel<D, 64, 3>), dim3(grid), dim3(C::NT), C::SMEM, st, a);
  INTEL_CHECK_LAUNCH();
  if (dbg_on) {
    unsigned long long h[16];
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h, dbg_buf, sizeof(h), hipMemcpyDeviceToHost);
    const int iters = (a.ntiles + grid - 1) / grid;
    fprintf(stderr, "enc_block_bwd grid=%d iters=%d cycles/tile: meta %llu ln2 %llu (+bar %llu) w2 %llu (+bar %llu) w1 %llu (+bar %llu) ln1+stage %llu (+bar %llu) attn %llu (+bar %llu)\n",
            grid, iters, h[0] / iters, h[1] / iters, h[2] / iters, h[3] / iters, h[4] / iters, h[5] / iters, h[6] / iters, h[7] / iters, h[8] / iters, h[9] / iters, h[10] / iters);
  }
  float* outs[4] = {f.dg2, f.db2, f.dg1, f.db1};
  const int accs[4] = {f.acc_g2, f.acc_b2, f.acc_g1, f.acc_b1};
  for (int k = 0; k < 4; ++k)
    if (outs[k]) redq_push(q, slab + (size_t)k * D, (size_t)4 * D, grid, 1, D, outs[k], D, accs[k]);
  return 0;
}

int launch_enc_last_bwd(const EncLastBwd& f, hipStream_t st, ReduceQueue* q) {
  if (f.B <= 0) return 0;
  INTEL_CHECK_ARG(enc_fused_supported(f.T, f.dm, f.heads), "enc_last_bwd: unsupported shape T=%d dm=%d heads=%d", f.T, f.dm, f.heads);
  INTEL_CHECK_ARG(q, "enc_last_bwd: needs the reduce queue");
  constexpr int D = 128;
  using C = EncLastBwdCfg<D>;
  EncLastBwdArgs a;
  a.dvec = f.dvec; a.ldv = f.ldv; a.KV = f.KV; a.off = f.off; a.len = f.len; a.B = f.B; a.T = f.T;
  a.W2T = reinterpret_cast<const uint4*>(f.W2T); a.W1T = reinterpret_cast<const uint4*>(f.W1T); a.WqT = reinterpret_cast<const uint4*>(f.WqT);
  a.g1 = f.g1; a.g2 = f.g2;
  a.XH2 = f.XH2; a.RSTD2 = f.RSTD2; a.F1 = f.F1; a.XH1 = f.XH1; a.RSTD1 = f.RSTD1; a.PL = f.PL; a.QL = f.QL;
  a.DZ2 = f.DZ2; a.DF1 = f.DF1; a.DQ = f.DQ; a.DXL = f.DXL; a.DKV = f.DKV;
  const int grid = cdiv(f.B, ENC_LAST_SPB);
  float* slab = redq_alloc(q, (size_t)grid * 4 * D);
  INTEL_CHECK_ARG(slab, "enc_last_bwd: reduction arena exhausted");
  a.slab = slab;
  if (gemm_planes() == 1) {
    allow_lds((enc_last_bwd_kernel<D, 64, 1>), C::SMEM);
    LAUNCH_S(f.B, D, 64, 2.0 * f.B * (double)D * D * 3, 4.0 * f.B * (double)D * (8.0 + 2.0 * f.T), (enc_last_bwd_kernel<D, 64, 1>), dim3(grid), dim3(C::NT), C::SMEM, st, a);
  } else {
    allow_lds((enc_last_bwd_kernel<D, 64, 3>), C::SMEM);
    LAUNCH_S(f.B, D, 64, 2.0 * f.B * (double)D * D * 3, 4.0 * f.B * (double)D * (8.0 + 2.0 * f.T), (enc_last_bwd_kernel<D, 64, 3>), dim3(grid), dim3(C::NT), C::SMEM, st, a);
  }
  INTEL_CHECK_LAUNCH();
  float* outs[4] = {f.dg2, f.db2, f.dg1, f.db1};
  const int accs[4] = {f.acc_g2, f.acc_b2, f.acc_g1, f.acc_b1};
  for (int k = 0; k < 4; ++k)
    if (outs[k]) redq_push(q, slab + (size_t)k * D, (size_t)4 * D, grid, 1, D, outs[k], D, accs[k]);
  return 0;
}
