// The MIDDLE of one tied tower layer's backward (models/IntEL/IntEL.py:182-188 / 191-197, torch autograd of
//
//     h = MHA(x, x, x);  f = relu(W1 h + b1);  z = W2 f + b2;  y = LayerNorm(z + x)
//
// ) as ONE kernel per layer: from dZ (the gradient behind the LayerNorm) to dQKV (the gradient of the fused q/k/v projection), with the
// weight gradients of both feed-forward linears accumulated in the workgroup.  A workgroup owns one session at a time (persistent over
// sessions, lists of up to 64 candidates, widths 64 / 128); nothing between dZ and dQKV touches HBM:
//
//   P0  A (the forward's attention-output stash) and dZ tiles  global -> three bf16 planes each in LDS
//   P1  R1 = relu(A W1^T + b1)                       recomputed on the bf16 pipe (six plane products = fp32 accuracy) -> planes
//   P2  dW2 += dZ^T R1, db2 += colsum dZ             (operands k-contiguous along the ROWS: ds_read_b64_tr_b16 from the row-major planes)
//       dF1 = (dZ W2) * [R1 > 0]                     -> planes, over R1
//   P3  dW1 += dF1^T A, db1 += colsum dF1;  dA = dF1 W1 -> fp32 rows
//   P4  X tile -> planes;  [Q | K | V] = X Wqkv^T    recomputed -> fp32 rows
//   P5  attention backward per head on exact fp32 MFMAs (bf16 mode: single bf16 MFMAs):
//         pass 1, wave = (16 keys, half of the queries): S and dP once, P from the forward's log-sum-exp, delta = rowsum(P * dP) summed over the key tiles
//                 through LDS, dS -> LDS, partial dV = P^T dA and
//                 dK = dS^T Q in registers; the two query halves are summed through the dead Q / dA columns
//         pass 2, wave = (16 queries, half of the head dim): dQ = dS K
//       dQ / dK / dV leave as rows of dQKV [B L, 3 D]
//
// What stays outside: the LayerNorm backward in front (fused into the pooling backward for the last layer: session.hip), and behind it the
// q/k/v weight gradient dQKV^T X and dX = dQKV Wqkv + dZ (gemm.hip; at D = 128 the 192 KB of accumulators of dWqkv do not fit beside the
// 128 KB of dW1 / dW2).  HBM per session and layer: A, dZ, X in, dQKV out = 24 L D bytes, against 100 L D of the kernel-per-op middle
// (R1, dF1 twice, dA twice, QKV, A twice, dZ three times ...); the forward stashes A and the log-sum-exp only.
// Padded rows (row >= L) are zero rows of A / dZ / X: they add nothing to any gradient; their keys are masked, their queries get P = 0.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "kernels.h"
#include "planes.h"

namespace {

using namespace planes;

struct TowerBwdArgs {
  const float* X;        // [B*L, D] layer input
  const float* A;        // [B*L, D] attention output (forward stash; bf16 array when a16)
  const float* LSE;      // [B*heads*L] natural-log softmax normalisers of the forward
  const float* dZ;       // [B*L, D] gradient behind the LayerNorm
  int B, L;
  int a16, dqkv16;       // bf16 mode: A is read / dQKV is written as a bf16 array of the same shape
  const uint4* Wqkv;     // three-plane images (launch_pack_b3): [D -> 3D] forward weights,
  const uint4* W1;       //   [D -> D] forward W1 (R1 = A W1^T),
  const uint4* W2T;      //   W2 transposed (dR1 = dZ W2),
  const uint4* W1T;      //   W1 transposed (dA = dF1 W1)
  const uint4* WqkvT;    //   [3D -> D] the stacked q/k/v weights transposed (dX = dQKV Wqkv), k extent 3D
  const float* b1;
  float* dQKV;           // [B*L, 3D] (not written where the kernel consumes it itself: TowerBwdScope)
  float* dX;             // [B*L, D] = dQKV Wqkv + dZ (scope bit 0)
  float* slabs;          // per workgroup: dW2 [D, D] | db2 [D] | dW1 [D, D] | db1 [D] (| dWq | dWk | dWv [D, D] each: scope bit 1)
  unsigned long long* dbg;   // INTEL_TOWER_DBG=1 (debug builds): per-phase shader-clock totals of workgroup 0's thread 0 (NULL otherwise)
};

template <int D, int NP>
struct BwdCfg {
  static constexpr int NW = 8, NT = 512;
  static constexpr int KB = D / 32, CTW = D / 16;
  static constexpr int RS = NW / CTW, RT = 4 / RS;          // row splits of a linear over the waves, row tiles per wave
  static constexpr int WPN = NW / CTW, KTL = CTW / WPN;      // weight gradient: waves per 16-row block of dW, 16-column tiles per wave
  static constexpr int LDP = D + 8, PLANE = 64 * LDP;        // bf16 plane pitch / elements
  static constexpr int LQ = D + 4;                            // fp32 row pitch
  static constexpr int DSP = 68;                              // dS [64 queries][64 keys] pitch
  static constexpr size_t P3 = (size_t)NP * PLANE * 2;
  static constexpr size_t F = (size_t)64 * LQ * 4;
  static constexpr size_t DS = (size_t)64 * DSP * 4;
  // bf16 mode (NP = 1, "PA"): the attention operands live as bf16 images too (pitch D + 16: row reads and the accumulator-order transposing reads are
  // conflict-free), every product of the attention backward is a v_mfma_f32_16x16x32_bf16 -- five slots of one image each + the dS image
  static constexpr bool PA = NP == 1;
  static constexpr int LDA = D + 16;
  static constexpr size_t SLOT = (size_t)64 * LDA * 2;
  static constexpr size_t DSI = (size_t)64 * 72 * 2;
  static constexpr size_t R12 = PA ? 5 * SLOT : (2 * P3 > 3 * F ? 2 * P3 : 3 * F);      // P(A) | P(R1 -> dF1), later fp32 Q | K | V
  static constexpr size_t R3 = PA ? DSI : (P3 > F + DS ? P3 : F + DS);                  // P(dZ), later fp32 dA + dS
  static constexpr size_t STAT = (size_t)(2 * 64 + 4 * 64) * 4;      // lse2 [2][64], dpart [4][64]
  // scope of the kernel beyond dQKV: FULLX = dX = dQKV Wqkv + dZ too, FULLW = the q/k/v weight gradients too (their 3 D^2 accumulators fit the registers
  // at D = 64 only; six-plane operands at D = 128 leave no room for either)
  static constexpr bool FULLX = D == 64 || NP == 1, FULLW = D == 64;
  static constexpr size_t R4 = (FULLX && NP == 3) ? 2 * P3 : 0;      // P(X) | P(dQ / dK / dV); one plane: over the dead K rows / dS instead
  static constexpr size_t SMEM = R12 + R3 + STAT + R4;
  static constexpr int NJ = 64 * (D / 4) / NT;                // float4 per thread per 64-row tile
  static constexpr size_t SLAB = (size_t)2 * D * (D + 1) + (FULLW ? (size_t)3 * D * D : 0);
};

// The weight fragments of a column tile's image that a linear holds BEFORE it starts: k-blocks 0 .. WN - 1 (issued one phase ahead: their L2 round trip is
// off the phase).  fp32 mode (three planes, registers are scarce): the first k-block, the rest streams one k-block ahead of its MFMAs; bf16 mode (one
// plane): ALL k-blocks -- a phase of 16 MFMAs per wave was four dependent L2 round trips long otherwise.
template <int NP, int WN>
struct WFrag { uint4 v[WN][NP]; };
template <int NP, int WN>
__device__ __forceinline__ void wload(const uint4* img, WFrag<NP, WN>& w) {
#pragma unroll
  for (int kb = 0; kb < WN; ++kb)
#pragma unroll
    for (int q = 0; q < NP; ++q) w.v[kb][q] = img[(kb * 3 + q) * 64];
}

// acc[rt] (lane (p, j): row (rt0 + rt) * 16 + p, columns ct * 16 + 4j .. 4j+3) += planes[row][:] . W[column][:]; img = the image of column tile ct (+ lane),
// w0 = its first WN k-blocks' fragments (wload); NEXT: the fragments of the image `nimg` are fetched meanwhile into wn (WN = all k-blocks: at the start of
// this linear, otherwise under its last k-block)
template <int D, int NP, int RT, bool NEXT, int LD = D + 8, int WN = 1>
__device__ __forceinline__ void lin(const __bf16* pl, const uint4* img, const WFrag<NP, WN>& w0, int rt0, int p, int j, f32x4 (&acc)[RT], const uint4* nimg,
                                    WFrag<NP, WN>& wn) {
  constexpr int KB = D / 32, LDP = LD, PLANE = 64 * (D + 8);
  const __bf16* frag = pl + (rt0 * 16 + p) * LDP + 8 * j;
  if (NEXT && WN == KB) wload<NP, WN>(nimg, wn);
  if constexpr (NP == 1 && WN == KB) {
    // one plane, every weight fragment in registers: ALL activation fragments are requested before the first product (16 b128 reads in flight instead
    // of a read -> wait -> MFMA chain per fragment: the phase was LDS-latency-bound)
    bf16x8 af[KB][RT];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) af[kb][rt] = *reinterpret_cast<const bf16x8*>(frag + rt * 16 * LDP + kb * 32);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w0.v[kb][0]), af[kb][rt], acc[rt], 0, 0, 0);
    return;
  }
  uint4 bw[2][NP];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    if (kb + 1 >= WN && kb + 1 < KB) {
#pragma unroll
      for (int q = 0; q < NP; ++q) bw[(kb + 1) & 1][q] = img[((kb + 1) * 3 + q) * 64];
    }
    if (NEXT && WN != KB && kb == KB - 1) wload<NP, WN>(nimg, wn);
    __builtin_amdgcn_sched_barrier(0);      // (pins the requests here: the scheduler otherwise sinks them behind this block's MFMAs -- no lookahead at all)
    uint4 cw[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) cw[q] = kb < WN ? w0.v[kb < WN ? kb : 0][q] : bw[kb & 1][q];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const __bf16* fp = frag + rt * 16 * LDP + kb * 32;
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(fp);
      const bf16x8 am = NP == 3 ? *reinterpret_cast<const bf16x8*>(fp + PLANE) : ah;
      const bf16x8 al = NP == 3 ? *reinterpret_cast<const bf16x8*>(fp + 2 * PLANE) : ah;
      acc[rt] = mma<NP>(__builtin_bit_cast(bf16x8, cw[0]), __builtin_bit_cast(bf16x8, cw[NP == 3 ? 1 : 0]), __builtin_bit_cast(bf16x8, cw[NP == 3 ? 2 : 0]), ah, am, al,
                        acc[rt]);
    }
  }
}

// acc[kt] (lane (p, j): dW[nt * 16 + 4j + r][(kt0 + kt) * 16 + p]) += sum over the 64 rows of Y[row][nt * 16 + .] X[row][(kt0 + kt) * 16 + .];
// dbp += this lane's share of colsum(Y[:, nt * 16 + p]) (rows 8j .. 8j+7 of every 32-row block)
template <int D, int NP, int KTL, int LDY = D + 8, int LDX = D + 8, bool P8 = false>
__device__ __forceinline__ void wgrad(const __bf16* Y, const __bf16* X, int nt, int kt0, int p, int j, f32x4 (&acc)[KTL], float& dbp) {
  constexpr int PLANE = 64 * (D + 8);
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const bf16x8 yh = tr_frag<LDY, P8>(Y, kb, nt, p, j);
    const bf16x8 ym = NP == 3 ? tr_frag<LDY, P8>(Y + PLANE, kb, nt, p, j) : yh;
    const bf16x8 yl = NP == 3 ? tr_frag<LDY, P8>(Y + 2 * PLANE, kb, nt, p, j) : yh;
    dbp += NP == 3 ? (sum8(yl) + sum8(ym)) + sum8(yh) : sum8(yh);
    if constexpr (NP == 1) {      // all of the k-block's X fragments in flight before the first product
      bf16x8 xf[KTL];
#pragma unroll
      for (int kt = 0; kt < KTL; ++kt) xf[kt] = tr_frag<LDX, P8>(X, kb, kt0 + kt, p, j);
#pragma unroll
      for (int kt = 0; kt < KTL; ++kt) acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yh, xf[kt], acc[kt], 0, 0, 0);
      continue;
    }
#pragma unroll
    for (int kt = 0; kt < KTL; ++kt) {
      const bf16x8 xh = tr_frag<LDX, P8>(X, kb, kt0 + kt, p, j);
      const bf16x8 xm = NP == 3 ? tr_frag<LDX, P8>(X + PLANE, kb, kt0 + kt, p, j) : xh;
      const bf16x8 xl = NP == 3 ? tr_frag<LDX, P8>(X + 2 * PLANE, kb, kt0 + kt, p, j) : xh;
      acc[kt] = mma<NP>(yh, ym, yl, xh, xm, xl, acc[kt]);
    }
  }
}

// 16-deep group of an attention product: four exact fp32 MFMAs, or (bf16 mode) one bf16 MFMA of the rounded operands (same element order)
template <int NP>
__device__ __forceinline__ f32x4 amma(const f32x4& a, const f32x4& b, f32x4 c) {
  if (NP == 1) return mma4_bf16(to_bf16x4(a), to_bf16x4(b), c);
  return mma4(a, b, c);
}

// IO16 (bf16 mode only): A is read and dQKV written as bf16 arrays
template <int D, int DK, int NP, bool IO16>
__global__ __launch_bounds__(512, 2) void tower_bwd_fused_kernel(TowerBwdArgs a) {
  using C = BwdCfg<D, NP>;
  constexpr int NT = C::NT, CTW = C::CTW, RT = C::RT, WPN = C::WPN, KTL = C::KTL, LDP = C::LDP, PLANE = C::PLANE, LQ = C::LQ, DSP = C::DSP, NJ = C::NJ;
  constexpr int HEADS = D / DK, DQ = DK / 64, KBT = 4;
  // the next session's A / dZ tiles are requested under the last attention pass where the registers allow it; the 128-wide fp32 form (64 registers of
  // weight-gradient accumulators per wave) has none to spare there -- a spilled register's reload would queue behind the tile loads (vmcnt is in
  // order) -- and requests them in front of the dQKV copy-out instead
  constexpr bool EARLY = NP == 1 || D == 64;
  constexpr bool WPF = EARLY;      // ... likewise the next phase's first weight fragments under the current phase's last k-block
  static_assert(DK == 64 || DK == 128, "head dim 64 / 128");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* pA = reinterpret_cast<__bf16*>(smem_raw);                          // P(A); later P(X)
  constexpr bool PA = C::PA;
  constexpr int LDA = C::LDA;
  constexpr int WN = NP == 1 ? C::KB : 1;      // k-blocks of a linear's weight fragments held ahead of it (lin)
  using WF = WFrag<NP, WN>;
  __bf16* pR = reinterpret_cast<__bf16*>(smem_raw + (PA ? C::SLOT : C::P3));                // P(R1) -> P(dF1)
  __bf16* pZ = reinterpret_cast<__bf16*>(smem_raw + (PA ? 2 * C::SLOT : C::R12));           // P(dZ)
  // PA slots: S0 = P(A) -> P(X), S1 = P(R1 / dF1) -> K, S2 = P(dZ) -> dA -> dK, S3 = Q -> dQ, S4 = V -> dV; dS image behind them
  __bf16* const S0 = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* const S1 = reinterpret_cast<__bf16*>(smem_raw + C::SLOT);
  __bf16* const S2 = reinterpret_cast<__bf16*>(smem_raw + 2 * C::SLOT);
  __bf16* const S3 = reinterpret_cast<__bf16*>(smem_raw + 3 * C::SLOT);
  __bf16* const S4 = reinterpret_cast<__bf16*>(smem_raw + 4 * C::SLOT);
  __bf16* const dSi = reinterpret_cast<__bf16*>(smem_raw + 5 * C::SLOT);
  float* fQ = reinterpret_cast<float*>(smem_raw);                            // fp32 rows Q | K | V over P(X) / P(dF1)
  float* fK = fQ + 64 * LQ;
  float* fV = fK + 64 * LQ;
  float* fdA = reinterpret_cast<float*>(smem_raw + C::R12);                  // fp32 rows dA over P(dZ)
  float* dSb = fdA + 64 * LQ;
  float* lse2 = reinterpret_cast<float*>(smem_raw + C::R12 + C::R3);
  float* dpart = lse2 + 2 * 64;      // [4 key tiles][64 queries]: shares of delta
  constexpr bool FULLX = C::FULLX, FULLW = C::FULLW;
  __bf16* pX2 = NP == 1 ? reinterpret_cast<__bf16*>(fK) : reinterpret_cast<__bf16*>(smem_raw + C::R12 + C::R3 + C::STAT);      // P(X) of the tail
  __bf16* pD = NP == 1 ? reinterpret_cast<__bf16*>(dSb) : pX2 + NP * PLANE;                                                      // P(dQ), P(dK), P(dV) in turn
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = a.L, NTL = (L + 15) >> 4;
  const float scale = 1.0f / sqrtf((float)DK);
  const float c2 = scale * 1.4426950408889634f;
  // linears: wave = column tile ct, row tiles rt0 .. rt0 + RT - 1
  const int ct = wave % CTW, rt0 = (wave / CTW) * RT;
  // weight gradients: wave = rows nt * 16 .. of dW, column tiles kt0 .. kt0 + KTL - 1
  const int wnt = wave / WPN, wkt0 = (wave % WPN) * KTL;
  // tile staging: float4 #i of a 64-row tile = (row i / (D/4), 4 * (i % (D/4)))
  // (recomputed from a laundered thread id wherever a tile is loaded / stored: hoisted, the 64-bit addresses of three arrays x NJ pieces would be spilled)
  auto tile_rc = [&](int jj, int& row, int& colx) {
    int t = tid;
    asm volatile("" : "+v"(t));
    const int i = t + NT * jj;
    row = i / (D / 4);
    colx = (i - row * (D / 4)) * 4;
  };
  // A tile's loads are issued long before their values are needed, so nothing here may LOOK at a loaded value (a select on it would make the wave
  // wait for the load where it is issued): rows are clamped to the list, the raw 16 (bf16 arrays: 8) bytes stay in registers, and padding rows are
  // zeroed / bf16 values widened when the tile is stored to LDS
  auto load_tile = [&](const float* src, int b, auto is16, f32x4 (&v)[NJ]) {
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      int tr, tc;
      tile_rc(jj, tr, tc);
      const int row = min(tr, L - 1);
      const size_t off = ((size_t)b * L + row) * D + tc;
      if constexpr (decltype(is16)::value) {
        const f32x2 u = *reinterpret_cast<const f32x2*>(reinterpret_cast<const __bf16*>(src) + off);
        v[jj][0] = u[0];
        v[jj][1] = u[1];
      } else {
        v[jj] = *reinterpret_cast<const f32x4*>(src + off);
      }
    }
  };
  auto store_tile = [&](__bf16* dst, auto is16, const f32x4 (&v)[NJ], auto pitch) {
    constexpr int LDT = decltype(pitch)::value;
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      int tr, tc;
      tile_rc(jj, tr, tc);
      f32x4 x = v[jj];
      if constexpr (decltype(is16)::value) {
        const bf16x4 h = __builtin_bit_cast(bf16x4, f32x2{v[jj][0], v[jj][1]});
        x = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
      }
      if (tr >= L) x = f32x4{0.f, 0.f, 0.f, 0.f};
      store4<NP, PLANE>(dst + tr * LDT + tc, x);
    }
  };
  using PitchP = std::integral_constant<int, LDP>;
  using PitchA = std::integral_constant<int, LDA>;

  f32x4 accW2[KTL], accW1[KTL];
#pragma unroll
  for (int kt = 0; kt < KTL; ++kt) accW2[kt] = accW1[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float db2p = 0.f, db1p = 0.f;
  f32x4 accWqkv[FULLW ? 3 : 1][KTL];
#pragma unroll
  for (int c = 0; c < (FULLW ? 3 : 1); ++c)
#pragma unroll
    for (int kt = 0; kt < KTL; ++kt) accWqkv[c][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
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
  // the session's A / dZ tiles and log-sum-exp travel during the previous session's last phase
  f32x4 va[NJ], vz[NJ];
  float lsev = 0.f;
  auto load_session = [&](int b) {
    load_tile(a.A, b, std::integral_constant<bool, IO16>{}, va);
    load_tile(a.dZ, b, std::false_type{}, vz);
    if (tid < 64 * HEADS) {
      const int h = tid >> 6, q = min(tid & 63, L - 1);
      lsev = a.LSE[((size_t)b * HEADS + h) * L + q];
    }
  };
  if ((int)blockIdx.x < a.B) load_session(blockIdx.x);

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    // The LDS / image addresses of a session's phases are a few hundred distinct values of (lane, wave); hoisted out of the session loop they would
    // all be spilled.  Laundering the lane id once per session keeps every address a couple of VALU instructions next to its use.
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int j = lane >> 4, p = lane & 15;
    const int col = ct * 16 + 4 * j;                    // this lane's four columns in every linear's epilogue
    const size_t wtile = ((size_t)ct * KBT * 3) * 64 + lane;
    // ---- P0: A and dZ -> planes; the forward's log-sum-exp (base 2; +inf for padded queries: P = 0)
    WF wf;
    wload<NP, WN>(launder(a.W1) + wtile, wf);
    {
      if (tid < 64 * HEADS) lse2[tid] = (tid & 63) < L ? lsev * 1.4426950408889634f : INFINITY;
      store_tile(pA, std::integral_constant<bool, IO16>{}, va, PitchP{});
      store_tile(pZ, std::false_type{}, vz, PitchP{});
    }
    lds_barrier();
    mark(0);
    // ---- P1: R1 = relu(A W1^T + b1) -> planes
    {
      f32x4 acc[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 bias = *reinterpret_cast<const f32x4*>(a.b1 + col);
      WF wn;
      lin<D, NP, RT, WPF, LDP, WN>(pA, a.W1 + wtile, wf, rt0, p, j, acc, launder(a.W2T) + wtile, wn);
      if (WPF) {
wf = wn;
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        f32x4 x = acc[rt] + bias;
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = fmaxf(x[r], 0.f);
        store4<NP, PLANE>(pR + ((rt0 + rt) * 16 + p) * LDP + col, x);
      }
    }
    lds_barrier();
    mark(1);
    // ---- P2: dW2 += dZ^T R1, db2;  dF1 = (dZ W2) * [R1 > 0] -> planes over R1
    {
      float dbp = 0.f;
      if (!WPF) wload<NP, WN>(launder(a.W2T) + wtile, wf);
      wgrad<D, NP, KTL>(pZ, pR, wnt, wkt0, p, j, accW2, dbp);
      if (wkt0 == 0) db2p += dbp;
      f32x4 acc[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      WF wn;
      lin<D, NP, RT, WPF, LDP, WN>(pZ, a.W2T + wtile, wf, rt0, p, j, acc, launder(a.W1T) + wtile, wn);
      if (WPF) {
wf = wn;
      }
      lds_barrier();      // every transposed read of R1 is done
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        __bf16* dst = pR + ((rt0 + rt) * 16 + p) * LDP + col;
        const bf16x4 hv = *reinterpret_cast<const bf16x4*>(dst);      // high plane of R1: > 0 exactly where R1 > 0
        f32x4 x = acc[rt];
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = (float)hv[r] > 0.f ? x[r] : 0.f;
        store4<NP, PLANE>(dst, x);
      }
    }
    lds_barrier();
    mark(2);
    // ---- P3: dW1 += dF1^T A, db1;  dA = dF1 W1 -> fp32 rows (over dZ);  the X tile of P4 travels meanwhile
    {
      f32x4 vx[NJ];
      load_tile(a.X, b, std::false_type{}, vx);
      float dbp = 0.f;
      if (!WPF) wload<NP, WN>(launder(a.W1T) + wtile, wf);
      wgrad<D, NP, KTL>(pR, pA, wnt, wkt0, p, j, accW1, dbp);
      if (wkt0 == 0) db1p += dbp;
      f32x4 acc[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      WF wn;
      lin<D, NP, RT, WPF, LDP, WN>(pR, a.W1T + wtile, wf, rt0, p, j, acc, launder(a.Wqkv) + wtile, wn);
      if (WPF) {
wf = wn;
      }
      lds_barrier();      // every read of P(A), P(dF1), P(dZ) is done
      if constexpr (PA) {      // dA as a bf16 image (its consumers round it anyway) and X in the attention pitch: both stay for the tail
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const f32x4& v = acc[rt];
          *reinterpret_cast<bf16x4*>(S2 + ((rt0 + rt) * 16 + p) * LDA + col) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        }
        store_tile(S0, std::false_type{}, vx, PitchA{});
      } else {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) *reinterpret_cast<f32x4*>(fdA + ((rt0 + rt) * 16 + p) * LQ + col) = acc[rt];
        store_tile(pA, std::false_type{}, vx, PitchP{});      // P(X)
      }
    }
    lds_barrier();
    mark(3);
    if constexpr (PA) {
    // ======== bf16 mode: P4 / P5 / tail on bf16 images, every product a v_mfma_f32_16x16x32_bf16 ========
    // ---- P4: [Q | K | V] = X Wqkv^T -> bf16 images (X stays in S0: nothing waits for its readers)
    {
      if (!WPF) wload<NP, WN>(launder(a.Wqkv) + wtile, wf);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        f32x4 acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const uint4* img = a.Wqkv + ((size_t)(c * CTW + ct) * KBT * 3) * 64 + lane;
        WF wn;
        if (c < 2) lin<D, NP, RT, true, LDA, WN>(S0, img, wf, rt0, p, j, acc, img + ((size_t)CTW * KBT * 3) * 64, wn);
        else lin<D, NP, RT, false, LDA, WN>(S0, img, wf, rt0, p, j, acc, img, wn);
        if (c < 2) {
wf = wn;
        }
        __bf16* dst = c == 0 ? S3 : (c == 1 ? S1 : S4);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const f32x4& v = acc[rt];
          *reinterpret_cast<bf16x4*>(dst + ((rt0 + rt) * 16 + p) * LDA + col) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        }
      }
    }
    lds_barrier();
    mark(4);
    // ---- P5: attention backward.  Pass 1: wave = (16 keys, half of the head's 16-dim tiles): S and dP for all queries (both halves compute them: a
    // handful of MFMAs -- cheaper than summing partial dK / dV across waves), P, delta, dS; dV / dK of its dim tiles with the probabilities / dS
    // values going from the accumulators straight into the B operand (tr_frag<.., true> delivers the A operand's rows in the accumulators' order).
    // Pass 2: wave = (16 queries, half of the dim tiles): dQ = dS K.  Outputs over the dead inputs: dQ -> Q's slot, dK -> dA's, dV -> V's.
#pragma unroll
    for (int h = 0; h < HEADS; ++h) {
      const int hc = h * DK;
      constexpr int NDT = DK / 16, HDT = NDT / 2;
      const int kt = wave & 3, dh = wave >> 2;
      f32x4 st[4], dp[4], dv[HDT], dk[HDT];
#pragma unroll
      for (int i = 0; i < 4; ++i) st[i] = dp[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < HDT; ++i) dv[i] = dk[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kt < NTL) {
#pragma unroll
        for (int kb = 0; kb < DK / 32; ++kb) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(S1 + (kt * 16 + p) * LDA + hc + 32 * kb + 8 * j);
          const bf16x8 vf = *reinterpret_cast<const bf16x8*>(S4 + (kt * 16 + p) * LDA + hc + 32 * kb + 8 * j);
          bf16x8 qf[4], df[4];      // the k-block's ten fragments are requested together, then the eight products
#pragma unroll
          for (int qt = 0; qt < 4; ++qt) {
            const int row = (qt < NTL ? qt : 0) * 16 + p;
            qf[qt] = *reinterpret_cast<const bf16x8*>(S3 + row * LDA + hc + 32 * kb + 8 * j);
            df[qt] = *reinterpret_cast<const bf16x8*>(S2 + row * LDA + hc + 32 * kb + 8 * j);
          }
#pragma unroll
          for (int qt = 0; qt < 4; ++qt) {
            if (qt < NTL) {
              st[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[qt], kf, st[qt], 0, 0, 0);      // S[query qt*16 + 4j + r][key kt*16 + p]
              dp[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df[qt], vf, dp[qt], 0, 0, 0);
            }
          }
        }
        const bool keyok = kt * 16 + p < L;
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
          const int q0 = qt * 16 + 4 * j;
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse2 + h * 64 + q0);
          f32x4 part;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = keyok ? __builtin_amdgcn_exp2f(__builtin_fmaf(st[qt][r], c2, -l4[r])) : 0.f;
            st[qt][r] = pv;
            part[r] = row16_sum(pv * dp[qt][r]);
          }
          if (p == 0 && dh == 0) *reinterpret_cast<f32x4*>(dpart + kt * 64 + q0) = part;
        }
      }
      lds_barrier();      // the key tiles' shares of delta
      if (kt < NTL) {
        // dS image: row = query, the 64 keys of a row stored in the order the pass-2 B operand wants them (key 32 a + 16 b + 4 c + r -> column 32 a + 8 c + 4 b + r)
        const int key = kt * 16 + p;
        const int kcol = 32 * (key >> 5) + 8 * ((key >> 2) & 3) + 4 * ((key >> 4) & 1) + (key & 3);
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
          const int q0 = qt * 16 + 4 * j;
          f32x4 d4 = *reinterpret_cast<const f32x4*>(dpart + q0);
#pragma unroll
          for (int k2 = 1; k2 < 4; ++k2)
            if (k2 < NTL) d4 += *reinterpret_cast<const f32x4*>(dpart + k2 * 64 + q0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float ds = st[qt][r] * (dp[qt][r] - d4[r]) * scale;
            dp[qt][r] = ds;
            if (dh == 0) dSi[(q0 + r) * 72 + kcol] = (__bf16)ds;
          }
        }
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2) {
          if (2 * kb2 < NTL) {
            const bf16x8 pb = bf16x8{(__bf16)st[2 * kb2][0], (__bf16)st[2 * kb2][1], (__bf16)st[2 * kb2][2], (__bf16)st[2 * kb2][3],
                                     (__bf16)st[2 * kb2 + 1][0], (__bf16)st[2 * kb2 + 1][1], (__bf16)st[2 * kb2 + 1][2], (__bf16)st[2 * kb2 + 1][3]};
            const bf16x8 db = bf16x8{(__bf16)dp[2 * kb2][0], (__bf16)dp[2 * kb2][1], (__bf16)dp[2 * kb2][2], (__bf16)dp[2 * kb2][3],
                                     (__bf16)dp[2 * kb2 + 1][0], (__bf16)dp[2 * kb2 + 1][1], (__bf16)dp[2 * kb2 + 1][2], (__bf16)dp[2 * kb2 + 1][3]};
            bf16x8 fa[HDT], fq[HDT];
#pragma unroll
            for (int dt = 0; dt < HDT; ++dt) {
              const int ctile = hc / 16 + dh * HDT + dt;
              fa[dt] = tr_frag<LDA, true>(S2, kb2, ctile, p, j);
              fq[dt] = tr_frag<LDA, true>(S3, kb2, ctile, p, j);
            }
#pragma unroll
            for (int dt = 0; dt < HDT; ++dt) {
              dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[dt], pb, dv[dt], 0, 0, 0);      // dV[key][dim] += P^T dA
              dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq[dt], db, dk[dt], 0, 0, 0);      // dK[key][dim] += dS^T Q
            }
          }
        }
      } else if (kt == NTL && (NTL & 1) && dh == 0) {
        // an odd number of key tiles: pass 2 reads the dS image in 32-key blocks, so the absent tile's 16 columns must hold zeros (whatever the
        // slot held before may be a NaN pattern, and NaN x K's zero rows is NaN)
        const int key = kt * 16 + p;
        const int kcol = 32 * (key >> 5) + 8 * ((key >> 2) & 3) + 4 * ((key >> 4) & 1) + (key & 3);
#pragma unroll
        for (int qt = 0; qt < 4; ++qt)
#pragma unroll
          for (int r = 0; r < 4; ++r) dSi[(qt * 16 + 4 * j + r) * 72 + kcol] = (__bf16)0.f;
      }
      lds_barrier();      // dS complete; this head's reads of Q / dA / V done
      mark(5);
      if (kt < NTL) {
#pragma unroll
        for (int dt = 0; dt < HDT; ++dt) {
          const int off = (kt * 16 + p) * LDA + hc + (dh * HDT + dt) * 16 + 4 * j;
          *reinterpret_cast<bf16x4*>(S4 + off) = bf16x4{(__bf16)dv[dt][0], (__bf16)dv[dt][1], (__bf16)dv[dt][2], (__bf16)dv[dt][3]};
          *reinterpret_cast<bf16x4*>(S2 + off) = bf16x4{(__bf16)dk[dt][0], (__bf16)dk[dt][1], (__bf16)dk[dt][2], (__bf16)dk[dt][3]};
        }
      }
      mark(6);
      if (h == HEADS - 1) load_session(min(b + (int)gridDim.x, a.B - 1));      // the next session's tiles travel under the last pass
      {
        const int qt = wave & 3;
        if (qt < NTL) {
          f32x4 o[HDT];
#pragma unroll
          for (int dt = 0; dt < HDT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kb2 = 0; kb2 < 2; ++kb2) {
            if (2 * kb2 < NTL) {
              const bf16x8 dsf = *reinterpret_cast<const bf16x8*>(dSi + (qt * 16 + p) * 72 + 32 * kb2 + 8 * j);
              bf16x8 fk[HDT];
#pragma unroll
              for (int dt = 0; dt < HDT; ++dt) fk[dt] = tr_frag<LDA, true>(S1, kb2, hc / 16 + dh * HDT + dt, p, j);
#pragma unroll
              for (int dt = 0; dt < HDT; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[dt], dsf, o[dt], 0, 0, 0);      // dQ[query][dim] += dS K
            }
          }
#pragma unroll
          for (int dt = 0; dt < HDT; ++dt)
            *reinterpret_cast<bf16x4*>(S3 + (qt * 16 + p) * LDA + hc + (dh * HDT + dt) * 16 + 4 * j) = bf16x4{(__bf16)o[dt][0], (__bf16)o[dt][1], (__bf16)o[dt][2], (__bf16)o[dt][3]};
        }
      }
      lds_barrier();      // pass 2's reads of dS / K done; dQ / dK / dV of this head complete
      mark(7);
    }
    // ---- dQKV rows (only where the q/k/v weight gradient stays outside): [dQ | dK | dV] from their slots, 16 bytes of bf16 per lane along the row
    if constexpr (!FULLW) {
      for (int i = tid; i < L * (D / 8); i += NT) {
        const int row = i / (D / 8), c8 = (i - row * (D / 8)) * 8;
        const bf16x8 q8 = *reinterpret_cast<const bf16x8*>(S3 + row * LDA + c8);
        const bf16x8 k8 = *reinterpret_cast<const bf16x8*>(S2 + row * LDA + c8);
        const bf16x8 v8 = *reinterpret_cast<const bf16x8*>(S4 + row * LDA + c8);
        const size_t g = ((size_t)b * L + row) * (3 * D) + c8;
        if constexpr (IO16) {
          __bf16* o16 = reinterpret_cast<__bf16*>(a.dQKV);
          *reinterpret_cast<bf16x8*>(o16 + g) = q8;
          *reinterpret_cast<bf16x8*>(o16 + g + D) = k8;
          *reinterpret_cast<bf16x8*>(o16 + g + 2 * D) = v8;
        } else {
#pragma unroll
          for (int hlf = 0; hlf < 2; ++hlf) {
            *reinterpret_cast<f32x4*>(a.dQKV + g + 4 * hlf) = f32x4{(float)q8[4 * hlf], (float)q8[4 * hlf + 1], (float)q8[4 * hlf + 2], (float)q8[4 * hlf + 3]};
            *reinterpret_cast<f32x4*>(a.dQKV + g + D + 4 * hlf) = f32x4{(float)k8[4 * hlf], (float)k8[4 * hlf + 1], (float)k8[4 * hlf + 2], (float)k8[4 * hlf + 3]};
            *reinterpret_cast<f32x4*>(a.dQKV + g + 2 * D + 4 * hlf) = f32x4{(float)v8[4 * hlf], (float)v8[4 * hlf + 1], (float)v8[4 * hlf + 2], (float)v8[4 * hlf + 3]};
          }
        }
      }
    }
    // ---- tail: dX = dQ Wq + dK Wk + dV Wv + dZ and (FULLW) dWq / dWk / dWv += dQ^T X, dK^T X, dV^T X straight from the images (X is still in S0)
    {
      constexpr int KBT3 = 4 * ((3 * D + 127) / 128), KB = D / 32;
      const uint4* imgT = launder(a.WqkvT) + ((size_t)ct * KBT3 * 3) * 64 + lane;
      WF wq;
      wload<NP, WN>(imgT, wq);
      f32x4 accX[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) accX[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const __bf16* src = c == 0 ? S3 : (c == 1 ? S2 : S4);
        WF wn;
        if (c < 2) lin<D, NP, RT, true, LDA, WN>(src, imgT + ((size_t)c * KB * 3) * 64, wq, rt0, p, j, accX, imgT + ((size_t)(c + 1) * KB * 3) * 64, wn);
        else lin<D, NP, RT, false, LDA, WN>(src, imgT + ((size_t)c * KB * 3) * 64, wq, rt0, p, j, accX, imgT, wn);
        if (c < 2) {
wq = wn;
        }
        if constexpr (FULLW) {
          float nodb = 0.f;
          wgrad<D, NP, KTL, LDA, LDA, true>(src, S0, wnt, wkt0, p, j, accWqkv[c], nodb);
        }
      }
      lds_barrier();      // every read of the three gradient images (and of K) is done: dX is staged as fp32 rows over the K / dK slots
      float* stage = reinterpret_cast<float*>(S1);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) *reinterpret_cast<f32x4*>(stage + ((rt0 + rt) * 16 + p) * LQ + col) = accX[rt];
      lds_barrier();
      for (int i = tid; i < L * (D / 4); i += NT) {
        const int row = i / (D / 4), c4 = (i - row * (D / 4)) * 4;
        const size_t g = ((size_t)b * L + row) * D + c4;
        *reinterpret_cast<f32x4*>(a.dX + g) = *reinterpret_cast<const f32x4*>(stage + row * LQ + c4) + *reinterpret_cast<const f32x4*>(a.dZ + g);
      }
    }
    } else {
    // ---- P4: [Q | K | V] = X Wqkv^T -> fp32 rows
    {
      if (!WPF) wload<NP, WN>(launder(a.Wqkv) + wtile, wf);
      f32x4 acc[3][RT];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[c][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const uint4* img = a.Wqkv + ((size_t)(c * CTW + ct) * KBT * 3) * 64 + lane;
        WF wn;
        if (c < 2) lin<D, NP, RT, true, LDP, WN>(pA, img, wf, rt0, p, j, acc[c], img + ((size_t)CTW * KBT * 3) * 64, wn);
        else lin<D, NP, RT, false, LDP, WN>(pA, img, wf, rt0, p, j, acc[c], img, wn);
        if (c < 2) {
wf = wn;
        }
      }
      lds_barrier();      // every read of P(X) is done
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) *reinterpret_cast<f32x4*>(fQ + c * 64 * LQ + ((rt0 + rt) * 16 + p) * LQ + col) = acc[c][rt];
    }
    lds_barrier();
    mark(4);
    // ---- P5: attention backward, head by head.  dQ is left in the dead V rows, dK in the dead Q rows, dV in the dead dA rows (fp32), and the
    // three matrices leave together as whole rows of dQKV at the end
#pragma unroll
    for (int h = 0; h < HEADS; ++h) {
      const int hc = h * DK;
      const int kt = wave & 3, qh = wave >> 2;
      f32x4 dk[DQ * 4], dv[DQ * 4];
#pragma unroll
      for (int i = 0; i < DQ * 4; ++i) dk[i] = dv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 st[2], dp[2];
      st[0] = st[1] = dp[0] = dp[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kt < NTL) {      // (wave-uniform; a key tile of padding only has nothing to add)
        const float* Kp = fK + (kt * 16 + p) * LQ + hc + 4 * j;
        const float* Vp = fV + (kt * 16 + p) * LQ + hc + 4 * j;
#pragma unroll
        for (int g = 0; g < DK / 16; ++g) {
          const f32x4 kf = *reinterpret_cast<const f32x4*>(Kp + 16 * g);
          const f32x4 vf = *reinterpret_cast<const f32x4*>(Vp + 16 * g);
#pragma unroll
          for (int qi = 0; qi < 2; ++qi) {
            const int qt = 2 * qh + qi;
            if (qt < NTL) {
              const f32x4 qf = *reinterpret_cast<const f32x4*>(fQ + (qt * 16 + p) * LQ + hc + 4 * j + 16 * g);
              const f32x4 df = *reinterpret_cast<const f32x4*>(fdA + (qt * 16 + p) * LQ + hc + 4 * j + 16 * g);
              st[qi] = amma<NP>(qf, kf, st[qi]);      // S[query qt*16 + 4j + r][key kt*16 + p]
              dp[qi] = amma<NP>(df, vf, dp[qi]);      // dP, same layout
            }
          }
        }
        // P from the forward's log-sum-exp; this key tile's share of delta[q] = sum_k P[q][k] dP[q][k] (the SAME P and dP the products below use: the rows
        // of dS then sum to zero exactly, whatever the arithmetic mode rounded on the way) -- summed over the 16 keys = the 16 lanes of a DPP row
        const bool keyok = kt * 16 + p < L;
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
          const int q0 = (2 * qh + qi) * 16 + 4 * j;
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse2 + h * 64 + q0);
          f32x4 part;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = keyok ? __builtin_amdgcn_exp2f(__builtin_fmaf(st[qi][r], c2, -l4[r])) : 0.f;
            st[qi][r] = pv;
            part[r] = row16_sum(pv * dp[qi][r]);
          }
          if (p == 0) *reinterpret_cast<f32x4*>(dpart + kt * 64 + q0) = part;
        }
      }
      lds_barrier();      // the key tiles' shares of delta
      if (kt < NTL) {
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
          const int q0 = (2 * qh + qi) * 16 + 4 * j;
          f32x4 d4 = *reinterpret_cast<const f32x4*>(dpart + q0);
#pragma unroll
          for (int k2 = 1; k2 < 4; ++k2)
            if (k2 < NTL) d4 += *reinterpret_cast<const f32x4*>(dpart + k2 * 64 + q0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float ds = st[qi][r] * (dp[qi][r] - d4[r]) * scale;
            dp[qi][r] = ds;
            dSb[(q0 + r) * DSP + kt * 16 + p] = ds;
          }
        }
        // dV[key][dim] += P^T dA, dK[key][dim] += dS^T Q: the row operand read as one b128 along the head dim feeds four MFMAs whose
        // output row i means dim 4 i + t (tower.hip, phase 2)
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
          const int qt = 2 * qh + qi;
          if (qt >= NTL) continue;
#pragma unroll
          for (int dq = 0; dq < DQ; ++dq) {
            {
              f32x4 vv[4];
#pragma unroll
              for (int s = 0; s < 4; ++s) vv[s] = *reinterpret_cast<const f32x4*>(fdA + (qt * 16 + 4 * j + s) * LQ + hc + dq * 64 + 4 * p);
#pragma unroll
              for (int t = 0; t < 4; ++t) dv[dq * 4 + t] = amma<NP>(f32x4{vv[0][t], vv[1][t], vv[2][t], vv[3][t]}, st[qi], dv[dq * 4 + t]);
            }
            __builtin_amdgcn_sched_barrier(0);      // (register budget: one operand block at a time)
            {
              f32x4 ww[4];
#pragma unroll
              for (int s = 0; s < 4; ++s) ww[s] = *reinterpret_cast<const f32x4*>(fQ + (qt * 16 + 4 * j + s) * LQ + hc + dq * 64 + 4 * p);
#pragma unroll
              for (int t = 0; t < 4; ++t) dk[dq * 4 + t] = amma<NP>(f32x4{ww[0][t], ww[1][t], ww[2][t], ww[3][t]}, dp[qi], dk[dq * 4 + t]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      lds_barrier();      // dS complete; this head's reads of Q / dA / V done
      mark(5);
      if (qh == 1 && kt < NTL) {      // park the second query half's partials in the head's dead Q / dA columns (rows = keys)
#pragma unroll
        for (int dq = 0; dq < DQ; ++dq)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int off = (kt * 16 + p) * LQ + hc + dq * 64 + 16 * j + 4 * r;
            *reinterpret_cast<f32x4*>(fQ + off) = f32x4{dk[dq * 4 + 0][r], dk[dq * 4 + 1][r], dk[dq * 4 + 2][r], dk[dq * 4 + 3][r]};
            *reinterpret_cast<f32x4*>(fdA + off) = f32x4{dv[dq * 4 + 0][r], dv[dq * 4 + 1][r], dv[dq * 4 + 2][r], dv[dq * 4 + 3][r]};
          }
      }
      lds_barrier();      // the parked partials are visible
      mark(6);
      if (qh == 0 && kt < NTL) {      // dK / dV of this key tile = both halves, left where the partials were parked
#pragma unroll
        for (int dq = 0; dq < DQ; ++dq)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int off = (kt * 16 + p) * LQ + hc + dq * 64 + 16 * j + 4 * r;
            const f32x4 k4 = f32x4{dk[dq * 4 + 0][r], dk[dq * 4 + 1][r], dk[dq * 4 + 2][r], dk[dq * 4 + 3][r]} + *reinterpret_cast<const f32x4*>(fQ + off);
            const f32x4 v4 = f32x4{dv[dq * 4 + 0][r], dv[dq * 4 + 1][r], dv[dq * 4 + 2][r], dv[dq * 4 + 3][r]} + *reinterpret_cast<const f32x4*>(fdA + off);
            *reinterpret_cast<f32x4*>(fQ + off) = k4;
            *reinterpret_cast<f32x4*>(fdA + off) = v4;
          }
      }
      if (EARLY && h == HEADS - 1) load_session(min(b + (int)gridDim.x, a.B - 1));      // the next session's tiles travel under the last pass (the last session re-reads itself: no stale registers to keep)
      {   // pass 2: dQ = dS K -> the dead V rows; wave = (query tile, half of the head dim: DK = 128 a 64-dim block, DK = 64 the dims 4 i + {2 sub, 2 sub + 1})
        const int qt = wave & 3, sub = wave >> 2;
        if (qt < NTL) {
          const int q = qt * 16 + p;
          if constexpr (DK == 128) {
            f32x4 o[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
              if (k2 < NTL) {
                const f32x4 ds4 = *reinterpret_cast<const f32x4*>(dSb + q * DSP + k2 * 16 + 4 * j);
                f32x4 vv[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) vv[s] = *reinterpret_cast<const f32x4*>(fK + (k2 * 16 + 4 * j + s) * LQ + hc + sub * 64 + 4 * p);
#pragma unroll
                for (int t = 0; t < 4; ++t) o[t] = amma<NP>(f32x4{vv[0][t], vv[1][t], vv[2][t], vv[3][t]}, ds4, o[t]);
              }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
              *reinterpret_cast<f32x4*>(fV + q * LQ + hc + sub * 64 + 16 * j + 4 * r) = f32x4{o[0][r], o[1][r], o[2][r], o[3][r]};
          } else {
            f32x4 o[2];
            o[0] = o[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
              if (k2 < NTL) {
                const f32x4 ds4 = *reinterpret_cast<const f32x4*>(dSb + q * DSP + k2 * 16 + 4 * j);
                f32x2 vv[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) vv[s] = *reinterpret_cast<const f32x2*>(fK + (k2 * 16 + 4 * j + s) * LQ + hc + 4 * p + 2 * sub);
#pragma unroll
                for (int t = 0; t < 2; ++t) o[t] = amma<NP>(f32x4{vv[0][t], vv[1][t], vv[2][t], vv[3][t]}, ds4, o[t]);
              }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x2*>(fV + q * LQ + hc + 16 * j + 4 * r + 2 * sub) = f32x2{o[0][r], o[1][r]};
          }
        }
      }
      lds_barrier();      // pass 2's reads of dS / K done (the next head's pass 1 rewrites dS); dQ / dK / dV of this head complete
      mark(7);
    }
    if (!EARLY) load_session(min(b + (int)gridDim.x, a.B - 1));
    if constexpr (!FULLW) {
      // ---- dQKV rows: [dQ | dK | dV] from the V / Q / dA regions, 16 bytes per lane along the row
      for (int i = tid; i < L * (D / 4); i += NT) {
        const int row = i / (D / 4), c4 = (i - row * (D / 4)) * 4;
        const f32x4 q4 = *reinterpret_cast<const f32x4*>(fV + row * LQ + c4);
        const f32x4 k4 = *reinterpret_cast<const f32x4*>(fQ + row * LQ + c4);
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(fdA + row * LQ + c4);
        const size_t g = ((size_t)b * L + row) * (3 * D) + c4;
        if constexpr (IO16) {
          __bf16* o16 = reinterpret_cast<__bf16*>(a.dQKV);
          *reinterpret_cast<bf16x4*>(o16 + g) = bf16x4{(__bf16)q4[0], (__bf16)q4[1], (__bf16)q4[2], (__bf16)q4[3]};
          *reinterpret_cast<bf16x4*>(o16 + g + D) = bf16x4{(__bf16)k4[0], (__bf16)k4[1], (__bf16)k4[2], (__bf16)k4[3]};
          *reinterpret_cast<bf16x4*>(o16 + g + 2 * D) = bf16x4{(__bf16)v4[0], (__bf16)v4[1], (__bf16)v4[2], (__bf16)v4[3]};
        } else {
          *reinterpret_cast<f32x4*>(a.dQKV + g) = q4;
          *reinterpret_cast<f32x4*>(a.dQKV + g + D) = k4;
          *reinterpret_cast<f32x4*>(a.dQKV + g + 2 * D) = v4;
        }
      }
    }
    if constexpr (FULLX) {
      // ---- tail: dX = dQ Wq + dK Wk + dV Wv + dZ, and (FULLW) dWq / dWk / dWv += dQ^T X, dK^T X, dV^T X.  The three gradients sit as fp32 rows in
      // the V / Q / dA regions (zero in padded rows); each goes through planes in turn, X once
      constexpr int KBT3 = 4 * ((3 * D + 127) / 128), KB = D / 32;
      const uint4* imgT = launder(a.WqkvT) + ((size_t)ct * KBT3 * 3) * 64 + lane;
      WF wq;
      wload<NP, WN>(imgT, wq);
      if constexpr (FULLW) {
        f32x4 vx[NJ];
        load_tile(a.X, b, std::false_type{}, vx);
        store_tile(pX2, std::false_type{}, vx, PitchP{});
      }
      f32x4 accX[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) accX[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* src = c == 0 ? fV : (c == 1 ? fQ : fdA);
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
          int tr, tc;
          tile_rc(jj, tr, tc);
          store4<NP, PLANE>(pD + tr * LDP + tc, *reinterpret_cast<const f32x4*>(src + tr * LQ + tc));
        }
        lds_barrier();
        WF wn;
        if (c < 2) lin<D, NP, RT, true, LDP, WN>(pD, imgT + ((size_t)c * KB * 3) * 64, wq, rt0, p, j, accX, imgT + ((size_t)(c + 1) * KB * 3) * 64, wn);
        else lin<D, NP, RT, false, LDP, WN>(pD, imgT + ((size_t)c * KB * 3) * 64, wq, rt0, p, j, accX, imgT, wn);
        if (c < 2) {
wq = wn;
        }
        if constexpr (FULLW) {
          float nodb = 0.f;
          wgrad<D, NP, KTL>(pD, pX2, wnt, wkt0, p, j, accWqkv[c], nodb);
        }
        lds_barrier();      // the next gradient's planes go over these; after the last one the three fp32 regions are free
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) *reinterpret_cast<f32x4*>(fQ + ((rt0 + rt) * 16 + p) * LQ + col) = accX[rt];
      lds_barrier();
      for (int i = tid; i < L * (D / 4); i += NT) {
        const int row = i / (D / 4), c4 = (i - row * (D / 4)) * 4;
        const size_t g = ((size_t)b * L + row) * D + c4;
        *reinterpret_cast<f32x4*>(a.dX + g) = *reinterpret_cast<const f32x4*>(fQ + row * LQ + c4) + *reinterpret_cast<const f32x4*>(a.dZ + g);
      }
    }
    }
    lds_barrier();      // the next session's tiles go over the regions the copy still read
  }
  const int j = lane0 >> 4, p = lane0 & 15;
  // ---- the workgroup's weight-gradient slab: dW2 [D, D] | db2 [D] | dW1 [D, D] | db1 [D]
  float* slab = a.slabs + (size_t)blockIdx.x * C::SLAB;
#pragma unroll
  for (int kt = 0; kt < KTL; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t o = (size_t)(wnt * 16 + 4 * j + r) * D + (wkt0 + kt) * 16 + p;
      slab[o] = accW2[kt][r];
      slab[(size_t)D * (D + 1) + o] = accW1[kt][r];
    }
  if constexpr (FULLW) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int kt = 0; kt < KTL; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          slab[(size_t)2 * D * (D + 1) + (size_t)c * D * D + (size_t)(wnt * 16 + 4 * j + r) * D + (wkt0 + kt) * 16 + p] = accWqkv[c][kt][r];
  }
  db2p = gsum16(db2p);
  db1p = gsum16(db1p);
  if (wkt0 == 0 && j == 0) {
    slab[(size_t)D * D + wnt * 16 + p] = db2p;
    slab[(size_t)D * (D + 1) + (size_t)D * D + wnt * 16 + p] = db1p;
  }
}

// INTEL_FUSE_TOWER_BWD: 0 = never, 1 = wherever the shape is supported, unset = where it is the faster step (tower_bwd_fused_wanted)
int bwd_mode() {
  static const int m = [] { const char* e = getenv("INTEL_FUSE_TOWER_BWD"); return !e || !e[0] || e[0] == 'a' ? 2 : (e[0] == '0' ? 0 : 1); }();
  return m;
}

// one workgroup per CU (the LDS tile of a session takes more than half a CU's); INTEL_TOWER_BWD_CUS=n: at most n of them (A/B: a tower that is
// not on the step's critical chain may leave CUs to the branches that are)
int bwd_grid(int B) {
  static const int cap = [] { const int v = INTEL_DEBUG_ENV("INTEL_TOWER_BWD_CUS", 0); return v > 0 ? v : 1 << 30; }();      // (A/B probe: debug builds only)
  int g = num_cus();
  if (g > cap) g = cap;
  return B < g ? B : g;
}

template <int D, int DK, int NP, bool IO16>
int launch_one(const TowerBwdArgs& a, hipStream_t st) {
  using C = BwdCfg<D, NP>;
  static_assert(C::SMEM <= 160 * 1024, "LDS budget");
  allow_lds((tower_bwd_fused_kernel<D, DK, NP, IO16>), C::SMEM);
  const int grid = bwd_grid(a.B);
  // algorithmic work per session: six D x D linears (R1, dR1, dA, q / k / v) + two weight gradients, five attention products
  // (+ dX: three more linears, + the q/k/v weight gradients: three more weight gradients)
  const double flops = (double)a.B * ((16.0 + (C::FULLX ? 6.0 : 0.0) + (C::FULLW ? 6.0 : 0.0)) * a.L * D * D + 10.0 * (double)a.L * a.L * D);
  const double bytes = (double)a.B * a.L * D * ((a.a16 ? 2.0 : 4.0) + 8.0 + (C::FULLW ? 0.0 : (a.dqkv16 ? 6.0 : 12.0)) + (C::FULLX ? 4.0 : 0.0));
  static const int dbg_on = INTEL_DEBUG_ENV("INTEL_TOWER_DBG", 0);      // phase clocks: debug builds only (common.h)
  TowerBwdArgs aa = a;
  static unsigned long long* dbg_buf = nullptr;
  if (dbg_on) {
    if (!dbg_buf) (void)hipMalloc(&dbg_buf, 8 * sizeof(unsigned long long));
    (void)hipMemsetAsync(dbg_buf, 0, 8 * sizeof(unsigned long long), st);
    aa.dbg = dbg_buf;
  }
  LAUNCH_S(a.B * a.L, D, DK, flops, bytes, (tower_bwd_fused_kernel<D, DK, NP, IO16>), dim3(grid), dim3(C::NT), C::SMEM, st, aa);
  INTEL_CHECK_LAUNCH();
  if (dbg_on) {          // tools/tower_probe.py
    unsigned long long h[8];
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h, dbg_buf, sizeof(h), hipMemcpyDeviceToHost);
    const int iters = (a.B + grid - 1) / grid;
    fprintf(stderr, "tower_bwd D=%d dk=%d np=%d grid=%d iters=%d  cycles/session: load %llu  r1 %llu  dw2+df1 %llu  dw1+da %llu  qkv %llu  attn-kv %llu  pair %llu  attn-q %llu\n",
            D, DK, NP, grid, iters, h[0] / iters, h[1] / iters, h[2] / iters, h[3] / iters, h[4] / iters, h[5] / iters, h[6] / iters, h[7] / iters);
  }
  return 0;
}

template <int D, int DK>
int launch_np(const TowerBwdArgs& a, hipStream_t st) {
  if (gemm_planes() == 1) return a.a16 ? launch_one<D, DK, 1, true>(a, st) : launch_one<D, DK, 1, false>(a, st);
  return launch_one<D, DK, 3, false>(a, st);
}

}  // namespace

bool tower_bwd_fused_supported(int L, int d, int heads) {
  if (bwd_mode() == 0 || L < 1 || L > 64 || heads < 1 || d % heads != 0) return false;
  const int dk = d / heads;
  return (d == 128 && (dk == 128 || dk == 64)) || (d == 64 && dk == 64);
}

// Policy (same-box A/B at the headline).  bf16 mode: on for both widths (one plane, no spill; 35 / 20 k cycles per session
// against ~52 / 26 k of the kernels it replaces; +2.3 ... +3.3 % sessions/s).  fp32: OFF -- the kernel's work is equal to (64-wide: 36 k cycles against
// 38 k) or more than (128-wide: 103 - 115 k against 76 k; 64 accumulator registers per wave next to six-plane operands spill) what it replaces, and a
// workgroup holds a whole CU (8 waves x 256 registers, > 80 KB of LDS), so the three other branches of the backward cannot share it: -1.5 ... -2 %
// sessions/s for the 64-wide tower alone, -9 ... -15 % with the 128-wide one, at 5 GB less HBM traffic per step.  INTEL_FUSE_TOWER_BWD=1 forces it.
bool tower_bwd_fused_wanted(int d) {
  (void)d;
  const int m = bwd_mode();
  if (m != 2) return m == 1;
  return gemm_planes() == 1;
}

// what one launch covers beyond dQKV: bit 0 = dX = dQKV Wqkv + dZ, bit 1 = the q/k/v weight gradients (then dQKV itself is not written)
int tower_bwd_fused_scope(int d) { return d == 64 ? 3 : (gemm_planes() == 1 ? 1 : 0); }

size_t tower_bwd_slab_floats(int B, int d) { return (size_t)bwd_grid(B) * ((size_t)2 * d * (d + 1) + (d == 64 ? (size_t)3 * d * d : 0)); }

int launch_tower_bwd_fused(const float* X, const float* A, const float* LSE, const float* dZ, int B, int L, int d, int heads, const void* Wqkv_b3,
                           const void* W1_b3, const void* W2T_b3, const void* W1T_b3, const void* WqkvT_b3, const float* b1, float* dQKV, float* dX,
                           float* const* grads, const int* accumulate, ReduceQueue* q, hipStream_t st, int a16, int dqkv16) {
  float *dW2 = grads[0], *db2 = grads[1], *dW1 = grads[2], *db1 = grads[3];
  if (B <= 0) return 0;
  INTEL_CHECK_ARG(tower_bwd_fused_supported(L, d, heads), "tower_bwd_fused: unsupported shape L=%d d=%d heads=%d", L, d, heads);
  INTEL_CHECK_ARG(q != nullptr, "tower_bwd_fused: needs the reduce queue");
  INTEL_CHECK_ARG(!(a16 || dqkv16) || gemm_planes() == 1, "tower_bwd_fused: bf16 operands need the bf16 mode");
  INTEL_CHECK_ARG((a16 != 0) == (dqkv16 != 0), "tower_bwd_fused: A and dQKV are bf16 arrays together or not at all");
  const int scope = tower_bwd_fused_scope(d);
  INTEL_CHECK_ARG(!(scope & 1) || (dX && WqkvT_b3), "tower_bwd_fused: this shape computes dX in the kernel (needs dX and the transposed q/k/v image)");
  const size_t slab = (size_t)2 * d * (d + 1) + ((scope & 2) ? (size_t)3 * d * d : 0);
  const int grid = bwd_grid(B);
  float* slabs = redq_alloc(q, (size_t)grid * slab);
  INTEL_CHECK_ARG(slabs != nullptr, "tower_bwd_fused: reduce arena exhausted");
  TowerBwdArgs a;
  a.X = X; a.A = A; a.LSE = LSE; a.dZ = dZ; a.B = B; a.L = L; a.a16 = a16 ? 1 : 0; a.dqkv16 = dqkv16 ? 1 : 0;
  a.Wqkv = reinterpret_cast<const uint4*>(Wqkv_b3); a.W1 = reinterpret_cast<const uint4*>(W1_b3);
  a.W2T = reinterpret_cast<const uint4*>(W2T_b3); a.W1T = reinterpret_cast<const uint4*>(W1T_b3);
  a.WqkvT = reinterpret_cast<const uint4*>(WqkvT_b3); a.dX = dX;
  a.b1 = b1; a.dQKV = dQKV; a.slabs = slabs; a.dbg = nullptr;
  const int dk = d / heads;
  int rc;
  if (d == 128 && dk == 128) rc = launch_np<128, 128>(a, st);
  else if (d == 128) rc = launch_np<128, 64>(a, st);
  else rc = launch_np<64, 64>(a, st);
  if (rc) return rc;
  const size_t dd = (size_t)d * d;
  if (dW2) redq_push(q, slabs, slab, grid, d, d, dW2, d, accumulate[0]);
  if (db2) redq_push(q, slabs + dd, slab, grid, 1, d, db2, d, accumulate[1]);
  if (dW1) redq_push(q, slabs + dd + d, slab, grid, d, d, dW1, d, accumulate[2]);
  if (db1) redq_push(q, slabs + 2 * dd + d, slab, grid, 1, d, db1, d, accumulate[3]);
  if (scope & 2)
    for (int c = 0; c < 3; ++c)
      if (grads[4 + c]) redq_push(q, slabs + 2 * dd + 2 * d + (size_t)c * dd, slab, grid, d, d, grads[4 + c], d, accumulate[4 + c]);
  return 0;
}
