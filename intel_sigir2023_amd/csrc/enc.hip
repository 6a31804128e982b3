// The BERT4Rec sequence encoder (models/GeneralSeq.py:89-106, blocks = modules/layers.py:82-88 with the attention of
// layers.py:31-60) on PACKED history rows as two kernels per forward pass instead of ~25:
//
//   enc_block_fwd_kernel   one full transformer block for a tile of up to 64 packed rows (several whole sessions):
//                            [Q | K | V] = X Wqkv^T + b            bf16 matrix pipe at fp32 accuracy (three planes, six products)
//                            A = softmax(Q K^T / sqrt(dk)) V        per session and head, keys = the session's own rows, exact fp32 MFMA
//                            C = LayerNorm1(A + X)                  in the attention's register layout (a wave owns both heads of its rows)
//                            F = relu(C W1^T + b1)
//                            E = LayerNorm2(F W2^T + b2 + C)
//                          and, when the NEXT block is the last one, its key / value projection [K' | V'] = E Wkv'^T + b' for all
//                          rows plus row len-1 of E per session (the only query row the pruned last block needs, GeneralSeq.py:103-105).
//                          The tile stays in LDS from the X rows to the K' / V' rows; HBM sees X in, K'V' (+ the training stash) out.
//   enc_last_fwd_kernel    the pruned last block for 16 sessions per workgroup: q = x_last Wq^T + b, one-row attention over the
//                          session's K' / V' rows, LayerNorm1, feed-forward, LayerNorm2 -> the encoder's output vector.
//
// Tiles: session b belongs to tile off[b] / R with R = 65 - T (T = the batch's longest history): a tile's sessions then span at
// most R - 1 + T = 64 rows.  enc_tiles_kernel writes the first session of every tile once per batch (no search in the kernels).
// The training stash has the layout of the kernel-per-op path (model.cpp: EncBlockBufs / EncLastBufs).
#include <stdio.h>
#include <stdlib.h>

#include "kernels.h"
#include "enc.h"
#include "planes.h"

namespace {

using namespace planes;

template <int D, int NP>
struct EncCfg {
  // fp32 mode: a wave per 16-column tile of the D-wide products.  bf16 mode: HALF as many waves with two column tiles each -- the
  // 70 KB tile lets two workgroups share a CU, and with 2 x 4 waves per CU each wave keeps a 256-register budget (8 waves per
  // workgroup would have to fit 128: the kernel then spills)
  static constexpr int NW = NP == 1 ? D / 32 : D / 16;
  static constexpr int CM = (D / 16) / NW;          // column tiles per wave in a D-wide product
  static constexpr int NT = NW * 64;
  static constexpr int KB = D / 32;                 // 32-deep k blocks
  static constexpr int KBT = 4;                     // k blocks per column tile in a weight image (k padded to 128)
  static constexpr int LDP = D + 8;                 // bf16 plane pitch (16-byte fragment reads conflict-free)
  static constexpr int PLANE = 64 * LDP;            // bf16 elements per plane
  static constexpr int LQ = D + 4;                  // fp32 row pitch
  static constexpr int NJ = 64 * (D / 4) / NT;      // float4 per thread per tile (= 4)
  static constexpr size_t P_BYTES = (size_t)NP * PLANE * 2;                                   // X planes -> C planes -> E planes
  static constexpr size_t R_BYTES = NP == 1 ? (size_t)3 * PLANE * 2 : (size_t)3 * 64 * LQ * 4; // q | k | v rows; later R1 planes + the LayerNorm tile
  static constexpr size_t R1_BYTES = (size_t)NP * PLANE * 2;
  static constexpr size_t ES_BYTES = (size_t)64 * LQ * 4;
  static_assert(R1_BYTES + ES_BYTES <= R_BYTES, "R1 planes + LayerNorm tile must fit the q/k/v region");
  static constexpr size_t SMEM = P_BYTES + R_BYTES;
  static constexpr int WPS = 2;                     // waves per SIMD the register budget is set for
};

// acc[c][rt] += A rows (planes at `frag` = planes + p * LDP + 8 * j, RT row tiles of 16) x column tiles ct0 .. ct0 + CT - 1 of a
// pre-split weight image (`img` already points at this lane); weight fragments one k block ahead of their MFMAs
template <int D, int NP, int CT, int RT, int ROWS>
__device__ __forceinline__ void gemm_planes(const __bf16* frag, const uint4* img, int ct0, f32x4 (&acc)[CT][RT]) {
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
      __builtin_amdgcn_sched_barrier(0);      // (pins the requests here: the scheduler otherwise sinks them behind this block's MFMAs -- no lookahead at all)
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const __bf16* fp = frag + rt * 16 * LDP + kb * 32;
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(fp);
      bf16x8 am = ah, al = ah;
      if (NP == 3) {
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

// the weight fragments of k block kb of column tiles ct0 .. ct0 + CT - 1
template <int NP, int CT>
__device__ __forceinline__ void load_w(uint4 (&bw)[CT][NP], const uint4* img, int ct0, int kb) {
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) bw[c][pl] = img[((size_t)((ct0 + c) * 4 + kb) * 3 + pl) * 64];
}
// gemm_planes with the fragments of k block 0 already in bw[0] (loaded ahead of the barrier in front of the product)
// `tail_loads` runs right behind the product's last fragment loads: global loads of the NEXT phase issued there do not delay this
// product (vmcnt retires in order: anything issued earlier would have to land before the fragments behind it count as arrived)
struct NoHook { __device__ __forceinline__ void operator()() const {} };
template <int D, int NP, int CT, int RT, int ROWS, typename Hook = NoHook>
__device__ __forceinline__ void gemm_planes_pre(const __bf16* frag, const uint4* img, int ct0, f32x4 (&acc)[CT][RT], uint4 (&bw)[2][CT][NP],
                                                Hook tail_loads = Hook()) {
  constexpr int KB = D / 32, LDP = D + 8, PLANE = ROWS * LDP;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    if (kb + 1 < KB) load_w<NP, CT>(bw[(kb + 1) & 1], img, ct0, kb + 1);
    if (kb == (KB >= 2 ? KB - 2 : 0)) tail_loads();
    __builtin_amdgcn_sched_barrier(0);        // (pins the requests here: the scheduler otherwise sinks them behind this block's MFMAs -- no lookahead at all)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const __bf16* fp = frag + rt * 16 * LDP + kb * 32;
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(fp);
      bf16x8 am = ah, al = ah;
      if (NP == 3) {
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

// bf16 mode (one plane): NCT column tiles one after the other, the K / 32 fragments of the NEXT tile requested while this one is
// multiplied (a tile's whole k extent is 4 x 16 bytes per lane); epi(ct, acc) consumes a finished tile
template <int D, int NCT, int RT, int ROWS, typename Epi>
__device__ __forceinline__ void gemm_cols_bf16(const __bf16* frag, const uint4* img, int ct0, Epi epi) {
  constexpr int KB = D / 32, KBT = 4, LDP = D + 8;
  uint4 bw[2][KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) bw[0][kb] = img[((size_t)(ct0 * KBT + kb) * 3) * 64];
  bf16x8 af[RT][KB];               // the activation fragments of the whole tile stay in registers for all column tiles (one LDS pass)
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) af[rt][kb] = *reinterpret_cast<const bf16x8*>(frag + rt * 16 * LDP + kb * 32);
#pragma unroll 2
  for (int c = 0; c < NCT; ++c) {
    if (c + 1 < NCT) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) bw[(c + 1) & 1][kb] = img[((size_t)((ct0 + c + 1) * KBT + kb) * 3) * 64];
    }
    f32x4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bw[c & 1][kb]), af[rt][kb], acc[rt], 0, 0, 0);
    epi(ct0 + c, acc);
  }
  (void)ROWS;
}

struct EncBlockArgs {
  const float* X;            // [rows, D] packed block input
  int rows, B, T, ntiles;
  const int* off;            // [B] first packed row of every session
  const int* tile_s;         // [ntiles + 1] first session of every tile (enc_tiles_kernel)
  const uint4* Wqkv; const uint4* W1; const uint4* W2;      // pre-split bf16 images (pack_b3 layout): [D -> 3D], [D -> D], [D -> D]
  const uint4* Wkv;          // image of the next block's [D -> 2D] key / value weights, or NULL
  const float* bqkv; const float* b1; const float* b2; const float* bkv;
  const float* g1; const float* be1; const float* g2; const float* be2;
  float* C;                  // [rows, D] LayerNorm1 output (always written: the LayerNorm2 residual is read back from it)
  float* out;                // [rows, D] block output, or NULL
  float* xlast;              // [B, D] row len-1 of the output per session, or NULL
  float* KV;                 // [rows, 2D] next block's [k | v] rows, or NULL
  // training stash (each may be NULL)
  float* QKV;                // [rows, 3D]
  float* LSE;                // [B * heads * T] natural-log softmax normalisers
  float* XH1; float* RSTD1;  // LayerNorm1 x-hat [rows, D], 1/std [rows]
  float* F1;                 // [rows, D] relu(W1 C + b1)
  float* XH2; float* RSTD2;
  unsigned long long* dbg;   // INTEL_ENC_DBG=1: per-phase shader-clock totals of workgroup 0's thread 0 (NULL otherwise)
};

// ---- attention + LayerNorm1 for one (session, 16-query tile): both heads by one wave, so that every row of the tile is complete in
// the wave's registers and LayerNorm1 needs no LDS round trip.  KT = key tiles of 16 (the session's length decides).
template <int D, int DK, int NP, int KT, bool TRAIN>
__device__ __forceinline__ void attn_ln1_item(const EncBlockArgs& a, const unsigned char* rbase, __bf16* cplanes, const float* s_par, int base, int len,
                                              int qt, int r0, int sess, int lane) {
  using C = EncCfg<D, NP>;
  constexpr int LQ = C::LQ, LDP = C::LDP, PLANE = C::PLANE, HEADS = D / DK;
  static_assert(DK == 64, "the b128 V read feeds four 16-dim output tiles: head dim 64");
  const int j = lane >> 4, p = lane & 15;
  const float scale = 1.0f / sqrtf((float)DK);
  const float c2 = scale * 1.4426950408889634f;
  const int q = qt * 16 + p;                       // query row inside the session
  const bool qok = q < len;
  const int qrow = min(base + q, 63);              // rows past the session are other sessions' (finite) rows: masked / never stored
  int krow[KT], vrow[KT][4];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    krow[kt] = min(base + kt * 16 + p, 63);
#pragma unroll
    for (int r = 0; r < 4; ++r) vrow[kt][r] = min(base + kt * 16 + 4 * j + r, 63);
  }
  // the residual rows of LayerNorm1 (the block input) travel while the attention is computed
  const size_t grow = (size_t)r0 + min(base + q, base + len - 1);       // clamped for the lanes without a query (loads only)
  constexpr bool HOIST = NP == 3 && !TRAIN;        // (training and the bf16 mode: registers -- loaded where they are used)
  f32x4 xres[HOIST ? HEADS : 1][4];
  if constexpr (HOIST) {
#pragma unroll
    for (int h = 0; h < HEADS; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) xres[h][r] = *reinterpret_cast<const f32x4*>(a.X + grow * D + h * DK + 16 * j + 4 * r);
  }
  f32x4 o[HEADS][4];                               // o[h][r] = columns h*DK + 16j + 4r .. +3 of query q
#pragma unroll
  for (int h = 0; h < HEADS; ++h) {
    f32x4 st[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (NP == 1) {
      const __bf16* Q16 = reinterpret_cast<const __bf16*>(rbase);
      const __bf16* K16 = Q16 + PLANE;
      const __bf16* Qh = Q16 + qrow * LDP + h * DK + 4 * j;
#pragma unroll
      for (int g = 0; g < DK / 16; ++g) {
        const s16x4 qf = *reinterpret_cast<const s16x4*>(Qh + 16 * g);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const s16x4 kf = *reinterpret_cast<const s16x4*>(K16 + krow[kt] * LDP + h * DK + 4 * j + 16 * g);
          st[kt] = mma4_bf16(kf, qf, st[kt]);
        }
      }
    } else {
      const float* Qs = reinterpret_cast<const float*>(rbase);
      const float* Ks = Qs + 64 * LQ;
      const float* Qp = Qs + qrow * LQ + h * DK + 4 * j;
      f32x4 sb[KT];                                 // second partial sum: two independent MFMA chains per key tile
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) sb[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < DK / 16; ++g) {
        const f32x4 qf = *reinterpret_cast<const f32x4*>(Qp + 16 * g);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + krow[kt] * LQ + h * DK + 4 * j + 16 * g);
          if (g & 1) sb[kt] = mma4(kf, qf, sb[kt]);
          else st[kt] = mma4(kf, qf, st[kt]);
        }
      }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) st[kt] += sb[kt];
    }
    // accumulator register r of tile kt at lane (j, p) = key kt*16 + 4j + r, query q
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = (kt * 16 + 4 * j + r) < len ? st[kt][r] : -INFINITY;
        st[kt][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = gmax16(mx);                               // len >= 1: key 0 is live for every query
    const float moff = -mx * c2;
    float ps = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][r], c2, moff));
        st[kt][r] = e;
        ps += e;
      }
    ps = gsum16(ps);
    const float inv = 1.f / ps;
    if (TRAIN && a.LSE && j == 0 && qok) a.LSE[((size_t)sess * HEADS + h) * a.T + q] = mx * scale + __logf(ps);
    // O^T = V^T P^T: V read as one b128 along the head dim (lane p takes dims 4p .. 4p+3 of its key rows) feeding four MFMAs
    // whose output row i means dim 4 i + t
    f32x4 oT[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) oT[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (NP == 1) {
      const __bf16* V16 = reinterpret_cast<const __bf16*>(rbase) + 2 * PLANE;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const s16x4 pb = to_bf16x4(st[kt]);        // the unnormalised probabilities rounded to bf16 (row sums stay fp32)
        s16x4 vv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) vv[r] = *reinterpret_cast<const s16x4*>(V16 + vrow[kt][r] * LDP + h * DK + 4 * p);
#pragma unroll
        for (int t = 0; t < 4; ++t) oT[t] = mma4_bf16(s16x4{vv[0][t], vv[1][t], vv[2][t], vv[3][t]}, pb, oT[t]);
      }
    } else {
      const float* Vs = reinterpret_cast<const float*>(rbase) + 2 * 64 * LQ;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        f32x4 vv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) vv[r] = *reinterpret_cast<const f32x4*>(Vs + vrow[kt][r] * LQ + h * DK + 4 * p);
#pragma unroll
        for (int t = 0; t < 4; ++t) oT[t] = mma4(f32x4{vv[0][t], vv[1][t], vv[2][t], vv[3][t]}, st[kt], oT[t]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) o[h][r] = f32x4{oT[0][r], oT[1][r], oT[2][r], oT[3][r]} * inv;
  }
  // ---- LayerNorm1(A + X): lane (p, j) holds 8 * HEADS column quads of query q, the other quads sit in the lanes p + 16 j'
  float s = 0.f;
#pragma unroll
  for (int h = 0; h < HEADS; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if constexpr (HOIST) o[h][r] += xres[h][r];
      else o[h][r] += *reinterpret_cast<const f32x4*>(a.X + grow * D + h * DK + 16 * j + 4 * r);
      s += (o[h][r][0] + o[h][r][1]) + (o[h][r][2] + o[h][r][3]);
    }
  const float mean = gsum16(s) * (1.f / (float)D);
  float v2 = 0.f;
#pragma unroll
  for (int h = 0; h < HEADS; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o[h][r] -= mean;
      v2 += (o[h][r][0] * o[h][r][0] + o[h][r][1] * o[h][r][1]) + (o[h][r][2] * o[h][r][2] + o[h][r][3] * o[h][r][3]);
    }
  const float rs = 1.f / sqrtf(gsum16(v2) * (1.f / (float)D) + 1e-5f);
  if (qok) {
    if (TRAIN && a.RSTD1 && j == 0) a.RSTD1[grow] = rs;
#pragma unroll
    for (int h = 0; h < HEADS; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = h * DK + 16 * j + 4 * r;
        const f32x4 xh = o[h][r] * rs;
        const f32x4 g = *reinterpret_cast<const f32x4*>(s_par + col), be = *reinterpret_cast<const f32x4*>(s_par + D + col);
        const f32x4 c = xh * g + be;
        if (TRAIN && a.XH1) *reinterpret_cast<f32x4*>(a.XH1 + grow * D + col) = xh;
        *reinterpret_cast<f32x4*>(a.C + grow * D + col) = c;
        store4<NP, PLANE>(cplanes + (base + q) * LDP + col, c);
      }
  }
}

// parameter vectors of a block staged once per workgroup in LDS (the attention / LayerNorm code reads them per tile)
template <int D>
struct EncPar {
  static constexpr int G1 = 0, BE1 = D, G2 = 2 * D, BE2 = 3 * D, B1 = 4 * D, B2 = 5 * D, BQKV = 6 * D, BKV = 9 * D, N = 11 * D;
};

template <int D, int DK, bool TRAIN, int NP>
__global__ __launch_bounds__((EncCfg<D, NP>::NT), (EncCfg<D, NP>::WPS)) void enc_block_fwd_kernel(EncBlockArgs a) {
  using C = EncCfg<D, NP>;
  using PR = EncPar<D>;
  constexpr int NW = C::NW, NT = C::NT, LDP = C::LDP, PLANE = C::PLANE, LQ = C::LQ, NJ = C::NJ, KBT = C::KBT, CM = C::CM;
  constexpr bool PF = NP == 3;          // weight fragments of a product's first k block requested across the barrier in front of it (one column tile per wave)
  constexpr bool PFX = NP == 3;         // the next tile's rows requested a tile ahead (bf16 mode: the CU's second workgroup covers that wait; registers)
  constexpr bool PFR = true;            // the LayerNorm2 residual rows requested a phase ahead
  static_assert(!PF || CM == 1, "the cross-phase fragment prefetch assumes one column tile per wave");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ int s_start[65];        // first row (inside the tile) of every session of the tile, then the tile's row count
  __shared__ int s_rowlast[64];      // session whose last row this is, or -1
  __shared__ int s_items[128];       // attention work items: session | query tile << 8
  __shared__ int s_nitems;
  __shared__ __attribute__((aligned(16))) float s_par[PR::N];
  __bf16* planes = reinterpret_cast<__bf16*>(smem_raw);                  // P
  unsigned char* rbase = smem_raw + C::P_BYTES;                           // R
  float* Qs = reinterpret_cast<float*>(rbase);
  __bf16* Q16 = reinterpret_cast<__bf16*>(rbase);
  __bf16* r1planes = reinterpret_cast<__bf16*>(rbase);
  float* Es = reinterpret_cast<float*>(rbase + C::R1_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, j = lane >> 4, p = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < PR::N; i += NT) {
    float v = 0.f;
    if (i < PR::BE1) v = a.g1[i];
    else if (i < PR::G2) v = a.be1[i - PR::BE1];
    else if (i < PR::BE2) v = a.g2[i - PR::G2];
    else if (i < PR::B1) v = a.be2[i - PR::BE2];
    else if (i < PR::B2) v = a.b1[i - PR::B1];
    else if (i < PR::BQKV) v = a.b2[i - PR::B2];
    else if (i < PR::BKV) v = a.bqkv[i - PR::BQKV];
    else if (a.bkv) v = a.bkv[i - PR::BKV];
    s_par[i] = v;
  }
  int trow[NJ], tcol[NJ];
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int i = tid + NT * jj;
    trow[jj] = i / (D / 4);
    tcol[jj] = (i - trow[jj] * (D / 4)) * 4;
  }
  unsigned long long tstamp = 0;
  const bool probe = a.dbg != nullptr && blockIdx.x == 0 && tid == 0;
  auto mark = [&](int ph) {
    if (probe) {
      const unsigned long long now = clock64();
      if (ph >= 0) a.dbg[ph] += now - tstamp;
      tstamp = now;
    }
  };
  // tile bookkeeping is read one tile ahead: the X rows of the next tile travel (into `pre`) while this one is computed
  struct Tile { int s_lo, ns, r0, nrows; };
  auto tile_meta = [&](int t) {
    Tile tl{0, 0, 0, 0};
    if (t < a.ntiles) {
      const int s_lo = a.tile_s[t], s_hi = a.tile_s[t + 1];
      tl.s_lo = s_lo;
      tl.ns = s_hi - s_lo;
      if (tl.ns > 0) {
        tl.r0 = a.off[s_lo];
        tl.nrows = (s_hi < a.B ? a.off[s_hi] : a.rows) - tl.r0;
      }
    }
    return tl;
  };
  f32x4 pre[NJ];
  int w_st = 0, w_en = 0;                                      // wave 0: first row / end row of session s_lo + lane
  auto load_tile = [&](const Tile& tl) {
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      const int row = min(trow[jj], max(tl.nrows - 1, 0));
      const f32x4 v = *reinterpret_cast<const f32x4*>(a.X + ((size_t)tl.r0 + row) * D + tcol[jj]);
      pre[jj] = trow[jj] < tl.nrows ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (wave == 0 && lane < tl.ns) {
      w_st = a.off[tl.s_lo + lane];
      w_en = tl.s_lo + lane + 1 < a.B ? a.off[tl.s_lo + lane + 1] : a.rows;
    }
  };
  Tile cur = tile_meta(blockIdx.x);
  if constexpr (PFX) load_tile(cur);
  mark(-1);
  for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
    if (cur.ns <= 0) break;            // only the last window can be without a session start (a window holds 65 - T > T rows)
    const int s_lo = cur.s_lo, ns = cur.ns, r0 = cur.r0, nrows = cur.nrows;
    if constexpr (!PFX) load_tile(cur);
    // ---- tile bookkeeping (wave 0) + X rows -> planes
    if (wave == 0) {
      s_rowlast[lane] = -1;
      int ln = 0;
      if (lane < ns) {
        const int start = w_st - r0;
        ln = w_en - w_st;
        s_start[lane] = start;
        if (ln > 0) s_rowlast[start + ln - 1] = s_lo + lane;      // an empty history has no last row (x_last stays unwritten: enc_last reads 0)
        s_items[lane] = lane;
      }
      if (lane == 0) s_start[ns] = nrows;
      // sessions longer than 16 rows have a second query tile: compacted behind the first tiles
      const int two = (lane < ns && ln > 16) ? 1 : 0;
      int incl = two;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
      }
      if (two) s_items[ns + incl - 1] = lane | (1 << 8);
      if (lane == 63) s_nitems = ns + incl;
    }
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) store4<NP, PLANE>(planes + trow[jj] * LDP + tcol[jj], pre[jj]);
    uint4 bwq[2][PF ? 3 : 1][NP];
    if constexpr (PF) load_w<NP, 3>(bwq[0], launder(a.Wqkv) + lane, 3 * wave, 0);
    mark(0);
    lds_barrier();
    mark(1);
    // ---- [Q | K | V] = X Wqkv^T + b; wave = column tiles 3 wave .. 3 wave + 2, all four row tiles
    {
      const uint4* img = launder(a.Wqkv) + lane;
      const __bf16* frag = planes + p * LDP + 8 * j;
      auto epilogue = [&](int n, const f32x4 (&acc)[4]) {      // n = column of [q | k | v]
        const f32x4 bias = *reinterpret_cast<const f32x4*>(s_par + PR::BQKV + n);
        const int which = n / D, col = n - which * D;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          const int row = rt * 16 + p;
          const f32x4 v = acc[rt] + bias;
          if (NP == 1) *reinterpret_cast<bf16x4*>(Q16 + which * PLANE + row * LDP + col) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
          else *reinterpret_cast<f32x4*>(Qs + which * 64 * LQ + row * LQ + col) = v;
          if (TRAIN && a.QKV && row < nrows) *reinterpret_cast<f32x4*>(a.QKV + ((size_t)r0 + row) * (3 * D) + n) = v;
        }
      };
      if constexpr (NP == 1) {                                 // register budget of four waves per SIMD: one column tile at a time
        gemm_cols_bf16<D, 3 * CM, 4, 64>(frag, img, 3 * CM * wave, [&](int ct, const f32x4 (&acc)[4]) { epilogue(ct * 16 + 4 * j, acc); });
      } else {
        f32x4 acc[3][4];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) acc[c][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm_planes_pre<D, NP, 3, 4, 64>(frag, img, 3 * wave, acc, bwq);
#pragma unroll
        for (int c = 0; c < 3; ++c) epilogue((3 * wave + c) * 16 + 4 * j, acc[c]);
      }
    }
    // the next tile's bookkeeping and rows travel from here on
    const Tile nxt = tile_meta(t + gridDim.x);
    if constexpr (PFX) { if (nxt.ns > 0) load_tile(nxt); }
    mark(2);
    lds_barrier();
    mark(3);
    // ---- attention + LayerNorm1 -> C planes (over the dead X planes), C rows to HBM
    {
      const int nitems = s_nitems;
      for (int it = wave; it < nitems; it += NW) {
        const int item = s_items[it];
        const int s = item & 255, qt = item >> 8;
        const int base = s_start[s], len = s_start[s + 1] - base;
        if (len > 16) attn_ln1_item<D, DK, NP, 2, TRAIN>(a, rbase, planes, s_par, base, len, qt, r0, s_lo + s, lane);
        else attn_ln1_item<D, DK, NP, 1, TRAIN>(a, rbase, planes, s_par, base, len, qt, r0, s_lo + s, lane);
      }
    }
    uint4 bw1[2][1][NP], bw2[2][1][NP];
    if constexpr (PF) load_w<NP, 1>(bw1[0], launder(a.W1) + lane, wave, 0);
    mark(4);
    __syncthreads();       // C planes in LDS and the C rows in HBM (read back as the LayerNorm2 residual) are complete
    mark(5);
    // the LayerNorm2 residual rows of this thread come back while the two feed-forward products run
    constexpr int CPL = D / 64;                      // float4 per lane per row (16 lanes per row)
    constexpr int ROUNDS = 64 / (NW * 4);
    f32x4 cres[PFR ? ROUNDS : 1][CPL];
    // ---- R1 = relu(C W1^T + b1) -> planes over the dead q / k / v rows; wave = one column tile
    {
      auto epi1 = [&](int ct, const f32x4 (&acc)[4]) {
        const int col = ct * 16 + 4 * j;
        const f32x4 bias = *reinterpret_cast<const f32x4*>(s_par + PR::B1 + col);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          const int row = rt * 16 + p;
          f32x4 x = acc[rt] + bias;
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] = fmaxf(x[r], 0.f);
          store4<NP, PLANE>(r1planes + row * LDP + col, x);
          if (TRAIN && a.F1 && row < nrows) *reinterpret_cast<f32x4*>(a.F1 + ((size_t)r0 + row) * D + col) = x;
        }
      };
      if constexpr (NP == 1) {
        gemm_cols_bf16<D, CM, 4, 64>(planes + p * LDP + 8 * j, launder(a.W1) + lane, wave * CM, epi1);
      } else {
        f32x4 acc[1][4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[0][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm_planes_pre<D, NP, 1, 4, 64>(planes + p * LDP + 8 * j, launder(a.W1) + lane, wave, acc, bw1);
        epi1(wave, acc[0]);
      }
    }
    if constexpr (PF) load_w<NP, 1>(bw2[0], launder(a.W2) + lane, wave, 0);
    mark(6);
    lds_barrier();
    mark(7);
    // ---- Z = R1 W2^T + b2 -> fp32 tile; the LayerNorm2 residual rows of this thread are requested behind the product's loads
    {
      auto res_loads = [&]() {
#pragma unroll
        for (int rnd = 0; rnd < (PFR ? ROUNDS : 1); ++rnd) {
          const int row = min((rnd * NW + wave) * 4 + j, nrows - 1);
#pragma unroll
          for (int cc = 0; cc < CPL; ++cc) cres[rnd][cc] = *reinterpret_cast<const f32x4*>(a.C + ((size_t)r0 + row) * D + (D / 16) * p + 4 * cc);
        }
      };
      auto epi2 = [&](int ct, const f32x4 (&acc)[4]) {
        const int col = ct * 16 + 4 * j;
        const f32x4 bias = *reinterpret_cast<const f32x4*>(s_par + PR::B2 + col);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) *reinterpret_cast<f32x4*>(Es + (rt * 16 + p) * LQ + col) = acc[rt] + bias;
      };
      if constexpr (NP == 1) {
        gemm_cols_bf16<D, CM, 4, 64>(r1planes + p * LDP + 8 * j, launder(a.W2) + lane, wave * CM, epi2);
        if constexpr (PFR) res_loads();
      } else {
        f32x4 acc[1][4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[0][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm_planes_pre<D, NP, 1, 4, 64>(r1planes + p * LDP + 8 * j, launder(a.W2) + lane, wave, acc, bw2, res_loads);
        epi2(wave, acc[0]);
      }
    }
    uint4 bwk[2][PF ? 2 : 1][NP];
    if constexpr (PF) { if (a.Wkv) load_w<NP, 2>(bwk[0], launder(a.Wkv) + lane, 2 * wave, 0); }
    mark(8);
    lds_barrier();
    mark(9);
    // ---- E = LayerNorm2(Z + C): 16 lanes per row (lane p = columns (D/16) p ..), four rows per wave at a time
    {
      const float inv_n = 1.f / (float)D;
#pragma unroll
      for (int rnd = 0; rnd < ROUNDS; ++rnd) {
        const int row = (rnd * NW + wave) * 4 + j;
        const bool rok = row < nrows;
        const size_t grow = (size_t)r0 + (rok ? row : 0);
        f32x4 v[CPL];
        float s = 0.f;
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) {
          const int col = (D / 16) * p + 4 * cc;
          const f32x4 res = PFR ? cres[PFR ? rnd : 0][cc] : *reinterpret_cast<const f32x4*>(a.C + grow * D + col);
          v[cc] = *reinterpret_cast<const f32x4*>(Es + row * LQ + col) + res;
          s += (v[cc][0] + v[cc][1]) + (v[cc][2] + v[cc][3]);
        }
        const float mean = row16_sum(s) * inv_n;
        float q2 = 0.f;
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) {
          v[cc] -= mean;
          q2 += (v[cc][0] * v[cc][0] + v[cc][1] * v[cc][1]) + (v[cc][2] * v[cc][2] + v[cc][3] * v[cc][3]);
        }
        const float rs = 1.f / sqrtf(row16_sum(q2) * inv_n + 1e-5f);
        const int lastof = rok ? s_rowlast[row] : -1;
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) {
          const int col = (D / 16) * p + 4 * cc;
          const f32x4 xh = v[cc] * rs;
          const f32x4 e = xh * *reinterpret_cast<const f32x4*>(s_par + PR::G2 + col) + *reinterpret_cast<const f32x4*>(s_par + PR::BE2 + col);
          if (a.Wkv) store4<NP, PLANE>(planes + row * LDP + col, rok ? e : f32x4{0.f, 0.f, 0.f, 0.f});
          if (rok) {
            if (TRAIN && a.XH2) *reinterpret_cast<f32x4*>(a.XH2 + grow * D + col) = xh;
            if (a.out) *reinterpret_cast<f32x4*>(a.out + grow * D + col) = e;
            if (a.xlast && lastof >= 0) *reinterpret_cast<f32x4*>(a.xlast + (size_t)lastof * D + col) = e;
          }
        }
        if (TRAIN && a.RSTD2 && rok && p == 0) a.RSTD2[grow] = rs;
      }
    }
    mark(10);
    lds_barrier();
    mark(11);
    // ---- the next (last) block's [K' | V'] = E Wkv'^T + b' straight to HBM; wave = column tiles 2 wave, 2 wave + 1 of 2D
    if (a.Wkv) {
      const uint4* img = launder(a.Wkv) + lane;
      const __bf16* frag = planes + p * LDP + 8 * j;
      auto epilogue = [&](int n, const f32x4 (&acc)[4]) {
        const f32x4 bias = *reinterpret_cast<const f32x4*>(s_par + PR::BKV + n);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          const int row = rt * 16 + p;
          if (row < nrows) *reinterpret_cast<f32x4*>(a.KV + ((size_t)r0 + row) * (2 * D) + n) = acc[rt] + bias;
        }
      };
      if constexpr (NP == 1) {
        gemm_cols_bf16<D, 2 * CM, 4, 64>(frag, img, 2 * CM * wave, [&](int ct, const f32x4 (&acc)[4]) { epilogue(ct * 16 + 4 * j, acc); });
      } else {
        f32x4 acc[2][4];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) acc[c][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm_planes_pre<D, NP, 2, 4, 64>(frag, img, 2 * wave, acc, bwk);
#pragma unroll
        for (int c = 0; c < 2; ++c) epilogue((2 * wave + c) * 16 + 4 * j, acc[c]);
      }
      mark(12);
      lds_barrier();       // the next tile's X planes go over the E planes
      mark(13);
    }
    cur = nxt;
    (void)KBT;
  }
}

// ---- the pruned last block: ENC_LAST_SPB sessions per workgroup (enc.h) in a 16-row MFMA tile ---------------------------
struct EncLastArgs {
  const float* xlast;        // [B, D] block input at row len-1 of every session
  const float* KV;           // [rows, 2D] this block's [k | v] rows (packed)
  const int* off; const int* len;
  int B, T;
  const uint4* Wq; const uint4* W1; const uint4* W2;      // images [D -> D]
  const float* bq; const float* b1; const float* b2;
  const float* g1; const float* be1; const float* g2; const float* be2;
  float* out; int ldo;       // [B, ldo] the encoder's output vector
  // training stash (each may be NULL)
  float* QL;                 // [B, D]
  float* PL;                 // [B * heads * T] attention weights (zero beyond len)
  float* CL; float* XH1; float* RSTD1; float* F1; float* XH2; float* RSTD2;
};

template <int D, int NP>
struct EncLastCfg {
  static constexpr int NW = D / 16, NT = NW * 64, LDP = D + 8, PLANE = 16 * LDP, LQ = D + 4;
  static constexpr size_t PL_BYTES = (size_t)NP * PLANE * 2;      // one set of 16-row planes
  static constexpr size_t T_BYTES = (size_t)16 * LQ * 4;          // one 16-row fp32 tile
  static constexpr size_t SMEM = 3 * PL_BYTES + 3 * T_BYTES;      // x / c / f planes; q rows, c rows, LayerNorm tile
};

template <int D, int DK, bool TRAIN, int NP>
__global__ __launch_bounds__((EncLastCfg<D, NP>::NT)) void enc_last_fwd_kernel(EncLastArgs a) {
  using C = EncLastCfg<D, NP>;
  constexpr int LDP = C::LDP, PLANE = C::PLANE, LQ = C::LQ, HEADS = D / DK, NW = C::NW, SPB = ENC_LAST_SPB;
  constexpr int SLOTS = 16 / NW;                     // tile rows per wave: row s = ss * NW + wave holds session b0 + s when s < SPB
  static_assert(HEADS == 2 && DK == 64, "one-row attention: lane = (head, key), 32 keys per head");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* xpl = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* cpl = reinterpret_cast<__bf16*>(smem_raw + C::PL_BYTES);
  __bf16* fpl = reinterpret_cast<__bf16*>(smem_raw + 2 * C::PL_BYTES);
  float* QLs = reinterpret_cast<float*>(smem_raw + 3 * C::PL_BYTES);
  float* CLs = QLs + 16 * LQ;
  float* Es = CLs + 16 * LQ;
  const int tid = threadIdx.x, lane = tid & 63, j = lane >> 4, p = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b0 = blockIdx.x * SPB;
  const float scale = 1.0f / sqrtf((float)DK);
  // ---- x_last rows -> planes
  for (int i = tid; i < 16 * (D / 4); i += C::NT) {
    const int row = i / (D / 4), col = (i - row * (D / 4)) * 4;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (row < SPB && b0 + row < a.B && a.len[b0 + row] > 0) v = *reinterpret_cast<const f32x4*>(a.xlast + (size_t)(b0 + row) * D + col);      // the block kernel writes x_last of non-empty histories only
    store4<NP, PLANE>(xpl + row * LDP + col, v);
  }
  lds_barrier();
  const int col = wave * 16 + 4 * j;                 // this lane's four output columns of every D -> D product
  const bool rowok = p < SPB && b0 + p < a.B;
  const size_t growp = (size_t)(b0 + p);
  // ---- q = x_last Wq^T + bq
  {
    f32x4 acc[1][1];
    acc[0][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm_planes<D, NP, 1, 1, 16>(xpl + p * LDP + 8 * j, launder(a.Wq) + lane, wave, acc);
    const f32x4 v = acc[0][0] + *reinterpret_cast<const f32x4*>(a.bq + col);
    *reinterpret_cast<f32x4*>(QLs + p * LQ + col) = v;
    if (TRAIN && a.QL && rowok) *reinterpret_cast<f32x4*>(a.QL + growp * D + col) = v;
  }
  lds_barrier();
  // ---- attention of the one query row (fp32 on the vector unit) + LayerNorm1; a wave owns whole sessions
#pragma unroll 1
  for (int ss = 0; ss < SLOTS; ++ss) {
    const int s = ss * NW + wave, b = b0 + s;
    if (s >= SPB || b >= a.B) {                       // wave-uniform: no session in this tile row -- finite values for the tile products
      CLs[s * LQ + lane] = 0.f;
      CLs[s * LQ + 64 + lane] = 0.f;
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) {
        cpl[pl * PLANE + s * LDP + lane] = (__bf16)0.f;
        cpl[pl * PLANE + s * LDP + 64 + lane] = (__bf16)0.f;
      }
      continue;
    }
    const int n = a.len[b];
    const size_t rb = (size_t)a.off[b];
    const int h = lane >> 5, t = lane & 31;
    float sc = -INFINITY;
    if (t < n) {
      const float* kr = a.KV + (rb + t) * (2 * D) + h * DK;
      const float* qr = QLs + s * LQ + h * DK;
      float d0 = 0.f, d1 = 0.f;
#pragma unroll
      for (int c = 0; c < DK; c += 8) {
        const f32x4 k0 = *reinterpret_cast<const f32x4*>(kr + c), k1 = *reinterpret_cast<const f32x4*>(kr + c + 4);
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(qr + c), q1 = *reinterpret_cast<const f32x4*>(qr + c + 4);
        d0 += (k0[0] * q0[0] + k0[1] * q0[1]) + (k0[2] * q0[2] + k0[3] * q0[3]);
        d1 += (k1[0] * q1[0] + k1[1] * q1[1]) + (k1[2] * q1[2] + k1[3] * q1[3]);
      }
      sc = (d0 + d1) * scale;
    }
    float mx = sc;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float e = t < n ? __expf(sc - mx) : 0.f;
    float sum = e;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) sum += __shfl_xor(sum, o);
    const float pw = n > 0 ? e / sum : 0.f;           // an empty history attends to nothing (the kernel-per-op path's NaN -> 0)
    if (TRAIN && a.PL && t < a.T) a.PL[((size_t)b * HEADS + h) * a.T + t] = pw;
    float o0 = 0.f, o1 = 0.f;                         // columns lane (head 0) and 64 + lane (head 1)
    const float* vr = a.KV + rb * (2 * D) + D;
    for (int tt = 0; tt < n; ++tt) {
      const float p0 = __shfl(pw, tt), p1 = __shfl(pw, 32 + tt);
      o0 = __builtin_fmaf(p0, vr[(size_t)tt * (2 * D) + lane], o0);
      o1 = __builtin_fmaf(p1, vr[(size_t)tt * (2 * D) + 64 + lane], o1);
    }
    float z0 = o0 + (n > 0 ? a.xlast[(size_t)b * D + lane] : 0.f), z1 = o1 + (n > 0 ? a.xlast[(size_t)b * D + 64 + lane] : 0.f);      // n == 0: x_last = 0 (select_last_kernel)
    const float mean = wave_sum(z0 + z1) * (1.f / (float)D);
    z0 -= mean; z1 -= mean;
    const float rs = 1.f / sqrtf(wave_sum(z0 * z0 + z1 * z1) * (1.f / (float)D) + 1e-5f);
    const float xh0 = z0 * rs, xh1 = z1 * rs;
    const float c0 = xh0 * a.g1[lane] + a.be1[lane], c1 = xh1 * a.g1[64 + lane] + a.be1[64 + lane];
    CLs[s * LQ + lane] = c0;
    CLs[s * LQ + 64 + lane] = c1;
    {
      __bf16 hh, mm, ll;
      split1(c0, hh, mm, ll);
      cpl[s * LDP + lane] = hh;
      if (NP == 3) { cpl[PLANE + s * LDP + lane] = mm; cpl[2 * PLANE + s * LDP + lane] = ll; }
      split1(c1, hh, mm, ll);
      cpl[s * LDP + 64 + lane] = hh;
      if (NP == 3) { cpl[PLANE + s * LDP + 64 + lane] = mm; cpl[2 * PLANE + s * LDP + 64 + lane] = ll; }
    }
    if (TRAIN) {
      if (a.CL) { a.CL[(size_t)b * D + lane] = c0; a.CL[(size_t)b * D + 64 + lane] = c1; }
      if (a.XH1) { a.XH1[(size_t)b * D + lane] = xh0; a.XH1[(size_t)b * D + 64 + lane] = xh1; }
      if (a.RSTD1 && lane == 0) a.RSTD1[b] = rs;
    }
  }
  lds_barrier();
  // ---- f = relu(c W1^T + b1)
  {
    f32x4 acc[1][1];
    acc[0][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm_planes<D, NP, 1, 1, 16>(cpl + p * LDP + 8 * j, launder(a.W1) + lane, wave, acc);
    f32x4 x = acc[0][0] + *reinterpret_cast<const f32x4*>(a.b1 + col);
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = fmaxf(x[r], 0.f);
    store4<NP, PLANE>(fpl + p * LDP + col, x);
    if (TRAIN && a.F1 && rowok) *reinterpret_cast<f32x4*>(a.F1 + growp * D + col) = x;
  }
  lds_barrier();
  // ---- z = f W2^T + b2
  {
    f32x4 acc[1][1];
    acc[0][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm_planes<D, NP, 1, 1, 16>(fpl + p * LDP + 8 * j, launder(a.W2) + lane, wave, acc);
    *reinterpret_cast<f32x4*>(Es + p * LQ + col) = acc[0][0] + *reinterpret_cast<const f32x4*>(a.b2 + col);
  }
  lds_barrier();
  // ---- out = LayerNorm2(z + c)
#pragma unroll 1
  for (int ss = 0; ss < SLOTS; ++ss) {
    const int s = ss * NW + wave, b = b0 + s;
    if (s >= SPB || b >= a.B) continue;
    float z0 = Es[s * LQ + lane] + CLs[s * LQ + lane], z1 = Es[s * LQ + 64 + lane] + CLs[s * LQ + 64 + lane];
    const float mean = wave_sum(z0 + z1) * (1.f / (float)D);
    z0 -= mean; z1 -= mean;
    const float rs = 1.f / sqrtf(wave_sum(z0 * z0 + z1 * z1) * (1.f / (float)D) + 1e-5f);
    const float xh0 = z0 * rs, xh1 = z1 * rs;
    a.out[(size_t)b * a.ldo + lane] = xh0 * a.g2[lane] + a.be2[lane];
    a.out[(size_t)b * a.ldo + 64 + lane] = xh1 * a.g2[64 + lane] + a.be2[64 + lane];
    if (TRAIN) {
      if (a.XH2) { a.XH2[(size_t)b * D + lane] = xh0; a.XH2[(size_t)b * D + 64 + lane] = xh1; }
      if (a.RSTD2 && lane == 0) a.RSTD2[b] = rs;
    }
  }
}

// tile_s[t] = first session b with off[b] >= R * t (t = 0 .. ntiles; B past the last one): session b starts tile off[b] / R when
// its predecessor lies in an earlier window, and every window in between without a session start gets b too (an empty tile)
__global__ void enc_tiles_kernel(const int* __restrict__ off, int B, int R, int ntiles, int* __restrict__ tile_s) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > B) return;
  const int hi = b < B ? off[b] / R : ntiles;                  // b == B: the sentinel and the empty windows behind the last session
  const int lo = b == 0 ? -1 : off[b - 1] / R;
  // (trailing EMPTY histories have off[b] == rows: when rows % R == 0 that is window `ntiles`, which belongs to the sentinel b == B alone --
  // sessions are cut at ntiles - 1, the sentinel always writes tile_s[ntiles])
  const int cap = b < B ? ntiles - 1 : ntiles;
  for (int t = lo + 1; t <= hi && t <= cap; ++t) tile_s[t] = b;
  if (b == B && lo + 1 > ntiles) tile_s[ntiles] = B;
}

template <int D, int DK, bool TRAIN, int NP>
int launch_block(const EncBlockArgs& a, hipStream_t st) {
  using C = EncCfg<D, NP>;
  const size_t smem = C::SMEM;
  allow_lds((enc_block_fwd_kernel<D, DK, TRAIN, NP>), smem);
  int per_cu = (int)((158 * 1024) / smem);             // 160 KB minus the static bookkeeping arrays
  if (per_cu > 4 * C::WPS / C::NW) per_cu = 4 * C::WPS / C::NW;
  if (per_cu < 1) per_cu = 1;
  int grid = num_cus() * per_cu;
  if (grid > a.ntiles) grid = a.ntiles;
  const double M = (double)a.rows;
  const double flops = 2.0 * M * D * D * (5 + (a.Wkv ? 2 : 0)) + 4.0 * M * a.T * D * 0.5;
  double bytes = 4.0 * M * D * (2.0 + (a.out ? 1.0 : 0.0) + (a.KV ? 2.0 : 0.0));
  if (TRAIN) bytes += 4.0 * M * D * ((a.QKV ? 3.0 : 0.0) + (a.XH1 ? 1.0 : 0.0) + (a.F1 ? 1.0 : 0.0) + (a.XH2 ? 1.0 : 0.0));
  static const int dbg_on = INTEL_DEBUG_ENV("INTEL_ENC_DBG", 0);      // phase clocks: debug builds only (common.h)
  EncBlockArgs aa = a;
  static unsigned long long* dbg_buf = nullptr;
  if (dbg_on) {
    if (!dbg_buf) (void)hipMalloc(&dbg_buf, 16 * sizeof(unsigned long long));
    (void)hipMemsetAsync(dbg_buf, 0, 16 * sizeof(unsigned long long), st);
    aa.dbg = dbg_buf;
  }
  LAUNCH_S(a.rows, D, DK, flops, bytes, (enc_block_fwd_kernel<D, DK, TRAIN, NP>), dim3(grid), dim3(C::NT), smem, st, aa);
  INTEL_CHECK_LAUNCH();
  if (dbg_on) {          // per-phase cycles of workgroup 0 (synchronises)
    unsigned long long h[16];
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h, dbg_buf, sizeof(h), hipMemcpyDeviceToHost);
    const int iters = (a.ntiles + grid - 1) / grid;
    fprintf(stderr, "enc_block_fwd train=%d NP=%d grid=%d iters=%d cycles/tile: stage %llu (+bar %llu) qkv %llu (+bar %llu) attn %llu (+bar %llu) w1 %llu (+bar %llu) w2 %llu (+bar %llu) ln2 %llu (+bar %llu) kv %llu (+bar %llu)\n",
            (int)TRAIN, NP, grid, iters, h[0] / iters, h[1] / iters, h[2] / iters, h[3] / iters, h[4] / iters, h[5] / iters, h[6] / iters, h[7] / iters,
            h[8] / iters, h[9] / iters, h[10] / iters, h[11] / iters, h[12] / iters, h[13] / iters);
  }
  return 0;
}

template <int D, int DK, bool TRAIN, int NP>
int launch_last(const EncLastArgs& a, hipStream_t st) {
  using C = EncLastCfg<D, NP>;
  const size_t smem = C::SMEM;
  allow_lds((enc_last_fwd_kernel<D, DK, TRAIN, NP>), smem);
  const double flops = 2.0 * a.B * (double)D * D * 3 + 4.0 * a.B * (double)a.T * D * 0.5;
  const double bytes = 4.0 * a.B * (double)D * (2.0 + a.T);
  LAUNCH_S(a.B, D, DK, flops, bytes, (enc_last_fwd_kernel<D, DK, TRAIN, NP>), dim3(cdiv(a.B, ENC_LAST_SPB)), dim3(C::NT), smem, st, a);
  INTEL_CHECK_LAUNCH();
  return 0;
}

}  // namespace

// INTEL_ENC_FUSED=0: the kernel-per-op encoder everywhere (read once per process)
bool enc_fused_supported(int T, int dm, int heads) {
  static const int on = [] { const char* e = getenv("INTEL_ENC_FUSED"); return (e && e[0] == '0') ? 0 : 1; }();
  return on && T >= 1 && T <= 32 && dm == 128 && heads == 2;
}

int enc_tile_rows(int T) { return 65 - T; }

int launch_enc_tiles(const int* off, int B, int T, int rows, int* tile_s, hipStream_t st) {
  const int R = enc_tile_rows(T), ntiles = cdiv(rows, R);
  LAUNCH(enc_tiles_kernel, dim3(cdiv(B + 1, 256)), dim3(256), 0, st, off, B, R, ntiles, tile_s);
  INTEL_CHECK_LAUNCH();
  return 0;
}

int launch_enc_block_fwd(const EncBlockFwd& f, hipStream_t st) {
  if (f.B <= 0 || f.rows <= 0) return 0;
  INTEL_CHECK_ARG(enc_fused_supported(f.T, f.dm, f.heads), "enc_block_fwd: unsupported shape T=%d dm=%d heads=%d", f.T, f.dm, f.heads);
  INTEL_CHECK_ARG(f.C, "enc_block_fwd: the LayerNorm1 output buffer is required");
  EncBlockArgs a;
  a.X = f.X; a.rows = f.rows; a.B = f.B; a.T = f.T; a.ntiles = cdiv(f.rows, enc_tile_rows(f.T));
  a.off = f.off; a.tile_s = f.tile_s;
  a.Wqkv = reinterpret_cast<const uint4*>(f.Wqkv); a.W1 = reinterpret_cast<const uint4*>(f.W1); a.W2 = reinterpret_cast<const uint4*>(f.W2);
  a.Wkv = reinterpret_cast<const uint4*>(f.Wkv);
  a.bqkv = f.bqkv; a.b1 = f.b1; a.b2 = f.b2; a.bkv = f.bkv; a.g1 = f.g1; a.be1 = f.be1; a.g2 = f.g2; a.be2 = f.be2;
  a.C = f.C; a.out = f.out; a.xlast = f.xlast; a.KV = f.Wkv ? f.KV : nullptr;
  INTEL_CHECK_ARG(!f.Wkv || (f.KV && f.bkv), "enc_block_fwd: key / value projection without its output or bias");
  const bool tr = f.train != 0;
  a.QKV = tr ? f.QKV : nullptr; a.LSE = tr ? f.LSE : nullptr; a.XH1 = tr ? f.XH1 : nullptr; a.RSTD1 = tr ? f.RSTD1 : nullptr;
  a.F1 = tr ? f.F1 : nullptr; a.XH2 = tr ? f.XH2 : nullptr; a.RSTD2 = tr ? f.RSTD2 : nullptr;
  a.dbg = nullptr;
  if (gemm_planes() == 1) return tr ? launch_block<128, 64, true, 1>(a, st) : launch_block<128, 64, false, 1>(a, st);
  return tr ? launch_block<128, 64, true, 3>(a, st) : launch_block<128, 64, false, 3>(a, st);
}

int launch_enc_last_fwd(const EncLastFwd& f, hipStream_t st) {
  if (f.B <= 0) return 0;
  INTEL_CHECK_ARG(enc_fused_supported(f.T, f.dm, f.heads), "enc_last_fwd: unsupported shape T=%d dm=%d heads=%d", f.T, f.dm, f.heads);
  EncLastArgs a;
  a.xlast = f.xlast; a.KV = f.KV; a.off = f.off; a.len = f.len; a.B = f.B; a.T = f.T;
  a.Wq = reinterpret_cast<const uint4*>(f.Wq); a.W1 = reinterpret_cast<const uint4*>(f.W1); a.W2 = reinterpret_cast<const uint4*>(f.W2);
  a.bq = f.bq; a.b1 = f.b1; a.b2 = f.b2; a.g1 = f.g1; a.be1 = f.be1; a.g2 = f.g2; a.be2 = f.be2;
  a.out = f.out; a.ldo = f.ldo;
  const bool tr = f.train != 0;
  a.QL = tr ? f.QL : nullptr; a.PL = tr ? f.PL : nullptr; a.CL = tr ? f.CL : nullptr; a.XH1 = tr ? f.XH1 : nullptr; a.RSTD1 = tr ? f.RSTD1 : nullptr;
  a.F1 = tr ? f.F1 : nullptr; a.XH2 = tr ? f.XH2 : nullptr; a.RSTD2 = tr ? f.RSTD2 : nullptr;
  if (gemm_planes() == 1) return tr ? launch_last<128, 64, true, 1>(a, st) : launch_last<128, 64, false, 1>(a, st);
  return tr ? launch_last<128, 64, true, 3>(a, st) : launch_last<128, 64, false, 3>(a, st);
}
