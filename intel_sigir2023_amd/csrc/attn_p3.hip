// The general attention kernels (attn.hip: lists / histories longer than 64, head dims 64 and 128) with EVERY product on the bf16 matrix pipe
// at fp32 accuracy: both operands of a product are split into hi + mid + lo bf16 planes and the six plane products of weight >= 2^-16 are
// summed smallest first (planes::mma<3>, the same arithmetic as the row GEMMs) -- 6 x v_mfma_f32_16x16x32_bf16 (16 cycles each) per 32-deep block
// instead of 8 x v_mfma_f32_16x16x4_f32 (32 cycles each): 2.7 x fewer matrix-pipe cycles.  Restates modules/layers.py:50-60 exactly as attn.hip does
// (same masks, same statistics, same dS scratch): the kernels are drop-ins for attn_fwd_kernel / attn_bwd_dkv_kernel / attn_bwd_dq_ds_kernel.
//
// What makes the split pay (an in-register split of every LDS fragment does not):
//   * the staged tile (32 rows of K / V, or of Q / dO) is split ONCE, on its way from the staging registers to LDS, into three bf16 plane images
//     (row-major, 16-byte chunks XOR-swizzled), and every wave reads finished plane fragments;
//   * "row" operands (S = K Q^T, dP = V dO^T: the k index runs along the head dim) are ONE ds_read_b128 per plane and 32-deep block;
//   * "transposed" operands (O^T = V^T P^T, dV^T = dO^T P, dK^T = Q^T dS, dQ^T = K^T dS^T: the k index runs over the 32 staged rows) come out of
//     LDS already transposed (ds_read_b64_tr_b16: a 16-lane group reads a 4-row x 16-column block column-major), two reads per plane;
//   * the probabilities / dS values leave the S-type accumulators in exactly the k order the transposed read delivers (lane group g holds staged rows
//     4g .. 4g+3 of the first 16-row tile and 16+4g .. 16+4g+3 of the second): eight values per lane, split in registers, are the B operand.
#include "kernels.h"
#include "planes.h"
#include <stdlib.h>

namespace {

#define AP_QB 64   // rows (queries, or keys in the dK/dV kernel) owned by a workgroup: 4 waves x 16
#define AP_KB 32   // rows staged per iteration = the k depth of one transposed product
#define LOG2E_F 1.4426950408889634f
#ifndef AP_DKV_DUAL
#define AP_DKV_DUAL 0      // the dK/dV kernel's products as two interleaved accumulator chains (0: one chain, one fragment set in flight)
#endif
#ifndef AP_DKV_PRE
#define AP_DKV_PRE 0      // (measured: the prefetch registers cost a wave per SIMD, 252 -> 320 us at head dim 64)
#endif
#ifndef AP_DKV_OCC
#define AP_DKV_OCC 3       // its waves per SIMD at head dim 64 (register budget 168)
#endif
#define LN2_F 0.6931471805599453f

typedef short ap_s16x4 __attribute__((ext_vector_type(4)));
typedef short ap_s16x8 __attribute__((ext_vector_type(8)));

// One staged tile = three bf16 planes of [32 rows][dk], dk = 16 DKT in {64, 128}.  Byte offset of 16-byte chunk ch of row `row` inside a plane:
// the chunk index is XORed with a function of the row chosen for the lane groups the LDS really services together (MI355X_MICROARCH.md, LDS):
//   (a) row operands, ds_read_b128, lane (p, g) reads row p at chunk 4c + g: the groups are lanes {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} (+32),
//       i.e. rows {0-3, 12-15} at chunk 4c + g and rows {4-11} at chunk 4c + (g ^ 1) -- the first set maps to swizzles 0..7, the second to 8..15;
//   (b) transposed operands, ds_read_b64_tr_b16, half a wave reads rows 0..7 (or 8..15) at a chunk PAIR: swz >> 1 differs over each of the two sets.
// Both hold for swz(r) = 2 (r & 3) | (r >> 3) | 8 ((r >> 2) ^ (r >> 3)) at 256-byte rows (the 256-byte bank window is one row), and for the same
// function of r >> 1 on three bits at 128-byte rows (two rows per window: the row's parity picks the half).
template <int DKT>
struct P3Tile {
  static constexpr int RB = DKT * 32;                 // bytes per row
  static constexpr int NCH = DKT * 2;                 // chunks per row
  static constexpr int PLANE = AP_KB * RB;
  static constexpr int BYTES = 3 * PLANE;
  __device__ __forceinline__ static int swz(int row) {
    if (DKT == 8) return ((row & 3) << 1) | ((row >> 3) & 1) | ((((row >> 2) ^ (row >> 3)) & 1) << 3);
    const int u = (row >> 1) & 7;
    return ((u & 1) << 1) | ((u >> 2) & 1) | ((((u >> 1) ^ (u >> 2)) & 1) << 2);
  }
  __device__ __forceinline__ static int off(int row, int ch) { return RB * row + 16 * (ch ^ swz(row)); }
};

__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    __bf16 hh, mm, ll;
    planes::split1(a[i], hh, mm, ll);
    h[i] = hh; m[i] = mm; l[i] = ll;
    planes::split1(b[i], hh, mm, ll);
    h[4 + i] = hh; m[4 + i] = mm; l[4 + i] = ll;
  }
}

// rows [r0, r0 + 32) of one third of qkv (or of dout), zero padded, global -> registers -> three plane images; a thread owns whole 8-dim chunks
template <int DKT>
struct P3Stage {
  static constexpr int NCH = DKT * 2, NCK = DKT / 4;      // chunks per thread: 32 * NCH / 256
  f32x4 v[NCK][2];
  __device__ __forceinline__ void load(const float* __restrict__ base, int ldg, int coff, int r0, int nrow, int tid) {
#pragma unroll
    for (int n = 0; n < NCK; ++n) {
      const int i = tid + n * 256, r = i / NCH, ch = i - r * NCH, row = r0 + r;
      v[n][0] = v[n][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (row < nrow) {
        const float* p = base + (size_t)row * ldg + coff + 8 * ch;
        v[n][0] = *reinterpret_cast<const f32x4*>(p);
        v[n][1] = *reinterpret_cast<const f32x4*>(p + 4);
      }
    }
  }
  __device__ __forceinline__ void store(unsigned char* tile, int tid) const {
    using TL = P3Tile<DKT>;
#pragma unroll
    for (int n = 0; n < NCK; ++n) {
      const int i = tid + n * 256, r = i / NCH, ch = i - r * NCH;
      bf16x8 h, m, l;
#if AP_ABLATE & 2
      for (int e = 0; e < 4; ++e) { h[e] = (__bf16)v[n][0][e]; h[4 + e] = (__bf16)v[n][1][e]; }
      m = h; l = h;
#else
      split8(v[n][0], v[n][1], h, m, l);
#endif
      unsigned char* d = tile + TL::off(r, ch);
      *reinterpret_cast<bf16x8*>(d) = h;
      *reinterpret_cast<bf16x8*>(d + TL::PLANE) = m;
      *reinterpret_cast<bf16x8*>(d + 2 * TL::PLANE) = l;
    }
  }
};

// the lane's own row (query / key) as the B operand of the S-type products: dims 32c + 8g .. +7, three planes per 32-deep block
struct Frag3 { bf16x8 h, m, l; };
#define AP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#ifndef AP_ABLATE
#define AP_ABLATE 0      // measurement builds (tools/): 1 = one plane product instead of six (results wrong); 2 = no staging split (hi plane only)
#endif
__device__ __forceinline__ f32x4 ap_mma3(const bf16x8& ah, const bf16x8& am, const bf16x8& al, const bf16x8& bh, const bf16x8& bm, const bf16x8& bl, f32x4 c) {
#if AP_ABLATE & 1
  return AP_MFMA(ah, bh, c);
#else
  return planes::mma<3>(ah, am, al, bh, bm, bl, c);
#endif
}
// Two independent six-product chains (planes::mma<3>'s order: smallest products first), interleaved: a v_mfma that accumulates into its
// predecessor's result waits for that result's passes, so the two accumulators alternate.
__device__ __forceinline__ void mma3x2(const Frag3& a1, const Frag3& b1, f32x4& c1, const Frag3& a2, const Frag3& b2, f32x4& c2) {
#if AP_ABLATE & 1
  c1 = AP_MFMA(a1.h, b1.h, c1); c2 = AP_MFMA(a2.h, b2.h, c2);
  return;
#endif
  c1 = AP_MFMA(a1.m, b1.m, c1); c2 = AP_MFMA(a2.m, b2.m, c2);
  c1 = AP_MFMA(a1.h, b1.l, c1); c2 = AP_MFMA(a2.h, b2.l, c2);
  c1 = AP_MFMA(a1.l, b1.h, c1); c2 = AP_MFMA(a2.l, b2.h, c2);
  c1 = AP_MFMA(a1.h, b1.m, c1); c2 = AP_MFMA(a2.h, b2.m, c2);
  c1 = AP_MFMA(a1.m, b1.h, c1); c2 = AP_MFMA(a2.m, b2.h, c2);
  c1 = AP_MFMA(a1.h, b1.h, c1); c2 = AP_MFMA(a2.h, b2.h, c2);
}

template <int DKT>
__device__ __forceinline__ void own_row_planes(const float* __restrict__ rowp, bool ok, int g, Frag3 (&f)[DKT / 2]) {
#pragma unroll
  for (int c = 0; c < DKT / 2; ++c) {
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, b = a;
    if (ok) {
      a = *reinterpret_cast<const f32x4*>(rowp + 32 * c + 8 * g);
      b = *reinterpret_cast<const f32x4*>(rowp + 32 * c + 8 * g + 4);
    }
    split8(a, b, f[c].h, f[c].m, f[c].l);
  }
}

// row fragment: staged row 16 t + (lane & 15), dims 32c + 8g .. +7, three planes.  roff = P3Tile::off(lane & 15, lane >> 4): chunk 4c + g of the
// row is roff ^ 64c (the low two chunk bits belong to g)
template <int DKT>
__device__ __forceinline__ Frag3 row_frag(const unsigned char* tile, int roff, int t, int c) {
  using TL = P3Tile<DKT>;
  const unsigned char* p = tile + (roff ^ (64 * c)) + t * 16 * TL::RB;
  return Frag3{*reinterpret_cast<const bf16x8*>(p), *reinterpret_cast<const bf16x8*>(p + TL::PLANE), *reinterpret_cast<const bf16x8*>(p + 2 * TL::PLANE)};
}
// S-type tile products of TWO 16-row tiles (tile rows: the A operand, k along the head dim) against B operands held in registers, 32 dims per step;
// the fragments of step c + 1 are requested before the products of step c
template <int DKT, bool PRE = true>
__device__ __forceinline__ void row_mma2(const unsigned char* tile1, int t1, const Frag3 (&b1)[DKT / 2], f32x4& acc1, const unsigned char* tile2, int t2,
                                         const Frag3 (&b2)[DKT / 2], f32x4& acc2, int roff) {
  Frag3 a1 = row_frag<DKT>(tile1, roff, t1, 0), a2 = row_frag<DKT>(tile2, roff, t2, 0);
#pragma unroll
  for (int c = 0; c < DKT / 2; ++c) {
    Frag3 n1 = a1, n2 = a2;
    if (PRE && c + 1 < DKT / 2) {
      n1 = row_frag<DKT>(tile1, roff, t1, c + 1);
      n2 = row_frag<DKT>(tile2, roff, t2, c + 1);
    }
    mma3x2(a1, b1[c], acc1, a2, b2[c], acc2);
    if (!PRE && c + 1 < DKT / 2) {
      n1 = row_frag<DKT>(tile1, roff, t1, c + 1);
      n2 = row_frag<DKT>(tile2, roff, t2, c + 1);
    }
    a1 = n1; a2 = n2;
  }
}

// transposed fragment: dims 16 dt + (lane & 15) on the accumulator rows, k = the 32 staged rows (slot (g, s): row 4g + s for s < 4, 16 + 4g + s - 4 above)
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* plane, int off_lo, int off_hi) {
  const ap_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ap_s16x4*)(plane + off_lo));
  const ap_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ap_s16x4*)(plane + off_hi));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
template <int DKT>
__device__ __forceinline__ Frag3 tr_frag3(const unsigned char* tile, int tlo, int thi, int dt) {      // tlo / thi: the lane's addresses for dt = 0 (dt: ^ 32 dt)
  using TL = P3Tile<DKT>;
  const int lo = tlo ^ (32 * dt), hi = thi ^ (32 * dt);
  return Frag3{tr_frag(tile, lo, hi), tr_frag(tile + TL::PLANE, lo, hi), tr_frag(tile + 2 * TL::PLANE, lo, hi)};
}
// acc1[dt] += (tile1^T)[dims of tile dt][32 rows] * b1[32 rows][the lane's column], acc2[dt] likewise from tile2 / b2: two chains per dim tile
template <int DKT, bool PRE = true>
__device__ __forceinline__ void tr_mma2(f32x4 (&acc1)[DKT], const unsigned char* tile1, const Frag3& b1, f32x4 (&acc2)[DKT], const unsigned char* tile2,
                                        const Frag3& b2, int tlo, int thi) {
  Frag3 a1 = tr_frag3<DKT>(tile1, tlo, thi, 0), a2 = tr_frag3<DKT>(tile2, tlo, thi, 0);
#pragma unroll
  for (int dt = 0; dt < DKT; ++dt) {
    Frag3 n1 = a1, n2 = a2;
    if (PRE && dt + 1 < DKT) {
      n1 = tr_frag3<DKT>(tile1, tlo, thi, dt + 1);
      n2 = tr_frag3<DKT>(tile2, tlo, thi, dt + 1);
    }
    mma3x2(a1, b1, acc1[dt], a2, b2, acc2[dt]);
    if (!PRE && dt + 1 < DKT) {
      n1 = tr_frag3<DKT>(tile1, tlo, thi, dt + 1);
      n2 = tr_frag3<DKT>(tile2, tlo, thi, dt + 1);
    }
    a1 = n1; a2 = n2;
  }
}
// single-chain forms (fewer fragment registers: one set in flight)
template <int DKT>
__device__ __forceinline__ f32x4 row_mma1(const unsigned char* tile, int t, const Frag3 (&b)[DKT / 2], f32x4 acc, int roff) {
#pragma unroll
  for (int c = 0; c < DKT / 2; ++c) {
    const Frag3 a = row_frag<DKT>(tile, roff, t, c);
    acc = ap_mma3(a.h, a.m, a.l, b[c].h, b[c].m, b[c].l, acc);
  }
  return acc;
}
template <int DKT>
__device__ __forceinline__ void tr_mma1(f32x4 (&acc)[DKT], const unsigned char* tile, const Frag3& b, int tlo, int thi) {
#pragma unroll
  for (int dt = 0; dt < DKT; ++dt) {
    const Frag3 a = tr_frag3<DKT>(tile, tlo, thi, dt);
    acc[dt] = ap_mma3(a.h, a.m, a.l, b.h, b.m, b.l, acc[dt]);
  }
}
// one tile, one B operand: the dim tiles go in pairs
template <int DKT>
__device__ __forceinline__ void tr_mma(f32x4 (&acc)[DKT], const unsigned char* tile, const Frag3& b, int tlo, int thi) {
  Frag3 a1 = tr_frag3<DKT>(tile, tlo, thi, 0), a2 = tr_frag3<DKT>(tile, tlo, thi, 1);
#pragma unroll
  for (int dt = 0; dt < DKT; dt += 2) {
    Frag3 n1 = a1, n2 = a2;
    if (dt + 2 < DKT) {
      n1 = tr_frag3<DKT>(tile, tlo, thi, dt + 2);
      n2 = tr_frag3<DKT>(tile, tlo, thi, dt + 3);
    }
    mma3x2(a1, b, acc[dt], a2, b, acc[dt + 1]);
    a1 = n1; a2 = n2;
  }
}
// the lane's addresses of the transposed reads for dim tile 0: lane 4q + pp of a 16-lane group names row q and columns 4pp .. 4pp+3 of its block
template <int DKT>
__device__ __forceinline__ void tr_addr(int lane, int& tlo, int& thi) {
  using TL = P3Tile<DKT>;
  const int p = lane & 15, g = lane >> 4, q = p >> 2, pp = p & 3;
  tlo = TL::off(4 * g + q, pp >> 1) + 8 * (pp & 1);
  thi = TL::off(16 + 4 * g + q, pp >> 1) + 8 * (pp & 1);
}

// the lane's share of one output row: accumulator dt holds dims 16 dt + 4g .. +3
template <int DKT>
__device__ __forceinline__ void dim_store(float* __restrict__ rowp, const f32x4 (&acc)[DKT], int g, float mul) {
#pragma unroll
  for (int dt = 0; dt < DKT; ++dt) *reinterpret_cast<f32x4*>(rowp + 16 * dt + 4 * g) = acc[dt] * mul;
}

struct ApBlock { int bh, y; };
__device__ __forceinline__ ApBlock ap_block(int nbh, int ny) {      // (as attn.hip: the blocks of one (session, head) pair share an XCD)
  const int id = blockIdx.x;
  if ((nbh & 7) == 0) {
    const int slot = id >> 3, g = slot / ny;
    return ApBlock{g * 8 + (id & 7), slot - g * ny};
  }
  const int bh = id / ny;
  return ApBlock{bh, id - bh * ny};
}

// ------------------------------------------------------------------------------------------
// forward (flash-style; attn_fwd_kernel's structure)
// ------------------------------------------------------------------------------------------
template <int DKT>
__global__ __launch_bounds__(256, DKT == 8 ? 2 : 3) void attn_fwd_p3_kernel(const float* __restrict__ qkv, int T, int d, int heads,
                                                                            const int* __restrict__ key_len, float scale, float* __restrict__ out,
                                                                            float* __restrict__ lse, const int* __restrict__ row_off) {
  using TL = P3Tile<DKT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_p3[];
  unsigned char* Ks = smem_p3;
  unsigned char* Vs = smem_p3 + TL::BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, g = lane >> 4;
  const int ny = (T + AP_QB - 1) / AP_QB;
  const ApBlock blk = ap_block((int)gridDim.x / ny, ny);
  const int b = blk.bh / heads, h = blk.bh - b * heads;
  const int dk = d / heads, ldg = 3 * d;
  const int nkeys = key_len ? min(key_len[b], T) : T;
  const int nrow = row_off ? nkeys : T;
  const size_t row0 = row_off ? (size_t)row_off[b] : (size_t)b * T;
  if (blk.y * AP_QB >= nrow) return;
  const float* base = qkv + row0 * ldg;
  const int q = blk.y * AP_QB + wave * 16 + p;
  Frag3 qf[DKT / 2];
  own_row_planes<DKT>(base + (size_t)q * ldg + h * dk, q < nrow, g, qf);
  f32x4 oT[DKT];
#pragma unroll
  for (int i = 0; i < DKT; ++i) oT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const int roff = TL::off(p, g);
  int tlo, thi;
  tr_addr<DKT>(lane, tlo, thi);
  const float scale2 = scale * LOG2E_F;

  P3Stage<DKT> kreg, vreg;
  kreg.load(base, ldg, d + h * dk, 0, nrow, tid);
  vreg.load(base, ldg, 2 * d + h * dk, 0, nrow, tid);
  for (int kb = 0; kb < nkeys; kb += AP_KB) {
    __syncthreads();
    kreg.store(Ks, tid);
    vreg.store(Vs, tid);
    __syncthreads();
    if (kb + AP_KB < nkeys) {
      kreg.load(base, ldg, d + h * dk, kb + AP_KB, nrow, tid);
      vreg.load(base, ldg, 2 * d + h * dk, kb + AP_KB, nrow, tid);
    }
    if (blk.y * AP_QB + wave * 16 >= nrow) continue;      // a wave whose 16 queries are all past the list only helps staging
    f32x4 st[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    row_mma2<DKT>(Ks, 0, qf, st[0], Ks, 1, qf, st[1], roff);      // (a second tile past the keys is zero rows / masked below)
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kb + kt * 16 + 4 * g + r;
        const float v = key < nkeys ? st[kt][r] * scale2 : -INFINITY;      // base-2 logits (v_exp_f32 is 2^x)
        st[kt][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = planes::gmax16(mx);
    const float m_new = fmaxf(m_run, mx);        // finite: this block holds >= 1 valid key
    const float corr = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
    float ps = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(st[kt][r] - m_new);   // 2^(-inf) = 0 for masked keys
        st[kt][r] = e;
        ps += e;
      }
    ps = planes::gsum16(ps);
    l_run = l_run * corr + ps;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < DKT; ++i) oT[i] *= corr;
    Frag3 pf;
    split8(st[0], st[1], pf.h, pf.m, pf.l);
    tr_mma<DKT>(oT, Vs, pf, tlo, thi);
  }
  if (q < nrow) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    dim_store<DKT>(out + (row0 + q) * d + h * dk, oT, g, inv);
    if (lane < 16) lse[((size_t)b * heads + h) * T + q] = l_run > 0.f ? m_run * LN2_F + logf(l_run) : INFINITY;      // natural-log statistic, as attn.hip stores it
  }
}

// ------------------------------------------------------------------------------------------
// backward, dK / dV (+ the dS tiles of the dQ kernel): wave owns 16 keys, sweeps query blocks (attn_bwd_dkv_kernel's structure)
// ------------------------------------------------------------------------------------------
template <int DKT>
__global__ __launch_bounds__(256, DKT == 8 ? 2 : AP_DKV_OCC) void attn_bwd_dkv_p3_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                 const float* __restrict__ lse, const float* __restrict__ dsum, int T, int d, int heads,
                                                                 const int* __restrict__ key_len, float scale, float* __restrict__ dqkv,
                                                                 float* __restrict__ dS, int ldS, const int* __restrict__ row_off) {
  using TL = P3Tile<DKT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_p3[];
  unsigned char* Qs = smem_p3;
  unsigned char* Os = smem_p3 + TL::BYTES;
  float* Ls = reinterpret_cast<float*>(smem_p3 + 2 * TL::BYTES);   // [32] lse
  float* Ds = Ls + AP_KB;                                           // [32] dsum
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, g = lane >> 4;
  const int ny = (T + AP_QB - 1) / AP_QB;
  const ApBlock blk = ap_block((int)gridDim.x / ny, ny);
  const int b = blk.bh / heads, h = blk.bh - b * heads;
  const int dk = d / heads, ldg = 3 * d;
  const int nkeys = key_len ? min(key_len[b], T) : T;
  const int nrow = row_off ? nkeys : T;
  const size_t row0 = row_off ? (size_t)row_off[b] : (size_t)b * T;
  if (blk.y * AP_QB >= nrow) return;
  const float* base = qkv + row0 * ldg;
  const int key = blk.y * AP_QB + wave * 16 + p;
  const bool kok = key < nrow;
  Frag3 kf[DKT / 2], vf[DKT / 2];
  own_row_planes<DKT>(base + (size_t)key * ldg + d + h * dk, kok, g, kf);
  own_row_planes<DKT>(base + (size_t)key * ldg + 2 * d + h * dk, kok, g, vf);
  f32x4 dkT[DKT], dvT[DKT];
#pragma unroll
  for (int i = 0; i < DKT; ++i) {
    dkT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    dvT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool key_live = key < nkeys;          // masked keys get exactly zero gradient
  const bool wave_live = (blk.y * AP_QB + wave * 16) < nkeys;
  const int roff = TL::off(p, g);
  int tlo, thi;
  tr_addr<DKT>(lane, tlo, thi);
  const float scale2 = scale * LOG2E_F;
  const float* dob = dout + row0 * d;
  // head dim 64: the next block's Q / dO rows and statistics are requested before this block's products (with the products on the bf16 pipe a
  // block's MFMAs no longer cover the load latency of the next); head dim 128 has no registers left for that (K / V planes: 96)
  constexpr bool PRE = AP_DKV_PRE && DKT == 4;
  P3Stage<DKT> qreg, oreg;
  float lreg = INFINITY, dreg = 0.f;
  auto load_block = [&](int qb) {
    qreg.load(base, ldg, h * dk, qb, nrow, tid);
    oreg.load(dob, d, h * dk, qb, nrow, tid);
    const int qq = qb + tid;
    const bool ok = tid < AP_KB && qq < nrow;
    lreg = ok ? lse[((size_t)b * heads + h) * T + qq] * LOG2E_F : INFINITY;      // base-2 statistics: p = exp2(s * scale * log2 e - lse * log2 e)
    dreg = ok ? dsum[((size_t)b * heads + h) * T + qq] : 0.f;
  };
  if (PRE) load_block(0);
  for (int qb = 0; qb < nrow; qb += AP_KB) {
    __syncthreads();
    if (!PRE) load_block(qb);
    qreg.store(Qs, tid);
    oreg.store(Os, tid);
    if (tid < AP_KB) {
      Ls[tid] = lreg;
      Ds[tid] = dreg;
    }
    __syncthreads();
    if (PRE && qb + AP_KB < nrow) load_block(qb + AP_KB);
    if (!wave_live) continue;
    f32x4 pr[2], ds[2];
    float* dSp = dS ? dS + ((size_t)blk.bh * T + qb) * ldS + key : nullptr;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, dp = sa;
      if (qb + qt * 16 < nrow) {     // S[query][key], dP[query][key]
#if AP_DKV_DUAL
        row_mma2<DKT, false>(Qs, qt, kf, sa, Os, qt, vf, dp, roff);
#else
        sa = row_mma1<DKT>(Qs, qt, kf, sa, roff);
        dp = row_mma1<DKT>(Os, qt, vf, dp, roff);
#endif
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = qt * 16 + 4 * g + r;
        const float pv = key_live ? __builtin_amdgcn_exp2f(sa[r] * scale2 - Ls[ql]) : 0.f;      // (rows past the list: Ls = inf -> 0)
        pr[qt][r] = pv;
        ds[qt][r] = pv * (dp[r] - Ds[ql]) * scale;
        // the dS tile for the dQ = dS K kernel (row = query, ldS floats per row); keys >= nkeys / queries >= T are never read
        if (dS && qb + ql < nrow && kok) dSp[ql * ldS] = ds[qt][r];
      }
    }
    Frag3 pf, df;
    split8(pr[0], pr[1], pf.h, pf.m, pf.l);
    split8(ds[0], ds[1], df.h, df.m, df.l);
    // dV^T[dim][key] += dO^T P, dK^T[dim][key] += Q^T dS
#if AP_DKV_DUAL
    tr_mma2<DKT, false>(dvT, Os, pf, dkT, Qs, df, tlo, thi);
#else
    tr_mma1<DKT>(dvT, Os, pf, tlo, thi);
    tr_mma1<DKT>(dkT, Qs, df, tlo, thi);
#endif
  }
  if (kok) {
    float* drow = dqkv + (row0 + key) * ldg + h * dk;
    dim_store<DKT>(drow + d, dkT, g, 1.f);
    dim_store<DKT>(drow + 2 * d, dvT, g, 1.f);
  }
}

// ------------------------------------------------------------------------------------------
// backward, dQ = dS K from the stored dS tiles (attn_bwd_dq_ds_kernel's structure)
// ------------------------------------------------------------------------------------------
template <int DKT>
__global__ __launch_bounds__(256, 4) void attn_bwd_dq_ds_p3_kernel(const float* __restrict__ qkv, const float* __restrict__ dS, int ldS, int T, int d,
                                                                   int heads, const int* __restrict__ key_len, float* __restrict__ dqkv,
                                                                   const int* __restrict__ row_off) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_p3[];
  unsigned char* Ks = smem_p3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, g = lane >> 4;
  const int ny = (T + AP_QB - 1) / AP_QB;
  const ApBlock blk = ap_block((int)gridDim.x / ny, ny);
  const int b = blk.bh / heads, h = blk.bh - b * heads;
  const int dk = d / heads, ldg = 3 * d;
  const int nkeys = key_len ? min(key_len[b], T) : T;
  const int nrow = row_off ? nkeys : T;
  const size_t row0 = row_off ? (size_t)row_off[b] : (size_t)b * T;
  if (blk.y * AP_QB >= nrow) return;
  const float* base = qkv + row0 * ldg;
  const int q = blk.y * AP_QB + wave * 16 + p;
  const bool qok = q < nrow;
  const float* dSq = dS + ((size_t)blk.bh * T + (qok ? q : 0)) * ldS;
  f32x4 dqT[DKT];
#pragma unroll
  for (int i = 0; i < DKT; ++i) dqT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  int tlo, thi;
  tr_addr<DKT>(lane, tlo, thi);
  P3Stage<DKT> kreg;
  kreg.load(base, ldg, d + h * dk, 0, nrow, tid);
  for (int kb = 0; kb < nkeys; kb += AP_KB) {
    __syncthreads();
    kreg.store(Ks, tid);
    __syncthreads();
    if (kb + AP_KB < nkeys) kreg.load(base, ldg, d + h * dk, kb + AP_KB, nrow, tid);
    if (blk.y * AP_QB + wave * 16 >= nrow) continue;
    f32x4 dsT[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int key0 = kb + kt * 16 + 4 * g;
      dsT[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (qok && key0 < ldS && key0 < nkeys) dsT[kt] = *reinterpret_cast<const f32x4*>(dSq + key0);
#pragma unroll
      for (int r = 0; r < 4; ++r) dsT[kt][r] = key0 + r < nkeys ? dsT[kt][r] : 0.f;      // masked / padding keys: nothing was stored
    }
    Frag3 df;
    split8(dsT[0], dsT[1], df.h, df.m, df.l);
    tr_mma<DKT>(dqT, Ks, df, tlo, thi);      // dQ^T[dim][query] += K^T dS^T
  }
  if (qok) dim_store<DKT>(dqkv + (row0 + q) * ldg + h * dk, dqT, g, 1.f);
}

}  // namespace

// fp32 parity mode (three-plane products), lists / histories longer than 64 (the shorter ones have the whole-sequence kernels), head dims 64 / 128
bool attn_p3_supported(int T, int dk) {
  static const int on = [] { const char* e = getenv("INTEL_ATTN_P3"); return (e && e[0] == '0') ? 0 : 1; }();
  return on && gemm_planes() == 3 && T > 64 && (dk == 64 || dk == 128);
}

#define AP_DISPATCH(DKT_RT, CALL) \
  if ((DKT_RT) == 4) { constexpr int DKT = 4; CALL; } else { constexpr int DKT = 8; CALL; }

int launch_attn_p3_fwd(const float* qkv, int B, int T, int d, int heads, const int* key_len, float* out, float* lse, hipStream_t st, const int* row_off) {
  const int dk = d / heads, dkt = dk / 16;
  INTEL_CHECK_ARG(dk == 64 || dk == 128, "attention (bf16-plane kernels): head dim %d", dk);
  const float scale = 1.0f / sqrtf((float)dk);
  dim3 grid(B * heads * cdiv(T, AP_QB));
  AP_DISPATCH(dkt, {
    size_t smem = (size_t)2 * P3Tile<DKT>::BYTES;
    allow_lds(attn_fwd_p3_kernel<DKT>, smem);
    LAUNCH_S(B * heads, T, dk, 4.0 * B * T * (double)T * d, 16.0 * B * T * (double)d, attn_fwd_p3_kernel<DKT>, grid, dim3(256), smem, st, qkv, T, d, heads, key_len, scale, out, lse, row_off);
  });
  INTEL_CHECK_LAUNCH();
  return 0;
}

// the dS scheme's second and third kernel (the row sums dsum are in place: attn_dsum_kernel)
int launch_attn_p3_bwd(const float* qkv, const float* dout, const float* lse, const float* dsum, int B, int T, int d, int heads, const int* key_len,
                       float* dqkv, float* dS, int ldS, hipStream_t st, const int* row_off) {
  const int dk = d / heads, dkt = dk / 16;
  INTEL_CHECK_ARG(dk == 64 || dk == 128, "attention (bf16-plane kernels): head dim %d", dk);
  const float scale = 1.0f / sqrtf((float)dk);
  dim3 grid(B * heads * cdiv(T, AP_QB));
  AP_DISPATCH(dkt, {
    size_t smem = (size_t)2 * P3Tile<DKT>::BYTES + 2 * AP_KB * sizeof(float);
    allow_lds(attn_bwd_dkv_p3_kernel<DKT>, smem);
    LAUNCH_S(B * heads, T, dk, 8.0 * B * T * (double)T * d, 24.0 * B * T * (double)d, attn_bwd_dkv_p3_kernel<DKT>, grid, dim3(256), smem, st, qkv, dout, lse, dsum, T, d, heads,
             key_len, scale, dqkv, dS, ldS, row_off);
  });
  INTEL_CHECK_LAUNCH();
  AP_DISPATCH(dkt, {
    size_t smem = (size_t)P3Tile<DKT>::BYTES;
    allow_lds(attn_bwd_dq_ds_p3_kernel<DKT>, smem);
    LAUNCH_S(B * heads, T, dk, 2.0 * B * T * (double)T * d, 8.0 * B * T * (double)d + 4.0 * B * heads * (double)T * T, attn_bwd_dq_ds_p3_kernel<DKT>, grid, dim3(256), smem, st, qkv, dS,
             ldS, T, d, heads, key_len, dqkv, row_off);
  });
  INTEL_CHECK_LAUNCH();
  return 0;
}
