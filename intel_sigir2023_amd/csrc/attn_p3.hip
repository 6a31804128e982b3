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

typedef short ap_s16x4 __attribute__((ext_vector_type(4)));
typedef short ap_s16x8 __attribute__((ext_vector_type(8)));

// One staged tile = three bf16 planes of [32 rows][dk], dk = 16 DKT in {64, 128}.  Byte offset of 16-byte chunk ch of row `row` inside a plane:
// the chunk index is XORed with a function of the row such that (a) 16 consecutive rows read at the same chunk (row operands) and (b) 8 consecutive
// rows read at a chunk pair (transposed operands, half a wave) fall into 16 different 16-byte bank groups.
template <int DKT>
struct P3Tile {
  static constexpr int RB = DKT * 32;                 // bytes per row
  static constexpr int NCH = DKT * 2;                 // chunks per row
  static constexpr int PLANE = AP_KB * RB;
  static constexpr int BYTES = 3 * PLANE;
  __device__ __forceinline__ static int swz(int row) {
    return DKT == 8 ? (((row & 7) << 1) | ((row >> 3) & 1)) : ((((row >> 1) & 3) << 1) | ((row >> 3) & 1));
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
      split8(v[n][0], v[n][1], h, m, l);
      unsigned char* d = tile + TL::off(r, ch);
      *reinterpret_cast<bf16x8*>(d) = h;
      *reinterpret_cast<bf16x8*>(d + TL::PLANE) = m;
      *reinterpret_cast<bf16x8*>(d + 2 * TL::PLANE) = l;
    }
  }
};

// the lane's own row (query / key) as the B operand of the S-type products: dims 32c + 8g .. +7, three planes per 32-deep block
template <int DKT>
__device__ __forceinline__ void own_row_planes(const float* __restrict__ rowp, bool ok, int g, bf16x8 (&h)[DKT / 2], bf16x8 (&m)[DKT / 2], bf16x8 (&l)[DKT / 2]) {
#pragma unroll
  for (int c = 0; c < DKT / 2; ++c) {
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, b = a;
    if (ok) {
      a = *reinterpret_cast<const f32x4*>(rowp + 32 * c + 8 * g);
      b = *reinterpret_cast<const f32x4*>(rowp + 32 * c + 8 * g + 4);
    }
    split8(a, b, h[c], m[c], l[c]);
  }
}

// S-type tile product: 16 staged rows (tile rows 16 t + (lane & 15): the A operand, k along the head dim) against the lane's own row planes.
// roff = P3Tile::off(lane & 15, lane >> 4): chunk 4c + g of the row is roff ^ 64c (the low two chunk bits belong to g), tile t adds 16 rows
template <int DKT>
__device__ __forceinline__ f32x4 row_mma(const unsigned char* tile, int roff, int t, const bf16x8 (&bh)[DKT / 2], const bf16x8 (&bm)[DKT / 2],
                                         const bf16x8 (&bl)[DKT / 2], f32x4 acc) {
  using TL = P3Tile<DKT>;
#pragma unroll
  for (int c = 0; c < DKT / 2; ++c) {
    const unsigned char* p = tile + (roff ^ (64 * c)) + t * 16 * TL::RB;
    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(p);
    const bf16x8 am = *reinterpret_cast<const bf16x8*>(p + TL::PLANE);
    const bf16x8 al = *reinterpret_cast<const bf16x8*>(p + 2 * TL::PLANE);
    acc = planes::mma<3>(ah, am, al, bh[c], bm[c], bl[c], acc);
  }
  return acc;
}

// transposed fragment of one plane: dims 16 dt + (lane & 15) on the accumulator rows, k = the 32 staged rows (slot (g, s): row 4g + s, s < 4; 16 + 4g + s - 4)
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* plane, int off_lo, int off_hi) {
  const ap_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ap_s16x4*)(plane + off_lo));
  const ap_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ap_s16x4*)(plane + off_hi));
  return __builtin_bit_cast(bf16x8, ap_s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}
// acc[dt] += (tile^T)[dims of tile dt][32 rows] * b[32 rows][the lane's column]; tlo / thi: the lane's read addresses for dt = 0 (dt: ^ 32 dt)
template <int DKT>
__device__ __forceinline__ void tr_mma(f32x4 (&acc)[DKT], const unsigned char* tile, int tlo, int thi, const bf16x8& bh, const bf16x8& bm, const bf16x8& bl) {
  using TL = P3Tile<DKT>;
#pragma unroll
  for (int dt = 0; dt < DKT; ++dt) {
    const int lo = tlo ^ (32 * dt), hi = thi ^ (32 * dt);
    const bf16x8 ah = tr_frag(tile, lo, hi);
    const bf16x8 am = tr_frag(tile + TL::PLANE, lo, hi);
    const bf16x8 al = tr_frag(tile + 2 * TL::PLANE, lo, hi);
    acc[dt] = planes::mma<3>(ah, am, al, bh, bm, bl, acc[dt]);
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
  bf16x8 qh[DKT / 2], qm[DKT / 2], ql[DKT / 2];
  own_row_planes<DKT>(base + (size_t)q * ldg + h * dk, q < nrow, g, qh, qm, ql);
  f32x4 oT[DKT];
#pragma unroll
  for (int i = 0; i < DKT; ++i) oT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const int roff = TL::off(p, g);
  int tlo, thi;
  tr_addr<DKT>(lane, tlo, thi);

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
    f32x4 st[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kb + kt * 16 < nkeys) st[kt] = row_mma<DKT>(Ks, roff, kt, qh, qm, ql, st[kt]);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kb + kt * 16 + 4 * g + r;
        const float v = key < nkeys ? st[kt][r] * scale : -INFINITY;
        st[kt][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = planes::gmax16(mx);
    const float m_new = fmaxf(m_run, mx);        // finite: this block holds >= 1 valid key
    const float corr = (m_run == -INFINITY) ? 0.f : expf(m_run - m_new);
    float ps = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = expf(st[kt][r] - m_new);   // exp(-inf) = 0 for masked keys
        st[kt][r] = e;
        ps += e;
      }
    ps = planes::gsum16(ps);
    l_run = l_run * corr + ps;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < DKT; ++i) oT[i] *= corr;
    bf16x8 ph, pm, pl;
    split8(st[0], st[1], ph, pm, pl);
    tr_mma<DKT>(oT, Vs, tlo, thi, ph, pm, pl);
  }
  if (q < nrow) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    dim_store<DKT>(out + (row0 + q) * d + h * dk, oT, g, inv);
    if (lane < 16) lse[((size_t)b * heads + h) * T + q] = l_run > 0.f ? m_run + logf(l_run) : INFINITY;
  }
}

// ------------------------------------------------------------------------------------------
// backward, dK / dV (+ the dS tiles of the dQ kernel): wave owns 16 keys, sweeps query blocks (attn_bwd_dkv_kernel's structure)
// ------------------------------------------------------------------------------------------
template <int DKT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_p3_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
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
  bf16x8 kh[DKT / 2], km[DKT / 2], kl[DKT / 2], vh[DKT / 2], vm[DKT / 2], vl[DKT / 2];
  own_row_planes<DKT>(base + (size_t)key * ldg + d + h * dk, kok, g, kh, km, kl);
  own_row_planes<DKT>(base + (size_t)key * ldg + 2 * d + h * dk, kok, g, vh, vm, vl);
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
  const float* dob = dout + row0 * d;
  for (int qb = 0; qb < nrow; qb += AP_KB) {
    __syncthreads();
    {
      P3Stage<DKT> qreg, oreg;
      qreg.load(base, ldg, h * dk, qb, nrow, tid);
      oreg.load(dob, d, h * dk, qb, nrow, tid);
      const int qq = qb + tid;
      const bool ok = tid < AP_KB && qq < nrow;
      const float lreg = ok ? lse[((size_t)b * heads + h) * T + qq] : INFINITY;
      const float dreg = ok ? dsum[((size_t)b * heads + h) * T + qq] : 0.f;
      qreg.store(Qs, tid);
      oreg.store(Os, tid);
      if (tid < AP_KB) {
        Ls[tid] = lreg;
        Ds[tid] = dreg;
      }
    }
    __syncthreads();
    if (!wave_live) continue;
    f32x4 pr[2], ds[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, dp = sa;
      if (qb + qt * 16 < nrow) {
        sa = row_mma<DKT>(Qs, roff, qt, kh, km, kl, sa);     // S[query][key]
        dp = row_mma<DKT>(Os, roff, qt, vh, vm, vl, dp);     // dP[query][key]
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = qt * 16 + 4 * g + r;
        const float pv = key_live ? expf(sa[r] * scale - Ls[ql]) : 0.f;      // (rows past the list: Ls = inf -> 0)
        pr[qt][r] = pv;
        ds[qt][r] = pv * (dp[r] - Ds[ql]) * scale;
        // the dS tile for the dQ = dS K kernel (row = query, ldS floats per row); keys >= nkeys / queries >= T are never read
        if (dS && qb + ql < nrow && kok) dS[((size_t)blk.bh * T + qb + ql) * ldS + key] = ds[qt][r];
      }
    }
    bf16x8 bh, bm, bl;
    split8(pr[0], pr[1], bh, bm, bl);
    tr_mma<DKT>(dvT, Os, tlo, thi, bh, bm, bl);      // dV^T[dim][key] += dO^T P
    split8(ds[0], ds[1], bh, bm, bl);
    tr_mma<DKT>(dkT, Qs, tlo, thi, bh, bm, bl);      // dK^T[dim][key] += Q^T dS
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
  using TL = P3Tile<DKT>;
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
    bf16x8 bh, bm, bl;
    split8(dsT[0], dsT[1], bh, bm, bl);
    tr_mma<DKT>(dqT, Ks, tlo, thi, bh, bm, bl);      // dQ^T[dim][query] += K^T dS^T
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
