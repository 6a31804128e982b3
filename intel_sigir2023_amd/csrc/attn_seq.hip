// Whole-sequence attention for T <= 64 and head dim 64 / 128 (the IntEL towers at list length 50, the
// BERT4Rec blocks at history length 20): restates modules/layers.py:50-60 like attn.hip, for the shapes where
// everything a (session, head) pair needs fits in LDS at once.
//
//   * wave = one 16-row tile of one pair; PW = 4 / ceil(T/16) pairs share a workgroup ("item" = those PW pairs);
//   * persistent workgroups loop over items; the operand rows of item i+1 are copied global -> LDS by the DMA
//     path (global_load_lds_dwordx4) while item i is computed: two LDS stages, no staging registers.  A DMA
//     instruction writes 1 KB linearly, so rows are unpadded and the 16-byte chunk index is XOR-swizzled with
//     (row & 15) on the per-lane source address and again when fragments are read.  Rows >= T are never
//     copied: the stages are zeroed once and T is the same for every item, so padding rows stay zero;
//   * rows are staged PERMUTED inside each 16-row tile (row 4a+b -> slot 4b+a): accumulator row 4j+r then is
//     key (query) 4r+j, so k-step s of the following product covers rows 4s..4s+3 and the k-steps that hold
//     only padding are dropped AT COMPILE TIME (template LS = live k-steps of the last tile; a runtime branch
//     around accumulating MFMAs makes the compiler shuttle every accumulator between AGPRs and VGPRs);
//   * "transposed" operands (V^T, dO^T, Q^T, K^T) are read as one b128 along the head dim: lane p takes dims
//     4p..4p+3 of its row and feeds FOUR MFMAs whose output row p means dim 4p+t -- the output rows of the
//     four tiles interleave and the epilogue stores float4s;
//   * softmax in base 2 (v_exp_f32 on fma(s, c, -m c), c = log2(e)/sqrt(dk)): 6 VALU instructions per logit;
//     at one or two waves per SIMD every VALU instruction is time the MFMA pipe idles.
//   * backward = ONE kernel per item (attn_seq_bwd_fused_kernel): the dK/dV sweep with wave = 16 keys computes S and dP
//     once and keeps its dS column block in registers; then the waves park K (from registers) and dS in the LDS space of
//     the dead Q / dO stages and become 16-query tiles for dQ = dS K: 5 tile products instead of the 7 of a recompute
//     scheme, no dS / K round trip through HBM.  delta = rowsum(dO * O) = rowsum(P * dP) is summed over the key-tile
//     waves through LDS, so the attention output O is not read at all.
#include <stdio.h>
#include <stdlib.h>
#include "kernels.h"

namespace {

__device__ __forceinline__ int perm16(int r) { return (r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3); }
__device__ __forceinline__ float gmax16(float v) {   // over the 4 lane groups sharing lane&15
  v = fmaxf(v, __shfl_xor(v, 16));
  return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float gsum16(float v) {
  v += __shfl_xor(v, 16);
  return v + __shfl_xor(v, 32);
}

// c += sum over the four k-steps s (and the four lane groups) of a[s] * b[s]: four exact fp32 MFMAs (16x16x4) in the parity mode;
// in bf16 mode ONE v_mfma_f32_16x16x16_bf16, whose lane (i, j) supplies k = 4j .. 4j+3 -- the same element order -- so the
// accumulator-to-operand tricks of these kernels carry over unchanged
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4s __attribute__((ext_vector_type(4)));
template <bool BF>
__device__ __forceinline__ f32x4 mma4(const f32x4& a, const f32x4& b, f32x4 c) {
  if (BF) {
    const bf16x4s ab = bf16x4s{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3]};
    const bf16x4s bb = bf16x4s{(__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, ab), __builtin_bit_cast(s16x4, bb), c, 0, 0, 0);
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) c = mfma16(a[s], b[s], c);
  return c;
}

// four consecutive elements at element offset `off` of an fp32 or (H) bf16 array
template <bool H>
__device__ __forceinline__ f32x4 ldx4(const float* __restrict__ base, size_t off) {
  if (H) {
    const bf16x4s v = *reinterpret_cast<const bf16x4s*>(reinterpret_cast<const __bf16*>(base) + off);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  }
  return *reinterpret_cast<const f32x4*>(base + off);
}
template <bool H>
__device__ __forceinline__ void stx4(float* __restrict__ base, size_t off, const f32x4& v) {
  if (H) *reinterpret_cast<bf16x4s*>(reinterpret_cast<__bf16*>(base) + off) = bf16x4s{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  else *reinterpret_cast<f32x4*>(base + off) = v;
}

template <int DKT, int NT>
struct SeqP {
  static constexpr int DK = DKT * 16;
  static constexpr int PW = 4 / NT;                   // pairs per item
  static constexpr int TP = NT * 16;                  // padded rows per pair
  static constexpr int ROWS = PW * TP;
  static constexpr int CPR = DKT * 4;                 // 16-byte chunks per row
  static constexpr int BUF = ROWS * DK;               // floats per matrix per stage
  static constexpr int IPW = ROWS * CPR / 256;        // DMA instructions per wave per matrix per stage
};

// per-lane constants of the DMA instructions of this wave: source row in the pair, source column, pair slot
template <int DKT, int NT, bool PERM>
struct DmaSlots {
  int rho[SeqP<DKT, NT>::IPW], col[SeqP<DKT, NT>::IPW], slot[SeqP<DKT, NT>::IPW];
  unsigned live;        // bit j: instruction j holds at least one row < T (wave-uniform)
  __device__ __forceinline__ void init(int wave, int lane, int T) {
    using C = SeqP<DKT, NT>;
    live = 0;
#pragma unroll
    for (int j = 0; j < C::IPW; ++j) {
      const int lin = (wave * C::IPW + j) * 64 + lane;
      const int R = lin / C::CPR, pch = lin - R * C::CPR;
      const int sl = R / C::TP, r = R - sl * C::TP;
      const int rr = PERM ? perm16(r) : r;
      slot[j] = sl;
      col[j] = (pch ^ (R & 15)) * 4;
      if (__ballot(rr < T) != 0ull) live |= 1u << j;
      rho[j] = min(rr, T - 1);          // lanes of a live instruction that sit on a padding row copy row T-1 (finite, unused)
    }
  }
};

// fragment of one row ("row on the lane" operand form): element s of group g = x[row][g*16 + 4*(lane>>4) + s]
template <int DKT>
__device__ __forceinline__ void load_row_frags(f32x4 (&f)[DKT], const float* __restrict__ rowp, bool rowok, int dk, int lane) {
#pragma unroll
  for (int g = 0; g < DKT; ++g) {
    const int col = g * 16 + 4 * (lane >> 4);
    f[g] = (rowok && col < dk) ? *reinterpret_cast<const f32x4*>(rowp + col) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

__device__ __forceinline__ void split_pair(int bh, int heads, int& b, int& h) {
  b = bh;
  h = 0;
  if (heads > 1) {
    b = bh / heads;
    h = bh - b * heads;
  }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int DKT, int NT, int LS, bool BF = false>
__global__ __launch_bounds__(256) void attn_seq_fwd_kernel(const float* __restrict__ qkv, int BH, int T, int d, int heads,
                                                           const int* __restrict__ key_len, const int* __restrict__ row_off, float c2,
                                                           float scale, float* __restrict__ out, float* __restrict__ lse) {
  // row_off != NULL: PACKED rows -- session b owns rows row_off[b] .. row_off[b] + key_len[b] - 1 of qkv / out (no padding rows
  // in memory); otherwise session b owns rows b*T .. b*T + T - 1.  lse stays indexed [pair][T].
  using C = SeqP<DKT, NT>;
  constexpr int DK = C::DK, DQ = DKT / 4, NSTEPS = (NT - 1) * 4 + LS;
  extern __shared__ __attribute__((aligned(16))) float smem[];      // [stage][K | V][ROWS][DK]
  const int tid = threadIdx.x, lane = tid & 63, j = lane >> 4, p = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ldg = 3 * d;
  const int nitems = (BH + C::PW - 1) / C::PW;
  const int slot = wave / NT, tile = wave - slot * NT;
  const bool wlive = slot < C::PW;
  for (int i = tid; i < C::BUF; i += 256) reinterpret_cast<f32x4*>(smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};   // 4 BUF floats
  DmaSlots<DKT, NT, true> ds;
  ds.init(wave, lane, T);
  auto dma_item = [&](int it, float* stage) {
#pragma unroll
    for (int jj = 0; jj < C::IPW; ++jj) {
      if (!((ds.live >> jj) & 1u)) continue;
      int b, h;
      split_pair(min(it * C::PW + ds.slot[jj], BH - 1), heads, b, h);
      const size_t base = row_off ? (size_t)row_off[b] : (size_t)b * T;
      const int tb = row_off ? max(min(key_len[b], T), 1) : T;
      // (pointers in plain locals: the builtin's argument check mishandles template-dependent expressions)
      const float* srck = qkv + (base + min(ds.rho[jj], tb - 1)) * ldg + d + h * DK + ds.col[jj];
      const float* srcv = srck + d;
      float* dstk = stage + (wave * C::IPW + jj) * 256;
      float* dstv = dstk + C::BUF;
      __builtin_amdgcn_global_load_lds(srck, dstk, 16, 0, 0);
      __builtin_amdgcn_global_load_lds(srcv, dstv, 16, 0, 0);
    }
  };
  auto load_q = [&](int it, f32x4 (&qf)[DKT]) {
    int b, h;
    split_pair(min(it * C::PW + slot, BH - 1), heads, b, h);
    const size_t base = row_off ? (size_t)row_off[b] : (size_t)b * T;
    const int tb = row_off ? max(min(key_len[b], T), 1) : T;
    const float* rowp = qkv + (base + min(tile * 16 + p, tb - 1)) * ldg + h * DK + 4 * j;
#pragma unroll
    for (int g = 0; g < DKT; ++g) qf[g] = *reinterpret_cast<const f32x4*>(rowp + g * 16);
  };
  int it = blockIdx.x;
  const int G = gridDim.x;
  __syncthreads();
  dma_item(it, smem);
  if (it + G < nitems) dma_item(it + G, smem + 2 * C::BUF);
  f32x4 qf[DKT], qn[DKT];
  load_q(it, qf);
  __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
  __syncthreads();
  int cur = 0;
  for (; it < nitems; it += G) {
    const float* Kp = smem + cur * 2 * C::BUF + slot * C::TP * DK;
    const float* Vp = Kp + C::BUF;
    if (it + G < nitems) load_q(it + G, qn);
    const int bh = it * C::PW + slot;
    const bool live = wlive && bh < BH;
    int b, h;
    split_pair(live ? bh : 0, heads, b, h);
    const int q = tile * 16 + p;
    const int nkeys = key_len ? min(key_len[b], T) : T;
    const bool qok = live && q < (row_off ? nkeys : T);
    const size_t obase = row_off ? (size_t)row_off[b] : (size_t)b * T;
    f32x4 st[NT];
    f32x4 oT[DKT];
    float ps = 0.f, mref = 0.f;
    if (wlive) {
      // S^T tiles; operand fragments are fetched one k-group ahead of the MFMAs that consume them
      f32x4 kf[2][NT];
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        kf[0][kt] = *reinterpret_cast<const f32x4*>(Kp + (kt * 16 + p) * DK + ((j ^ p) << 2));
      }
#pragma unroll
      for (int g = 0; g < DKT; ++g) {
        if (g + 1 < DKT) {
#pragma unroll
          for (int kt = 0; kt < NT; ++kt)
            kf[(g + 1) & 1][kt] = *reinterpret_cast<const f32x4*>(Kp + (kt * 16 + p) * DK + ((((g + 1) * 4 + j) ^ p) << 2));
        }
        __builtin_amdgcn_sched_barrier(0);      // the prefetch stays above the MFMAs it overlaps
        if (BF) {
#pragma unroll
          for (int kt = 0; kt < NT; ++kt) st[kt] = mma4<true>(kf[g & 1][kt], qf[g], st[kt]);
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) st[kt] = mfma16(kf[g & 1][kt][s], qf[g][s], st[kt]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // first V fragment rides under the softmax
      f32x4 vv[2][DQ];
      auto load_v = [&](int step, f32x4 (&dst)[DQ]) {
        const int row = (step >> 2) * 16 + 4 * j + (step & 3);
#pragma unroll
        for (int dq = 0; dq < DQ; ++dq) dst[dq] = *reinterpret_cast<const f32x4*>(Vp + row * DK + (((dq * 16 + p) ^ (row & 15)) << 2));
      };
      load_v(0, vv[0]);
      // accumulator row 4j+r of tile kt is key kt*16 + 4r + j: valid iff j < nkeys - (kt*16 + 4r)
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = j < nkeys - (kt * 16 + 4 * r) ? st[kt][r] : -INFINITY;
          st[kt][r] = v;
          mx = fmaxf(mx, v);
        }
      mx = gmax16(mx);
      mref = mx == -INFINITY ? 0.f : mx;
      const float moff = -mref * c2;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][r], c2, moff));
          st[kt][r] = e;
          ps += e;
        }
      ps = gsum16(ps);
#pragma unroll
      for (int i = 0; i < DKT; ++i) oT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (BF) {
        // one bf16 MFMA per (key tile, output tile): the four k-steps of a key tile are the four registers of st[kt]; the
        // steps of the last tile that hold only padding carry P = 0 and zero V rows
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
          f32x4 v4[4][DQ];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (kt == 0 && r == 0) {
#pragma unroll
              for (int dq = 0; dq < DQ; ++dq) v4[0][dq] = vv[0][dq];
            } else {
              load_v(kt * 4 + r, v4[r]);
            }
          }
#pragma unroll
          for (int dq = 0; dq < DQ; ++dq)
#pragma unroll
            for (int t = 0; t < 4; ++t)
              oT[dq * 4 + t] = mma4<true>(f32x4{v4[0][dq][t], v4[1][dq][t], v4[2][dq][t], v4[3][dq][t]}, st[kt], oT[dq * 4 + t]);
        }
      } else {
#pragma unroll
      for (int step = 0; step < NSTEPS; ++step) {
        if (step + 1 < NSTEPS) load_v(step + 1, vv[(step + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dq = 0; dq < DQ; ++dq)
#pragma unroll
          for (int t = 0; t < 4; ++t) oT[dq * 4 + t] = mfma16(vv[step & 1][dq][t], st[step >> 2][step & 3], oT[dq * 4 + t]);
        __builtin_amdgcn_sched_barrier(0);
      }
      }
    }
    // the copy into the other stage (issued one item ago) and the next query fragments had this whole compute
    // phase to land
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_sched_barrier(0);
    if (qok) {
      const float inv = ps > 0.f ? 1.f / ps : 0.f;
      float* orow = out + (obase + q) * d + h * DK;
#pragma unroll
      for (int dq = 0; dq < DQ; ++dq)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f32x4 o = f32x4{oT[dq * 4 + 0][r], oT[dq * 4 + 1][r], oT[dq * 4 + 2][r], oT[dq * 4 + 3][r]} * inv;
          *reinterpret_cast<f32x4*>(orow + dq * 64 + 16 * j + 4 * r) = o;
        }
      // natural-log lse, as the backward kernels and the general path use it
      if (lane < 16) lse[(size_t)bh * T + q] = ps > 0.f ? mref * scale + __logf(ps) : INFINITY;
    }
    __syncthreads();
    if (it + 2 * G < nitems) dma_item(it + 2 * G, smem + cur * 2 * C::BUF);
    cur ^= 1;
#pragma unroll
    for (int g = 0; g < DKT; ++g) qf[g] = qn[g];
  }
}

// ------------------------------------------------------------------------------------------
// backward (register-staged, one item per workgroup)
// ------------------------------------------------------------------------------------------
template <int DKT, int NT>
struct SeqCfg {
  static constexpr int LD = DKT * 16 + 4;
  static constexpr int DK = DKT * 16;
  static constexpr int PW = 4 / NT;          // pairs per workgroup
  static constexpr int TP = NT * 16;         // padded rows per pair
  static constexpr int ROWS = PW * TP;
  static constexpr int C4 = DKT * 4;         // float4 per row
  static constexpr int ITERS = ROWS * C4 / 256;
  static constexpr int TPD = TP + 8;         // pitch of the dS rows the fused backward parks in LDS
  static constexpr int OSZ = ROWS * LD > PW * TP * TPD ? ROWS * LD : PW * TP * TPD;   // dO stage, later the dS tiles
};

// address of float4 #i of the staged block: LDS row rl (pair slot, row in pair), source row, validity
template <int DKT, int NT, bool PERM>
__device__ __forceinline__ bool seq_src(int i, int bh0, int BH, int T, int heads, int& rl, int& c4, size_t& grow, int& hcol,
                                        const int* __restrict__ row_off = nullptr, const int* __restrict__ key_len = nullptr) {
  using C = SeqCfg<DKT, NT>;
  rl = i / C::C4;
  c4 = i - rl * C::C4;
  const int sl = rl / C::TP, r = rl - sl * C::TP;
  const int rho = PERM ? perm16(r) : r;
  const int bh = bh0 + sl;
  int b = bh, h = 0;
  if (heads > 1) { b = bh / heads; h = bh - b * heads; }
  hcol = h * C::DK + c4 * 4;
  if (row_off) {       // packed rows: session b owns row_off[b] .. + key_len[b] - 1
    if (bh >= BH) { grow = 0; return false; }
    grow = (size_t)row_off[b] + rho;
    return rho < min(key_len[b], T);
  }
  grow = (size_t)b * T + rho;
  return bh < BH && rho < T;
}

// stage the same rows of two matrices (all loads in flight before the first LDS store)
template <int DKT, int NT, bool PERM, bool TWO>
__device__ __forceinline__ void stage_seq2(float* dst0, const float* __restrict__ src0, int ld0, int coff0, float* dst1,
                                           const float* __restrict__ src1, int ld1, int coff1, int bh0, int BH, int T,
                                           int heads, int tid) {
  using C = SeqCfg<DKT, NT>;
  f32x4 v0[C::ITERS], v1[C::ITERS];
#pragma unroll
  for (int it = 0; it < C::ITERS; ++it) {
    int rl, c4, hcol;
    size_t grow;
    const bool ok = seq_src<DKT, NT, PERM>(tid + it * 256, bh0, BH, T, heads, rl, c4, grow, hcol);
    // unconditional loads from a clamped (always valid) address + select: keeps the staging registers scalarised
    const size_t gr = ok ? grow : 0;
    const int hc = ok ? hcol : 0;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 t0 = *reinterpret_cast<const f32x4*>(src0 + gr * ld0 + coff0 + hc);
    v0[it] = ok ? t0 : zero;
    if (TWO) {
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(src1 + gr * ld1 + coff1 + hc);
      v1[it] = ok ? t1 : zero;
    }
  }
#pragma unroll
  for (int it = 0; it < C::ITERS; ++it) {
    const int i = tid + it * 256;
    const int rl = i / C::C4, c4 = i - rl * C::C4;
    *reinterpret_cast<f32x4*>(dst0 + rl * C::LD + c4 * 4) = v0[it];
    if (TWO) *reinterpret_cast<f32x4*>(dst1 + rl * C::LD + c4 * 4) = v1[it];
  }
}

// The whole backward of one item in ONE kernel: the dK/dV sweep (wave = 16 keys), then -- the staged Q / dO rows being dead -- every
// wave parks its 16 K rows (still in registers) and its dS column block in their LDS space and turns into a 16-QUERY
// tile for dQ = dS K.  Neither dS nor K makes a round trip through HBM (bwd_kv + bwd_q: 0.6 GB per Tmall-shape step).
// BF: the products as single bf16 MFMAs (bf16 mode); H16 (with BF): q/k/v and dO are read and dq/dk/dv written as bf16 arrays
template <int DKT, int NT, int LS, bool BF = false, bool H16 = false>
__global__ __launch_bounds__(256, 2) void attn_seq_bwd_fused_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                              const float* __restrict__ dout, const float* __restrict__ lse,
                                                              int BH, int T, int d, int heads, const int* __restrict__ key_len,
                                                              const int* __restrict__ row_off, float c2, float scale,
                                                              float* __restrict__ dqkv) {
  // row_off != NULL: packed rows (see attn_seq_fwd_kernel): qkv / dout / dqkv rows of session b start at row_off[b]
  using C = SeqCfg<DKT, NT>;
  constexpr int LD = C::LD, DK = C::DK, DQ = DKT / 4, TP = C::TP, NSTEPS = (NT - 1) * 4 + LS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;
  float* Os = smem + C::ROWS * LD;
  float* Ls = smem + C::ROWS * LD + C::OSZ;
  float* Ds = Ls + C::ROWS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane >> 4, p = lane & 15;
  const int bh0 = blockIdx.x * C::PW, ldg = 3 * d;
  const int slot = wave / NT, tile = wave - slot * NT, bh = bh0 + slot;
  const bool live = slot < C::PW && bh < BH;
  int b = live ? bh : 0, h = 0;
  if (heads > 1) { b = (live ? bh : 0) / heads; h = (live ? bh : 0) - b * heads; }
  const int key = tile * 16 + p;
  const size_t base = row_off ? (size_t)row_off[b] : (size_t)b * T;
  const int tb = row_off ? min(key_len[b], T) : T;            // rows of this session that exist in memory
  const bool kok = live && key < tb;
  f32x4 kf[DKT], vf[DKT];
#pragma unroll
  for (int g = 0; g < DKT; ++g) {
    const int col = g * 16 + 4 * j;
    const bool ok = kok && col < DK;
    const size_t off = ok ? (base + key) * ldg + d + h * DK + col : 0;
    const f32x4 tk = ldx4<H16>(qkv, off), tv = ldx4<H16>(qkv, off + (ok ? d : 0));
    kf[g] = ok ? tk : f32x4{0.f, 0.f, 0.f, 0.f};
    vf[g] = ok ? tv : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  {
    // stage Q and dO (permuted rows)
    f32x4 vq[C::ITERS], vo[C::ITERS];
#pragma unroll
    for (int it = 0; it < C::ITERS; ++it) {
      int rl, c4, hcol;
      size_t grow;
      const bool ok = seq_src<DKT, NT, true>(tid + it * 256, bh0, BH, T, heads, rl, c4, grow, hcol, row_off, key_len);
      const size_t gr = ok ? grow : 0;
      const int hc = ok ? hcol : 0;
      const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 tq = ldx4<H16>(qkv, gr * ldg + hc);
      const f32x4 to = ldx4<H16>(dout, gr * d + hc);
      vq[it] = ok ? tq : zero;
      vo[it] = ok ? to : zero;
    }
#pragma unroll
    for (int it = 0; it < C::ITERS; ++it) {
      const int i = tid + it * 256;
      const int rl = i / C::C4, c4 = i - rl * C::C4;
      *reinterpret_cast<f32x4*>(Qs + rl * LD + c4 * 4) = vq[it];
      *reinterpret_cast<f32x4*>(Os + rl * LD + c4 * 4) = vo[it];
    }
    if (tid < C::ROWS) {
      const int sl = tid / TP, r = tid - sl * TP, rho = perm16(r), bb = bh0 + sl;
      const int tbb = (row_off && bb < BH) ? min(key_len[heads > 1 ? bb / heads : bb], T) : T;
      Ls[tid] = (bb < BH && rho < tbb) ? -1.44269504088896340736f * lse[(size_t)bb * T + rho] : -INFINITY;   // -lse in base 2
    }
  }
  __syncthreads();
  constexpr int TPD = C::TPD;                    // dS rows in LDS: b128 reads of 16 queries x 4 key groups are conflict-free
  // delta[q] = rowsum(dO * O) = sum_key P[q][key] dP[q][key]: every wave sums its 16 keys, the NT key-tile waves of a pair
  // meet in LDS (Ds[slot row][tile]) -- the attention output O is not read at all
  f32x4 prs[NT], dsk[NT];                        // P, then dP -> dS, of this wave's key tile, query tile by query tile
  f32x4 dkT[DKT];
  const int nkeys = live ? (key_len ? min(key_len[b], T) : T) : 0;
  const bool key_live = key < nkeys;             // masked keys get exactly zero gradient
  const float* Qp = Qs + slot * TP * LD;
  const float* Op = Os + slot * TP * LD;
  const float* Lp = Ls + slot * TP;
  float* Dp = Ds + slot * TP * NT;
  if (live) {
    f32x4 dvT[DKT];
#pragma unroll
    for (int i = 0; i < DKT; ++i) dvT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
      f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < DKT; ++g) {
        const int off = (qt * 16 + p) * LD + g * 16 + 4 * j;
        const f32x4 qa = *reinterpret_cast<const f32x4*>(Qp + off);
        const f32x4 oa = *reinterpret_cast<const f32x4*>(Op + off);
        if (BF) {
          sa = mma4<true>(qa, kf[g], sa);
          dp = mma4<true>(oa, vf[g], dp);
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            sa = mfma16(qa[s], kf[g][s], sa);     // S[query slot][key]
            dp = mfma16(oa[s], vf[g][s], dp);     // dP[query slot][key]
          }
        }
      }
      f32x4 pr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = qt * 16 + 4 * j + r;     // staged slot of accumulator row 4j+r; its query is 4r+j
        const float pv = key_live ? __builtin_amdgcn_exp2f(__builtin_fmaf(sa[r], c2, Lp[rl])) : 0.f;
        pr[r] = pv;
        float x = pv * dp[r];                   // this key's share of delta[rl]; sum over the 16 keys of the tile
        x += __shfl_xor(x, 1);
        x += __shfl_xor(x, 2);
        x += __shfl_xor(x, 4);
        x += __shfl_xor(x, 8);
        if (p == 0) Dp[rl * NT + tile] = x;
      }
      prs[qt] = pr;
      dsk[qt] = dp;
      if (BF) {        // the four k-steps (queries 4j+s of the tile) in one bf16 MFMA; padded queries carry P = 0 and zero dO rows
#pragma unroll
        for (int dq = 0; dq < DQ; ++dq) {
          f32x4 ov[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) ov[s] = *reinterpret_cast<const f32x4*>(Op + (qt * 16 + 4 * j + s) * LD + dq * 64 + 4 * p);
#pragma unroll
          for (int t = 0; t < 4; ++t) dvT[dq * 4 + t] = mma4<true>(f32x4{ov[0][t], ov[1][t], ov[2][t], ov[3][t]}, pr, dvT[dq * 4 + t]);
        }
      } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (qt * 4 + s >= NSTEPS) continue;     // queries 4s..4s+3 of the tile are padding (compile-time)
#pragma unroll
        for (int dq = 0; dq < DQ; ++dq) {
          const f32x4 ov = *reinterpret_cast<const f32x4*>(Op + (qt * 16 + 4 * j + s) * LD + dq * 64 + 4 * p);
#pragma unroll
          for (int t = 0; t < 4; ++t) dvT[dq * 4 + t] = mfma16(ov[t], pr[s], dvT[dq * 4 + t]);   // dV^T[dim][key] += dO^T P
        }
      }
      }
    }
    if (kok) {
      const size_t drow = (base + key) * ldg + h * DK + 2 * d;
#pragma unroll
      for (int dq = 0; dq < DQ; ++dq)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          stx4<H16>(dqkv, drow + dq * 64 + 16 * j + 4 * r, f32x4{dvT[dq * 4 + 0][r], dvT[dq * 4 + 1][r], dvT[dq * 4 + 2][r], dvT[dq * 4 + 3][r]});
    }
  }
  __syncthreads();                               // the delta shares of all key tiles are in LDS
  if (live) {
#pragma unroll
    for (int i = 0; i < DKT; ++i) dkT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
      f32x4 ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = qt * 16 + 4 * j + r;
        float delta = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) delta += Dp[rl * NT + t];
        ds[r] = prs[qt][r] * (dsk[qt][r] - delta) * scale;
      }
      dsk[qt] = ds;
      if (BF) {
#pragma unroll
        for (int dq = 0; dq < DQ; ++dq) {
          f32x4 qv[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) qv[s] = *reinterpret_cast<const f32x4*>(Qp + (qt * 16 + 4 * j + s) * LD + dq * 64 + 4 * p);
#pragma unroll
          for (int t = 0; t < 4; ++t) dkT[dq * 4 + t] = mma4<true>(f32x4{qv[0][t], qv[1][t], qv[2][t], qv[3][t]}, ds, dkT[dq * 4 + t]);
        }
      } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (qt * 4 + s >= NSTEPS) continue;
#pragma unroll
        for (int dq = 0; dq < DQ; ++dq) {
          const f32x4 qv = *reinterpret_cast<const f32x4*>(Qp + (qt * 16 + 4 * j + s) * LD + dq * 64 + 4 * p);
#pragma unroll
          for (int t = 0; t < 4; ++t) dkT[dq * 4 + t] = mfma16(qv[t], ds[s], dkT[dq * 4 + t]);   // dK^T[dim][key] += Q^T dS
        }
      }
      }
    }
    if (kok) {
      const size_t drow = (base + key) * ldg + h * DK + d;
#pragma unroll
      for (int dq = 0; dq < DQ; ++dq)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          stx4<H16>(dqkv, drow + dq * 64 + 16 * j + 4 * r, f32x4{dkT[dq * 4 + 0][r], dkT[dq * 4 + 1][r], dkT[dq * 4 + 2][r], dkT[dq * 4 + 3][r]});
    }
  }   // live
  __syncthreads();                               // every wave is done with the staged Q / dO rows
  float* Ks = Qs;                                // [ROWS][LD], plain row order
  float* Dsh = Os;                               // [PW][TP][TPD], row = query (plain order)
  if (live) {
#pragma unroll
    for (int g = 0; g < DKT; ++g) *reinterpret_cast<f32x4*>(Ks + (slot * TP + key) * LD + g * 16 + 4 * j) = kf[g];
#pragma unroll
    for (int qt = 0; qt < NT; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) Dsh[(slot * TP + qt * 16 + 4 * r + j) * TPD + key] = dsk[qt][r];   // accumulator row 4j+r is query 4r+j
  }
  __syncthreads();
  if (!live) return;
  {
    const int q = tile * 16 + p;
    const float* Kp = Ks + slot * TP * LD;
    const float* Dq = Dsh + (size_t)(slot * TP + q) * TPD;
    f32x4 dqT[DKT];
#pragma unroll
    for (int i = 0; i < DKT; ++i) dqT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      const f32x4 dsT = *reinterpret_cast<const f32x4*>(Dq + kt * 16 + 4 * j);
      if (BF) {
#pragma unroll
        for (int dq = 0; dq < DQ; ++dq) {
          f32x4 kv[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) kv[s] = *reinterpret_cast<const f32x4*>(Kp + (kt * 16 + 4 * j + s) * LD + dq * 64 + 4 * p);
#pragma unroll
          for (int t = 0; t < 4; ++t) dqT[dq * 4 + t] = mma4<true>(f32x4{kv[0][t], kv[1][t], kv[2][t], kv[3][t]}, dsT, dqT[dq * 4 + t]);
        }
      } else {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int dq = 0; dq < DQ; ++dq) {
          const f32x4 kv = *reinterpret_cast<const f32x4*>(Kp + (kt * 16 + 4 * j + s) * LD + dq * 64 + 4 * p);
#pragma unroll
          for (int t = 0; t < 4; ++t) dqT[dq * 4 + t] = mfma16(kv[t], dsT[s], dqT[dq * 4 + t]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (q < tb) {
      const size_t drow = (base + q) * ldg + h * DK;
#pragma unroll
      for (int dq = 0; dq < DQ; ++dq)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          stx4<H16>(dqkv, drow + dq * 64 + 16 * j + 4 * r, f32x4{dqT[dq * 4 + 0][r], dqT[dq * 4 + 1][r], dqT[dq * 4 + 2][r], dqT[dq * 4 + 3][r]});
    }
  }
}


}  // namespace

#define SEQ_DISPATCH3(DKT_RT, NT_RT, LS_RT, ...)                                        \
  do {                                                                                  \
    auto with_ls = [&](auto dkt_c, auto nt_c) {                                         \
      constexpr int DKT = decltype(dkt_c)::value;                                       \
      constexpr int NT = decltype(nt_c)::value;                                         \
      switch (LS_RT) {                                                                  \
        case 1: { constexpr int LS = 1; __VA_ARGS__; } break;                           \
        case 2: { constexpr int LS = 2; __VA_ARGS__; } break;                           \
        case 3: { constexpr int LS = 3; __VA_ARGS__; } break;                           \
        default: { constexpr int LS = 4; __VA_ARGS__; } break;                          \
      }                                                                                 \
    };                                                                                  \
    auto with_nt = [&](auto dkt_c) {                                                    \
      switch (NT_RT) {                                                                  \
        case 1: with_ls(dkt_c, std::integral_constant<int, 1>()); break;                \
        case 2: with_ls(dkt_c, std::integral_constant<int, 2>()); break;                \
        case 3: with_ls(dkt_c, std::integral_constant<int, 3>()); break;                \
        default: with_ls(dkt_c, std::integral_constant<int, 4>()); break;               \
      }                                                                                 \
    };                                                                                  \
    if ((DKT_RT) == 4) with_nt(std::integral_constant<int, 4>());                       \
    else with_nt(std::integral_constant<int, 8>());                                     \
  } while (0)

bool attn_seq_supported(int T, int dk) {
  static const int off = [] { const char* e = getenv("INTEL_ATTN_SEQ"); return (e && e[0] == '0') ? 1 : 0; }();
  return !off && T >= 1 && T <= 64 && (dk == 64 || dk == 128);
}

int launch_attn_seq_fwd(const float* qkv, int B, int T, int d, int heads, const int* key_len, float* out, float* lse,
                        hipStream_t st, const int* row_off) {
  INTEL_CHECK_ARG(!row_off || key_len, "attn_seq: packed rows need the session lengths");
  const int dk = d / heads, dkt = dk / 16, BH = B * heads;
  const int nt = cdiv(T, 16), ls = cdiv(T - (nt - 1) * 16, 4);
  const float scale = 1.0f / sqrtf((float)dk);
  SEQ_DISPATCH3(dkt, nt, ls, {
    using C = SeqP<DKT, NT>;
    const size_t smem = (size_t)4 * C::BUF * sizeof(float);
    const int per_cu = (int)((size_t)160 * 1024 / smem);
    const int grid = min(cdiv(BH, C::PW), num_cus() * (per_cu < 1 ? 1 : per_cu));
    if (gemm_planes() == 1) {       // bf16 mode: the two products as single bf16 MFMAs (softmax stays fp32)
      allow_lds((attn_seq_fwd_kernel<DKT, NT, LS, true>), smem);
      LAUNCH_S(BH, T, dk, 4.0 * B * T * (double)T * d, 16.0 * B * T * (double)d, (attn_seq_fwd_kernel<DKT, NT, LS, true>), dim3(grid), dim3(256), smem, st, qkv, BH, T, d, heads, key_len, row_off, scale * 1.44269504088896340736f, scale, out, lse);
    } else {
      allow_lds((attn_seq_fwd_kernel<DKT, NT, LS>), smem);
      LAUNCH_S(BH, T, dk, 4.0 * B * T * (double)T * d, 16.0 * B * T * (double)d, (attn_seq_fwd_kernel<DKT, NT, LS>), dim3(grid), dim3(256), smem, st, qkv, BH, T, d, heads, key_len, row_off, scale * 1.44269504088896340736f, scale, out, lse);
    }
  });
  INTEL_CHECK_LAUNCH();
  return 0;
}

size_t attn_seq_bwd_scratch_floats(int B, int T, int heads) {
  const size_t tp = (size_t)cdiv(T, 16) * 16;
  return (size_t)B * heads * tp * tp;
}

// scratch: attn_seq_bwd_scratch_floats(B, T, heads) floats (the dS tiles)
bool attn_seq_packed_supported(int T, int dk) {
  return attn_seq_supported(T, dk);
}

bool attn_seq_h16_supported(int T, int dk) { return attn_seq_packed_supported(T, dk) && gemm_planes() == 1; }

int launch_attn_seq_bwd(const float* qkv, const float* out, const float* dout, const float* lse, int B, int T, int d,
                        int heads, const int* key_len, float* dqkv, float* dS, hipStream_t st, const int* row_off, int h16) {
  INTEL_CHECK_ARG(!h16 || attn_seq_h16_supported(T, d / heads), "attn_seq: bf16-stored q/k/v need the fused backward in bf16 mode");
  INTEL_CHECK_ARG(!row_off || (key_len && attn_seq_packed_supported(T, d / heads)), "attn_seq: packed rows need the fused backward and the session lengths");
  const int dk = d / heads, dkt = dk / 16, BH = B * heads;
  const int nt = cdiv(T, 16), ls = cdiv(T - (nt - 1) * 16, 4);
  const float scale = 1.0f / sqrtf((float)dk);
  {
    SEQ_DISPATCH3(dkt, nt, ls, {
      using C = SeqCfg<DKT, NT>;
      const size_t smem = (size_t)(C::ROWS * C::LD + C::OSZ + (1 + NT) * C::ROWS) * sizeof(float);
      if (h16) {
        allow_lds((attn_seq_bwd_fused_kernel<DKT, NT, LS, true, true>), smem);
        LAUNCH_S(BH, T, dk, 10.0 * B * T * (double)T * d, 16.0 * B * T * (double)d, (attn_seq_bwd_fused_kernel<DKT, NT, LS, true, true>), dim3(cdiv(BH, C::PW)), dim3(256), smem, st, qkv, out, dout, lse, BH, T, d, heads, key_len, row_off, scale * 1.44269504088896340736f, scale, dqkv);
      } else if (gemm_planes() == 1) {
        allow_lds((attn_seq_bwd_fused_kernel<DKT, NT, LS, true>), smem);
        LAUNCH_S(BH, T, dk, 10.0 * B * T * (double)T * d, 24.0 * B * T * (double)d, (attn_seq_bwd_fused_kernel<DKT, NT, LS, true>), dim3(cdiv(BH, C::PW)), dim3(256), smem, st, qkv, out, dout, lse, BH, T, d, heads, key_len, row_off, scale * 1.44269504088896340736f, scale, dqkv);
      } else {
        allow_lds((attn_seq_bwd_fused_kernel<DKT, NT, LS>), smem);
        LAUNCH_S(BH, T, dk, 10.0 * B * T * (double)T * d, 24.0 * B * T * (double)d, (attn_seq_bwd_fused_kernel<DKT, NT, LS>), dim3(cdiv(BH, C::PW)), dim3(256), smem, st, qkv, out, dout, lse, BH, T, d, heads, key_len, row_off, scale * 1.44269504088896340736f, scale, dqkv);
      }
    });
    INTEL_CHECK_LAUNCH();
    return 0;
  }
}
