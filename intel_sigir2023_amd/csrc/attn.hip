// Per-(session, head) softmax attention, forward and backward, fp32 MFMA (gfx950).
//
// Restates modules/layers.py:50-60 (scaled_dot_product_attention) for
//   - the item / score towers: NO mask, padded rows are keys and queries (IntEL.py:184,193),
//   - the BERT4Rec blocks: key mask j < history_len (models/GeneralSeq.py:100).
// The reference's "minus tensor-global max" (layers.py:57) is a mathematical no-op and is not
// reproduced; rows with no valid key give 0 (NaN->0, layers.py:58).
//
// Layout: one packed activation buffer qkv[B*T, 3d] = [q | k | v] per row (the fused QKV GEMM's
// output); head h owns columns h*dk..(h+1)*dk-1 of each third.
//
// Forward (flash-style, one workgroup = 64 queries of one (session, head), wave = 16 queries):
//   S^T = K Q^T is computed with the KEY on the accumulator rows and the QUERY on the lane, so the
//   probabilities sit in registers exactly in the B-operand layout of the following O^T = V^T P^T
//   product: no LDS round trip for P (cdna guide §3 "accumulator tile as the next operand").
// Backward = two kernels with the same structure (S and dP are recomputed from the saved
//   log-sum-exp): attn_bwd_dq (wave owns 16 queries, sweeps keys) and attn_bwd_dkv (wave owns 16
//   keys, sweeps queries); neither needs a cross-wave reduction or atomics.
#include "kernels.h"
#include "planes.h"
#include <stdio.h>
#include <stdlib.h>

#define AT_QB 64   // rows (queries, or keys in the dK/dV kernel) owned by a workgroup: 4 waves x 16
#define AT_KB 32   // rows staged in LDS per iteration: 2 x 32 x (dk+4) floats = 33 KB at dk = 128 -> 4 workgroups / CU

template <int DKT>
struct AttnSmem {
  static constexpr int LD = DKT * 16 + 4;   // ld % 8 == 4: b128 row reads and b32 column reads conflict-free
};

// rows [r0, r0 + AT_KB) of one third of qkv (column offset coff), zero padded, on their way to LDS through registers: the loads of the NEXT
// block are issued before the current block's products and land under them
template <int DKT>
struct StageRegs {
  static constexpr int C4 = DKT * 4;                            // float4 per row
  static constexpr int NV = (AT_KB * C4 + 255) / 256;           // float4 per thread
  f32x4 v[NV];
  __device__ __forceinline__ void load(const float* __restrict__ base, int ldg, int coff, int dk, int r0, int T, int tid) {
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int i = tid + n * 256, r = i / C4, c4 = i - r * C4;
      const int row = r0 + r, col = c4 * 4;
      v[n] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < AT_KB * C4 && row < T && col < dk) v[n] = *reinterpret_cast<const f32x4*>(base + (size_t)row * ldg + coff + col);
    }
  }
  __device__ __forceinline__ void store(float* dst, int tid) const {
    constexpr int LD = AttnSmem<DKT>::LD;
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int i = tid + n * 256, r = i / C4, c4 = i - r * C4;
      if (i < AT_KB * C4) *reinterpret_cast<f32x4*>(dst + r * LD + c4 * 4) = v[n];
    }
  }
};

// fragment of one row (B-operand / "row on the lane" form): element s of group g = x[row][g*16+4*(lane>>4)+s]
template <int DKT>
__device__ __forceinline__ void load_row_frags(f32x4 (&f)[DKT], const float* __restrict__ rowp, bool rowok, int dk, int lane) {
#pragma unroll
  for (int g = 0; g < DKT; ++g) {
    const int col = g * 16 + 4 * (lane >> 4);
    f[g] = (rowok && col < dk) ? *reinterpret_cast<const f32x4*>(rowp + col) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

__device__ __forceinline__ float group_max16(float v) {   // over the 4 lane groups sharing lane&15
  v = fmaxf(v, __shfl_xor(v, 16));
  return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float group_sum16(float v) {
  v += __shfl_xor(v, 16);
  return v + __shfl_xor(v, 32);
}

// "Head dim on the accumulator rows" products (O^T = V^T P^T, dV^T = dO^T P, dK^T = Q^T dS, dQ^T = K^T dS^T): the A operand is a staged row read
// ALONG the head dim.  When the head dim is whole 64-column chunks (DKT % 4 == 0: head dims 64 and 128) a lane reads four consecutive dims as ONE
// b128 and feeds four MFMAs — accumulator 4c+t row i is dim 64c + 4i + t — instead of one b32 read per MFMA (accumulator dt row i = dim 16dt + i).
#ifndef ATTN_WIDE
#define ATTN_WIDE 1
#endif
template <int DKT>
__device__ __forceinline__ void dimT_mma(f32x4 (&acc)[DKT], const float* __restrict__ rowp, float bval, int lane) {
  if constexpr (ATTN_WIDE && DKT % 4 == 0) {
#pragma unroll
    for (int c = 0; c < DKT / 4; ++c) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(rowp + 64 * c + 4 * (lane & 15));
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[4 * c + t] = mfma16(x[t], bval, acc[4 * c + t]);
    }
  } else {
#pragma unroll
    for (int dt = 0; dt < DKT; ++dt) acc[dt] = mfma16(rowp[dt * 16 + (lane & 15)], bval, acc[dt]);
  }
}
// the lane's share of one output row (16 dims per 64-column chunk / per 16-column tile), scaled
template <int DKT>
__device__ __forceinline__ void dimT_store(float* __restrict__ rowp, const f32x4 (&acc)[DKT], int dk, int lane, float mul) {
  if constexpr (ATTN_WIDE && DKT % 4 == 0) {
#pragma unroll
    for (int c = 0; c < DKT / 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = 64 * c + 16 * (lane >> 4) + 4 * r;
        if (col < dk) *reinterpret_cast<f32x4*>(rowp + col) = f32x4{acc[4 * c][r], acc[4 * c + 1][r], acc[4 * c + 2][r], acc[4 * c + 3][r]} * mul;
      }
  } else {
#pragma unroll
    for (int dt = 0; dt < DKT; ++dt) {
      const int col = dt * 16 + 4 * (lane >> 4);
      if (col < dk) *reinterpret_cast<f32x4*>(rowp + col) = acc[dt] * mul;
    }
  }
}

// bf16 mode (gemm_planes() == 1, lists longer than 64, head dims 64 / 128): the same products as ONE v_mfma_f32_16x16x16_bf16 per four fp32 steps, both
// operands rounded to bf16 as they leave LDS / the accumulators (lane (i, j) supplies k = 4j..4j+3: the element order of the four fp32 steps, so the
// accumulator-as-operand layouts carry over).  tile = the 16 staged rows the k index runs over; b = their four probabilities / dS values of this lane.
template <int DKT>
__device__ __forceinline__ void dimT_mma_bf(f32x4 (&acc)[DKT], const float* __restrict__ tile, int LD, s16x4 b, int lane) {
  const float* r0 = tile + 4 * (lane >> 4) * LD;
  if constexpr (DKT % 4 == 0) {
#pragma unroll
    for (int c = 0; c < DKT / 4; ++c) {
      f32x4 x[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) x[s] = *reinterpret_cast<const f32x4*>(r0 + s * LD + 64 * c + 4 * (lane & 15));
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[4 * c + t] = planes::mma4_bf16(planes::to_bf16x4(f32x4{x[0][t], x[1][t], x[2][t], x[3][t]}), b, acc[4 * c + t]);
    }
  } else {
#pragma unroll
    for (int dt = 0; dt < DKT; ++dt) {
      const float* cp = r0 + dt * 16 + (lane & 15);
      acc[dt] = planes::mma4_bf16(planes::to_bf16x4(f32x4{cp[0], cp[LD], cp[2 * LD], cp[3 * LD]}), b, acc[dt]);
    }
  }
}
// S-type tile product: 16 staged rows (A operand, read along the head dim) against the lane's own row fragments
template <int DKT, bool BF>
__device__ __forceinline__ f32x4 rowT_mma(const float* __restrict__ tile, int LD, const f32x4 (&bf32)[DKT], const s16x4 (&bbf)[DKT], f32x4 acc, int lane) {
#pragma unroll
  for (int g = 0; g < DKT; ++g) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(tile + (lane & 15) * LD + g * 16 + 4 * (lane >> 4));
    if constexpr (BF) {
      acc = planes::mma4_bf16(planes::to_bf16x4(a), bbf[g], acc);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = mfma16(a[s], bf32[g][s], acc);
    }
  }
  return acc;
}

// Workgroup -> ((session, head) pair, 64-row block) for the 1-D grids of the flash-style kernels.  The blocks of one pair re-read the same
// K / V (forward, dQ) or Q / dO (dK / dV) rows; consecutive workgroup ids land on consecutive XCDs (id % 8), each with its own L2, so the
// blocks of a pair are given ids 8 apart: same XCD, dispatched together, the second reader is served by that L2.
struct AttnBlock { int bh, y; };
__device__ __forceinline__ AttnBlock attn_block(int nbh, int ny) {
  const int id = blockIdx.x;
  if ((nbh & 7) == 0) {
    const int slot = id >> 3, g = slot / ny;
    return AttnBlock{g * 8 + (id & 7), slot - g * ny};
  }
  const int bh = id / ny;
  return AttnBlock{bh, id - bh * ny};
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int DKT>
__global__ __launch_bounds__(256, DKT >= 6 ? 3 : 4) void attn_fwd_kernel(const float* __restrict__ qkv, int T, int d, int heads,
                                                       const int* __restrict__ key_len, float scale,
                                                       float* __restrict__ out, float* __restrict__ lse, const int* __restrict__ row_off) {
  constexpr int LD = AttnSmem<DKT>::LD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;
  float* Vs = smem + AT_KB * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ny = (T + AT_QB - 1) / AT_QB;
  const AttnBlock blk = attn_block((int)gridDim.x / ny, ny);
  const int b = blk.bh / heads, h = blk.bh - b * heads;
  const int dk = d / heads, ldg = 3 * d;
  const int nkeys = key_len ? min(key_len[b], T) : T;
  // packed rows (row_off): the session's nkeys valid rows start at row_off[b] and nothing else of it is in the buffers; the per-(session, head)
  // statistics (log-sum-exp, dsum, the dS scratch) keep their padded [.., T] index either way
  const int nrow = row_off ? nkeys : T;
  const size_t row0 = row_off ? (size_t)row_off[b] : (size_t)b * T;
  if (blk.y * AT_QB >= nrow) return;
  const float* base = qkv + row0 * ldg;
  const int q = blk.y * AT_QB + wave * 16 + (lane & 15);
  f32x4 qf[DKT];
  load_row_frags<DKT>(qf, base + (size_t)q * ldg + h * dk, q < nrow, dk, lane);
  f32x4 oT[DKT];
#pragma unroll
  for (int i = 0; i < DKT; ++i) oT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  StageRegs<DKT> kreg, vreg;
  kreg.load(base, ldg, d + h * dk, dk, 0, nrow, tid);
  vreg.load(base, ldg, 2 * d + h * dk, dk, 0, nrow, tid);
  for (int kb = 0; kb < nkeys; kb += AT_KB) {
    __syncthreads();
    kreg.store(Ks, tid);
    vreg.store(Vs, tid);
    __syncthreads();
    if (kb + AT_KB < nkeys) {
      kreg.load(base, ldg, d + h * dk, dk, kb + AT_KB, nrow, tid);
      vreg.load(base, ldg, 2 * d + h * dk, dk, kb + AT_KB, nrow, tid);
    }
    if (blk.y * AT_QB + wave * 16 >= nrow) continue;      // a wave whose 16 queries are all past the list only helps staging
    f32x4 st[AT_KB / 16];
#pragma unroll
    for (int kt = 0; kt < AT_KB / 16; ++kt) {
      st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kb + kt * 16 < nkeys) {
#pragma unroll
        for (int g = 0; g < DKT; ++g) {
          const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + (kt * 16 + (lane & 15)) * LD + g * 16 + 4 * (lane >> 4));
#pragma unroll
          for (int s = 0; s < 4; ++s) st[kt] = mfma16(kf[s], qf[g][s], st[kt]);
        }
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < AT_KB / 16; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kb + kt * 16 + 4 * (lane >> 4) + r;
        const float v = key < nkeys ? st[kt][r] * scale : -INFINITY;
        st[kt][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = group_max16(mx);
    const float m_new = fmaxf(m_run, mx);        // finite: this block holds >= 1 valid key
    const float corr = (m_run == -INFINITY) ? 0.f : expf(m_run - m_new);
    float ps = 0.f;
#pragma unroll
    for (int kt = 0; kt < AT_KB / 16; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = expf(st[kt][r] - m_new);   // exp(-inf) = 0 for masked keys
        st[kt][r] = p;
        ps += p;
      }
    ps = group_sum16(ps);
    l_run = l_run * corr + ps;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < DKT; ++i) oT[i] *= corr;
#pragma unroll
    for (int kt = 0; kt < AT_KB / 16; ++kt) {
      if (kb + kt * 16 < nkeys) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          dimT_mma<DKT>(oT, Vs + (kt * 16 + 4 * (lane >> 4) + s) * LD, st[kt][s], lane);
        }
      }
    }
  }
  if (q < nrow) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    dimT_store<DKT>(out + (row0 + q) * d + h * dk, oT, dk, lane, inv);
    if (lane < 16) lse[((size_t)b * heads + h) * T + q] = l_run > 0.f ? m_run + logf(l_run) : INFINITY;
  }
}

// bf16 mode forward.  The mode's arithmetic is O = bf(e) bf(V) / sum(e) with e = exp(s - rowmax) (the tests' emulation restates it): the
// rounding of e depends on the row's FINAL maximum, which an online softmax does not know when it rounds.  Hence two sweeps over the keys: the first
// computes S = bf(Q) bf(K)^T alone for the row maximum (a sixteenth of the fp32 MFMA cycles), the second e, its fp32 row sum and the O product.
template <int DKT>
__global__ __launch_bounds__(256, DKT >= 6 ? 3 : 4) void attn_fwd_bf_kernel(const float* __restrict__ qkv, int T, int d, int heads,
                                                       const int* __restrict__ key_len, float scale,
                                                       float* __restrict__ out, float* __restrict__ lse, const int* __restrict__ row_off) {
  constexpr int LD = AttnSmem<DKT>::LD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;
  float* Vs = smem + AT_KB * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ny = (T + AT_QB - 1) / AT_QB;
  const AttnBlock blk = attn_block((int)gridDim.x / ny, ny);
  const int b = blk.bh / heads, h = blk.bh - b * heads;
  const int dk = d / heads, ldg = 3 * d;
  const int nkeys = key_len ? min(key_len[b], T) : T;
  const int nrow = row_off ? nkeys : T;
  const size_t row0 = row_off ? (size_t)row_off[b] : (size_t)b * T;
  if (blk.y * AT_QB >= nrow) return;
  const float* base = qkv + row0 * ldg;
  const int q = blk.y * AT_QB + wave * 16 + (lane & 15);
  const bool wave_live = blk.y * AT_QB + wave * 16 < nrow;
  f32x4 qf[DKT];
  s16x4 qb[DKT];
  load_row_frags<DKT>(qf, base + (size_t)q * ldg + h * dk, q < nrow, dk, lane);
#pragma unroll
  for (int g = 0; g < DKT; ++g) qb[g] = planes::to_bf16x4(qf[g]);
  StageRegs<DKT> kreg, vreg;
  // sweep 1: row maximum
  float mx = -INFINITY;
  kreg.load(base, ldg, d + h * dk, dk, 0, nrow, tid);
  for (int kb = 0; kb < nkeys; kb += AT_KB) {
    __syncthreads();
    kreg.store(Ks, tid);
    __syncthreads();
    if (kb + AT_KB < nkeys) kreg.load(base, ldg, d + h * dk, dk, kb + AT_KB, nrow, tid);
    if (!wave_live) continue;
#pragma unroll
    for (int kt = 0; kt < AT_KB / 16; ++kt) {
      if (kb + kt * 16 >= nkeys) continue;
      const f32x4 st = rowT_mma<DKT, true>(Ks + kt * 16 * LD, LD, qf, qb, f32x4{0.f, 0.f, 0.f, 0.f}, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (kb + kt * 16 + 4 * (lane >> 4) + r < nkeys) mx = fmaxf(mx, st[r] * scale);
    }
  }
  const float m = group_max16(mx);
  const float m_use = (m == -INFINITY) ? 0.f : m;
  // sweep 2: e, its row sum, O
  f32x4 oT[DKT];
#pragma unroll
  for (int i = 0; i < DKT; ++i) oT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float ps = 0.f;
  kreg.load(base, ldg, d + h * dk, dk, 0, nrow, tid);
  vreg.load(base, ldg, 2 * d + h * dk, dk, 0, nrow, tid);
  for (int kb = 0; kb < nkeys; kb += AT_KB) {
    __syncthreads();
    kreg.store(Ks, tid);
    vreg.store(Vs, tid);
    __syncthreads();
    if (kb + AT_KB < nkeys) {
      kreg.load(base, ldg, d + h * dk, dk, kb + AT_KB, nrow, tid);
      vreg.load(base, ldg, 2 * d + h * dk, dk, kb + AT_KB, nrow, tid);
    }
    if (!wave_live) continue;
#pragma unroll
    for (int kt = 0; kt < AT_KB / 16; ++kt) {
      if (kb + kt * 16 >= nkeys) continue;
      f32x4 st = rowT_mma<DKT, true>(Ks + kt * 16 * LD, LD, qf, qb, f32x4{0.f, 0.f, 0.f, 0.f}, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = (kb + kt * 16 + 4 * (lane >> 4) + r < nkeys) ? expf(st[r] * scale - m_use) : 0.f;
        st[r] = e;
        ps += e;
      }
      dimT_mma_bf<DKT>(oT, Vs + kt * 16 * LD, LD, planes::to_bf16x4(st), lane);
    }
  }
  const float l = group_sum16(ps);
  if (q < nrow) {
    const float inv = l > 0.f ? 1.f / l : 0.f;
    dimT_store<DKT>(out + (row0 + q) * d + h * dk, oT, dk, lane, inv);
    if (lane < 16) lse[((size_t)b * heads + h) * T + q] = l > 0.f ? m_use + logf(l) : INFINITY;
  }
}

// ------------------------------------------------------------------------------------------
// backward, dQ: wave owns 16 queries, sweeps key blocks.  Also writes dsum[q] = sum_d dO*O.
// ------------------------------------------------------------------------------------------
template <int DKT>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                          const float* __restrict__ dout, const float* __restrict__ lse,
                                                          int T, int d, int heads, const int* __restrict__ key_len,
                                                          float scale, float* __restrict__ dqkv, float* __restrict__ dsum, const int* __restrict__ row_off) {
  constexpr int LD = AttnSmem<DKT>::LD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;
  float* Vs = smem + AT_KB * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ny = (T + AT_QB - 1) / AT_QB;
  const AttnBlock blk = attn_block((int)gridDim.x / ny, ny);
  const int b = blk.bh / heads, h = blk.bh - b * heads;
  const int dk = d / heads, ldg = 3 * d;
  const int nkeys = key_len ? min(key_len[b], T) : T;
  // packed rows (row_off): the session's nkeys valid rows start at row_off[b] and nothing else of it is in the buffers; the per-(session, head)
  // statistics (log-sum-exp, dsum, the dS scratch) keep their padded [.., T] index either way
  const int nrow = row_off ? nkeys : T;
  const size_t row0 = row_off ? (size_t)row_off[b] : (size_t)b * T;
  if (blk.y * AT_QB >= nrow) return;
  const float* base = qkv + row0 * ldg;
  const int q = blk.y * AT_QB + wave * 16 + (lane & 15);
  const bool qok = q < nrow;
  f32x4 qf[DKT], dof[DKT];
  load_row_frags<DKT>(qf, base + (size_t)q * ldg + h * dk, qok, dk, lane);
  load_row_frags<DKT>(dof, dout + (row0 + q) * d + h * dk, qok, dk, lane);
  float dsm = 0.f;
  {
    f32x4 of[DKT];
    load_row_frags<DKT>(of, out + (row0 + q) * d + h * dk, qok, dk, lane);
#pragma unroll
    for (int g = 0; g < DKT; ++g)
#pragma unroll
      for (int s = 0; s < 4; ++s) dsm += of[g][s] * dof[g][s];
    dsm = group_sum16(dsm);
  }
  const float lq = qok ? lse[((size_t)b * heads + h) * T + q] : INFINITY;
  if (qok && lane < 16) dsum[((size_t)b * heads + h) * T + q] = dsm;
  f32x4 dqT[DKT];
#pragma unroll
  for (int i = 0; i < DKT; ++i) dqT[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  StageRegs<DKT> kreg, vreg;
  kreg.load(base, ldg, d + h * dk, dk, 0, nrow, tid);
  vreg.load(base, ldg, 2 * d + h * dk, dk, 0, nrow, tid);
  for (int kb = 0; kb < nkeys; kb += AT_KB) {
    __syncthreads();
    kreg.store(Ks, tid);
    vreg.store(Vs, tid);
    __syncthreads();
    if (kb + AT_KB < nkeys) {
      kreg.load(base, ldg, d + h * dk, dk, kb + AT_KB, nrow, tid);
      vreg.load(base, ldg, 2 * d + h * dk, dk, kb + AT_KB, nrow, tid);
    }
#pragma unroll
    for (int kt = 0; kt < AT_KB / 16; ++kt) {
      if (kb + kt * 16 >= nkeys) continue;
      f32x4 sT = f32x4{0.f, 0.f, 0.f, 0.f}, dpT = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < DKT; ++g) {
        const int off = (kt * 16 + (lane & 15)) * LD + g * 16 + 4 * (lane >> 4);
        const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + off);
        const f32x4 vf = *reinterpret_cast<const f32x4*>(Vs + off);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          sT = mfma16(kf[s], qf[g][s], sT);
          dpT = mfma16(vf[s], dof[g][s], dpT);
        }
      }
      f32x4 dsT;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kb + kt * 16 + 4 * (lane >> 4) + r;
        const float p = key < nkeys ? expf(sT[r] * scale - lq) : 0.f;
        dsT[r] = p * (dpT[r] - dsm) * scale;
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        dimT_mma<DKT>(dqT, Ks + (kt * 16 + 4 * (lane >> 4) + s) * LD, dsT[s], lane);
      }
    }
  }
  if (qok) {
    dimT_store<DKT>(dqkv + (row0 + q) * ldg + h * dk, dqT, dk, lane, 1.f);
  }
}

// ------------------------------------------------------------------------------------------
// backward, dK/dV: wave owns 16 keys, sweeps query blocks (Q and dO staged in LDS).
// ------------------------------------------------------------------------------------------
template <int DKT, bool BF>
__device__ __forceinline__ void attn_bwd_dkv_body(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                           const float* __restrict__ lse, const float* __restrict__ dsum,
                                                           int T, int d, int heads, const int* __restrict__ key_len,
                                                           float scale, float* __restrict__ dqkv, float* __restrict__ dS, int ldS, const int* __restrict__ row_off) {
  constexpr int LD = AttnSmem<DKT>::LD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;
  float* Os = smem + AT_KB * LD;
  float* Ls = smem + 2 * AT_KB * LD;   // [64] lse
  float* Ds = Ls + AT_KB;              // [64] dsum
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ny = (T + AT_QB - 1) / AT_QB;
  const AttnBlock blk = attn_block((int)gridDim.x / ny, ny);
  const int b = blk.bh / heads, h = blk.bh - b * heads;
  const int dk = d / heads, ldg = 3 * d;
  const int nkeys = key_len ? min(key_len[b], T) : T;
  // packed rows (row_off): the session's nkeys valid rows start at row_off[b] and nothing else of it is in the buffers; the per-(session, head)
  // statistics (log-sum-exp, dsum, the dS scratch) keep their padded [.., T] index either way
  const int nrow = row_off ? nkeys : T;
  const size_t row0 = row_off ? (size_t)row_off[b] : (size_t)b * T;
  if (blk.y * AT_QB >= nrow) return;
  const float* base = qkv + row0 * ldg;
  const int key = blk.y * AT_QB + wave * 16 + (lane & 15);
  const bool kok = key < nrow;
  f32x4 kf[DKT], vf[DKT];
  load_row_frags<DKT>(kf, base + (size_t)key * ldg + d + h * dk, kok, dk, lane);
  load_row_frags<DKT>(vf, base + (size_t)key * ldg + 2 * d + h * dk, kok, dk, lane);
  s16x4 kb16[DKT], vb16[DKT];      // bf16 mode: the wave's K / V rows rounded once
#pragma unroll
  for (int g = 0; g < DKT; ++g) {
    kb16[g] = planes::to_bf16x4(kf[g]);
    vb16[g] = planes::to_bf16x4(vf[g]);
  }
  f32x4 dkT[DKT], dvT[DKT];
#pragma unroll
  for (int i = 0; i < DKT; ++i) {
    dkT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    dvT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool key_live = key < nkeys;          // masked keys get exactly zero gradient
  const bool wave_live = (blk.y * AT_QB + wave * 16) < nkeys;

  StageRegs<DKT> qreg, oreg;
  float lreg = INFINITY, dreg = 0.f;
  const float* dob = dout + row0 * d;
  // (prefetching the next block's Q / dO rows under the products was measured: no gain — two to four resident workgroups already cover the
  // staging — and it costs 16-32 registers, i.e. a wave per SIMD at head dim 64)
  for (int qb = 0; qb < nrow; qb += AT_KB) {
    __syncthreads();
    qreg.load(base, ldg, h * dk, dk, qb, nrow, tid);
    oreg.load(dob, d, h * dk, dk, qb, nrow, tid);
    {
      const int qq = qb + tid;
      const bool ok = tid < AT_KB && qq < nrow;
      lreg = ok ? lse[((size_t)b * heads + h) * T + qq] : INFINITY;
      dreg = ok ? dsum[((size_t)b * heads + h) * T + qq] : 0.f;
    }
    qreg.store(Qs, tid);
    oreg.store(Os, tid);
    if (tid < AT_KB) {
      Ls[tid] = lreg;
      Ds[tid] = dreg;
    }
    __syncthreads();
    if (!wave_live) continue;
#pragma unroll
    for (int qt = 0; qt < AT_KB / 16; ++qt) {
      if (qb + qt * 16 >= nrow) continue;
      const f32x4 sa = rowT_mma<DKT, BF>(Qs + qt * 16 * LD, LD, kf, kb16, f32x4{0.f, 0.f, 0.f, 0.f}, lane);     // S[query][key]
      const f32x4 dp = rowT_mma<DKT, BF>(Os + qt * 16 * LD, LD, vf, vb16, f32x4{0.f, 0.f, 0.f, 0.f}, lane);     // dP[query][key]
      f32x4 pr, ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = qt * 16 + 4 * (lane >> 4) + r;
        const float p = key_live ? expf(sa[r] * scale - Ls[ql]) : 0.f;
        pr[r] = p;
        ds[r] = p * (dp[r] - Ds[ql]) * scale;
        // the dS tile for the dQ = dS K kernel (row = query, ldS floats per row); keys >= nkeys / queries >= T are never read
        if (dS && qb + ql < nrow && kok) dS[((size_t)blk.bh * T + qb + ql) * ldS + key] = ds[r];
      }
      if constexpr (BF) {
        dimT_mma_bf<DKT>(dvT, Os + qt * 16 * LD, LD, planes::to_bf16x4(pr), lane);
        dimT_mma_bf<DKT>(dkT, Qs + qt * 16 * LD, LD, planes::to_bf16x4(ds), lane);
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int roff = (qt * 16 + 4 * (lane >> 4) + s) * LD;
          dimT_mma<DKT>(dvT, Os + roff, pr[s], lane);   // dV^T[dim][key] += dO^T P
          dimT_mma<DKT>(dkT, Qs + roff, ds[s], lane);   // dK^T[dim][key] += Q^T dS
        }
      }
    }
  }
  if (kok) {
    float* drow = dqkv + (row0 + key) * ldg + h * dk;
    dimT_store<DKT>(drow + d, dkT, dk, lane, 1.f);
    dimT_store<DKT>(drow + 2 * d, dvT, dk, lane, 1.f);
  }
}

#define DKV_PARAMS const float* __restrict__ qkv, const float* __restrict__ dout, const float* __restrict__ lse, const float* __restrict__ dsum, int T, int d, \
                   int heads, const int* __restrict__ key_len, float scale, float* __restrict__ dqkv, float* __restrict__ dS, int ldS, const int* __restrict__ row_off
#define DKV_ARGS qkv, dout, lse, dsum, T, d, heads, key_len, scale, dqkv, dS, ldS, row_off
template <int DKT>
__global__ __launch_bounds__(256, DKT >= 6 ? 2 : (DKT >= 3 ? 3 : 4)) void attn_bwd_dkv_kernel(DKV_PARAMS) { attn_bwd_dkv_body<DKT, false>(DKV_ARGS); }
template <int DKT>
__global__ __launch_bounds__(256, DKT >= 6 ? 2 : (DKT >= 3 ? 3 : 4)) void attn_bwd_dkv_bf_kernel(DKV_PARAMS) { attn_bwd_dkv_body<DKT, true>(DKV_ARGS); }

// ------------------------------------------------------------------------------------------
// backward without recomputation in the dQ pass: dsum = rowsum(dO * O) first (one wave per row), the dK/dV kernel above
// stores its dS tiles, and dQ = dS K is one product per key tile (5 tile products per pair instead of 7; the dS
// scratch costs 8 T^2 bytes of traffic per (session, head)).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_dsum_kernel(const float* __restrict__ out, const float* __restrict__ dout, int T, int d,
                                                        int heads, long long rows, float* __restrict__ dsum,
                                                        const int* __restrict__ key_len, const int* __restrict__ row_off) {
  // 16 lanes per (row, head), four of them per wave, 16-byte loads.  rows = B * T positions; packed rows: position (b, t) is data row
  // row_off[b] + t when t < key_len[b] and absent otherwise
  const int lane = threadIdx.x & 63, sub = lane & 15;
  const long long i = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);      // (b*T + t, h) flattened
  bool ok = i < rows * heads;
  const long long ic = ok ? i : 0;
  const long long pos = ic / heads;
  const int h = (int)(ic - pos * heads), dk = d / heads;
  const long long b = pos / T, t = pos - b * T;
  long long row = pos;
  if (row_off) {
    ok = ok && t < min(key_len[b], T);
    row = ok ? (long long)row_off[b] + t : 0;
  }
  float s = 0.f;
  for (int c = sub * 4; c < dk; c += 64) {
    const f32x4 o = *reinterpret_cast<const f32x4*>(out + row * d + h * dk + c);
    const f32x4 g = *reinterpret_cast<const f32x4*>(dout + row * d + h * dk + c);
    s += (o[0] * g[0] + o[1] * g[1]) + (o[2] * g[2] + o[3] * g[3]);
  }
  s += __shfl_xor(s, 1);
  s += __shfl_xor(s, 2);
  s += __shfl_xor(s, 4);
  s += __shfl_xor(s, 8);
  if (ok && sub == 0) dsum[(b * heads + h) * T + t] = s;
}

template <int DKT, bool BF>
__device__ __forceinline__ void attn_bwd_dq_ds_body(const float* __restrict__ qkv, const float* __restrict__ dS, int ldS, int T,
                                                             int d, int heads, const int* __restrict__ key_len,
                                                             float* __restrict__ dqkv, const int* __restrict__ row_off) {
  constexpr int LD = AttnSmem<DKT>::LD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ny = (T + AT_QB - 1) / AT_QB;
  const AttnBlock blk = attn_block((int)gridDim.x / ny, ny);
  const int b = blk.bh / heads, h = blk.bh - b * heads;
  const int dk = d / heads, ldg = 3 * d;
  const int nkeys = key_len ? min(key_len[b], T) : T;
  // packed rows (row_off): the session's nkeys valid rows start at row_off[b] and nothing else of it is in the buffers; the per-(session, head)
  // statistics (log-sum-exp, dsum, the dS scratch) keep their padded [.., T] index either way
  const int nrow = row_off ? nkeys : T;
  const size_t row0 = row_off ? (size_t)row_off[b] : (size_t)b * T;
  if (blk.y * AT_QB >= nrow) return;
  const float* base = qkv + row0 * ldg;
  const int q = blk.y * AT_QB + wave * 16 + (lane & 15);
  const bool qok = q < nrow;
  const float* dSq = dS + ((size_t)blk.bh * T + (qok ? q : 0)) * ldS;
  f32x4 dqT[DKT];
#pragma unroll
  for (int i = 0; i < DKT; ++i) dqT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  StageRegs<DKT> kreg;
  kreg.load(base, ldg, d + h * dk, dk, 0, nrow, tid);
  for (int kb = 0; kb < nkeys; kb += AT_KB) {
    __syncthreads();
    kreg.store(Ks, tid);
    __syncthreads();
    if (kb + AT_KB < nkeys) kreg.load(base, ldg, d + h * dk, dk, kb + AT_KB, nrow, tid);
    if (blk.y * AT_QB + wave * 16 >= nrow) continue;
#pragma unroll
    for (int kt = 0; kt < AT_KB / 16; ++kt) {
      if (kb + kt * 16 >= nkeys) continue;
      const int key0 = kb + kt * 16 + 4 * (lane >> 4);
      f32x4 dsT = f32x4{0.f, 0.f, 0.f, 0.f};
      if (qok && key0 < ldS) dsT = *reinterpret_cast<const f32x4*>(dSq + key0);
#pragma unroll
      for (int r = 0; r < 4; ++r) dsT[r] = key0 + r < nkeys ? dsT[r] : 0.f;      // masked / padding keys: nothing was stored
      if constexpr (BF) {
        dimT_mma_bf<DKT>(dqT, Ks + kt * 16 * LD, LD, planes::to_bf16x4(dsT), lane);
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) dimT_mma<DKT>(dqT, Ks + (kt * 16 + 4 * (lane >> 4) + s) * LD, dsT[s], lane);
      }
    }
  }
  if (qok) {
    dimT_store<DKT>(dqkv + (row0 + q) * ldg + h * dk, dqT, dk, lane, 1.f);
  }
}

#define DQDS_PARAMS const float* __restrict__ qkv, const float* __restrict__ dS, int ldS, int T, int d, int heads, const int* __restrict__ key_len, \
                    float* __restrict__ dqkv, const int* __restrict__ row_off
template <int DKT>
__global__ __launch_bounds__(256, 4) void attn_bwd_dq_ds_kernel(DQDS_PARAMS) { attn_bwd_dq_ds_body<DKT, false>(qkv, dS, ldS, T, d, heads, key_len, dqkv, row_off); }
template <int DKT>
__global__ __launch_bounds__(256, 4) void attn_bwd_dq_ds_bf_kernel(DQDS_PARAMS) { attn_bwd_dq_ds_body<DKT, true>(qkv, dS, ldS, T, d, heads, key_len, dqkv, row_off); }

static inline int attn_ds_pitch(int T) { return (T + 3) & ~3; }
static inline bool attn_ds_scheme() { return true; }      // (the recompute form below stays for small problems only: launch_attn_bwd)

static inline bool attn_seq_path(int T, int dk) { return attn_seq_supported(T, dk); }
bool attn_packed_supported(int T, int dk) { return attn_seq_path(T, dk) ? attn_seq_packed_supported(T, dk) : true; }

size_t attn_bwd_scratch_floats(int B, int T, int d, int heads) {
  size_t f = rup_sz((size_t)B * heads * T, 64);
  if (heads > 0 && attn_seq_path(T, d / heads)) f += attn_seq_bwd_scratch_floats(B, T, heads);
  else if (attn_ds_scheme()) f += (size_t)B * heads * T * attn_ds_pitch(T);
  return f;
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static int check_attn_shape(int T, int d, int heads) {
  INTEL_CHECK_ARG(heads > 0 && d % heads == 0, "attention: d=%d not divisible by heads=%d", d, heads);
  const int dk = d / heads;
  INTEL_CHECK_ARG(dk % 4 == 0 && dk <= 128 && d % 4 == 0, "attention: head dim %d unsupported (need multiple of 4, <= 128)", dk);
  INTEL_CHECK_ARG(T > 0, "attention: empty sequence");
  return 0;
}

// bf16 mode on the general path: lists / histories longer than 64 (the shorter ones are the whole-sequence and one-kernel paths' business and have
// their own rules), head dims 64 and 128, the dS scheme.  The tests' bf16 emulation restates exactly this condition.
static inline bool attn_bf16_products(int T, int dk) { return gemm_planes() == 1 && T > 64 && (dk == 64 || dk == 128) && attn_ds_scheme(); }
#define ATTN_DISPATCH_BF(DKT_RT, CALL)            \
  if ((DKT_RT) == 4) { constexpr int DKT = 4; CALL; } else { constexpr int DKT = 8; CALL; }

#define ATTN_DISPATCH(DKT_RT, CALL)               \
  switch (DKT_RT) {                               \
    case 1: { constexpr int DKT = 1; CALL; } break; \
    case 2: { constexpr int DKT = 2; CALL; } break; \
    case 3: { constexpr int DKT = 3; CALL; } break; \
    case 4: { constexpr int DKT = 4; CALL; } break; \
    case 5: case 6: { constexpr int DKT = 6; CALL; } break; \
    default: { constexpr int DKT = 8; CALL; } break; \
  }

int launch_attn_fwd(const float* qkv, int B, int T, int d, int heads, const int* key_len, float* out, float* lse,
                    hipStream_t st, const int* row_off) {
  if (B <= 0) return 0;
  int rc = check_attn_shape(T, d, heads);
  if (rc) return rc;
  const int dk = d / heads, dkt = cdiv(dk, 16);
  const float scale = 1.0f / sqrtf((float)dk);
  if (attn_seq_path(T, dk)) return launch_attn_seq_fwd(qkv, B, T, d, heads, key_len, out, lse, st, row_off);
  INTEL_CHECK_ARG(!row_off || key_len, "attention: packed rows need the session lengths");
  dim3 grid(B * heads * cdiv(T, AT_QB));
  if (attn_p3_supported(T, dk)) return launch_attn_p3_fwd(qkv, B, T, d, heads, key_len, out, lse, st, row_off);
  if (attn_bf16_products(T, dk)) {
    ATTN_DISPATCH_BF(dkt, {
      size_t smem = (size_t)2 * AT_KB * AttnSmem<DKT>::LD * sizeof(float);
      allow_lds(attn_fwd_bf_kernel<DKT>, smem);
      LAUNCH_S(B * heads, T, dk, 4.0 * B * T * (double)T * d, 16.0 * B * T * (double)d, attn_fwd_bf_kernel<DKT>, grid, dim3(256), smem, st, qkv, T, d, heads, key_len, scale, out, lse, row_off);
    });
    INTEL_CHECK_LAUNCH();
    return 0;
  }
  ATTN_DISPATCH(dkt, {
    size_t smem = (size_t)2 * AT_KB * AttnSmem<DKT>::LD * sizeof(float);
    allow_lds(attn_fwd_kernel<DKT>, smem);
    LAUNCH_S(B * heads, T, dk, 4.0 * B * T * (double)T * d, 16.0 * B * T * (double)d, attn_fwd_kernel<DKT>, grid, dim3(256), smem, st, qkv, T, d, heads, key_len, scale, out, lse, row_off);
  });
  INTEL_CHECK_LAUNCH();
  return 0;
}

int launch_attn_bwd(const float* qkv, const float* out, const float* dout, const float* lse, int B, int T, int d,
                    int heads, const int* key_len, float* dqkv, float* scratch, hipStream_t st, const int* row_off, int h16) {
  if (B <= 0) return 0;
  int rc = check_attn_shape(T, d, heads);
  if (rc) return rc;
  const int dk = d / heads, dkt = cdiv(dk, 16);
  const float scale = 1.0f / sqrtf((float)dk);
  float* dsum = scratch;
  if (attn_seq_path(T, dk))
    return launch_attn_seq_bwd(qkv, out, dout, lse, B, T, d, heads, key_len, dqkv, scratch + rup_sz((size_t)B * heads * T, 64), st, row_off, h16);
  INTEL_CHECK_ARG(!h16, "attention: bf16-stored q,k,v are supported by the whole-sequence kernels only");
  INTEL_CHECK_ARG(!row_off || key_len, "attention: packed rows need the session lengths");
  dim3 grid(B * heads * cdiv(T, AT_QB));
  // small problems (short lists, few rows: the 32-wide towers of the published hyper-parameters) are bound by the number of dependent launches, not by
  // their products: there the recompute form's two kernels beat the three of the dS scheme
  const bool small = T <= 64 && (long long)B * T <= 32768;
  if (attn_ds_scheme() && !small) {
    float* dS = scratch + rup_sz((size_t)B * heads * T, 64);
    const int ldS = attn_ds_pitch(T);
    const long long rows = (long long)B * T;
    LAUNCH_W(0.0, 8.0 * (double)rows * d, attn_dsum_kernel, dim3((unsigned)((rows * heads + 15) / 16)), dim3(256), 0, st, out, dout, T, d, heads, rows, dsum, key_len, row_off);
    INTEL_CHECK_LAUNCH();
    if (attn_p3_supported(T, dk)) return launch_attn_p3_bwd(qkv, dout, lse, dsum, B, T, d, heads, key_len, dqkv, dS, ldS, st, row_off);
    if (attn_bf16_products(T, dk)) {
      ATTN_DISPATCH_BF(dkt, {
        size_t smem = (size_t)(2 * AT_KB * AttnSmem<DKT>::LD + 2 * AT_KB) * sizeof(float);
        allow_lds(attn_bwd_dkv_bf_kernel<DKT>, smem);
        LAUNCH_S(B * heads, T, dk, 8.0 * B * T * (double)T * d, 24.0 * B * T * (double)d, attn_bwd_dkv_bf_kernel<DKT>, grid, dim3(256), smem, st, qkv, dout, lse, dsum, T, d, heads, key_len,
                           scale, dqkv, dS, ldS, row_off);
      });
      INTEL_CHECK_LAUNCH();
      ATTN_DISPATCH_BF(dkt, {
        size_t smem = (size_t)AT_KB * AttnSmem<DKT>::LD * sizeof(float);
        allow_lds(attn_bwd_dq_ds_bf_kernel<DKT>, smem);
        LAUNCH_S(B * heads, T, dk, 2.0 * B * T * (double)T * d, 8.0 * B * T * (double)d + 4.0 * B * heads * (double)T * T, attn_bwd_dq_ds_bf_kernel<DKT>, grid, dim3(256), smem, st, qkv, dS, ldS, T, d, heads, key_len, dqkv, row_off);
      });
      INTEL_CHECK_LAUNCH();
      return 0;
    }
    ATTN_DISPATCH(dkt, {
      size_t smem = (size_t)(2 * AT_KB * AttnSmem<DKT>::LD + 2 * AT_KB) * sizeof(float);
      allow_lds(attn_bwd_dkv_kernel<DKT>, smem);
      LAUNCH_S(B * heads, T, dk, 8.0 * B * T * (double)T * d, 24.0 * B * T * (double)d, attn_bwd_dkv_kernel<DKT>, grid, dim3(256), smem, st, qkv, dout, lse, dsum, T, d, heads, key_len,
                         scale, dqkv, dS, ldS, row_off);
    });
    INTEL_CHECK_LAUNCH();
    ATTN_DISPATCH(dkt, {
      size_t smem = (size_t)AT_KB * AttnSmem<DKT>::LD * sizeof(float);
      allow_lds(attn_bwd_dq_ds_kernel<DKT>, smem);
      LAUNCH_S(B * heads, T, dk, 2.0 * B * T * (double)T * d, 8.0 * B * T * (double)d + 4.0 * B * heads * (double)T * T, attn_bwd_dq_ds_kernel<DKT>, grid, dim3(256), smem, st, qkv, dS, ldS, T, d, heads, key_len, dqkv, row_off);
    });
    INTEL_CHECK_LAUNCH();
    return 0;
  }
    ATTN_DISPATCH(dkt, {
    size_t smem = (size_t)2 * AT_KB * AttnSmem<DKT>::LD * sizeof(float);
    allow_lds(attn_bwd_dq_kernel<DKT>, smem);
    LAUNCH_S(B * heads, T, dk, 6.0 * B * T * (double)T * d, 24.0 * B * T * (double)d, attn_bwd_dq_kernel<DKT>, grid, dim3(256), smem, st, qkv, out, dout, lse, T, d, heads, key_len,
                       scale, dqkv, dsum, row_off);
  });
  INTEL_CHECK_LAUNCH();
  ATTN_DISPATCH(dkt, {
    size_t smem = (size_t)(2 * AT_KB * AttnSmem<DKT>::LD + 2 * AT_KB) * sizeof(float);
    allow_lds(attn_bwd_dkv_kernel<DKT>, smem);
    LAUNCH_S(B * heads, T, dk, 8.0 * B * T * (double)T * d, 24.0 * B * T * (double)d, attn_bwd_dkv_kernel<DKT>, grid, dim3(256), smem, st, qkv, dout, lse, dsum, T, d, heads, key_len,
                       scale, dqkv, nullptr, 0, row_off);
  });
  INTEL_CHECK_LAUNCH();
  return 0;
}
