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

#define AT_QB 64   // rows (queries, or keys in the dK/dV kernel) owned by a workgroup: 4 waves x 16
#define AT_KB 32   // rows staged in LDS per iteration: 2 x 32 x (dk+4) floats = 33 KB at dk = 128 -> 4 workgroups / CU

template <int DKT>
struct AttnSmem {
  static constexpr int LD = DKT * 16 + 4;   // ld % 8 == 4: b128 row reads and b32 column reads conflict-free
};

// stage rows [r0, r0+64) of one third of qkv (column offset coff) into LDS, zero padded
template <int DKT>
__device__ __forceinline__ void stage_rows(float* dst, const float* __restrict__ base, int ldg, int coff, int dk,
                                           int r0, int T, int tid) {
  constexpr int LD = AttnSmem<DKT>::LD;
  constexpr int C4 = DKT * 4;   // float4 per row
  for (int i = tid; i < AT_KB * C4; i += 256) {
    const int r = i / C4, c4 = i - r * C4;
    const int row = r0 + r, col = c4 * 4;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (row < T && col < dk) v = *reinterpret_cast<const f32x4*>(base + (size_t)row * ldg + coff + col);
    *reinterpret_cast<f32x4*>(dst + r * LD + col) = v;
  }
}

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

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int DKT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, int T, int d, int heads,
                                                       const int* __restrict__ key_len, float scale,
                                                       float* __restrict__ out, float* __restrict__ lse) {
  constexpr int LD = AttnSmem<DKT>::LD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;
  float* Vs = smem + AT_KB * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
  const int dk = d / heads, ldg = 3 * d;
  const int nkeys = key_len ? min(key_len[b], T) : T;
  const float* base = qkv + (size_t)b * T * ldg;
  const int q = blockIdx.y * AT_QB + wave * 16 + (lane & 15);
  f32x4 qf[DKT];
  load_row_frags<DKT>(qf, base + (size_t)q * ldg + h * dk, q < T, dk, lane);
  f32x4 oT[DKT];
#pragma unroll
  for (int i = 0; i < DKT; ++i) oT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  for (int kb = 0; kb < nkeys; kb += AT_KB) {
    __syncthreads();
    stage_rows<DKT>(Ks, base, ldg, d + h * dk, dk, kb, T, tid);
    stage_rows<DKT>(Vs, base, ldg, 2 * d + h * dk, dk, kb, T, tid);
    __syncthreads();
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
          const float* vrow = Vs + (kt * 16 + 4 * (lane >> 4) + s) * LD + (lane & 15);
#pragma unroll
          for (int dt = 0; dt < DKT; ++dt) oT[dt] = mfma16(vrow[dt * 16], st[kt][s], oT[dt]);
        }
      }
    }
  }
  if (q < T) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    float* orow = out + ((size_t)b * T + q) * d + h * dk;
#pragma unroll
    for (int dt = 0; dt < DKT; ++dt) {
      const int col = dt * 16 + 4 * (lane >> 4);
      if (col < dk) *reinterpret_cast<f32x4*>(orow + col) = oT[dt] * inv;
    }
    if (lane < 16) lse[((size_t)b * heads + h) * T + q] = l_run > 0.f ? m_run + logf(l_run) : INFINITY;
  }
}

// ------------------------------------------------------------------------------------------
// backward, dQ: wave owns 16 queries, sweeps key blocks.  Also writes dsum[q] = sum_d dO*O.
// ------------------------------------------------------------------------------------------
template <int DKT>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                          const float* __restrict__ dout, const float* __restrict__ lse,
                                                          int T, int d, int heads, const int* __restrict__ key_len,
                                                          float scale, float* __restrict__ dqkv, float* __restrict__ dsum) {
  constexpr int LD = AttnSmem<DKT>::LD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;
  float* Vs = smem + AT_KB * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
  const int dk = d / heads, ldg = 3 * d;
  const int nkeys = key_len ? min(key_len[b], T) : T;
  const float* base = qkv + (size_t)b * T * ldg;
  const int q = blockIdx.y * AT_QB + wave * 16 + (lane & 15);
  const bool qok = q < T;
  f32x4 qf[DKT], dof[DKT];
  load_row_frags<DKT>(qf, base + (size_t)q * ldg + h * dk, qok, dk, lane);
  load_row_frags<DKT>(dof, dout + ((size_t)b * T + q) * d + h * dk, qok, dk, lane);
  float dsm = 0.f;
  {
    f32x4 of[DKT];
    load_row_frags<DKT>(of, out + ((size_t)b * T + q) * d + h * dk, qok, dk, lane);
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

  for (int kb = 0; kb < nkeys; kb += AT_KB) {
    __syncthreads();
    stage_rows<DKT>(Ks, base, ldg, d + h * dk, dk, kb, T, tid);
    stage_rows<DKT>(Vs, base, ldg, 2 * d + h * dk, dk, kb, T, tid);
    __syncthreads();
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
        const float* krow = Ks + (kt * 16 + 4 * (lane >> 4) + s) * LD + (lane & 15);
#pragma unroll
        for (int dt = 0; dt < DKT; ++dt) dqT[dt] = mfma16(krow[dt * 16], dsT[s], dqT[dt]);
      }
    }
  }
  if (qok) {
    float* drow = dqkv + ((size_t)b * T + q) * ldg + h * dk;
#pragma unroll
    for (int dt = 0; dt < DKT; ++dt) {
      const int col = dt * 16 + 4 * (lane >> 4);
      if (col < dk) *reinterpret_cast<f32x4*>(drow + col) = dqT[dt];
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward, dK/dV: wave owns 16 keys, sweeps query blocks (Q and dO staged in LDS).
// ------------------------------------------------------------------------------------------
template <int DKT>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                           const float* __restrict__ lse, const float* __restrict__ dsum,
                                                           int T, int d, int heads, const int* __restrict__ key_len,
                                                           float scale, float* __restrict__ dqkv) {
  constexpr int LD = AttnSmem<DKT>::LD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;
  float* Os = smem + AT_KB * LD;
  float* Ls = smem + 2 * AT_KB * LD;   // [64] lse
  float* Ds = Ls + AT_KB;              // [64] dsum
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
  const int dk = d / heads, ldg = 3 * d;
  const int nkeys = key_len ? min(key_len[b], T) : T;
  const float* base = qkv + (size_t)b * T * ldg;
  const int key = blockIdx.y * AT_QB + wave * 16 + (lane & 15);
  const bool kok = key < T;
  f32x4 kf[DKT], vf[DKT];
  load_row_frags<DKT>(kf, base + (size_t)key * ldg + d + h * dk, kok, dk, lane);
  load_row_frags<DKT>(vf, base + (size_t)key * ldg + 2 * d + h * dk, kok, dk, lane);
  f32x4 dkT[DKT], dvT[DKT];
#pragma unroll
  for (int i = 0; i < DKT; ++i) {
    dkT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    dvT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool key_live = key < nkeys;          // masked keys get exactly zero gradient
  const bool wave_live = (blockIdx.y * AT_QB + wave * 16) < nkeys;

  for (int qb = 0; qb < T; qb += AT_KB) {
    __syncthreads();
    stage_rows<DKT>(Qs, base, ldg, h * dk, dk, qb, T, tid);
    stage_rows<DKT>(Os, dout + (size_t)b * T * d, d, h * dk, dk, qb, T, tid);
    if (tid < AT_KB) {
      const int qq = qb + tid;
      Ls[tid] = qq < T ? lse[((size_t)b * heads + h) * T + qq] : INFINITY;
      Ds[tid] = qq < T ? dsum[((size_t)b * heads + h) * T + qq] : 0.f;
    }
    __syncthreads();
    if (!wave_live) continue;
#pragma unroll
    for (int qt = 0; qt < AT_KB / 16; ++qt) {
      if (qb + qt * 16 >= T) continue;
      f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < DKT; ++g) {
        const int off = (qt * 16 + (lane & 15)) * LD + g * 16 + 4 * (lane >> 4);
        const f32x4 qf = *reinterpret_cast<const f32x4*>(Qs + off);
        const f32x4 of = *reinterpret_cast<const f32x4*>(Os + off);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          sa = mfma16(qf[s], kf[g][s], sa);     // S[query][key]
          dp = mfma16(of[s], vf[g][s], dp);     // dP[query][key]
        }
      }
      f32x4 pr, ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = qt * 16 + 4 * (lane >> 4) + r;
        const float p = key_live ? expf(sa[r] * scale - Ls[ql]) : 0.f;
        pr[r] = p;
        ds[r] = p * (dp[r] - Ds[ql]) * scale;
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int roff = (qt * 16 + 4 * (lane >> 4) + s) * LD + (lane & 15);
#pragma unroll
        for (int dt = 0; dt < DKT; ++dt) {
          dvT[dt] = mfma16(Os[roff + dt * 16], pr[s], dvT[dt]);   // dV^T[dim][key] += dO^T P
          dkT[dt] = mfma16(Qs[roff + dt * 16], ds[s], dkT[dt]);   // dK^T[dim][key] += Q^T dS
        }
      }
    }
  }
  if (kok) {
    float* drow = dqkv + ((size_t)b * T + key) * ldg + h * dk;
#pragma unroll
    for (int dt = 0; dt < DKT; ++dt) {
      const int col = dt * 16 + 4 * (lane >> 4);
      if (col < dk) {
        *reinterpret_cast<f32x4*>(drow + d + col) = dkT[dt];
        *reinterpret_cast<f32x4*>(drow + 2 * d + col) = dvT[dt];
      }
    }
  }
}

// ==========================================================================================
// Whole-sequence kernels: T <= 64 and head dim 64 / 128 (the towers at list length 50, the BERT4Rec
// blocks at history length 20).  Everything a (session, head) pair needs fits in LDS at once, so
//   * a wave = one 16-row tile of one pair; 4 / ceil(T/16) pairs share a workgroup (no idle waves at T = 20);
//   * the operand rows are staged once, one barrier, no key-block loop and no running softmax;
//   * rows are staged PERMUTED inside each 16-row tile (row 4a+b -> slot 4b+a): accumulator row 4j+r then
//     is key (query) 4r+j, so k-step s of the following product covers rows 4s..4s+3 and the steps that only
//     hold padding (rows >= T, 14 of 16 in the last tile at T = 50) are skipped;
//   * "transposed" operands (V^T, dO^T, Q^T, K^T) are read as one b128 along the head dim: lane p takes dims
//     4p..4p+3 of its row and feeds FOUR MFMAs whose output row p means dim 4p+t -- the output rows of the four
//     tiles interleave, and the epilogue stores float4s;
//   * backward = dK/dV kernel that also writes dS[q][key] (5 tile products, not 7: S and dP are computed
//     once) + a dQ = dS K kernel; delta = rowsum(dO * O) is folded into the staging of dO.
// ==========================================================================================
template <int DKT, int NT>
struct SeqCfg {
  static constexpr int LD = DKT * 16 + 4;
  static constexpr int DK = DKT * 16;
  static constexpr int PW = 4 / NT;          // pairs per workgroup
  static constexpr int TP = NT * 16;         // padded rows per pair
  static constexpr int ROWS = PW * TP;
  static constexpr int C4 = DKT * 4;         // float4 per row
  static constexpr int ITERS = ROWS * C4 / 256;
};

__device__ __forceinline__ int perm16(int r) { return (r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3); }

// address of float4 #i of the staged block: LDS row rl (pair slot, row in pair), source row, validity
template <int DKT, int NT, bool PERM>
__device__ __forceinline__ bool seq_src(int i, int bh0, int BH, int T, int heads, int& rl, int& c4, size_t& grow, int& hcol) {
  using C = SeqCfg<DKT, NT>;
  rl = i / C::C4;
  c4 = i - rl * C::C4;
  const int sl = rl / C::TP, r = rl - sl * C::TP;
  const int rho = PERM ? perm16(r) : r;
  const int bh = bh0 + sl;
  int b = bh, h = 0;
  if (heads > 1) { b = bh / heads; h = bh - b * heads; }
  grow = (size_t)b * T + rho;
  hcol = h * C::DK + c4 * 4;
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

template <int DKT, int NT>
__global__ __launch_bounds__(256) void attn_fwd_seq_kernel(const float* __restrict__ qkv, int BH, int T, int d, int heads,
                                                           const int* __restrict__ key_len, float scale,
                                                           float* __restrict__ out, float* __restrict__ lse) {
  using C = SeqCfg<DKT, NT>;
  constexpr int LD = C::LD, DK = C::DK, DQ = DKT / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;
  float* Vs = smem + C::ROWS * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane >> 4, p = lane & 15;
  const int bh0 = blockIdx.x * C::PW, ldg = 3 * d;
  const int slot = wave / NT, tile = wave - slot * NT, bh = bh0 + slot;
  const bool live = slot < C::PW && bh < BH;
  int b = live ? bh : 0, h = 0;
  if (heads > 1) { b = (live ? bh : 0) / heads; h = (live ? bh : 0) - b * heads; }
  const int q = tile * 16 + p;
  const bool qok = live && q < T;
  f32x4 qf[DKT];
  load_row_frags<DKT>(qf, qkv + ((size_t)b * T + q) * ldg + h * DK, qok, DK, lane);
  stage_seq2<DKT, NT, true, true>(Ks, qkv, ldg, d, Vs, qkv, ldg, 2 * d, bh0, BH, T, heads, tid);
  __syncthreads();
  if (!live) return;
  const int nkeys = key_len ? min(key_len[b], T) : T;
  const float* Kp = Ks + slot * C::TP * LD;
  const float* Vp = Vs + slot * C::TP * LD;
  f32x4 st[NT];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < DKT; ++g) {
    f32x4 kf[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) kf[kt] = *reinterpret_cast<const f32x4*>(Kp + (kt * 16 + p) * LD + g * 16 + 4 * j);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) st[kt] = mfma16(kf[kt][s], qf[g][s], st[kt]);
  }
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kt * 16 + 4 * r + j;                  // accumulator row 4j+r holds key 4r+j of the tile
      const float v = key < nkeys ? st[kt][r] * scale : -INFINITY;
      st[kt][r] = v;
      mx = fmaxf(mx, v);
    }
  mx = group_max16(mx);
  const float mref = mx == -INFINITY ? 0.f : mx;
  float ps = 0.f;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = expf(st[kt][r] - mref);
      st[kt][r] = e;
      ps += e;
    }
  ps = group_sum16(ps);
  f32x4 oT[DKT];
#pragma unroll
  for (int i = 0; i < DKT; ++i) oT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (kt * 16 + 4 * s >= nkeys) continue;               // keys 4s..4s+3 of the tile are all masked / padding
#pragma unroll
      for (int dq = 0; dq < DQ; ++dq) {
        const f32x4 vv = *reinterpret_cast<const f32x4*>(Vp + (kt * 16 + 4 * j + s) * LD + dq * 64 + 4 * p);
#pragma unroll
        for (int t = 0; t < 4; ++t) oT[dq * 4 + t] = mfma16(vv[t], st[kt][s], oT[dq * 4 + t]);
      }
    }
  if (qok) {
    const float inv = ps > 0.f ? 1.f / ps : 0.f;
    float* orow = out + ((size_t)b * T + q) * d + h * DK;
#pragma unroll
    for (int dq = 0; dq < DQ; ++dq)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4 o = f32x4{oT[dq * 4 + 0][r], oT[dq * 4 + 1][r], oT[dq * 4 + 2][r], oT[dq * 4 + 3][r]} * inv;
        *reinterpret_cast<f32x4*>(orow + dq * 64 + 16 * j + 4 * r) = o;
      }
    if (lane < 16) lse[(size_t)bh * T + q] = ps > 0.f ? mx + logf(ps) : INFINITY;
  }
}

// dK, dV of a 16-key tile + the dS tile column for the dQ kernel.  dS layout: [BH][TP][TP], row = query.
template <int DKT, int NT>
__global__ __launch_bounds__(256, 2) void attn_bwd_kv_seq_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                              const float* __restrict__ dout, const float* __restrict__ lse,
                                                              int BH, int T, int d, int heads, const int* __restrict__ key_len,
                                                              float scale, float* __restrict__ dqkv, float* __restrict__ dS) {
  using C = SeqCfg<DKT, NT>;
  constexpr int LD = C::LD, DK = C::DK, DQ = DKT / 4, TP = C::TP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;
  float* Os = smem + C::ROWS * LD;
  float* Ls = smem + 2 * C::ROWS * LD;
  float* Ds = Ls + C::ROWS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane >> 4, p = lane & 15;
  const int bh0 = blockIdx.x * C::PW, ldg = 3 * d;
  const int slot = wave / NT, tile = wave - slot * NT, bh = bh0 + slot;
  const bool live = slot < C::PW && bh < BH;
  int b = live ? bh : 0, h = 0;
  if (heads > 1) { b = (live ? bh : 0) / heads; h = (live ? bh : 0) - b * heads; }
  const int key = tile * 16 + p;
  const bool kok = live && key < T;
  f32x4 kf[DKT], vf[DKT];
  load_row_frags<DKT>(kf, qkv + ((size_t)b * T + key) * ldg + d + h * DK, kok, DK, lane);
  load_row_frags<DKT>(vf, qkv + ((size_t)b * T + key) * ldg + 2 * d + h * DK, kok, DK, lane);
  {
    // stage Q and dO (permuted rows); delta[row] = sum_d dO * O rides along: the C4 lanes of a row are adjacent
    f32x4 vq[C::ITERS], vo[C::ITERS];
    float dot[C::ITERS];
#pragma unroll
    for (int it = 0; it < C::ITERS; ++it) {
      int rl, c4, hcol;
      size_t grow;
      const bool ok = seq_src<DKT, NT, true>(tid + it * 256, bh0, BH, T, heads, rl, c4, grow, hcol);
      const size_t gr = ok ? grow : 0;
      const int hc = ok ? hcol : 0;
      const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 tq = *reinterpret_cast<const f32x4*>(qkv + gr * ldg + hc);
      const f32x4 to = *reinterpret_cast<const f32x4*>(dout + gr * d + hc);
      const f32x4 w = *reinterpret_cast<const f32x4*>(out + gr * d + hc);
      vq[it] = ok ? tq : zero;
      vo[it] = ok ? to : zero;
      dot[it] = ok ? (to[0] * w[0] + to[1] * w[1] + to[2] * w[2] + to[3] * w[3]) : 0.f;
    }
#pragma unroll
    for (int it = 0; it < C::ITERS; ++it) {
      const int i = tid + it * 256;
      const int rl = i / C::C4, c4 = i - rl * C::C4;
      *reinterpret_cast<f32x4*>(Qs + rl * LD + c4 * 4) = vq[it];
      *reinterpret_cast<f32x4*>(Os + rl * LD + c4 * 4) = vo[it];
      float s = dot[it];
#pragma unroll
      for (int m = 1; m < C::C4; m <<= 1) s += __shfl_xor(s, m);
      if (c4 == 0) Ds[rl] = s;
    }
    if (tid < C::ROWS) {
      const int sl = tid / TP, r = tid - sl * TP, rho = perm16(r), bb = bh0 + sl;
      Ls[tid] = (bb < BH && rho < T) ? lse[(size_t)bb * T + rho] : INFINITY;
    }
  }
  __syncthreads();
  if (!live) return;
  const int nkeys = key_len ? min(key_len[b], T) : T;
  const bool key_live = key < nkeys;             // masked keys get exactly zero gradient
  const float* Qp = Qs + slot * TP * LD;
  const float* Op = Os + slot * TP * LD;
  const float* Lp = Ls + slot * TP;
  const float* Dp = Ds + slot * TP;
  float* dSp = dS + (size_t)bh * TP * TP;
  f32x4 dkT[DKT], dvT[DKT];
#pragma unroll
  for (int i = 0; i < DKT; ++i) {
    dkT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    dvT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (tile * 16 < nkeys) {
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
      f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < DKT; ++g) {
        const int off = (qt * 16 + p) * LD + g * 16 + 4 * j;
        const f32x4 qa = *reinterpret_cast<const f32x4*>(Qp + off);
        const f32x4 oa = *reinterpret_cast<const f32x4*>(Op + off);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          sa = mfma16(qa[s], kf[g][s], sa);     // S[query slot][key]
          dp = mfma16(oa[s], vf[g][s], dp);     // dP[query slot][key]
        }
      }
      f32x4 pr, ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = qt * 16 + 4 * j + r;     // staged slot of accumulator row 4j+r; its query is 4r+j
        const int qg = qt * 16 + 4 * r + j;
        const float pv = key_live ? expf(sa[r] * scale - Lp[rl]) : 0.f;
        pr[r] = pv;
        ds[r] = pv * (dp[r] - Dp[rl]) * scale;
        if (qg < T && key < T) dSp[(size_t)qg * TP + key] = ds[r];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (qt * 16 + 4 * s >= T) continue;     // queries 4s..4s+3 of the tile are padding
#pragma unroll
        for (int dq = 0; dq < DQ; ++dq) {
          const int off = (qt * 16 + 4 * j + s) * LD + dq * 64 + 4 * p;
          const f32x4 ov = *reinterpret_cast<const f32x4*>(Op + off);
          const f32x4 qv = *reinterpret_cast<const f32x4*>(Qp + off);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            dvT[dq * 4 + t] = mfma16(ov[t], pr[s], dvT[dq * 4 + t]);   // dV^T[dim][key] += dO^T P
            dkT[dq * 4 + t] = mfma16(qv[t], ds[s], dkT[dq * 4 + t]);   // dK^T[dim][key] += Q^T dS
          }
        }
      }
    }
  }
  if (kok) {
    float* drow = dqkv + ((size_t)b * T + key) * ldg + h * DK;
#pragma unroll
    for (int dq = 0; dq < DQ; ++dq)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = dq * 64 + 16 * j + 4 * r;
        *reinterpret_cast<f32x4*>(drow + d + col) = f32x4{dkT[dq * 4 + 0][r], dkT[dq * 4 + 1][r], dkT[dq * 4 + 2][r], dkT[dq * 4 + 3][r]};
        *reinterpret_cast<f32x4*>(drow + 2 * d + col) = f32x4{dvT[dq * 4 + 0][r], dvT[dq * 4 + 1][r], dvT[dq * 4 + 2][r], dvT[dq * 4 + 3][r]};
      }
  }
}

// dQ[q][dim] = sum_key dS[q][key] K[key][dim]  (dS already carries 1/sqrt(dk))
template <int DKT, int NT>
__global__ __launch_bounds__(256, 3) void attn_bwd_q_seq_kernel(const float* __restrict__ qkv, const float* __restrict__ dS,
                                                             int BH, int T, int d, int heads, const int* __restrict__ key_len,
                                                             float* __restrict__ dqkv) {
  using C = SeqCfg<DKT, NT>;
  constexpr int LD = C::LD, DK = C::DK, DQ = DKT / 4, TP = C::TP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane >> 4, p = lane & 15;
  const int bh0 = blockIdx.x * C::PW, ldg = 3 * d;
  const int slot = wave / NT, tile = wave - slot * NT, bh = bh0 + slot;
  const bool live = slot < C::PW && bh < BH;
  int b = live ? bh : 0, h = 0;
  if (heads > 1) { b = (live ? bh : 0) / heads; h = (live ? bh : 0) - b * heads; }
  const int q = tile * 16 + p;
  const bool qok = live && q < T;
  const int nkeys = live ? (key_len ? min(key_len[b], T) : T) : 0;
  f32x4 dsT[NT];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
    dsT[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (qok && kt * 16 < nkeys) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(dS + ((size_t)bh * TP + q) * TP + kt * 16 + 4 * j);
#pragma unroll
      for (int s = 0; s < 4; ++s) dsT[kt][s] = (kt * 16 + 4 * j + s < nkeys) ? v[s] : 0.f;
    }
  }
  stage_seq2<DKT, NT, false, false>(Ks, qkv, ldg, d, nullptr, nullptr, 0, 0, bh0, BH, T, heads, tid);
  __syncthreads();
  if (!live) return;
  const float* Kp = Ks + slot * TP * LD;
  f32x4 dqT[DKT];
#pragma unroll
  for (int i = 0; i < DKT; ++i) dqT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
    if (kt * 16 >= nkeys) continue;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int dq = 0; dq < DQ; ++dq) {
        const f32x4 kv = *reinterpret_cast<const f32x4*>(Kp + (kt * 16 + 4 * j + s) * LD + dq * 64 + 4 * p);
#pragma unroll
        for (int t = 0; t < 4; ++t) dqT[dq * 4 + t] = mfma16(kv[t], dsT[kt][s], dqT[dq * 4 + t]);
      }
    __builtin_amdgcn_sched_barrier(0);       // keep the LDS reads of later tiles from being hoisted (register pressure)
  }
  if (qok) {
    float* drow = dqkv + ((size_t)b * T + q) * ldg + h * DK;
#pragma unroll
    for (int dq = 0; dq < DQ; ++dq)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<f32x4*>(drow + dq * 64 + 16 * j + 4 * r) =
            f32x4{dqT[dq * 4 + 0][r], dqT[dq * 4 + 1][r], dqT[dq * 4 + 2][r], dqT[dq * 4 + 3][r]};
  }
}

static inline bool attn_seq_path(int T, int dk) {
  static const int off = [] { const char* e = getenv("INTEL_ATTN_SEQ"); return (e && e[0] == '0') ? 1 : 0; }();
  return !off && T <= 64 && (dk == 64 || dk == 128);
}

#define ATTN_SEQ_DISPATCH(DKT_RT, NT_RT, ...)                                             \
  do {                                                                                    \
    if ((DKT_RT) == 4) {                                                                  \
      constexpr int DKT = 4;                                                              \
      switch (NT_RT) {                                                                    \
        case 1: { constexpr int NT = 1; __VA_ARGS__; } break;                                    \
        case 2: { constexpr int NT = 2; __VA_ARGS__; } break;                                    \
        case 3: { constexpr int NT = 3; __VA_ARGS__; } break;                                    \
        default: { constexpr int NT = 4; __VA_ARGS__; } break;                                   \
      }                                                                                   \
    } else {                                                                              \
      constexpr int DKT = 8;                                                              \
      switch (NT_RT) {                                                                    \
        case 1: { constexpr int NT = 1; __VA_ARGS__; } break;                                    \
        case 2: { constexpr int NT = 2; __VA_ARGS__; } break;                                    \
        case 3: { constexpr int NT = 3; __VA_ARGS__; } break;                                    \
        default: { constexpr int NT = 4; __VA_ARGS__; } break;                                   \
      }                                                                                   \
    }                                                                                     \
  } while (0)

size_t attn_bwd_scratch_floats(int B, int T, int d, int heads) {
  size_t f = rup_sz((size_t)B * heads * T, 64);
  if (heads > 0 && attn_seq_path(T, d / heads)) {
    const size_t tp = (size_t)cdiv(T, 16) * 16;
    f += (size_t)B * heads * tp * tp;
  }
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
                    hipStream_t st) {
  if (B <= 0) return 0;
  int rc = check_attn_shape(T, d, heads);
  if (rc) return rc;
  const int dk = d / heads, dkt = cdiv(dk, 16);
  const float scale = 1.0f / sqrtf((float)dk);
  if (attn_seq_path(T, dk)) {
    const int BH = B * heads;
    ATTN_SEQ_DISPATCH(dkt, cdiv(T, 16), {
      using C = SeqCfg<DKT, NT>;
      size_t smem = (size_t)2 * C::ROWS * C::LD * sizeof(float);
      allow_lds((attn_fwd_seq_kernel<DKT, NT>), smem);
      LAUNCH_S(BH, T, dk, 4.0 * B * T * (double)T * d, 16.0 * B * T * (double)d, (attn_fwd_seq_kernel<DKT, NT>), dim3(cdiv(BH, C::PW)), dim3(256), smem, st, qkv, BH, T, d, heads, key_len, scale, out, lse);
    });
    INTEL_CHECK_LAUNCH();
    return 0;
  }
  dim3 grid(B * heads, cdiv(T, AT_QB));
  ATTN_DISPATCH(dkt, {
    size_t smem = (size_t)2 * AT_KB * AttnSmem<DKT>::LD * sizeof(float);
    allow_lds(attn_fwd_kernel<DKT>, smem);
    LAUNCH_S(B * heads, T, dk, 4.0 * B * T * (double)T * d, 16.0 * B * T * (double)d, attn_fwd_kernel<DKT>, grid, dim3(256), smem, st, qkv, T, d, heads, key_len, scale, out, lse);
  });
  INTEL_CHECK_LAUNCH();
  return 0;
}

int launch_attn_bwd(const float* qkv, const float* out, const float* dout, const float* lse, int B, int T, int d,
                    int heads, const int* key_len, float* dqkv, float* scratch, hipStream_t st) {
  if (B <= 0) return 0;
  int rc = check_attn_shape(T, d, heads);
  if (rc) return rc;
  const int dk = d / heads, dkt = cdiv(dk, 16);
  const float scale = 1.0f / sqrtf((float)dk);
  float* dsum = scratch;
  if (attn_seq_path(T, dk)) {
    const int BH = B * heads;
    float* dS = scratch + rup_sz((size_t)BH * T, 64);
    ATTN_SEQ_DISPATCH(dkt, cdiv(T, 16), {
      using C = SeqCfg<DKT, NT>;
      size_t smem = (size_t)(2 * C::ROWS * C::LD + 2 * C::ROWS) * sizeof(float);
      allow_lds((attn_bwd_kv_seq_kernel<DKT, NT>), smem);
      LAUNCH_S(BH, T, dk, 8.0 * B * T * (double)T * d, 28.0 * B * T * (double)d, (attn_bwd_kv_seq_kernel<DKT, NT>), dim3(cdiv(BH, C::PW)), dim3(256), smem, st, qkv, out, dout, lse, BH, T, d, heads, key_len, scale, dqkv, dS);
    });
    INTEL_CHECK_LAUNCH();
    ATTN_SEQ_DISPATCH(dkt, cdiv(T, 16), {
      using C = SeqCfg<DKT, NT>;
      size_t smem = (size_t)C::ROWS * C::LD * sizeof(float);
      allow_lds((attn_bwd_q_seq_kernel<DKT, NT>), smem);
      LAUNCH_S(BH, T, dk, 2.0 * B * T * (double)T * d, 8.0 * B * T * (double)d, (attn_bwd_q_seq_kernel<DKT, NT>), dim3(cdiv(BH, C::PW)), dim3(256), smem, st, qkv, dS, BH, T, d, heads, key_len, dqkv);
    });
    INTEL_CHECK_LAUNCH();
    return 0;
  }
  dim3 grid(B * heads, cdiv(T, AT_QB));
  ATTN_DISPATCH(dkt, {
    size_t smem = (size_t)2 * AT_KB * AttnSmem<DKT>::LD * sizeof(float);
    allow_lds(attn_bwd_dq_kernel<DKT>, smem);
    LAUNCH_S(B * heads, T, dk, 6.0 * B * T * (double)T * d, 24.0 * B * T * (double)d, attn_bwd_dq_kernel<DKT>, grid, dim3(256), smem, st, qkv, out, dout, lse, T, d, heads, key_len,
                       scale, dqkv, dsum);
  });
  INTEL_CHECK_LAUNCH();
  ATTN_DISPATCH(dkt, {
    size_t smem = (size_t)(2 * AT_KB * AttnSmem<DKT>::LD + 2 * AT_KB) * sizeof(float);
    allow_lds(attn_bwd_dkv_kernel<DKT>, smem);
    LAUNCH_S(B * heads, T, dk, 8.0 * B * T * (double)T * d, 24.0 * B * T * (double)d, attn_bwd_dkv_kernel<DKT>, grid, dim3(256), smem, st, qkv, dout, lse, dsum, T, d, heads, key_len,
                       scale, dqkv);
  });
  INTEL_CHECK_LAUNCH();
  return 0;
}
