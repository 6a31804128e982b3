// Device helpers shared by the fused sequence-encoder kernels (enc.hip, enc_bwd.hip): bf16 plane splitting, the plane
// products on the bf16 matrix pipe, exact-fp32 tile products, 16-lane row reductions and LDS-only barriers.
// (tower.hip keeps its own copies: it predates this header.)
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace planes {

// x = hi + mid + lo with bf16 planes (24 significant bits; both subtractions are exact)
__device__ __forceinline__ void split4(const f32x4& x, bf16x4& h, bf16x4& m, bf16x4& l) {
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
__device__ __forceinline__ void split1(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}

// four consecutive values into the planes of one row (NP = 3: hi / mid / lo, PLANE elements apart; NP = 1: the bf16 rounding)
template <int NP, int PLANE>
__device__ __forceinline__ void store4(__bf16* dst, const f32x4& x) {
  if (NP == 1) {
    *reinterpret_cast<bf16x4*>(dst) = bf16x4{(__bf16)x[0], (__bf16)x[1], (__bf16)x[2], (__bf16)x[3]};
  } else {
    bf16x4 h, m, l;
    split4(x, h, m, l);
    *reinterpret_cast<bf16x4*>(dst) = h;
    *reinterpret_cast<bf16x4*>(dst + PLANE) = m;
    *reinterpret_cast<bf16x4*>(dst + 2 * PLANE) = l;
  }
}

// six plane products of weight >= 2^-16 (hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid), smallest first;
// NP = 1 (bf16 mode): the one product of the bf16-rounded operands
template <int NP>
__device__ __forceinline__ f32x4 mma(const bf16x8& wh, const bf16x8& wm, const bf16x8& wl, const bf16x8& ah, const bf16x8& am,
                                     const bf16x8& al, f32x4 c) {
  if (NP == 1) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, am, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, am, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, ah, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah, c, 0, 0, 0);
  return c;
}

// c += sum over the four k-steps s and the four lane groups j of a[s] * b[s]: four exact fp32 MFMAs (16x16x4); in a step the k
// index of lane group j is element 4j + s of a 16-deep group, for both operands
__device__ __forceinline__ f32x4 mma4(const f32x4& a, const f32x4& b, f32x4 c) {
#pragma unroll
  for (int s = 0; s < 4; ++s) c = mfma16(a[s], b[s], c);
  return c;
}
// the same 16-deep group as ONE bf16 MFMA (v_mfma_f32_16x16x16_bf16: lane (i, j) supplies k = 4j .. 4j+3, the same element order)
__device__ __forceinline__ f32x4 mma4_bf16(const s16x4& a, const s16x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ s16x4 to_bf16x4(const f32x4& v) {
  return __builtin_bit_cast(s16x4, bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]});
}

// over the 4 lane groups sharing lane & 15
__device__ __forceinline__ float gmax16(float v) {
  v = fmaxf(v, __shfl_xor(v, 16));
  return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float gsum16(float v) {
  v += __shfl_xor(v, 16);
  return v + __shfl_xor(v, 32);
}

// sum over the 16 lanes of a DPP row (lanes sharing lane >> 4): every lane of the row gets the sum
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);     // row_half_mirror
  v += dpp_mov<0x140>(v);     // row_mirror
  return v;
}

// Workgroup barrier over the LDS only (no wait for the wave's global accesses)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// keeps loads of loop-invariant weight images where they are written (streamed from L2) instead of hoisted into registers
// (the asm makes the pointer's address space unknown to the compiler -- its loads would become FLAT loads, which count on lgkmcnt as well and drag every
// LDS read into their waits; laundering the address as an INTEGER and casting it to a global-address-space pointer keeps global_load)
typedef const uint4 __attribute__((address_space(1)))* global_uint4_ptr;
__device__ __forceinline__ const uint4* launder(const uint4* p) {
  unsigned long long u = (unsigned long long)p;
  asm volatile("" : "+s"(u));
  return (const uint4*)(global_uint4_ptr)u;
}

// ---- transposing LDS reads (shared by tower_bwd.hip and pair.hip) ----
typedef short tb_s16x4 __attribute__((ext_vector_type(4)));
typedef short tb_s16x8 __attribute__((ext_vector_type(8)));

// The operand of a product whose reduction runs over the ROWS of a row-major plane image: lane (p, j) gets column ct*16 + p at the eight rows
// 32 kb + 8 j + {0, 2, 4, 6, 1, 3, 5, 7} from two transposing reads (ds_read_b64_tr_b16: lane 4q + pp of a 16-lane group names row q, columns
// 4pp .. 4pp+3 of the group's 4 x 16 block and receives column p of its four rows).  Both operands of a product use the same row order.  The 32 lanes
// one LDS cycle services touch rows {0, 2, .. 14} + half of a 32-row block: with a pitch of D + 8 bf16 (68 / 36 dwords) they cover all 64 banks once.
// P8 = false: rows 32 kb + 8 j + {0, 2, 4, 6, 1, 3, 5, 7} (conflict-free at a pitch of D + 8 bf16: 68 / 36 dwords); P8 = true: rows 32 kb + 4 j + {0 .. 3}
// and 32 kb + 16 + 4 j + {0 .. 3} -- the order in which an S-type accumulator pair (two 16-row tiles) holds its rows, so that probabilities / dS values go
// from the accumulators straight into the other operand (conflict-free at a pitch of D + 16 bf16: 72 / 40 dwords: eight consecutive rows cover all banks)
template <int LD, bool P8 = false>
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* plane, int kb, int ct, int p, int j) {
  const int q = p >> 2, pp = p & 3;
  const __bf16* a0 = plane + (P8 ? (32 * kb + 4 * j + q) : (32 * kb + 8 * j + 2 * q)) * LD + ct * 16 + 4 * pp;
  const tb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tb_s16x4*)(a0));
  const tb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tb_s16x4*)(a0 + (P8 ? 16 : 1) * LD));
  return __builtin_bit_cast(bf16x8, tb_s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}

__device__ __forceinline__ float sum8(const bf16x8& v) {
  return (((float)v[0] + (float)v[1]) + ((float)v[2] + (float)v[3])) + (((float)v[4] + (float)v[5]) + ((float)v[6] + (float)v[7]));
}

}  // namespace planes
