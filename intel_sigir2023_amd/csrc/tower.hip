// One tied tower layer of IntEL.predict_ensemble (models/IntEL/IntEL.py:182-188 / 191-197) as ONE kernel per layer:
//
//     res = h;  h = MHA(h, h, h)  (no mask, no output projection: modules/layers.py:31-60)
//     h = W1 h + b1;  h = W2 relu(h) + b2;  h = LayerNorm(h + res)
//
// A workgroup owns one session at a time (persistent over sessions): the session's [L <= 64, D] tile is read from HBM
// once and everything between the read and the LayerNorm output stays on chip --
//
//   phase 0  X tile global -> registers (prefetched during the previous session) -> three bf16 planes in LDS
//   phase 1  [Q | K | V] = X Wqkv^T on the bf16 matrix pipe at fp32 accuracy (hi/mid/lo planes, six plane products, as
//            gemm_rows_b3_kernel); weight fragments stream from the pre-split bf16 image in L2 (pack_b3), one k-block
//            ahead of the MFMAs; the result goes to LDS as fp32 rows
//   phase 2  softmax(Q K^T / sqrt(dk)) V per head in exact fp32 MFMA (v_mfma_f32_16x16x4_f32): wave = one 16-query tile
//            of one head, S^T in accumulators, base-2 softmax, P^T fed to O^T = V^T P^T straight from registers;
//            the attention output is split into the planes that phase 3 reads (the X planes are dead by then)
//   phase 3  R1 = relu(A W1^T + b1) -> bf16 planes (over the dead K / V rows)
//   phase 4  Z = R1 W2^T + b2 -> fp32 tile (over the dead Q rows)
//   phase 5  LayerNorm(Z + X) row-wise (the residual rows come back from L2), output / x-hat / rstd to HBM
//
// HBM traffic per session and layer: the X tile in, the output tile out (inference); training adds the stash the
// backward reads (QKV, A, LSE, R1, x-hat / rstd).  The kernel-per-op pipeline it replaces moved 14 tiles.
// Padded rows (row >= L) are zero rows: their K / V rows are zero (the tower linears have no bias, IntEL.py:60,68) and
// their keys are masked out of the softmax.
#include <stdio.h>
#include <stdlib.h>

#include "kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

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

// six plane products of weight >= 2^-16 (hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid), smallest first
__device__ __forceinline__ f32x4 mma6(const bf16x8& wh, const bf16x8& wm, const bf16x8& wl, const bf16x8& ah, const bf16x8& am,
                                      const bf16x8& al, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, am, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, am, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, ah, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah, c, 0, 0, 0);
  return c;
}

// The weight images are the same for every session, so the compiler would hoist their fragment loads out of the session
// loop and keep 240 registers of weights live (spilling everything else).  Laundering the base pointer once per session
// keeps the loads where they are written: streamed from L2, one k-block ahead of their MFMAs.
__device__ __forceinline__ const uint4* per_session(const uint4* p) {
  unsigned long long u = (unsigned long long)p;      // (as an integer, then a global-address-space pointer: global_load, not flat_load -- see planes.h: launder)
  asm volatile("" : "+s"(u));
  return (const uint4*)(const uint4 __attribute__((address_space(1)))*)u;
}

template <int NP, int PLANE>
__device__ __forceinline__ void store_planes(__bf16* dst, const bf16x4& h, const bf16x4& m, const bf16x4& l) {
  *reinterpret_cast<bf16x4*>(dst) = h;
  if (NP == 3) {
    *reinterpret_cast<bf16x4*>(dst + PLANE) = m;
    *reinterpret_cast<bf16x4*>(dst + 2 * PLANE) = l;
  }
}

// c += sum over the four k-steps s and the four lane groups j of a[s] * b[s]: four exact fp32 MFMAs (16x16x4) in the parity mode,
// ONE bf16 MFMA (v_mfma_f32_16x16x16_bf16: lane (i, j) supplies k = 4j .. 4j+3, the same element order) in bf16 mode
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int NP>
__device__ __forceinline__ f32x4 mma4(const f32x4& a, const f32x4& b, f32x4 c) {
  if (NP == 1) {
    const bf16x4 ab = bf16x4{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3]};
    const bf16x4 bb = bf16x4{(__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, ab), __builtin_bit_cast(s16x4, bb), c, 0, 0, 0);
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) c = mfma16(a[s], b[s], c);
  return c;
}

// NP = 3: the six plane products; NP = 1 (bf16 mode): the one product of the bf16-rounded operands
template <int NP>
__device__ __forceinline__ f32x4 mma(const bf16x8& wh, const bf16x8& wm, const bf16x8& wl, const bf16x8& ah, const bf16x8& am,
                                     const bf16x8& al, f32x4 c) {
  if (NP == 1) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah, c, 0, 0, 0);
  return mma6(wh, wm, wl, ah, am, al, c);
}

__device__ __forceinline__ float gmax16(float v) {   // over the 4 lane groups sharing lane&15
  v = fmaxf(v, __shfl_xor(v, 16));
  return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float gsum16(float v) {
  v += __shfl_xor(v, 16);
  return v + __shfl_xor(v, 32);
}

// sum over the 64 lanes without the LDS crossbar: four DPP steps give every lane its 16-lane row's sum (quad_perm 1,0,3,2 /
// 2,3,0,1, row_half_mirror, row_mirror), four v_readlane gather the rows.  (__shfl_xor is ds_bpermute: ~100 cycles a step)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  const int iv = __builtin_bit_cast(int, v);
  return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16))) +
         (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48)));
}

struct TowerFwdArgs {
  const float* X;            // [B*L, D] layer input
  int B, L, heads;
  const uint4* Wqkv;         // pre-split bf16 images (pack_b3 layout, k extent padded to 128): [D -> 3D], [D -> D], [D -> D]
  const uint4* W1;
  const uint4* W2;
  const float* b1; const float* b2; const float* gamma; const float* beta;
  float* out;                // [B*L, D] or NULL (the caller rebuilds it from x-hat: fused tail)
  float* QKV;                // training stash (each may be NULL): [B*L, 3D]
  int qkv16;                 // bf16 mode: the QKV / A / R1 stashes are bf16 arrays (their only readers are matrix products / a sign test)
  float* A;                  // [B*L, D] attention output
  float* LSE;                // [B*heads*L] natural-log softmax normalisers
  float* R1;                 // [B*L, D] relu(W1 A + b1)
  float* XH;                 // [B*L, D] LayerNorm x-hat
  float* RSTD;               // [B*L]
  unsigned long long* dbg;   // INTEL_TOWER_DBG=1: per-phase shader-clock totals of workgroup 0's thread 0 (NULL otherwise)
  // The first layer's input built INSIDE the kernel (X == NULL, template parameter INP; inference): the session's <= 64 rows go from the embedding tables
  // straight into the plane image, the [B*L, D] input tensor never exists in HBM.  INP = 1 (item tower, IntEL.py:170-173): columns 0 .. d0-1 =
  // tab0[idx0[row]], columns d0 .. D-1 = tab1[idx1[row]] (a negative id = a zero row, as gather_rows).  (The score tower's K-wide input linear was
  // tried inside the kernel as well and dropped: 137 -> 183 us per launch for the 21 us the separate linear takes.)
  const float* tab0; const int* idx0; int d0;
  const float* tab1; const int* idx1;
};

template <int D, int NP = 3>
struct TowerCfg {
  static constexpr int NW = D / 16;              // waves: 8 (D = 128) or 4 (D = 64)
  static constexpr int NT = NW * 64;
  static constexpr int KB = D / 32;              // 32-deep k blocks
  static constexpr int LDP = D + 8;              // bf16 plane pitch (16-byte fragment reads conflict-free)
  static constexpr int PLANE = 64 * LDP;         // bf16 elements per plane
  static constexpr int LQ = D + 4;               // fp32 row pitch of Q / K / V / the LayerNorm tile
  static constexpr int NJ = 64 * (D / 4) / NT;   // float4 per thread per X tile (= 4)
  // parity mode: three planes, fp32 q/k/v rows.  bf16 mode (NP = 1): one plane, q/k/v rows as bf16 (the attention rounds them
  // anyway) -- 69.6 KB instead of 153.6 KB at D = 128, so two workgroups share a CU (four at D = 64)
  static constexpr size_t P_BYTES = (size_t)NP * PLANE * 2;
  static constexpr size_t R_BYTES = NP == 1 ? (size_t)3 * PLANE * 2 : (size_t)3 * 64 * LQ * 4;
  static constexpr size_t ES_BYTES = (size_t)64 * LQ * 4;          // the LayerNorm tile (over the dead q rows)
  static constexpr size_t SMEM = P_BYTES + R_BYTES;
  static_assert(NP != 1 || ES_BYTES + (size_t)PLANE * 2 <= R_BYTES, "bf16 mode: LayerNorm tile + R1 plane must fit the q/k/v region");
  static constexpr int WAVES_PER_SIMD = NP == 1 ? 4 : 2;           // bf16 mode: <= 128 registers
};

// training stash store: four consecutive values at element offset `off`, as fp32 or (bf16 mode) bf16
template <int NP>
__device__ __forceinline__ void stash4(float* base, size_t off, const f32x4& v, int h16) {
  if (NP == 1 && h16)
    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(base) + off) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  else
    *reinterpret_cast<f32x4*>(base + off) = v;
}

// Workgroup barrier over the LDS tile only.  __syncthreads() also waits for every global access of the wave (s_waitcnt
// vmcnt(0)): here that would be the next session's X prefetch and the stash stores, whose latency the phases exist to hide.
// Nothing one wave writes to global memory is read by another wave of this kernel.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// bf16 mode: the training stashes are the bf16 rows the phases leave in LDS -- copied out row-contiguous, 16 bytes per lane
// (src: [rows][ldp] bf16 in LDS, nmat matrices PLANE elements apart side by side in the destination row)
template <int D, int LDP, int PLANE, int NT>
__device__ __forceinline__ void stash_rows16(const __bf16* src, int nmat, __bf16* dst, int ldd, int rows, int tid) {
  constexpr int CPR = D / 8;                       // 16-byte chunks per matrix row
  const int total = rows * nmat * CPR;
  for (int i = tid; i < total; i += NT) {
    const int row = i / (nmat * CPR), c = i - row * (nmat * CPR);
    const int which = c / CPR, cc = c - which * CPR;
    *reinterpret_cast<uint4*>(dst + (size_t)row * ldd + which * D + cc * 8) =
        *reinterpret_cast<const uint4*>(src + which * PLANE + row * LDP + cc * 8);
  }
}

template <int D, int DK, bool TRAIN, int NP = 3, int INP = 0>
__global__ __launch_bounds__((TowerCfg<D, NP>::NT), (TowerCfg<D, NP>::WAVES_PER_SIMD)) void tower_fwd_fused_kernel(TowerFwdArgs a) {
  using C = TowerCfg<D, NP>;
  constexpr int NW = C::NW, NT = C::NT, KB = C::KB, LDP = C::LDP, PLANE = C::PLANE, LQ = C::LQ, NJ = C::NJ;
  constexpr int HEADS = D / DK, DKT = DK / 16;
  constexpr int KBT = 4;                         // k blocks per column tile in the image (k padded to 128)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* planes = reinterpret_cast<__bf16*>(smem_raw);                 // P: X planes, later the attention-output planes
  float* Qs = reinterpret_cast<float*>(smem_raw + C::P_BYTES);          // R: Q | K | V rows; later Es (over Q) and R1 planes (over K, V)
  float* Ks = Qs + 64 * LQ;
  float* Vs = Ks + 64 * LQ;
  float* Es = Qs;
  __bf16* Q16 = reinterpret_cast<__bf16*>(smem_raw + C::P_BYTES);       // bf16 mode: q | k | v rows with the plane pitch
  __bf16* K16 = Q16 + PLANE;
  __bf16* V16 = K16 + PLANE;
  __bf16* r1planes = NP == 1 ? reinterpret_cast<__bf16*>(smem_raw + C::P_BYTES + C::ES_BYTES) : reinterpret_cast<__bf16*>(Ks);
  const int tid = threadIdx.x, lane = tid & 63, j = lane >> 4, p = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = a.L;
  const float scale = 1.0f / sqrtf((float)DK);
  const float c2 = scale * 1.4426950408889634f;

  // X tile staging: float4 #i of the tile = (row i / (D/4), 4 * (i % (D/4)))
  f32x4 pre[NJ];
  int trow[NJ], tcol[NJ];
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int i = tid + NT * jj;
    trow[jj] = i / (D / 4);
    tcol[jj] = (i - trow[jj] * (D / 4)) * 4;
  }
  // INP = 1: the session's ids are staged in LDS one session ahead of its rows (ids[slot][table][row], three slots): a row load whose address waits
  // for an id load in the same phase is two dependent HBM round trips on the wave's critical path (measured: -10 % evaluation sessions/s)
  int* ids = reinterpret_cast<int*>(smem_raw + C::SMEM);
  auto load_id = [&](int bb) {      // threads 0 .. 127: (table tid >> 6, row tid & 63) of session bb
    const size_t g = (size_t)bb * L + min(tid & 63, L - 1);
    return (tid >> 6) == 0 ? a.idx0[g] : a.idx1[g];
  };
  int idreg = 0;
  // Nothing in load_x may LOOK at a loaded value: a select on it (padding rows, negative ids) makes the wave wait for the
  // load where it is issued -- the random table rows of INP = 1 take an HBM round trip, not the Infinity-Cache hit of a tile the gather kernel just
  // wrote (measured: +40 us per launch).  The raw values stay in registers; stage_x applies the masks when the tile goes to LDS.
  auto load_x = [&](int b, int slot) {
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      const int row = min(trow[jj], L - 1);
      if constexpr (INP == 1) {           // two gathered tables side by side
        const bool first = tcol[jj] < a.d0;
        const int id = ids[slot * 128 + (first ? 0 : 64) + row];
        const float* src = first ? a.tab0 + (size_t)max(id, 0) * a.d0 + tcol[jj] : a.tab1 + (size_t)max(id, 0) * (D - a.d0) + (tcol[jj] - a.d0);
        pre[jj] = *reinterpret_cast<const f32x4*>(src);
      } else {
        pre[jj] = *reinterpret_cast<const f32x4*>(a.X + ((size_t)b * L + row) * D + tcol[jj]);
      }
    }
  };
  // piece jj of the tile being staged (session slot `slot` of the id table): the input values, padding rows (and rows with a negative id) zeroed
  auto stage_x = [&](int jj, int slot) {
    bool live = trow[jj] < L;
    if constexpr (INP == 1) live = live && ids[slot * 128 + (tcol[jj] < a.d0 ? 0 : 64) + min(trow[jj], L - 1)] >= 0;
    return live ? pre[jj] : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  int b = blockIdx.x;
  if (b >= a.B) return;
  if constexpr (INP == 1) {
    if (tid < 128) {
      ids[tid] = load_id(b);
      if (b + (int)gridDim.x < a.B) idreg = load_id(b + gridDim.x);
    }
    __syncthreads();
  }
  int it = 0;
  load_x(b, 0);
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
  for (; b < a.B; b += gridDim.x, ++it) {
    // ---- phase 0: X -> planes
    if constexpr (INP == 1) {      // the next session's ids (requested an iteration ago) -> their slot; the barrier below publishes them
      if (tid < 128) ids[((it + 1) % 3) * 128 + tid] = idreg;
    }
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      bf16x4 h, m, l;
      split4(stage_x(jj, it % 3), h, m, l);
      const int off = trow[jj] * LDP + tcol[jj];
      store_planes<NP, PLANE>(planes + off, h, m, l);
    }
    lds_barrier();
    mark(0);
    // ---- phase 1: [Q | K | V] = X Wqkv^T; wave = column tiles 3 wave .. 3 wave + 2, all four row tiles
    if constexpr (NP == 1) {
      // bf16 mode runs four waves per SIMD (<= 128 registers): one column tile at a time, the A fragments are re-read
      const uint4* Bimg = per_session(a.Wqkv) + ((size_t)(3 * wave) * KBT * 3) * 64 + lane;
      const __bf16* frag = planes + p * LDP + 8 * j;
#pragma unroll 1
      for (int c = 0; c < 3; ++c) {
        uint4 bw[2];                               // weight fragments one k block ahead (register budget)
        bw[0] = Bimg[((size_t)(c * KBT) * 3) * 64];
        f32x4 acc[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          if (kb + 1 < KB) bw[(kb + 1) & 1] = Bimg[((size_t)(c * KBT + kb + 1) * 3) * 64];
#pragma unroll
          for (int rt = 0; rt < 4; ++rt)
            acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bw[kb & 1]),
                                                              *reinterpret_cast<const bf16x8*>(frag + rt * 16 * LDP + kb * 32), acc[rt], 0, 0, 0);
        }
        const int n = (3 * wave + c) * 16 + 4 * j;           // column of [q | k | v]
        const int which = n / D, col = n - which * D;
        __bf16* dst16 = Q16 + which * PLANE + col;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          const int row = rt * 16 + p;
          const f32x4& v = acc[rt];
          *reinterpret_cast<bf16x4*>(dst16 + row * LDP) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        }
      }
    } else {
      f32x4 acc[3][4];
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[c][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const uint4* Bimg = per_session(a.Wqkv) + ((size_t)(3 * wave) * KBT * 3) * 64 + lane;
      uint4 bw[2][3][3];
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bw[0][c][pl] = Bimg[((size_t)(c * KBT + 0) * 3 + pl) * 64];
      const __bf16* frag = planes + p * LDP + 8 * j;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        if (kb + 1 < KB) {
#pragma unroll
          for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bw[(kb + 1) & 1][c][pl] = Bimg[((size_t)(c * KBT + kb + 1) * 3 + pl) * 64];
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          const __bf16* fp = frag + rt * 16 * LDP + kb * 32;
          const bf16x8 ah = *reinterpret_cast<const bf16x8*>(fp);
          const bf16x8 am = *reinterpret_cast<const bf16x8*>(fp + PLANE);
          const bf16x8 al = *reinterpret_cast<const bf16x8*>(fp + 2 * PLANE);
#pragma unroll
          for (int c = 0; c < 3; ++c)
            acc[c][rt] = mma<NP>(__builtin_bit_cast(bf16x8, bw[kb & 1][c][0]), __builtin_bit_cast(bf16x8, bw[kb & 1][c][1]),
                              __builtin_bit_cast(bf16x8, bw[kb & 1][c][2]), ah, am, al, acc[c][rt]);
        }
      }
      // epilogue: fp32 rows to LDS (Q | K | V side by side, heads side by side inside each); training: the q/k/v stash
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int n = (3 * wave + c) * 16 + 4 * j;           // column of [q | k | v]
        const int which = n / D, col = n - which * D;
        float* dst = Qs + which * 64 * LQ + col;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          const int row = rt * 16 + p;
          *reinterpret_cast<f32x4*>(dst + row * LQ) = acc[c][rt];
          if (TRAIN && a.QKV && row < L) stash4<NP>(a.QKV, ((size_t)b * L + row) * (3 * D) + n, acc[c][rt], a.qkv16);
        }
      }
    }
    // the next session's rows travel while this one is computed
    // (INP = 1: the gathered rows are random 256-byte HBM reads -- slower to return than a streamed tile -- and vector-memory results come back in issue
    // order: requested here, the W1 fragments below would queue behind them; they are requested after those fragments instead)
    if (INP != 1 && b + (int)gridDim.x < a.B) load_x(b + gridDim.x, (it + 1) % 3);
    mark(1);
    lds_barrier();
    mark(6);
    if constexpr (NP == 1 && TRAIN) {
      if (a.QKV) stash_rows16<D, LDP, PLANE, NT>(Q16, 3, reinterpret_cast<__bf16*>(a.QKV) + (size_t)b * L * (3 * D), 3 * D, L, tid);
    }
    // ---- phase 2: attention; (query tile, head) pairs over the waves.  The W1 fragments of phase 3 travel meanwhile.
    uint4 bw1[KB][3];
    if constexpr (NP != 1) {
      const uint4* Bimg = per_session(a.W1) + ((size_t)wave * KBT * 3) * 64 + lane;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bw1[kb][pl] = Bimg[((size_t)kb * 3 + pl) * 64];
    }
    if constexpr (INP == 1) {
      if (b + (int)gridDim.x < a.B) load_x(b + gridDim.x, (it + 1) % 3);
      if (tid < 128 && b + 2 * (int)gridDim.x < a.B) idreg = load_id(b + 2 * gridDim.x);
    }
    for (int pair = wave; pair < 4 * HEADS; pair += NW) {
      const int tile = pair & 3, h = pair >> 2;
      if (tile * 16 >= L) continue;                          // wave-uniform: a tile of padding only
      const float* Qp = Qs + (tile * 16 + p) * LQ + h * DK + 4 * j;
      const float* Kp = Ks + p * LQ + h * DK + 4 * j;
      f32x4 st[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (NP == 1) {
        const __bf16* Qh = Q16 + (tile * 16 + p) * LDP + h * DK + 4 * j;
        const __bf16* Kh = K16 + p * LDP + h * DK + 4 * j;
#pragma unroll
        for (int g = 0; g < DKT; ++g) {
          const s16x4 qf = *reinterpret_cast<const s16x4*>(Qh + 16 * g);
          s16x4 kf[4];
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) kf[kt] = *reinterpret_cast<const s16x4*>(Kh + kt * 16 * LDP + 16 * g);
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) st[kt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kf[kt], qf, st[kt], 0, 0, 0);
        }
      } else {
#pragma unroll
      for (int g = 0; g < DKT; ++g) {
        const f32x4 qf = *reinterpret_cast<const f32x4*>(Qp + 16 * g);
        f32x4 kf[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) kf[kt] = *reinterpret_cast<const f32x4*>(Kp + kt * 16 * LQ + 16 * g);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) st[kt] = mma4<NP>(kf[kt], qf, st[kt]);
      }
      }
      // accumulator register r of tile kt at lane (j, p) = key kt*16 + 4j + r, query tile*16 + p
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = (kt * 16 + 4 * j + r) < L ? st[kt][r] : -INFINITY;
          st[kt][r] = v;
          mx = fmaxf(mx, v);
        }
      mx = gmax16(mx);
      const float moff = -mx * c2;
      float ps = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][r], c2, moff));
          st[kt][r] = e;
          ps += e;
        }
      ps = gsum16(ps);
      const float inv = 1.f / ps;                            // L >= 1: at least one live key
      const int q = tile * 16 + p;
      if (TRAIN && a.LSE && j == 0 && q < L) a.LSE[((size_t)b * HEADS + h) * L + q] = mx * scale + __logf(ps);     // (before P V: mx / ps die here)
      if constexpr (DK >= 64) {
        // V read as one b128 along the head dim: lane p takes dims 4p .. 4p+3 of its key row and feeds four MFMAs whose
        // output row i means dim 4 i + t
        constexpr int DQ = DK >= 64 ? DK / 64 : 1;
        f32x4 oT[DQ * 4];
#pragma unroll
        for (int i = 0; i < DQ * 4; ++i) oT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* Vp = Vs + (4 * j) * LQ + h * DK + 4 * p;
        if constexpr (NP == 1) {
          // one 64-dim block of the head at a time (register budget): P as bf16 once, V rows read as 8-byte bf16 groups
          const __bf16* Vh = V16 + (4 * j) * LDP + h * DK + 4 * p;
          s16x4 pb[4];
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
            pb[kt] = __builtin_bit_cast(s16x4, bf16x4{(__bf16)st[kt][0], (__bf16)st[kt][1], (__bf16)st[kt][2], (__bf16)st[kt][3]});
#pragma unroll
          for (int dq = 0; dq < DQ; ++dq) {
            f32x4 o4[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) o4[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
              s16x4 vv[4];                 // the four keys kt*16 + 4j + r of this lane group, dims 4p .. 4p+3
#pragma unroll
              for (int r = 0; r < 4; ++r) vv[r] = *reinterpret_cast<const s16x4*>(Vh + (kt * 16 + r) * LDP + dq * 64);
#pragma unroll
              for (int t = 0; t < 4; ++t)
                o4[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(s16x4{vv[0][t], vv[1][t], vv[2][t], vv[3][t]}, pb[kt], o4[t], 0, 0, 0);
            }
            // this 64-dim block of the output row goes straight into the plane of the W1 product
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const f32x4 o = f32x4{o4[0][r], o4[1][r], o4[2][r], o4[3][r]} * inv;
              const int col = h * DK + dq * 64 + 16 * j + 4 * r;
              *reinterpret_cast<bf16x4*>(planes + q * LDP + col) = bf16x4{(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int dq = 0; dq < DQ; ++dq) {
            f32x4 vv[4];                   // the four keys kt*16 + 4j + r of this lane group, dims 4p .. 4p+3
#pragma unroll
            for (int r = 0; r < 4; ++r) vv[r] = *reinterpret_cast<const f32x4*>(Vp + (kt * 16 + r) * LQ + dq * 64);
#pragma unroll
            for (int t = 0; t < 4; ++t)
              oT[dq * 4 + t] = mma4<NP>(f32x4{vv[0][t], vv[1][t], vv[2][t], vv[3][t]}, st[kt], oT[dq * 4 + t]);
          }
        }
        if constexpr (NP != 1) {
#pragma unroll
        for (int dq = 0; dq < DQ; ++dq)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const f32x4 o = f32x4{oT[dq * 4 + 0][r], oT[dq * 4 + 1][r], oT[dq * 4 + 2][r], oT[dq * 4 + 3][r]} * inv;
            const int col = h * DK + dq * 64 + 16 * j + 4 * r;
            bf16x4 hh, mm, ll;
            split4(o, hh, mm, ll);
            const int off = q * LDP + col;
            store_planes<NP, PLANE>(planes + off, hh, mm, ll);
            if (TRAIN && a.A && q < L) stash4<NP>(a.A, ((size_t)b * L + q) * D + col, o, a.qkv16);
          }
        }
      } else {
        // narrow heads (dk = 32): two output tiles of 16 dims, V read as scalars (lane p = dim, j = key of the k-step)
        f32x4 oT[2];
        oT[0] = oT[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* Vp = Vs + (4 * j) * LQ + h * DK + p;
        const __bf16* Vh = V16 + (4 * j) * LDP + h * DK + p;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            if (NP == 1)
              oT[t] = mma4<NP>(f32x4{(float)Vh[(kt * 16 + 0) * LDP + t * 16], (float)Vh[(kt * 16 + 1) * LDP + t * 16], (float)Vh[(kt * 16 + 2) * LDP + t * 16],
                                     (float)Vh[(kt * 16 + 3) * LDP + t * 16]}, st[kt], oT[t]);
            else
              oT[t] = mma4<NP>(f32x4{Vp[(kt * 16 + 0) * LQ + t * 16], Vp[(kt * 16 + 1) * LQ + t * 16], Vp[(kt * 16 + 2) * LQ + t * 16],
                                     Vp[(kt * 16 + 3) * LQ + t * 16]}, st[kt], oT[t]);
          }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f32x4 o = oT[t] * inv;                       // register r = dim t*16 + 4j + r of query p
          const int col = h * DK + t * 16 + 4 * j;
          bf16x4 hh, mm, ll;
          split4(o, hh, mm, ll);
          const int off = q * LDP + col;
          store_planes<NP, PLANE>(planes + off, hh, mm, ll);
          if (NP != 1 && TRAIN && a.A && q < L) stash4<NP>(a.A, ((size_t)b * L + q) * D + col, o, a.qkv16);
        }
      }
    }
    // query tiles of padding only were skipped above: their planes still hold X (rows >= L are zero there already)
    mark(2);
    lds_barrier();
    mark(7);
    if constexpr (NP == 1 && TRAIN) {
      if (a.A) stash_rows16<D, LDP, PLANE, NT>(planes, 1, reinterpret_cast<__bf16*>(a.A) + (size_t)b * L * D, D, L, tid);
    }
    // ---- phase 3: R1 = relu(A W1^T + b1); wave = one column tile, four row tiles (the W2 fragments of phase 4 travel meanwhile)
    uint4 bw2[KB][3];
    if constexpr (NP != 1) {
      const uint4* Bimg = per_session(a.W2) + ((size_t)wave * KBT * 3) * 64 + lane;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bw2[kb][pl] = Bimg[((size_t)kb * 3 + pl) * 64];
    } else {      // bf16 mode (register budget): the fragments of both products are fetched where they are used
      const uint4* Bimg = per_session(a.W1) + ((size_t)wave * KBT * 3) * 64 + lane;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) bw1[kb][0] = Bimg[((size_t)kb * 3) * 64];
    }
    {
      f32x4 acc[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const __bf16* frag = planes + p * LDP + 8 * j;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          const __bf16* fp = frag + rt * 16 * LDP + kb * 32;
          const bf16x8 ah = *reinterpret_cast<const bf16x8*>(fp);
          const bf16x8 am = *reinterpret_cast<const bf16x8*>(fp + PLANE);
          const bf16x8 al = *reinterpret_cast<const bf16x8*>(fp + 2 * PLANE);
          acc[rt] = mma<NP>(__builtin_bit_cast(bf16x8, bw1[kb][0]), __builtin_bit_cast(bf16x8, bw1[kb][1]), __builtin_bit_cast(bf16x8, bw1[kb][2]),
                         ah, am, al, acc[rt]);
        }
      const int col = wave * 16 + 4 * j;
      const f32x4 bias = *reinterpret_cast<const f32x4*>(a.b1 + col);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const int row = rt * 16 + p;
        f32x4 x = acc[rt] + bias;
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = fmaxf(x[r], 0.f);
        bf16x4 hh, mm, ll;
        split4(x, hh, mm, ll);
        const int off = row * LDP + col;
        store_planes<NP, PLANE>(r1planes + off, hh, mm, ll);
        if (NP != 1 && TRAIN && a.R1 && row < L) stash4<NP>(a.R1, ((size_t)b * L + row) * D + col, x, a.qkv16);
      }
    }
    lds_barrier();
    mark(3);
    if constexpr (NP == 1 && TRAIN) {
      if (a.R1) stash_rows16<D, LDP, PLANE, NT>(r1planes, 1, reinterpret_cast<__bf16*>(a.R1) + (size_t)b * L * D, D, L, tid);
    }
    // the residual rows of phase 5 come back from L2 while the W2 product runs (bf16 mode: loaded in phase 5 -- the 128-register
    // budget of its four waves per SIMD has no room for them here, and a spilling kernel's occupancy depends on the scratch the
    // runtime happens to have allocated: 2-4x run-to-run differences were measured)
    constexpr int RPW = 64 / NW, CPL = D / 64;
    float res[RPW][CPL];
    auto load_res = [&]() {
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int row = min(wave * RPW + rr, L - 1);
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) {
          const int c = lane + 64 * cc;
          if constexpr (INP == 1) {       // (the rows are L2-hot: this session's phase 0 gathered them; the ids sit in LDS)
            const bool first = c < a.d0;
            const int id = ids[(it % 3) * 128 + (first ? 0 : 64) + row];
            const float v = first ? a.tab0[(size_t)max(id, 0) * a.d0 + c] : a.tab1[(size_t)max(id, 0) * (D - a.d0) + (c - a.d0)];
            res[rr][cc] = id < 0 ? 0.f : v;
          } else {
            res[rr][cc] = a.X[((size_t)b * L + row) * D + c];
          }
        }
      }
    };
    if (NP != 1) load_res();
    // ---- phase 4: Z = R1 W2^T + b2 -> fp32 tile
    if constexpr (NP == 1) {
      const uint4* Bimg = per_session(a.W2) + ((size_t)wave * KBT * 3) * 64 + lane;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) bw2[kb][0] = Bimg[((size_t)kb * 3) * 64];
    }
    {
      f32x4 acc[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const __bf16* frag = r1planes + p * LDP + 8 * j;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          const __bf16* fp = frag + rt * 16 * LDP + kb * 32;
          const bf16x8 ah = *reinterpret_cast<const bf16x8*>(fp);
          const bf16x8 am = *reinterpret_cast<const bf16x8*>(fp + PLANE);
          const bf16x8 al = *reinterpret_cast<const bf16x8*>(fp + 2 * PLANE);
          acc[rt] = mma<NP>(__builtin_bit_cast(bf16x8, bw2[kb][0]), __builtin_bit_cast(bf16x8, bw2[kb][1]), __builtin_bit_cast(bf16x8, bw2[kb][2]),
                         ah, am, al, acc[rt]);
        }
      const int col = wave * 16 + 4 * j;
      const f32x4 bias = *reinterpret_cast<const f32x4*>(a.b2 + col);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) *reinterpret_cast<f32x4*>(Es + (rt * 16 + p) * LQ + col) = acc[rt] + bias;
    }
    lds_barrier();
    mark(4);
    // ---- phase 5: LayerNorm(Z + X) over the D columns; wave = 64 / NW rows, lane = columns lane (and lane + 64)
    {
      if (NP == 1) load_res();
      float g[CPL], be[CPL];
#pragma unroll
      for (int cc = 0; cc < CPL; ++cc) { g[cc] = a.gamma[lane + 64 * cc]; be[cc] = a.beta[lane + 64 * cc]; }
      const float inv_n = 1.f / (float)D;
      // all rows of the wave advance together: the reductions of different rows are independent and overlap
      float v[RPW][CPL], mean[RPW], rs[RPW];
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int row = wave * RPW + rr;
        float s = 0.f;
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) { v[rr][cc] = Es[row * LQ + lane + 64 * cc] + res[rr][cc]; s += v[rr][cc]; }
        mean[rr] = s;
      }
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) mean[rr] = wave_sum_dpp(mean[rr]) * inv_n;
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        float q2 = 0.f;
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) { v[rr][cc] -= mean[rr]; q2 += v[rr][cc] * v[rr][cc]; }
        rs[rr] = q2;
      }
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) rs[rr] = 1.f / sqrtf(wave_sum_dpp(rs[rr]) * inv_n + 1e-5f);
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int row = wave * RPW + rr;
        if (row < L) {                                      // wave-uniform
          const size_t grow = (size_t)b * L + row;
          if (TRAIN && a.RSTD && lane == 0) a.RSTD[grow] = rs[rr];
#pragma unroll
          for (int cc = 0; cc < CPL; ++cc) {
            const float xh = v[rr][cc] * rs[rr];
            if (TRAIN && a.XH) a.XH[grow * D + lane + 64 * cc] = xh;
            if (a.out) a.out[grow * D + lane + 64 * cc] = xh * g[cc] + be[cc];
          }
        }
      }
    }
    mark(5);
    // the next iteration's phase 0 writes the X planes (last read in phase 3) and its first barrier orders the
    // LayerNorm reads of Es before the next q/k/v rows are stored over them
  }
}

// INTEL_FUSE_TOWER: 0 = never, 1 = always, unset / auto = where it is the faster path (measured on one box, same process):
//   inference:              fused   (+4 % sessions/s in fp32, +30 % in bf16 mode)
//   training, bf16 mode:    fused   (two workgroups per CU at D = 128: -0.1 ms per step, 1.9 GB less HBM traffic)
//   training, fp32 mode:    fused for the 64-wide tower (same step time, 0.4 GB less traffic), kernel-per-op pipeline for the
//                           128-wide one (its six-product MFMA work, the L2 -> CU weight stream and one workgroup per CU
//                           make the step 2.4 % slower: DESIGN.md 6)
int fused_mode() {
  static const int m = [] { const char* e = getenv("INTEL_FUSE_TOWER"); return !e || !e[0] || e[0] == 'a' ? 2 : (e[0] == '0' ? 0 : 1); }();
  return m;
}

template <int D, int DK, bool TRAIN, int NP = 3, int INP = 0>
int launch_one(const TowerFwdArgs& a, hipStream_t st) {
  using C = TowerCfg<D, NP>;
  const size_t smem = C::SMEM + (INP == 1 ? 3 * 128 * sizeof(int) : 0);      // INP = 1: three slots of 2 x 64 ids
  static_assert(C::SMEM + 3 * 128 * sizeof(int) <= 160 * 1024, "LDS budget");
  if constexpr (NP == 3 && INP == 0) {
    if (gemm_planes() == 1) return launch_one<D, DK, TRAIN, 1, 0>(a, st);      // bf16 mode
  }
  if constexpr (!TRAIN && INP == 0) {      // inference with the input built in the kernel (TowerInput: no X)
    if (!a.X) {
      // fp32 mode only: the bf16 mode's variants run four waves per SIMD on 128 registers and the gather's addressing spilled 2 - 7 of them
      if constexpr (NP == 3) return launch_one<D, DK, TRAIN, NP, 1>(a, st);
      intel_set_error("tower_fwd_fused: the in-kernel input is the fp32 mode's");
      return -1;
    }
  }
  allow_lds((tower_fwd_fused_kernel<D, DK, TRAIN, NP, INP>), smem);      // (smem: + the id slots of the in-kernel gather, below)
  // resident workgroups per CU: LDS (160 KB) and wave slots (NW waves each, WAVES_PER_SIMD per SIMD by the launch bounds)
  int per_cu = (int)((160 * 1024) / smem);
  if (per_cu > 4 * C::WAVES_PER_SIMD / C::NW) per_cu = 4 * C::WAVES_PER_SIMD / C::NW;
  if (per_cu < 1) per_cu = 1;
  int grid = num_cus() * per_cu;
  if (grid > a.B) grid = a.B;
  const double M = (double)a.B * a.L;
  // algorithmic work: 5 D x D linears per row + the two attention products; bytes: X in, the output (or x-hat) out, the stash
  const double flops = 2.0 * M * D * D * 5 + 4.0 * (double)a.B * a.L * a.L * D;
  double bytes = 4.0 * M * D * (1.0 + (a.out ? 1.0 : 0.0));
  if (TRAIN) bytes += 4.0 * M * D * (((a.QKV ? 3.0 : 0.0) + (a.A ? 1.0 : 0.0) + (a.R1 ? 1.0 : 0.0)) * (a.qkv16 ? 0.5 : 1.0) + (a.XH ? 1.0 : 0.0));
  static const int dbg_on = INTEL_DEBUG_ENV("INTEL_TOWER_DBG", 0);      // phase clocks: debug builds only (common.h)
  TowerFwdArgs aa = a;
  static unsigned long long* dbg_buf = nullptr;
  if (dbg_on) {
    if (!dbg_buf) (void)hipMalloc(&dbg_buf, 8 * sizeof(unsigned long long));
    (void)hipMemsetAsync(dbg_buf, 0, 8 * sizeof(unsigned long long), st);
    aa.dbg = dbg_buf;
  }
  LAUNCH_S(a.B * a.L, D, DK, flops, bytes, (tower_fwd_fused_kernel<D, DK, TRAIN, NP, INP>), dim3(grid), dim3(C::NT), smem, st, aa);
  INTEL_CHECK_LAUNCH();
  if (dbg_on) {          // tools/tower_probe.py
    unsigned long long h[8];
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h, dbg_buf, sizeof(h), hipMemcpyDeviceToHost);
    const int iters = (a.B + grid - 1) / grid;
    fprintf(stderr, "tower_fwd D=%d dk=%d train=%d grid=%d iters=%d  cycles/session: p0 %llu  qkv %llu (+bar %llu)  attn %llu (+bar %llu)  w1 %llu  w2 %llu  ln %llu\n",
            D, DK, (int)TRAIN, grid, iters, h[0] / iters, h[1] / iters, h[6] / iters, h[2] / iters, h[7] / iters, h[3] / iters, h[4] / iters, h[5] / iters);
  }
  return 0;
}

}  // namespace

bool tower_fused_wanted(int train, int d) {
  const int m = fused_mode();
  if (m != 2) return m == 1;
  // fp32 training: the 64-wide tower only (80 KB of LDS: two workgroups per CU; same step time as the kernel-per-op pipeline,
  // 0.4 GB less traffic); the 128-wide one (154 KB, one workgroup per CU) stays on the pipeline (-2.4 % otherwise)
  return !train || gemm_planes() == 1 || d == 64;
}

bool tower_fused_supported(int L, int d, int heads) {
  if (fused_mode() == 0) return false;
  if (L < 1 || L > 64 || heads < 1 || d % heads != 0) return false;
  const int dk = d / heads;
  return (d == 128 && (dk == 128 || dk == 64)) || (d == 64 && (dk == 64 || dk == 32));
}

int launch_tower_fwd_fused(const float* X, int B, int L, int d, int heads, const void* Wqkv_b3, const void* W1_b3, const void* W2_b3,
                           const float* b1, const float* b2, const float* gamma, const float* beta, float* out, int train,
                           float* QKV, float* A, float* LSE, float* R1, float* XH, float* RSTD, hipStream_t st, int qkv16, const TowerInput* in) {
  if (B <= 0) return 0;
  INTEL_CHECK_ARG(tower_fused_supported(L, d, heads), "tower_fwd_fused: unsupported shape L=%d d=%d heads=%d", L, d, heads);
  TowerFwdArgs a;
  a.X = X; a.B = B; a.L = L; a.heads = heads;
  a.tab0 = a.tab1 = nullptr; a.idx0 = a.idx1 = nullptr; a.d0 = 0;
  if (in) {
    INTEL_CHECK_ARG(!train && !X, "tower_fwd_fused: the in-kernel input is the inference path's (no X, train = 0)");
    INTEL_CHECK_ARG(in->tab0 && in->idx0 && in->d0 > 0 && in->d0 % 4 == 0 && in->d0 <= d && (in->d0 == d || (in->tab1 && in->idx1)),
                    "tower_fwd_fused: gathered input needs tab0 / idx0 (width d0 %% 4 == 0) and, for d0 < d, tab1 / idx1");
    a.tab0 = in->tab0; a.idx0 = in->idx0; a.d0 = in->d0;
    a.tab1 = in->d0 == d ? in->tab0 : in->tab1; a.idx1 = in->d0 == d ? in->idx0 : in->idx1;
  } else {
    INTEL_CHECK_ARG(X != nullptr, "tower_fwd_fused: no input");
  }
  a.Wqkv = reinterpret_cast<const uint4*>(Wqkv_b3); a.W1 = reinterpret_cast<const uint4*>(W1_b3); a.W2 = reinterpret_cast<const uint4*>(W2_b3);
  a.b1 = b1; a.b2 = b2; a.gamma = gamma; a.beta = beta; a.out = out;
  a.qkv16 = (qkv16 && gemm_planes() == 1) ? 1 : 0;
  INTEL_CHECK_ARG(!qkv16 || a.qkv16, "tower_fwd: the bf16 q/k/v stash needs the bf16 mode");
  INTEL_CHECK_ARG(!(train && gemm_planes() == 1 && (QKV || A || R1)) || a.qkv16, "tower_fwd: in bf16 mode the training stashes are bf16 arrays (qkv16 = 1)");
  a.QKV = QKV; a.A = A; a.LSE = LSE; a.R1 = R1; a.XH = XH; a.RSTD = RSTD; a.dbg = nullptr;
  const int dk = d / heads;
  if (d == 128 && dk == 128) return train ? launch_one<128, 128, true>(a, st) : launch_one<128, 128, false>(a, st);
  if (d == 128 && dk == 64) return train ? launch_one<128, 64, true>(a, st) : launch_one<128, 64, false>(a, st);
  if (d == 64 && dk == 64) return train ? launch_one<64, 64, true>(a, st) : launch_one<64, 64, false>(a, st);
  return train ? launch_one<64, 32, true>(a, st) : launch_one<64, 32, false>(a, st);
}
