// The WHOLE tied tower of IntEL.predict_ensemble (models/IntEL/IntEL.py:182-188 / 191-197) at the reference's OWN widths -- 32-wide
// towers (its defaults IntEL.py:20-26 and its published runs script/IntEL.sh:15,21: --i_emb_size 16 --im_emb_size 16 --s_emb_size 32),
// 1-2 heads, any number of tied layers, lists of up to 128 candidates (training: up to 96) -- as ONE kernel per direction:
//
//     repeat n times with the same weights:  res = h;  h = MHA(h, h, h)  (no mask, no output projection: modules/layers.py:31-60)
//                                             h = W1 h + b1;  h = W2 relu(h) + b2;  h = LayerNorm(h + res)
//
// At these widths nothing is HBM- or MFMA-bound: a tower layer is 0.8 MFLOP per session, and the kernel-per-op pipeline spent its
// time in ~38 dependent launches per tower and training step (9 us each whatever they compute).  Here a workgroup owns one session,
// wave w owns the 16 rows 16w .. 16w+15 of its list, and everything row-wise stays in REGISTERS from the layer input to the LayerNorm:
// an MFMA accumulator of Y^T = W X^T puts row i of the tile on lane (i, j) with four consecutive columns in its registers, which is
// exactly the B operand of the next product -- Q feeds S^T = K Q^T, P feeds O^T = V^T P^T, the attention output feeds W1, relu feeds
// W2.  Only what other waves read goes through LDS: the K / V rows (forward), plus the Q and dO rows and two row statistics (backward).
// All products are exact fp32 MFMAs (v_mfma_f32_16x16x4_f32): 150 of them per wave and layer, nothing to split.
//
// Backward = torch autograd of the same lines (helpers/BaseRunner.py:288), hand-derived, in one kernel for ALL layers: the forward
// of every layer is recomputed from the tower input (no activation stash at all: the training forward is the inference forward),
// the attention backward runs as a query-tile pass (dQ) and a key-tile pass (dK, dV) so that no gradient is summed across waves,
// and the five d x d weight gradients + bias / LayerNorm gradients are accumulated in the workgroup's registers over its layers and
// sessions (the reduction over rows is an MFMA with the 16 rows as its k index; the operands take one trip through a wave-private
// LDS tile to get there) and leave ONCE per workgroup as a slab for the batched, fixed-order slab reduction (kernels.h: ReduceQueue).
#include <stdio.h>
#include <stdlib.h>

#include "kernels.h"

namespace {

constexpr int D = 32;
constexpr int LDW = 36;       // weight row pitch in LDS (floats)
constexpr int LDR = 36;       // activation row pitch in LDS (floats)
constexpr int WMAT = 32 * LDW;
constexpr float LOG2E = 1.4426950408889634f;

struct Tw32Params {
  const float *Wq, *Wk, *Wv, *W1, *b1, *W2, *b2, *gamma, *beta;      // raw reference layouts: W [out, in], vectors [32]
  // nn.Dropout in front of the residual add (IntEL.py:187,196), training only: keep / (1 - p) per element of the layer's [B*L, 32] output,
  // keep from `drop_ext` (0/1 floats, layer-major: parity tests pin the reference's draw) or from the counter-based generator of
  // rowops.hip's dropout_mask_kernel keyed by (seed, stream0 + layer, element) -- the same draw as the kernel-per-op path
  float drop_p;
  unsigned long long drop_seed;
  unsigned drop_stream0;
  const float* drop_ext;
  long long drop_layer_stride;      // elements between two layers' keep flags in drop_ext (= B * L * 32)
};

template <bool DROP>
__device__ __forceinline__ f32x4 drop_mask4(const Tw32Params& p, int layer, long long elem) {
  f32x4 m = {1.f, 1.f, 1.f, 1.f};
  if (DROP) {
    const float scale = 1.f / (1.f - p.drop_p);
    if (p.drop_ext) {
      const f32x4 k = *reinterpret_cast<const f32x4*>(p.drop_ext + (long long)layer * p.drop_layer_stride + elem);
#pragma unroll
      for (int r = 0; r < 4; ++r) m[r] = k[r] * scale;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        unsigned long long z = p.drop_seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)(p.drop_stream0 + layer) * 0x100000000ull + (unsigned long long)(elem + r) + 1ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
        m[r] = u >= p.drop_p ? scale : 0.f;
      }
    }
  }
  return m;
}

struct Tw32FwdArgs {
  const float* X;      // [B*L, 32] tower input rows
  float* out;          // [B*L, 32]
  int B, L, layers;
  Tw32Params p;
};

struct Tw32BwdArgs {
  const float* X;      // [B*L, 32] tower input rows
  const float* dout;   // [B*L, 32] gradient w.r.t. the tower output
  float* dX;           // [B*L, 32] gradient w.r.t. the tower input
  float* slabs;        // [grid, TW32_SLAB] per-workgroup partial parameter gradients
  int B, L, layers;
  Tw32Params p;
};

// lane (i, j) of a wave: i = row of the wave's 16-row tile, j = which four consecutive columns of every 16-column group.
// A register tile `f32x4 x[2]` holds X[row i][16 g + 4 j + r], g = 0, 1, r = 0 .. 3.

// Y^T = W X^T: y[mt][r] = sum_k W[16 mt + 4 j + r][k] x[k]   (W rows in LDS, pitch LDW)
__device__ __forceinline__ void lin32(const float* Ws, int i, int j, const f32x4 (&x)[2], f32x4 (&y)[2]) {
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(Ws + (16 * mt + i) * LDW + 16 * g + 4 * j);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = mfma16(w[s], x[g][s], acc);
    }
    y[mt] = acc;
  }
}

// dX^T = W^T dY^T: dx[mt][r] (+)= sum_n dY[n] W[n][16 mt + 4 j + r]
__device__ __forceinline__ void linT32(const float* Ws, int i, int j, const f32x4 (&dy)[2], f32x4 (&dx)[2]) {
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    f32x4 acc = dx[mt];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = mfma16(Ws[(16 * g + 4 * j + s) * LDW + 16 * mt + i], dy[g][s], acc);
    dx[mt] = acc;
  }
}

__device__ __forceinline__ float row_sum(float v) {      // over the four lanes (i, j = 0 .. 3) of a row
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ float row_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}

// softmax(Q K^T / sqrt(dk)) V for this wave's 16 queries and head `h` (column groups h * CG .. h * CG + CG - 1): S^T = K Q^T in
// accumulators (key on the accumulator row, query on the lane), base-2 softmax, P^T fed to O^T = V^T P^T straight from registers.
// lse2 = row maximum + log2(row sum) of the base-2 logits (what the backward rebuilds P from).
template <int HEADS, int NT>
__device__ __forceinline__ void attn32_fwd(const float* Ks, const float* Vs, int L, int i, int j, int h, const f32x4 (&q)[2], f32x4 (&a)[2],
                                           float& lse2) {
  constexpr int CG = 2 / HEADS;                       // 16-column groups per head
  const float sc = LOG2E / sqrtf((float)(D / HEADS));
  f32x4 p[NT];
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int gg = 0; gg < CG; ++gg) {
      const int g = h * CG + gg;
      const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + (16 * kt + i) * LDR + 16 * g + 4 * j);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = mfma16(kf[s], q[g][s], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 16 * kt + 4 * j + r;
      acc[r] = key < L ? acc[r] * sc : -INFINITY;
      mx = fmaxf(mx, acc[r]);
    }
    p[kt] = acc;
  }
  mx = row_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[kt][r] = __builtin_amdgcn_exp2f(p[kt][r] - mx);
      sum += p[kt][r];
    }
  sum = row_sum(sum);
  const float inv = 1.0f / sum;
  lse2 = mx + __builtin_amdgcn_logf(sum);             // v_log_f32 is log2
#pragma unroll
  for (int gg = 0; gg < CG; ++gg) {
    const int ct = h * CG + gg;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = mfma16(Vs[(16 * kt + 4 * j + s) * LDR + 16 * ct + i], p[kt][s] * inv, acc);
    a[ct] = acc;
  }
}

// the five 32 x 32 weights -> LDS (rows padded to LDW) and the four vectors b1 | b2 | gamma | beta behind them, all threads of the workgroup
constexpr int WS_FLOATS = 5 * WMAT + 4 * 32;
constexpr int C_B1 = 5 * WMAT, C_B2 = C_B1 + 32, C_G = C_B2 + 32, C_BE = C_G + 32;
__device__ __forceinline__ void stage_weights(const Tw32Params& p, float* Ws, int tid, int nthreads) {
  const float* src[5] = {p.Wq, p.Wk, p.Wv, p.W1, p.W2};
  for (int e = tid; e < 5 * 32 * 8; e += nthreads) {
    const int m = e / 256, rem = e - m * 256, row = rem >> 3, c4 = (rem & 7) * 4;
    *reinterpret_cast<f32x4*>(Ws + m * WMAT + row * LDW + c4) = *reinterpret_cast<const f32x4*>(src[m] + row * 32 + c4);
  }
  if (tid < 128) {
    const float* vs[4] = {p.b1, p.b2, p.gamma, p.beta};
    Ws[5 * WMAT + tid] = vs[tid >> 5][tid & 31];
  }
}
__device__ __forceinline__ f32x4 cvec(const float* Ws, int off, int g, int j) { return *reinterpret_cast<const f32x4*>(Ws + off + 16 * g + 4 * j); }

// One layer forward on the wave's row tile.  x -> x (in place).  Two workgroup barriers (K / V rows of all waves).
// KEEP: leave what the backward needs in the out-parameters (q / k / v / attention output / relu output / x-hat / rstd / lse2).
template <int HEADS, int NT, bool KEEP, bool DROP>
__device__ __forceinline__ void layer32_fwd(const Tw32Params& prm, int layer, long long erow, const float* Ws, float* Ks, float* Vs, float* Qs, int L, int row, int i, int j,
                                            f32x4 (&x)[2], f32x4 (&q)[2], f32x4 (&k)[2], f32x4 (&v)[2], f32x4 (&a)[2], f32x4 (&r1)[2],
                                            f32x4 (&xh)[2], float& rstd, float (&lse2)[HEADS]) {
  lin32(Ws + 0 * WMAT, i, j, x, q);
  lin32(Ws + 1 * WMAT, i, j, x, k);
  lin32(Ws + 2 * WMAT, i, j, x, v);
  __syncthreads();                                    // everybody is done with the previous K / V (/ Q) rows
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    *reinterpret_cast<f32x4*>(Ks + row * LDR + 16 * g + 4 * j) = k[g];
    *reinterpret_cast<f32x4*>(Vs + row * LDR + 16 * g + 4 * j) = v[g];
    if (KEEP) *reinterpret_cast<f32x4*>(Qs + row * LDR + 16 * g + 4 * j) = q[g];
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < HEADS; ++h) attn32_fwd<HEADS, NT>(Ks, Vs, L, i, j, h, q, a, lse2[h]);
  f32x4 f[2];
  lin32(Ws + 3 * WMAT, i, j, a, f);
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const f32x4 b1 = cvec(Ws, C_B1, g, j);
#pragma unroll
    for (int r = 0; r < 4; ++r) r1[g][r] = fmaxf(f[g][r] + b1[r], 0.f);
  }
  f32x4 z[2];
  lin32(Ws + 4 * WMAT, i, j, r1, z);
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const f32x4 b2 = cvec(Ws, C_B2, g, j);
    const f32x4 dm = drop_mask4<DROP>(prm, layer, erow + 16 * g + 4 * j);      // erow: element index of this row's column 0 in a layer's [B*L, 32] tensor
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      z[g][r] = (z[g][r] + b2[r]) * dm[r] + x[g][r];
      s += z[g][r];
    }
  }
  const float mean = row_sum(s) * (1.0f / D);
  float s2 = 0.f;
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      z[g][r] -= mean;
      s2 += z[g][r] * z[g][r];
    }
  rstd = 1.0f / sqrtf(row_sum(s2) * (1.0f / D) + 1e-5f);
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const f32x4 ga = cvec(Ws, C_G, g, j), be = cvec(Ws, C_BE, g, j);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xh[g][r] = z[g][r] * rstd;
      x[g][r] = xh[g][r] * ga[r] + be[r];
    }
  }
}

template <int HEADS, int NT, bool DROP>
__global__ __launch_bounds__(64 * NT) void tw32_fwd_kernel(Tw32FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;
  float* Ks = Ws + WS_FLOATS;
  float* Vs = Ks + NT * 16 * LDR;
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, j = lane >> 4;
  const int wave = tid >> 6, row = 16 * wave + i;
  stage_weights(a.p, Ws, tid, 64 * NT);      // (visible behind the first layer's barriers; its first reader is the q / k / v product: barrier below)
  __syncthreads();
  const bool rok = row < a.L;
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    const size_t base = ((size_t)b * a.L + row) * D;
    f32x4 x[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) x[g] = rok ? *reinterpret_cast<const f32x4*>(a.X + base + 16 * g + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < a.layers; ++l) {
      f32x4 q[2], k[2], v[2], at[2], r1[2], xh[2];
      float rstd, lse2[HEADS];
      layer32_fwd<HEADS, NT, false, DROP>(a.p, l, (long long)(rok ? base : 0), Ws, Ks, Vs, nullptr, a.L, row, i, j, x, q, k, v, at, r1, xh, rstd, lse2);
    }
    if (rok) {
#pragma unroll
      for (int g = 0; g < 2; ++g) *reinterpret_cast<f32x4*>(a.out + base + 16 * g + 4 * j) = x[g];
    }
  }
}

// ---- backward -------------------------------------------------------------------------------------------------------------------
// slab layout (floats): dWq | dWk | dWv | dW1 | dW2 (1024 each, [out, in]) | db1 | db2 | dgamma | dbeta (32 each)
constexpr int TW32_SLAB = 5 * 1024 + 4 * 32;

// dW[n][k] += sum over the tile's 16 rows of dY[row][n] X[row][k]: the row index is the MFMA's k index, so both operands go through a
// wave-private LDS tile (written row-major from the register layout, read column-wise).  acc[mt][nt] reg r = dW[16 mt + 4 j + r][16 nt + i].
// cs (optional): cs[mt] += this lane's share of colsum(dY) for column 16 mt + i (rows 4 j .. 4 j + 3; summed over j at the very end).
template <bool COLSUM>
__device__ __forceinline__ void wgrad32(float* T, int i, int j, const f32x4 (&dy)[2], const f32x4 (&x)[2], f32x4 (&acc)[2][2], float (&cs)[2]) {
  float ya[2][4];
#pragma unroll
  for (int g = 0; g < 2; ++g) *reinterpret_cast<f32x4*>(T + i * LDR + 16 * g + 4 * j) = dy[g];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int s = 0; s < 4; ++s) ya[mt][s] = T[(4 * j + s) * LDR + 16 * mt + i];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int g = 0; g < 2; ++g) *reinterpret_cast<f32x4*>(T + i * LDR + 16 * g + 4 * j) = x[g];
  __builtin_amdgcn_wave_barrier();
  if (COLSUM) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) cs[mt] += (ya[mt][0] + ya[mt][1]) + (ya[mt][2] + ya[mt][3]);
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    float xb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) xb[s] = T[(4 * j + s) * LDR + 16 * nt + i];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[mt][nt] = mfma16(ya[mt][s], xb[s], acc[mt][nt]);
  }
  __builtin_amdgcn_wave_barrier();
}
// cs[mt] += this lane's share of colsum(Y) (as above) for a tile that has no weight gradient of its own (LayerNorm gamma / beta)
__device__ __forceinline__ void colsum32(float* T, int i, int j, const f32x4 (&y)[2], float (&cs)[2]) {
#pragma unroll
  for (int g = 0; g < 2; ++g) *reinterpret_cast<f32x4*>(T + i * LDR + 16 * g + 4 * j) = y[g];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int s = 0; s < 4; ++s) cs[mt] += T[(4 * j + s) * LDR + 16 * mt + i];
  __builtin_amdgcn_wave_barrier();
}

template <int HEADS, int NT, bool DROP>
__global__ __launch_bounds__(64 * NT) void tw32_bwd_kernel(Tw32BwdArgs a) {
  constexpr int CG = 2 / HEADS, ROWS = NT * 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                              // 5 weights + 4 vectors
  float* Ks = Ws + WS_FLOATS;                    // [ROWS][LDR] each
  float* Vs = Ks + ROWS * LDR;
  float* Qs = Vs + ROWS * LDR;
  float* dAs = Qs + ROWS * LDR;
  float* LSEs = dAs + ROWS * LDR;                // [HEADS][ROWS] base-2 log-sum-exp of every query row
  float* DLs = LSEs + HEADS * ROWS;              // [HEADS][ROWS] delta = rowsum(dO * O) per head
  float* Ts = DLs + HEADS * ROWS;                // [NT][16][LDR] wave-private transposition tiles
  float* Xl = Ts + NT * 16 * LDR;                // [layers - 1][ROWS][32] inputs of layers 1 .. n-1 (layer 0 reads the tower input again)
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, j = lane >> 4;
  const int wave = tid >> 6, row = 16 * wave + i;
  float* T = Ts + wave * 16 * LDR;
  stage_weights(a.p, Ws, tid, 64 * NT);
  __syncthreads();
  const bool rok = row < a.L;
  const float scl = 1.0f / sqrtf((float)(D / HEADS));       // d(logit) -> d(q . k)
  const float sc2 = LOG2E * scl;
  f32x4 gW[5][2][2];
#pragma unroll
  for (int w = 0; w < 5; ++w)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) gW[w][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float gb1[2] = {0.f, 0.f}, gb2[2] = {0.f, 0.f}, gg[2] = {0.f, 0.f}, gbe[2] = {0.f, 0.f}, nocs[2] = {0.f, 0.f};

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    const size_t base = ((size_t)b * a.L + row) * D;
    f32x4 dy[2];
    {
      // ---- the inputs of every layer (forward recompute; the last layer's forward is part of its backward below)
      f32x4 x[2], q[2], k[2], v[2], at[2], r1[2], xh[2];
      float rstd, lse2[HEADS];
#pragma unroll
      for (int g = 0; g < 2; ++g) x[g] = rok ? *reinterpret_cast<const f32x4*>(a.X + base + 16 * g + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
      for (int l = 0; l + 1 < a.layers; ++l) {
        layer32_fwd<HEADS, NT, false, DROP>(a.p, l, (long long)(rok ? base : 0), Ws, Ks, Vs, nullptr, a.L, row, i, j, x, q, k, v, at, r1, xh, rstd, lse2);
#pragma unroll
        for (int g = 0; g < 2; ++g) *reinterpret_cast<f32x4*>(Xl + ((size_t)l * ROWS + row) * D + 16 * g + 4 * j) = x[g];
      }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) dy[g] = rok ? *reinterpret_cast<const f32x4*>(a.dout + base + 16 * g + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int l = a.layers - 1; l >= 0; --l) {
      // this lane's share of the layer input (re-read where needed): the tower input for layer 0 (a pad-tile row reads row 0: its
      // values only meet zero gradients), the stored output of layer l - 1 otherwise
      const float* xrow = l == 0 ? a.X + (rok ? base : (size_t)b * a.L * D) + 4 * j : Xl + ((size_t)(l - 1) * ROWS + row) * D + 4 * j;
      f32x4 q[2], at[2], dz[2], da[2];
      float lse2[HEADS];
      {
        f32x4 x[2], k[2], v[2], r1[2], xh[2];
        float rstd;
#pragma unroll
        for (int g = 0; g < 2; ++g) x[g] = (l > 0 || rok) ? *reinterpret_cast<const f32x4*>(xrow + 16 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
        layer32_fwd<HEADS, NT, true, DROP>(a.p, l, (long long)(rok ? base : 0), Ws, Ks, Vs, Qs, a.L, row, i, j, x, q, k, v, at, r1, xh, rstd, lse2);
        // ---- LayerNorm backward (rows of width 32 over the lanes (i, 0..3))
        float m1 = 0.f, m2 = 0.f;
        f32x4 t[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const f32x4 ga = cvec(Ws, C_G, g, j);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float gd = dy[g][r] * ga[r];
            m1 += gd;
            m2 += gd * xh[g][r];
            t[g][r] = dy[g][r] * xh[g][r];
            dz[g][r] = gd;
          }
        }
        colsum32(T, i, j, t, gg);
        colsum32(T, i, j, dy, gbe);
        m1 = row_sum(m1) * (1.0f / D);
        m2 = row_sum(m2) * (1.0f / D);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) dz[g][r] = (dz[g][r] - m1 - xh[g][r] * m2) * rstd;
        // ---- feed-forward backward (the dropout sits between W2 and the residual add: the W2 path sees dz * mask, the residual dz)
        f32x4 dzd[2] = {dz[0], dz[1]};
        if (DROP) {
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            const f32x4 dm = drop_mask4<DROP>(a.p, l, (long long)(rok ? base : 0) + 16 * g + 4 * j);
#pragma unroll
            for (int r = 0; r < 4; ++r) dzd[g][r] *= dm[r];
          }
        }
        wgrad32<true>(T, i, j, dzd, r1, gW[4], gb2);
        f32x4 df[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        linT32(Ws + 4 * WMAT, i, j, dzd, df);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) df[g][r] = r1[g][r] > 0.f ? df[g][r] : 0.f;
        wgrad32<true>(T, i, j, df, at, gW[3], gb1);
        da[0] = da[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        linT32(Ws + 3 * WMAT, i, j, df, da);
      }
      // ---- attention backward, pass 1: this wave's 16 QUERIES -> dQ; dO rows, delta and the log-sum-exp go to LDS for pass 2
      f32x4 dq[2];
#pragma unroll
      for (int g = 0; g < 2; ++g) *reinterpret_cast<f32x4*>(dAs + row * LDR + 16 * g + 4 * j) = da[g];
#pragma unroll
      for (int h = 0; h < HEADS; ++h) {
        float dl = 0.f;
#pragma unroll
        for (int gq = 0; gq < CG; ++gq)
#pragma unroll
          for (int r = 0; r < 4; ++r) dl += da[h * CG + gq][r] * at[h * CG + gq][r];
        dl = row_sum(dl);
        if (j == 0) {
          LSEs[h * ROWS + row] = lse2[h];
          DLs[h * ROWS + row] = dl;
        }
        f32x4 ds[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
          f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int gq = 0; gq < CG; ++gq) {
            const int g = h * CG + gq;
            const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + (16 * kt + i) * LDR + 16 * g + 4 * j);
            const f32x4 vf = *reinterpret_cast<const f32x4*>(Vs + (16 * kt + i) * LDR + 16 * g + 4 * j);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              st = mfma16(kf[s], q[g][s], st);          // S^T[key][query]
              dp = mfma16(vf[s], da[g][s], dp);         // dP^T[key][query] = sum_c V[key][c] dO[query][c]
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * j + r;
            const float p = key < a.L ? __builtin_amdgcn_exp2f(st[r] * sc2 - lse2[h]) : 0.f;
            ds[kt][r] = p * (dp[r] - dl) * scl;
          }
        }
#pragma unroll
        for (int gq = 0; gq < CG; ++gq) {
          const int ct = h * CG + gq;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = mfma16(Ks[(16 * kt + 4 * j + s) * LDR + 16 * ct + i], ds[kt][s], acc);      // dQ^T = K^T dS^T
          dq[ct] = acc;
        }
      }
      {   // q projection backward (its operands die here)
        f32x4 xin[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) xin[g] = *reinterpret_cast<const f32x4*>(xrow + 16 * g);
        wgrad32<false>(T, i, j, dq, xin, gW[0], nocs);
        linT32(Ws + 0 * WMAT, i, j, dq, dz);           // dz becomes the gradient of the layer input: residual + the three projections
      }
      __syncthreads();                                  // dO rows, delta, log-sum-exp of every query are in LDS
      // ---- pass 2: this wave's 16 KEYS -> dK, dV (S = Q K^T recomputed with the query on the accumulator row)
      f32x4 dk[2], dv[2];
      {
        f32x4 k[2], v[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          k[g] = *reinterpret_cast<const f32x4*>(Ks + row * LDR + 16 * g + 4 * j);
          v[g] = *reinterpret_cast<const f32x4*>(Vs + row * LDR + 16 * g + 4 * j);
        }
#pragma unroll
        for (int h = 0; h < HEADS; ++h) {
          f32x4 adk[CG], adv[CG];
#pragma unroll
          for (int gq = 0; gq < CG; ++gq) adk[gq] = adv[gq] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int qt = 0; qt < NT; ++qt) {
            f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int gq = 0; gq < CG; ++gq) {
              const int g = h * CG + gq;
              const f32x4 qf = *reinterpret_cast<const f32x4*>(Qs + (16 * qt + i) * LDR + 16 * g + 4 * j);
              const f32x4 of = *reinterpret_cast<const f32x4*>(dAs + (16 * qt + i) * LDR + 16 * g + 4 * j);
#pragma unroll
              for (int s = 0; s < 4; ++s) {
                st = mfma16(qf[s], k[g][s], st);          // S[query][key]
                dp = mfma16(of[s], v[g][s], dp);          // dP[query][key] = sum_c dO[query][c] V[key][c]
              }
            }
            const f32x4 ls = *reinterpret_cast<const f32x4*>(LSEs + h * ROWS + 16 * qt + 4 * j);
            const f32x4 dl = *reinterpret_cast<const f32x4*>(DLs + h * ROWS + 16 * qt + 4 * j);
            f32x4 p, ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              p[r] = rok ? __builtin_amdgcn_exp2f(st[r] * sc2 - ls[r]) : 0.f;      // a key row past the list is masked
              ds[r] = p[r] * (dp[r] - dl[r]) * scl;
            }
#pragma unroll
            for (int gq = 0; gq < CG; ++gq) {
              const int ct = h * CG + gq;
#pragma unroll
              for (int s = 0; s < 4; ++s) {
                adv[gq] = mfma16(dAs[(16 * qt + 4 * j + s) * LDR + 16 * ct + i], p[s], adv[gq]);       // dV^T = dO^T P
                adk[gq] = mfma16(Qs[(16 * qt + 4 * j + s) * LDR + 16 * ct + i], ds[s], adk[gq]);       // dK^T = Q^T dS
              }
            }
          }
#pragma unroll
          for (int gq = 0; gq < CG; ++gq) {
            dk[h * CG + gq] = adk[gq];
            dv[h * CG + gq] = adv[gq];
          }
        }
      }
      // ---- k / v projections backward + residual
      {
        f32x4 xin[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) xin[g] = *reinterpret_cast<const f32x4*>(xrow + 16 * g);
        wgrad32<false>(T, i, j, dk, xin, gW[1], nocs);
        wgrad32<false>(T, i, j, dv, xin, gW[2], nocs);
      }
      linT32(Ws + 1 * WMAT, i, j, dk, dz);
      linT32(Ws + 2 * WMAT, i, j, dv, dz);
      dy[0] = dz[0];
      dy[1] = dz[1];
      // (the next layer_fwd's first barrier keeps its K / V / Q stores behind this layer's pass-2 reads)
    }
    if (rok) {
#pragma unroll
      for (int g = 0; g < 2; ++g) *reinterpret_cast<f32x4*>(a.dX + base + 16 * g + 4 * j) = dy[g];
    }
  }
  // ---- the workgroup's parameter gradients: waves summed in wave order through LDS, one slab per workgroup
  __syncthreads();
  float* R = Ks;                                   // [TW32_SLAB] (the K / V / Q / dO rows, statistics and transposition tiles are dead)
  float* slab = a.slabs + (size_t)blockIdx.x * TW32_SLAB;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {                // vectors: the four lanes (i, 0..3) hold the shares of rows 4 j .. 4 j + 3
    gb1[mt] = row_sum(gb1[mt]);
    gb2[mt] = row_sum(gb2[mt]);
    gg[mt] = row_sum(gg[mt]);
    gbe[mt] = row_sum(gbe[mt]);
  }
  for (int w = 0; w < NT; ++w) {
    if (wave == w) {
#pragma unroll
      for (int m = 0; m < 5; ++m)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float* p = R + m * 1024 + (16 * mt + 4 * j + r) * 32 + 16 * nt + i;
              *p = (w == 0 ? 0.f : *p) + gW[m][mt][nt][r];
            }
      if (j == 0) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          float* p = R + 5 * 1024 + 16 * mt + i;
          p[0] = (w == 0 ? 0.f : p[0]) + gb1[mt];
          p[32] = (w == 0 ? 0.f : p[32]) + gb2[mt];
          p[64] = (w == 0 ? 0.f : p[64]) + gg[mt];
          p[96] = (w == 0 ? 0.f : p[96]) + gbe[mt];
        }
      }
    }
    __syncthreads();
  }
  for (int e = tid * 4; e < TW32_SLAB; e += 64 * NT * 4) *reinterpret_cast<f32x4*>(slab + e) = *reinterpret_cast<const f32x4*>(R + e);
}

// =================================================================================================================================
// BERT4Rec at the reference's DEFAULT widths (models/GeneralSeq.py:80-106 with modules/layers.py:63-88 blocks; dm = context_emb_size +
// intent_emb_size = i_emb_size + intent_emb_size = 32 with the defaults of IntEL.py:20-26 / GeneralSeq.py, 2 blocks x 2 heads hard-coded
// at IntEL.py:108-109): the whole encoder of one session history (<= 32 events) as one kernel per direction, with the machinery of the
// tower kernels above.  Differences from a tower layer: q / k / v have biases, keys are masked at the history length (GeneralSeq.py:100),
// a LayerNorm follows the attention (x = LN1(attention + x)), the blocks have their own weights (nothing tied), and the encoder's output
// is row len-1 of the last block (GeneralSeq.py:103-105) -- so the backward starts from ONE non-zero row.
// Per-block weight image in LDS: Wq | Wk | Wv | W1 | W2 (pitch LDW) | bq | bk | bv | b1 | b2 | g1 | be1 | g2 | be2.
constexpr int EB = 5 * WMAT + 9 * 32;
enum { EV_BQ = 5 * WMAT, EV_BK = EV_BQ + 32, EV_BV = EV_BK + 32, EV_B1 = EV_BV + 32, EV_B2 = EV_B1 + 32, EV_G1 = EV_B2 + 32, EV_BE1 = EV_G1 + 32,
       EV_G2 = EV_BE1 + 32, EV_BE2 = EV_G2 + 32 };
constexpr int ENC32_MAXL = 2;
constexpr int ENC32_SLAB = 5 * 1024 + 9 * 32;      // per block: dWq dWk dWv dW1 dW2 | dbq dbk dbv db1 db2 dg1 dbe1 dg2 dbe2

struct Enc32Args {
  const float* X;            // [rows, 32] input rows (position embedding already added); packed (off) or padded [B, T]
  const int* off;            // packed: first row of session b, else null
  const int* len;            // [B]
  int B, T, layers;
  const float* W[ENC32_MAXL][5];      // Wq Wk Wv W1 W2, raw [32, 32]
  const float* V[ENC32_MAXL][9];      // bq bk bv b1 b2 g1 be1 g2 be2
  float* out; int ldo;       // forward: out[b * ldo + c] = encoder output (row len-1)
  const float* dout; int ldd;// backward: gradient of that vector
  float* dX;                 // backward: [rows, 32] gradient of the input rows
  float* slabs;              // backward: [grid, layers * ENC32_SLAB]
};

__device__ __forceinline__ void enc32_stage(const Enc32Args& a, float* Ws, int tid, int nthreads) {
  for (int l = 0; l < a.layers; ++l) {
    for (int e = tid; e < 5 * 32 * 8; e += nthreads) {
      const int m = e / 256, rem = e - m * 256, row = rem >> 3, c4 = (rem & 7) * 4;
      *reinterpret_cast<f32x4*>(Ws + l * EB + m * WMAT + row * LDW + c4) = *reinterpret_cast<const f32x4*>(a.W[l][m] + row * 32 + c4);
    }
    for (int e = tid; e < 9 * 32; e += nthreads) Ws[l * EB + 5 * WMAT + e] = a.V[l][e >> 5][e & 31];
  }
}

// rows of width 32 over the lanes (i, 0..3): y = LayerNorm(z) (in place), x-hat and 1/std out
__device__ __forceinline__ void ln32(f32x4 (&z)[2], const float* Wb, int og, int ob, int j, f32x4 (&xh)[2], float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) s += z[g][r];
  const float mean = row_sum(s) * (1.0f / D);
  float s2 = 0.f;
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      z[g][r] -= mean;
      s2 += z[g][r] * z[g][r];
    }
  rstd = 1.0f / sqrtf(row_sum(s2) * (1.0f / D) + 1e-5f);
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const f32x4 ga = cvec(Wb, og, g, j), be = cvec(Wb, ob, g, j);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xh[g][r] = z[g][r] * rstd;
      z[g][r] = xh[g][r] * ga[r] + be[r];
    }
  }
}
// dz = LayerNorm backward of dy (in place): dz = (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat)) * rstd
__device__ __forceinline__ void ln32_bwd(f32x4 (&dy)[2], const float* Wb, int og, int j, const f32x4 (&xh)[2], float rstd) {
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const f32x4 ga = cvec(Wb, og, g, j);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dy[g][r] *= ga[r];
      m1 += dy[g][r];
      m2 += dy[g][r] * xh[g][r];
    }
  }
  m1 = row_sum(m1) * (1.0f / D);
  m2 = row_sum(m2) * (1.0f / D);
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) dy[g][r] = (dy[g][r] - m1 - xh[g][r] * m2) * rstd;
}

// one block forward on the wave's row tile: x -> x.  n = history length (keys >= n are masked).  Two workgroup barriers.
template <int HEADS, int NT, bool KEEP>
__device__ __forceinline__ void enc32_block_fwd(const float* Wb, float* Ks, float* Vs, float* Qs, int n, int row, int i, int j, f32x4 (&x)[2],
                                                f32x4 (&q)[2], f32x4 (&at)[2], f32x4 (&c)[2], f32x4 (&xh1)[2], float& rstd1, f32x4 (&f)[2],
                                                f32x4 (&xh2)[2], float& rstd2, float (&lse2)[HEADS]) {
  f32x4 k[2], v[2];
  lin32(Wb + 0 * WMAT, i, j, x, q);
  lin32(Wb + 1 * WMAT, i, j, x, k);
  lin32(Wb + 2 * WMAT, i, j, x, v);
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    q[g] += cvec(Wb, EV_BQ, g, j);
    k[g] += cvec(Wb, EV_BK, g, j);
    v[g] += cvec(Wb, EV_BV, g, j);
  }
  __syncthreads();
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    *reinterpret_cast<f32x4*>(Ks + row * LDR + 16 * g + 4 * j) = k[g];
    *reinterpret_cast<f32x4*>(Vs + row * LDR + 16 * g + 4 * j) = v[g];
    if (KEEP) *reinterpret_cast<f32x4*>(Qs + row * LDR + 16 * g + 4 * j) = q[g];
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < HEADS; ++h) attn32_fwd<HEADS, NT>(Ks, Vs, n, i, j, h, q, at, lse2[h]);
#pragma unroll
  for (int g = 0; g < 2; ++g) c[g] = at[g] + x[g];
  ln32(c, Wb, EV_G1, EV_BE1, j, xh1, rstd1);
  f32x4 t[2];
  lin32(Wb + 3 * WMAT, i, j, c, t);
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const f32x4 b1 = cvec(Wb, EV_B1, g, j);
#pragma unroll
    for (int r = 0; r < 4; ++r) f[g][r] = fmaxf(t[g][r] + b1[r], 0.f);
  }
  lin32(Wb + 4 * WMAT, i, j, f, x);
#pragma unroll
  for (int g = 0; g < 2; ++g) x[g] += cvec(Wb, EV_B2, g, j) + c[g];
  ln32(x, Wb, EV_G2, EV_BE2, j, xh2, rstd2);
}

template <int HEADS, int NT>
__global__ __launch_bounds__(64 * NT) void enc32_fwd_kernel(Enc32Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;
  float* Ks = Ws + a.layers * EB;
  float* Vs = Ks + NT * 16 * LDR;
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, j = lane >> 4;
  const int wave = tid >> 6, row = 16 * wave + i;
  enc32_stage(a, Ws, tid, 64 * NT);
  __syncthreads();
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    const int n = min(a.len[b], a.T);
    float* orow = a.out + (size_t)b * a.ldo;
    if (n <= 0) {                                   // (outside the reference's domain: GeneralSeq.py:48-52 gives every session one event)
      if (tid < 8) *reinterpret_cast<f32x4*>(orow + 4 * tid) = f32x4{0.f, 0.f, 0.f, 0.f};
      continue;
    }
    const size_t base = ((a.off ? (size_t)a.off[b] : (size_t)b * a.T) + row) * D;
    const bool rok = row < n;
    f32x4 x[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) x[g] = rok ? *reinterpret_cast<const f32x4*>(a.X + base + 16 * g + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < a.layers; ++l) {
      f32x4 q[2], at[2], c[2], xh1[2], f[2], xh2[2];
      float rstd1, rstd2, lse2[HEADS];
      enc32_block_fwd<HEADS, NT, false>(Ws + l * EB, Ks, Vs, nullptr, n, row, i, j, x, q, at, c, xh1, rstd1, f, xh2, rstd2, lse2);
    }
    if (row == n - 1) {
#pragma unroll
      for (int g = 0; g < 2; ++g) *reinterpret_cast<f32x4*>(orow + 16 * g + 4 * j) = x[g];
    }
  }
}

// One session per workgroup; the blocks are walked last to first and every block's parameter gradients leave as soon as the block is done
// (one slab per session and block): both blocks' accumulators at once would not fit the register file next to the data path.
template <int HEADS, int NT>
__global__ __launch_bounds__(64 * NT) void enc32_bwd_kernel(Enc32Args a) {
  constexpr int CG = 2 / HEADS, ROWS = NT * 16;
  const int LAYERS = a.layers;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;
  float* Xl = Ws + LAYERS * EB;                  // [LAYERS][ROWS][32]: the input rows of every block (live across the blocks' reductions below)
  float* Ks = Xl + LAYERS * ROWS * D;
  float* Vs = Ks + ROWS * LDR;
  float* Qs = Vs + ROWS * LDR;
  float* dAs = Qs + ROWS * LDR;
  float* LSEs = dAs + ROWS * LDR;
  float* DLs = LSEs + HEADS * ROWS;
  float* Ts = DLs + HEADS * ROWS;                // [NT][16][LDR]
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, j = lane >> 4;
  const int wave = tid >> 6, row = 16 * wave + i;
  float* T = Ts + wave * 16 * LDR;
  enc32_stage(a, Ws, tid, 64 * NT);
  __syncthreads();
  const float scl = 1.0f / sqrtf((float)(D / HEADS));
  const float sc2 = LOG2E * scl;
  f32x4 gW[5][2][2];
  float gv[9][2];                                // bq bk bv b1 b2 g1 be1 g2 be2 (column shares, summed over j at the end of a block)
  float* R = Ks;                                 // cross-wave reduction of a block's gradients (the activation tiles are dead then)

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    const int n = min(a.len[b], a.T);
    const size_t row0 = a.off ? (size_t)a.off[b] : (size_t)b * a.T;
    const size_t base = (row0 + row) * D;
    const bool rok = row < n;
    float* slab = a.slabs + (size_t)b * LAYERS * ENC32_SLAB;
    if (n <= 0) {
      if (!a.off && row < a.T) {
#pragma unroll
        for (int g = 0; g < 2; ++g) *reinterpret_cast<f32x4*>(a.dX + base + 16 * g + 4 * j) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      for (int e = tid * 4; e < LAYERS * ENC32_SLAB; e += 64 * NT * 4) *reinterpret_cast<f32x4*>(slab + e) = f32x4{0.f, 0.f, 0.f, 0.f};
      continue;
    }
    {   // forward recompute: the input rows of every block
      f32x4 x[2], q[2], at[2], c[2], xh1[2], f[2], xh2[2];
      float rstd1, rstd2, lse2[HEADS];
#pragma unroll
      for (int g = 0; g < 2; ++g) x[g] = rok ? *reinterpret_cast<const f32x4*>(a.X + base + 16 * g + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
      for (int l = 0; l < LAYERS; ++l) {
#pragma unroll
        for (int g = 0; g < 2; ++g) *reinterpret_cast<f32x4*>(Xl + ((size_t)l * ROWS + row) * D + 16 * g + 4 * j) = x[g];
        if (l + 1 < LAYERS) enc32_block_fwd<HEADS, NT, false>(Ws + l * EB, Ks, Vs, nullptr, n, row, i, j, x, q, at, c, xh1, rstd1, f, xh2, rstd2, lse2);
      }
    }
    f32x4 dy[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) dy[g] = row == n - 1 ? *reinterpret_cast<const f32x4*>(a.dout + (size_t)b * a.ldd + 16 * g + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int l = LAYERS - 1; l >= 0; --l) {
      const float* Wb = Ws + l * EB;
      const float* xrow = Xl + ((size_t)l * ROWS + row) * D + 4 * j;
      f32x4 q[2], at[2], dz1[2], da[2];
      float lse2[HEADS];
#pragma unroll
      for (int w = 0; w < 5; ++w)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int nn = 0; nn < 2; ++nn) gW[w][m][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < 9; ++w) gv[w][0] = gv[w][1] = 0.f;
      {
        f32x4 x[2], c[2], xh1[2], f[2], xh2[2];
        float rstd1, rstd2;
#pragma unroll
        for (int g = 0; g < 2; ++g) x[g] = *reinterpret_cast<const f32x4*>(xrow + 16 * g);
        enc32_block_fwd<HEADS, NT, true>(Wb, Ks, Vs, Qs, n, row, i, j, x, q, at, c, xh1, rstd1, f, xh2, rstd2, lse2);
        // LayerNorm-2 backward
        f32x4 t[2];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) t[g][r] = dy[g][r] * xh2[g][r];
        colsum32(T, i, j, t, gv[7]);
        colsum32(T, i, j, dy, gv[8]);
        ln32_bwd(dy, Wb, EV_G2, j, xh2, rstd2);             // dy = dz2
        // feed-forward backward; the residual c receives dz2 too
        wgrad32<true>(T, i, j, dy, f, gW[4], gv[4]);
        f32x4 df[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        linT32(Wb + 4 * WMAT, i, j, dy, df);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) df[g][r] = f[g][r] > 0.f ? df[g][r] : 0.f;
        wgrad32<true>(T, i, j, df, c, gW[3], gv[3]);
        linT32(Wb + 3 * WMAT, i, j, df, dy);                 // dy = dc = dz2 + df W1
        // LayerNorm-1 backward
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) t[g][r] = dy[g][r] * xh1[g][r];
        colsum32(T, i, j, t, gv[5]);
        colsum32(T, i, j, dy, gv[6]);
        ln32_bwd(dy, Wb, EV_G1, j, xh1, rstd1);              // dy = dz1 = d(attention output) = residual gradient into x
        dz1[0] = dy[0]; dz1[1] = dy[1];
        da[0] = dy[0]; da[1] = dy[1];
      }
      // attention backward, pass 1 (this wave's 16 queries -> dQ)
      f32x4 dq[2];
#pragma unroll
      for (int g = 0; g < 2; ++g) *reinterpret_cast<f32x4*>(dAs + row * LDR + 16 * g + 4 * j) = da[g];
#pragma unroll
      for (int h = 0; h < HEADS; ++h) {
        float dl = 0.f;
#pragma unroll
        for (int gq = 0; gq < CG; ++gq)
#pragma unroll
          for (int r = 0; r < 4; ++r) dl += da[h * CG + gq][r] * at[h * CG + gq][r];
        dl = row_sum(dl);
        if (j == 0) {
          LSEs[h * ROWS + row] = lse2[h];
          DLs[h * ROWS + row] = dl;
        }
        f32x4 ds[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
          f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int gq = 0; gq < CG; ++gq) {
            const int g = h * CG + gq;
            const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + (16 * kt + i) * LDR + 16 * g + 4 * j);
            const f32x4 vf = *reinterpret_cast<const f32x4*>(Vs + (16 * kt + i) * LDR + 16 * g + 4 * j);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              st = mfma16(kf[s], q[g][s], st);
              dp = mfma16(vf[s], da[g][s], dp);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * j + r;
            const float p = key < n ? __builtin_amdgcn_exp2f(st[r] * sc2 - lse2[h]) : 0.f;
            ds[kt][r] = p * (dp[r] - dl) * scl;
          }
        }
#pragma unroll
        for (int gq = 0; gq < CG; ++gq) {
          const int ct = h * CG + gq;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = mfma16(Ks[(16 * kt + 4 * j + s) * LDR + 16 * ct + i], ds[kt][s], acc);
          dq[ct] = acc;
        }
      }
      {
        f32x4 xin[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) xin[g] = *reinterpret_cast<const f32x4*>(xrow + 16 * g);
        wgrad32<true>(T, i, j, dq, xin, gW[0], gv[0]);
        linT32(Wb + 0 * WMAT, i, j, dq, dz1);
      }
      __syncthreads();
      // pass 2 (this wave's 16 keys -> dK, dV)
      f32x4 dk[2], dv[2];
      {
        f32x4 k[2], v[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          k[g] = *reinterpret_cast<const f32x4*>(Ks + row * LDR + 16 * g + 4 * j);
          v[g] = *reinterpret_cast<const f32x4*>(Vs + row * LDR + 16 * g + 4 * j);
        }
#pragma unroll
        for (int h = 0; h < HEADS; ++h) {
          f32x4 adk[CG], adv[CG];
#pragma unroll
          for (int gq = 0; gq < CG; ++gq) adk[gq] = adv[gq] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int qt = 0; qt < NT; ++qt) {
            f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int gq = 0; gq < CG; ++gq) {
              const int g = h * CG + gq;
              const f32x4 qf = *reinterpret_cast<const f32x4*>(Qs + (16 * qt + i) * LDR + 16 * g + 4 * j);
              const f32x4 of = *reinterpret_cast<const f32x4*>(dAs + (16 * qt + i) * LDR + 16 * g + 4 * j);
#pragma unroll
              for (int s = 0; s < 4; ++s) {
                st = mfma16(qf[s], k[g][s], st);
                dp = mfma16(of[s], v[g][s], dp);
              }
            }
            const f32x4 ls = *reinterpret_cast<const f32x4*>(LSEs + h * ROWS + 16 * qt + 4 * j);
            const f32x4 dl = *reinterpret_cast<const f32x4*>(DLs + h * ROWS + 16 * qt + 4 * j);
            f32x4 p, ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              p[r] = rok ? __builtin_amdgcn_exp2f(st[r] * sc2 - ls[r]) : 0.f;      // a key past the history is masked
              ds[r] = p[r] * (dp[r] - dl[r]) * scl;
            }
#pragma unroll
            for (int gq = 0; gq < CG; ++gq) {
              const int ct = h * CG + gq;
#pragma unroll
              for (int s = 0; s < 4; ++s) {
                adv[gq] = mfma16(dAs[(16 * qt + 4 * j + s) * LDR + 16 * ct + i], p[s], adv[gq]);
                adk[gq] = mfma16(Qs[(16 * qt + 4 * j + s) * LDR + 16 * ct + i], ds[s], adk[gq]);
              }
            }
          }
#pragma unroll
          for (int gq = 0; gq < CG; ++gq) {
            dk[h * CG + gq] = adk[gq];
            dv[h * CG + gq] = adv[gq];
          }
        }
      }
      {
        f32x4 xin[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) xin[g] = *reinterpret_cast<const f32x4*>(xrow + 16 * g);
        wgrad32<true>(T, i, j, dk, xin, gW[1], gv[1]);
        wgrad32<true>(T, i, j, dv, xin, gW[2], gv[2]);
      }
      linT32(Wb + 1 * WMAT, i, j, dk, dz1);
      linT32(Wb + 2 * WMAT, i, j, dv, dz1);
      dy[0] = dz1[0];
      dy[1] = dz1[1];
      // ---- this block's parameter gradients: waves summed in order through LDS (over the dead activation tiles), one slab
      __syncthreads();
#pragma unroll
      for (int w = 0; w < 9; ++w)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) gv[w][mt] = row_sum(gv[w][mt]);
      for (int w = 0; w < NT; ++w) {
        if (wave == w) {
#pragma unroll
          for (int m = 0; m < 5; ++m)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
              for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  float* p = R + m * 1024 + (16 * mt + 4 * j + r) * 32 + 16 * nt + i;
                  *p = (w == 0 ? 0.f : *p) + gW[m][mt][nt][r];
                }
          if (j == 0) {
#pragma unroll
            for (int v = 0; v < 9; ++v)
#pragma unroll
              for (int mt = 0; mt < 2; ++mt) {
                float* p = R + 5 * 1024 + v * 32 + 16 * mt + i;
                *p = (w == 0 ? 0.f : *p) + gv[v][mt];
              }
          }
        }
        __syncthreads();
      }
      for (int e = tid * 4; e < ENC32_SLAB; e += 64 * NT * 4) *reinterpret_cast<f32x4*>(slab + l * ENC32_SLAB + e) = *reinterpret_cast<const f32x4*>(R + e);
      // (the next block's forward recompute starts with a barrier before it rewrites the K / V / Q tiles this reduction sat on)
    }
    // rows past the history carry no gradient (queries of rows >= n fed nothing that reached the output); padded layout: zeros there
    if (rok || (!a.off && row < a.T)) {
#pragma unroll
      for (int g = 0; g < 2; ++g) *reinterpret_cast<f32x4*>(a.dX + base + 16 * g + 4 * j) = rok ? dy[g] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
}

size_t enc32_fwd_smem(int nt, int layers) { return sizeof(float) * ((size_t)layers * EB + 2 * (size_t)nt * 16 * LDR); }
size_t enc32_bwd_smem(int nt, int heads, int layers) {
  const size_t rows = (size_t)nt * 16;
  return sizeof(float) * ((size_t)layers * EB + 4 * rows * LDR + 2 * heads * rows + (size_t)nt * 16 * LDR + (size_t)layers * rows * D);
}

size_t fwd_smem(int nt) { return sizeof(float) * (WS_FLOATS + 2 * (size_t)nt * 16 * LDR); }
size_t bwd_smem(int nt, int heads, int layers) {
  const size_t rows = (size_t)nt * 16;
  return sizeof(float) * (WS_FLOATS + 4 * rows * LDR + 2 * heads * rows + (size_t)nt * 16 * LDR + (size_t)(layers - 1) * rows * D);
}
int tiles_for(int L) { return L <= 32 ? 2 : (L <= 64 ? 4 : (L <= 96 ? 6 : 8)); }

int on_switch() {
  static const int on = [] { const char* e = getenv("INTEL_TOWER32"); return (e && e[0] == '0') ? 0 : 1; }();
  return on;
}

}  // namespace

// INTEL_TOWER32=0: the kernel-per-op pipeline for 32-wide towers too (read once per process)
bool tower32_supported(int L, int d, int heads, int layers, int train) {
  if (!on_switch() || d != 32 || (heads != 1 && heads != 2) || L < 1 || L > 128 || layers < 1) return false;
  const int nt = tiles_for(L);
  if (train && bwd_smem(nt, heads, layers) > 160 * 1024 - 512) return false;
  if (train && nt == 8) return false;      // lists of 97 .. 128: the backward kernel's eight-wave form spills ~100 registers; kernel-per-op pipeline
  return true;
}

// small batches (<= 4 sessions per CU): one workgroup per CU (launch_tower32_bwd) on `share8` eighths of the CUs -- the two towers' backward kernels
// run side by side with the sequence encoders' kernels of the step's critical chain and must leave them room.  How much was re-measured on the round's
// final code (published hyper-parameters, 512 sessions, three runs each, sessions/s): GRU4Rec encoders 96 / 128 / 160 / 192 / 256 workgroups =
// 644 / 652 / 674 / 674 / 694 k (the recurrence kernels are 32 workgroups: every CU may carry a tower workgroup); BERT4Rec encoders (one-kernel,
// a workgroup per session) 665-764 / 711-740 / 788-790 / 721-768 / 712-734 k: five eighths.  640 ... 1024 sessions: all CUs in both (GRU4Rec +6 ... +7 %).
int tower32_grid(int B, int share8) {
  if (B > 4 * num_cus()) return 1024;
  const int g = num_cus() * (share8 < 1 ? 1 : (share8 > 8 ? 8 : share8)) / 8;
  return B < g ? B : g;
}
size_t tower32_slab_floats(int B) { return (size_t)tower32_grid(B, 8) * TW32_SLAB; }

#define TW32_DISPATCH_D(KERNEL, DROP_)                                \
  do {                                                               \
    if (heads == 1) {                                                \
      if (nt == 2) KERNEL(1, 2, DROP_);                              \
      else if (nt == 4) KERNEL(1, 4, DROP_);                         \
      else if (nt == 6) KERNEL(1, 6, DROP_);                         \
      else KERNEL(1, 8, DROP_);                                      \
    } else {                                                         \
      if (nt == 2) KERNEL(2, 2, DROP_);                              \
      else if (nt == 4) KERNEL(2, 4, DROP_);                         \
      else if (nt == 6) KERNEL(2, 6, DROP_);                         \
      else KERNEL(2, 8, DROP_);                                      \
    }                                                                \
  } while (0)
#define TW32_DISPATCH(KERNEL, drop)                                  \
  do {                                                               \
    if (drop) TW32_DISPATCH_D(KERNEL, true);                         \
    else TW32_DISPATCH_D(KERNEL, false);                             \
  } while (0)

int launch_tower32_fwd(const float* X, int B, int L, int heads, int layers, const float* Wq, const float* Wk, const float* Wv, const float* W1,
                       const float* b1, const float* W2, const float* b2, const float* gamma, const float* beta, float* out, hipStream_t st,
                       const Tower32Dropout* drop) {
  if (B <= 0) return 0;
  INTEL_CHECK_ARG(tower32_supported(L, 32, heads, layers, 0), "tower32_fwd: unsupported shape L=%d heads=%d layers=%d", L, heads, layers);
  Tw32FwdArgs a;
  a.X = X; a.out = out; a.B = B; a.L = L; a.layers = layers;
  const bool dropping = drop && drop->p > 0.f;
  a.p = Tw32Params{Wq, Wk, Wv, W1, b1, W2, b2, gamma, beta, dropping ? drop->p : 0.f, dropping ? drop->seed : 0ull, dropping ? drop->stream0 : 0u,
                   dropping ? drop->ext : nullptr, (long long)B * L * D};
  const int nt = tiles_for(L);
  const size_t smem = fwd_smem(nt);
  const int grid = B < 2048 ? B : 2048;
  const double rows = (double)B * L;
  const double flops = layers * (rows * 2.0 * D * D * 5 + 4.0 * rows * L * D);
  const double bytes = rows * D * 4 * 2;
#define FWD_K(H_, NT_, DR_)                                                                                            \
  do {                                                                                                                 \
    allow_lds((tw32_fwd_kernel<H_, NT_, DR_>), smem);                                                                  \
    LAUNCH_S(B * L, D, layers, flops, bytes, (tw32_fwd_kernel<H_, NT_, DR_>), dim3(grid), dim3(64 * NT_), smem, st, a); \
  } while (0)
  TW32_DISPATCH(FWD_K, dropping);
#undef FWD_K
  INTEL_CHECK_LAUNCH();
  return 0;
}

// grads: dWq, dWk, dWv, dW1, db1, dW2, db2, dgamma, dbeta (any may be NULL); accumulate[9]: add to / overwrite the destination.
// The partial sums go to `q`'s arena and are valid after its flush (redq_flush / redq_flush_tag).
int launch_tower32_bwd(const float* X, const float* dout, int B, int L, int heads, int layers, const float* Wq, const float* Wk, const float* Wv,
                       const float* W1, const float* b1, const float* W2, const float* b2, const float* gamma, const float* beta, float* dX,
                       float* const* grads, const int* accumulate, ReduceQueue* q, hipStream_t st, const Tower32Dropout* drop, int share8) {
  if (B <= 0) return 0;
  INTEL_CHECK_ARG(tower32_supported(L, 32, heads, layers, 1), "tower32_bwd: unsupported shape L=%d heads=%d layers=%d", L, heads, layers);
  INTEL_CHECK_ARG(q, "tower32_bwd: needs the reduce queue");
  const int grid = tower32_grid(B, share8);
  float* slabs = redq_alloc(q, (size_t)grid * TW32_SLAB);
  INTEL_CHECK_ARG(slabs, "tower32_bwd: reduction arena exhausted");
  Tw32BwdArgs a;
  a.X = X; a.dout = dout; a.dX = dX; a.slabs = slabs; a.B = B; a.L = L; a.layers = layers;
  const bool dropping = drop && drop->p > 0.f;
  a.p = Tw32Params{Wq, Wk, Wv, W1, b1, W2, b2, gamma, beta, dropping ? drop->p : 0.f, dropping ? drop->seed : 0ull, dropping ? drop->stream0 : 0u,
                   dropping ? drop->ext : nullptr, (long long)B * L * D};
  const int nt = tiles_for(L);
  size_t smem = bwd_smem(nt, heads, layers);
  // Small batches (the reference trains at 512 sessions): the step is a chain of small dependent launches on other streams (sequence
  // encoders, session head) and this kernel is OFF that chain -- but two of its workgroups per CU take all of a CU's LDS for the whole
  // kernel, and a 8-workgroup GEMM of the critical chain then waits for one of them to finish (measured: the GRU branch 90 us later).
  // Asking for more than half of the LDS keeps it to one workgroup per CU; the rest of the CU stays available.
  if (B <= 4 * num_cus() && smem < 82 * 1024) smem = 82 * 1024;
  const double rows = (double)B * L;
  const double flops = layers * 2.0 * (rows * 2.0 * D * D * 5 + 4.0 * rows * L * D);      // algorithmic: data + weight gradients (the forward recompute is overhead, not work)
  const double bytes = rows * D * 4 * 3;
#define BWD_K(H_, NT_, DR_)                                                                                            \
  do {                                                                                                                 \
    allow_lds((tw32_bwd_kernel<H_, NT_, DR_>), smem);                                                                  \
    LAUNCH_S(B * L, D, layers, flops, bytes, (tw32_bwd_kernel<H_, NT_, DR_>), dim3(grid), dim3(64 * NT_), smem, st, a); \
  } while (0)
  TW32_DISPATCH(BWD_K, dropping);
#undef BWD_K
  INTEL_CHECK_LAUNCH();
  // slab -> destination jobs: dWq dWk dWv dW1 db1 dW2 db2 dgamma dbeta
  const int off[9] = {0, 1024, 2048, 3072, 5 * 1024, 4096, 5 * 1024 + 32, 5 * 1024 + 64, 5 * 1024 + 96};
  const int isw[9] = {1, 1, 1, 1, 0, 1, 0, 0, 0};
  for (int p = 0; p < 9; ++p) {
    if (!grads[p]) continue;
    if (isw[p]) redq_push(q, slabs + off[p], TW32_SLAB, grid, 32, 32, grads[p], 32, accumulate[p]);
    else redq_push(q, slabs + off[p], TW32_SLAB, grid, 1, 32, grads[p], 32, accumulate[p]);
  }
  return 0;
}

// ---- BERT4Rec at width 32 (enc32_* kernels above) ------------------------------------------------------------------------------
constexpr int ENC32_MAXB = 1024;      // one slab per session and block: larger batches take the kernel-per-op encoder.  Measured on the published IntEL-MSE
                                       // configuration (ms per step with / without this kernel family): 256 sessions 0.72 / 1.03, 512: 0.72 / 0.79, 768: 0.77 / 0.84, 1024: 0.89 / 0.91,
                                       // 1536: 1.24 / 0.97, 2048: 1.45 / 1.10 -- one workgroup and one 21.6 KB slab per session stop paying where the step stops being launch-bound
bool enc32_supported(int T, int dm, int heads, int layers, int train) {
  static const int on = [] { const char* e = getenv("INTEL_ENC32"); return (e && e[0] == '0') ? 0 : 1; }();
  (void)train;
  return on && dm == 32 && (heads == 1 || heads == 2) && T >= 1 && T <= 32 && layers >= 1 && layers <= ENC32_MAXL;
}
bool enc32_batch_ok(int B, int train) { return !train || B <= ENC32_MAXB;
}
size_t enc32_slab_floats(int B, int layers) { return (size_t)(B < ENC32_MAXB ? B : ENC32_MAXB) * layers * ENC32_SLAB; }

static void enc32_fill(Enc32Args& a, const Enc32Block* blk, int layers) {
  for (int l = 0; l < layers; ++l) {
    const Enc32Block& k = blk[l];
    const float* W[5] = {k.Wq, k.Wk, k.Wv, k.W1, k.W2};
    const float* V[9] = {k.bq, k.bk, k.bv, k.b1, k.b2, k.g1, k.be1, k.g2, k.be2};
    for (int i = 0; i < 5; ++i) a.W[l][i] = W[i];
    for (int i = 0; i < 9; ++i) a.V[l][i] = V[i];
  }
}

int launch_enc32_fwd(const float* X, const int* off, const int* len, int B, int T, int heads, int layers, const Enc32Block* blk, float* out, int ldo,
                     hipStream_t st) {
  if (B <= 0) return 0;
  INTEL_CHECK_ARG(enc32_supported(T, 32, heads, layers, 0), "enc32_fwd: unsupported shape T=%d heads=%d layers=%d", T, heads, layers);
  Enc32Args a{};
  a.X = X; a.off = off; a.len = len; a.B = B; a.T = T; a.layers = layers; a.out = out; a.ldo = ldo;
  enc32_fill(a, blk, layers);
  const int nt = T <= 16 ? 1 : 2;
  const size_t smem = enc32_fwd_smem(nt, layers);
  const int grid = B < 2048 ? B : 2048;
  const double rows = (double)B * T * 0.5;
  const double flops = layers * (rows * 2.0 * D * D * 5 + 4.0 * rows * T * D * 0.5);
  const double bytes = rows * D * 4 + (double)B * D * 4;
#define E32F(H_, NT_)                                                                                                \
  do {                                                                                                               \
    allow_lds((enc32_fwd_kernel<H_, NT_>), smem);                                                                    \
    LAUNCH_S(B * T, D, layers, flops, bytes, (enc32_fwd_kernel<H_, NT_>), dim3(grid), dim3(64 * NT_), smem, st, a);  \
  } while (0)
  if (heads == 1) { if (nt == 1) E32F(1, 1); else E32F(1, 2); }
  else { if (nt == 1) E32F(2, 1); else E32F(2, 2); }
#undef E32F
  INTEL_CHECK_LAUNCH();
  return 0;
}

// grads[l][14]: dWq dbq dWk dbk dWv dbv dg1 dbe1 dW1 db1 dW2 db2 dg2 dbe2 (the order of the INTEL_ENC_* slots; NULL = not wanted);
// accumulate likewise.  The partial sums join the reduce queue (valid after its flush).
int launch_enc32_bwd(const float* X, const int* off, const int* len, int B, int T, int heads, int layers, const Enc32Block* blk, const float* dout,
                     int ldd, float* dX, float* const (*grads)[14], const int (*accumulate)[14], ReduceQueue* q, hipStream_t st) {
  if (B <= 0) return 0;
  INTEL_CHECK_ARG(enc32_supported(T, 32, heads, layers, 1), "enc32_bwd: unsupported shape T=%d heads=%d layers=%d", T, heads, layers);
  INTEL_CHECK_ARG(q, "enc32_bwd: needs the reduce queue");
  INTEL_CHECK_ARG(B <= ENC32_MAXB, "enc32_bwd: batch %d > %d", B, ENC32_MAXB);
  const int grid = B;
  float* slabs = redq_alloc(q, (size_t)grid * layers * ENC32_SLAB);
  INTEL_CHECK_ARG(slabs, "enc32_bwd: reduction arena exhausted");
  Enc32Args a{};
  a.X = X; a.off = off; a.len = len; a.B = B; a.T = T; a.layers = layers; a.dout = dout; a.ldd = ldd; a.dX = dX; a.slabs = slabs;
  enc32_fill(a, blk, layers);
  const int nt = T <= 16 ? 1 : 2;
  size_t smem = enc32_bwd_smem(nt, heads, layers);
  const size_t need = sizeof(float) * ((size_t)layers * EB + (size_t)layers * nt * 16 * D + ENC32_SLAB);      // the cross-wave reduction reuses the activation tiles (not the blocks' input rows)
  if (smem < need) smem = need;
  const double rows = (double)B * T * 0.5;
  const double flops = layers * 2.0 * (rows * 2.0 * D * D * 5 + 4.0 * rows * T * D * 0.5);
  const double bytes = rows * D * 4 * 2 + (double)B * D * 4;
#define E32B(H_, NT_)                                                                                                  \
  do {                                                                                                                 \
    allow_lds((enc32_bwd_kernel<H_, NT_>), smem);                                                                      \
    LAUNCH_S(B * T, D, layers, flops, bytes, (enc32_bwd_kernel<H_, NT_>), dim3(grid), dim3(64 * NT_), smem, st, a);    \
  } while (0)
  if (heads == 1) { if (nt == 1) E32B(1, 1); else E32B(1, 2); }
  else { if (nt == 1) E32B(2, 1); else E32B(2, 2); }
#undef E32B
  INTEL_CHECK_LAUNCH();
  // slab -> destination: slab order dWq dWk dWv dW1 dW2 | dbq dbk dbv db1 db2 dg1 dbe1 dg2 dbe2
  static const int slot_of_w[5] = {0, 2, 4, 8, 10};                                   // index into the 14-entry INTEL_ENC_* order
  static const int slot_of_v[9] = {1, 3, 5, 9, 11, 6, 7, 12, 13};
  const size_t stride = (size_t)layers * ENC32_SLAB;
  for (int l = 0; l < layers; ++l) {
    const float* sl = slabs + (size_t)l * ENC32_SLAB;
    for (int w = 0; w < 5; ++w)
      if (grads[l][slot_of_w[w]]) redq_push(q, sl + w * 1024, stride, grid, 32, 32, grads[l][slot_of_w[w]], 32, accumulate[l][slot_of_w[w]]);
    for (int v = 0; v < 9; ++v)
      if (grads[l][slot_of_v[v]]) redq_push(q, sl + 5 * 1024 + v * 32, stride, grid, 1, 32, grads[l][slot_of_v[v]], 32, accumulate[l][slot_of_v[v]]);
  }
  return 0;
}
