// The backward of one nn.Linear of the towers (models/IntEL/IntEL.py:182-197 under torch autograd) in ONE pass over its rows:
//
//     dXout = (dY W) [* (X > 0)]          data gradient (optionally through the relu whose OUTPUT X is: layers W2 <- relu <- W1)
//     dW   += dY^T X,  db += colsum(dY)    weight gradient
//
// The kernel-per-op backward reads dY twice (gemm_rows_b3 for the data gradient, wgrad_b3 for the weight gradient), splits it into bf16 planes
// twice and reads the relu output a second time as the mask.  Here a tile of dY is staged ONCE as a row-major three-plane image in LDS and serves
// both products: row reads (ds_read_b128 along the reduction index = dY's columns) feed the data gradient, transposing reads (ds_read_b64_tr_b16:
// reduction index = the rows) feed the weight gradient; the X tile's image is read transposed only (and its high plane as the relu mask).
// Only ONE weight's accumulator (d x d floats over the workgroup = 32 registers per lane at d = 128) is resident next to the weight's own
// stationary fragments (48 registers), so -- unlike the one-kernel layer backward of tower_bwd.hip -- nothing spills with six-plane operands.
//
// Pipeline: persistent workgroups of 8 waves over tiles of TR rows; two LDS stages.  In iteration i a wave (1) splits the rows of tile i + 1 --
// requested from HBM one iteration earlier and sitting in registers -- into the other stage, (2) requests tile i + 2, (3) runs the MFMAs of tile i:
// one LDS-only barrier per tile, the global loads have a whole iteration to land, and with two waves per SIMD the split of one wave overlaps
// the products of the other.
// Arithmetic: fp32 accuracy, six bf16 plane products per MFMA block (planes.h); the relu mask is the sign of the high plane (> 0 exactly where X > 0).
#include <stdio.h>
#include <stdlib.h>

#include "kernels.h"
#include "planes.h"

namespace {

using namespace planes;

struct PairArgs {
  const float* dY;      // [M, D] (row stride ldy)
  const float* X;       // [M, D] (row stride ldx)
  const uint4* W;       // three-plane image (launch_pack_b3) of the packed TRANSPOSED weight: dXout = dY W
  float* out;           // [M, D] (row stride ldo)
  float* slabs;         // per workgroup: dW [D, D] | db [D]
  int M, ldy, ldx, ldo;
  int ntiles;
};

template <int D>
struct PairCfg {
  static constexpr int NW = 8, NT = 512;
  static constexpr int TR = D == 128 ? 32 : 64;         // rows per tile
  static constexpr int KB = D / 32;                      // 32-deep k-blocks of the data gradient
  static constexpr int KBT = 4;                          // k-blocks per column tile in the image (K padded to 128)
  static constexpr int CTW = D / 16;                     // 16-column tiles
  static constexpr int RS = NW / CTW;                    // row splits of the data gradient over the waves
  static constexpr int RT = (TR / 16) / RS;              // its row tiles per wave
  static constexpr int LDP = D + 16, PLANE = TR * LDP;   // bf16 row pitch / plane elements (72 / 40 dwords: the b128 row reads AND the transposing reads in the P8 row order are conflict-free)
  static constexpr size_t IMG = (size_t)3 * PLANE * 2;
  static constexpr size_t STAGE = 2 * IMG;               // P(dY) | P(X)
  static constexpr size_t SMEM = 2 * STAGE;
  // weight gradient: the CTW x CTW tiles of dW over the 8 waves as WNT x WKT blocks
  static constexpr int WNT = D == 128 ? 2 : 1, WKT = D == 128 ? 4 : 2;
  static constexpr int WPG = CTW / WKT;                  // waves side by side along dW's columns
  static constexpr int NJ = TR * (D / 4) / NT;           // float4 per thread per tile
  static constexpr size_t SLAB = (size_t)D * (D + 1);
  static_assert(RT >= 1 && RT * RS * 16 == TR, "row tiles");
  static_assert((NW / WPG) * WNT == CTW, "dW tiles");
};

// ABL (debug builds, tools/pair_bench.py with INTEL_PAIR_ABL): 1 = no data-gradient products, 2 = no weight-gradient products, 4 = no output store,
// 8 = no tile requests after the prologue, 16 = no split / plane stores after the prologue
template <int D, bool MASK, int ABL = 0>
__global__ __launch_bounds__(512, 2) void linear_bwd_pair_kernel(PairArgs a) {
  using C = PairCfg<D>;
  constexpr int NT = C::NT, TR = C::TR, KB = C::KB, KBT = C::KBT, CTW = C::CTW, RT = C::RT, LDP = C::LDP, PLANE = C::PLANE, NJ = C::NJ;
  constexpr int WNT = C::WNT, WKT = C::WKT, WPG = C::WPG;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane >> 4, p = lane & 15;
  // data gradient: wave = column tile ct, row tiles rt0 .. rt0 + RT - 1
  const int ct = wave % CTW, rt0 = (wave / CTW) * RT;
  const int col = ct * 16 + 4 * j;
  // weight gradient: wave = rows (nt0 .. nt0 + WNT - 1) * 16 of dW, column tiles kt0 .. kt0 + WKT - 1
  const int nt0 = (wave / WPG) * WNT, kt0 = (wave % WPG) * WKT;
  const int G = gridDim.x;

  // the transposed weight's fragments of column tile ct stay in registers for the whole sweep
  uint4 wf[KB][3];
  {
    const uint4* img = a.W + ((size_t)ct * KBT * 3) * 64 + lane;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int q = 0; q < 3; ++q) wf[kb][q] = img[(kb * 3 + q) * 64];
  }
  f32x4 accW[WNT][WKT];
#pragma unroll
  for (int n = 0; n < WNT; ++n)
#pragma unroll
    for (int k = 0; k < WKT; ++k) accW[n][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbp[WNT];
#pragma unroll
  for (int n = 0; n < WNT; ++n) dbp[n] = 0.f;

  // tile staging: float4 #i of a TR-row tile = (row i / (D/4), column 4 * (i % (D/4))); rows past M are clamped for the load and zeroed when split
  // (the loop body is branch-free: past the last tile it re-stages zeros / re-reads the last rows, so that the compiler can interleave the split's
  // VALU work and the requests with the matrix products of one basic block)
  f32x4 vy[NJ], vx[NJ];
  auto load_tile = [&](int t) {
    const int tt = min(t, a.ntiles - 1);
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      const int i = tid + NT * jj;
      const int tr = i / (D / 4), tc = (i - tr * (D / 4)) * 4;
      const size_t row = (size_t)min(tt * TR + tr, a.M - 1);
      vy[jj] = *reinterpret_cast<const f32x4*>(a.dY + row * a.ldy + tc);
      vx[jj] = *reinterpret_cast<const f32x4*>(a.X + row * a.ldx + tc);
    }
  };
  // the rows in vy / vx (tile t) -> plane registers; store_planes writes them to a stage
  bf16x4 py[NJ][3], px[NJ][3];
  auto split_tile = [&](int t) {
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      const int i = tid + NT * jj;
      const int tr = i / (D / 4);
      const bool live = t * TR + tr < a.M;      // (t >= ntiles: rows past M as well)
      const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
      split4(live ? vy[jj] : z, py[jj][0], py[jj][1], py[jj][2]);
      split4(live ? vx[jj] : z, px[jj][0], px[jj][1], px[jj][2]);
    }
  };
  auto store_planes = [&](int stage) {
    __bf16* pY = reinterpret_cast<__bf16*>(smem_raw + (size_t)stage * C::STAGE);
    __bf16* pX = reinterpret_cast<__bf16*>(smem_raw + (size_t)stage * C::STAGE + C::IMG);
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      const int i = tid + NT * jj;
      const int tr = i / (D / 4), tc = (i - tr * (D / 4)) * 4;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        *reinterpret_cast<bf16x4*>(pY + q * PLANE + tr * LDP + tc) = py[jj][q];
        *reinterpret_cast<bf16x4*>(pX + q * PLANE + tr * LDP + tc) = px[jj][q];
      }
    }
  };

  int t = blockIdx.x;
  load_tile(t);
  split_tile(t);
  store_planes(0);
  load_tile(t + G);
  lds_barrier();
  for (int it = 0; t < a.ntiles; t += G, ++it) {
    const int cur = it & 1;
    // (1) the next tile's rows (requested one iteration ago) -> plane registers; they go to the other stage (its readers finished before the last
    // barrier) behind the data gradient's products; (2) the tile after it leaves HBM
    if (!(ABL & 16)) split_tile(t + G);
    if (!(ABL & 8)) load_tile(t + 2 * G);
    __builtin_amdgcn_sched_barrier(0);      // (pins the requests at the top of the iteration: the scheduler otherwise sinks them behind the products, to the end of the loop body)
    const __bf16* pY = reinterpret_cast<const __bf16*>(smem_raw + (size_t)cur * C::STAGE);
    const __bf16* pX = reinterpret_cast<const __bf16*>(smem_raw + (size_t)cur * C::STAGE + C::IMG);
    // (3a) data gradient: acc[rt] (lane (p, j): row (rt0 + rt) * 16 + p, columns col .. col + 3) = dY[row][:] . W[:][column]; the fragments of
    // k-block kb + 1 are requested before the products of kb, the row tiles' accumulator chains alternate
    {
      f32x4 acc[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const __bf16* frag = pY + (rt0 * 16 + p) * LDP + 8 * j;
      bf16x8 f[2][RT][3];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int q = 0; q < 3; ++q) f[0][rt][q] = *reinterpret_cast<const bf16x8*>(frag + rt * 16 * LDP + q * PLANE);
#pragma unroll
      for (int kb = 0; kb < ((ABL & 1) ? 0 : KB); ++kb) {
        if (kb + 1 < KB) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int q = 0; q < 3; ++q) f[(kb + 1) & 1][rt][q] = *reinterpret_cast<const bf16x8*>(frag + rt * 16 * LDP + (kb + 1) * 32 + q * PLANE);
        }
        const bf16x8 wh = __builtin_bit_cast(bf16x8, wf[kb][0]), wm = __builtin_bit_cast(bf16x8, wf[kb][1]), wl = __builtin_bit_cast(bf16x8, wf[kb][2]);
        auto& g = f[kb & 1];
        // six plane products, smallest first (planes::mma), one row tile after the other per product
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, g[rt][1], acc[rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, g[rt][2], acc[rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, g[rt][0], acc[rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, g[rt][1], acc[rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, g[rt][0], acc[rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, g[rt][0], acc[rt], 0, 0, 0);
      }
      if (!(ABL & 16)) store_planes(cur ^ 1);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int lr = (rt0 + rt) * 16 + p;
        f32x4 x = acc[rt];
        if (MASK) {
          const bf16x4 hv = *reinterpret_cast<const bf16x4*>(pX + lr * LDP + col);      // high plane of the relu output: > 0 exactly where it is
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] = (float)hv[r] > 0.f ? x[r] : 0.f;
        }
        const long long row = (long long)t * TR + lr;
        if (row < a.M && (!(ABL & 4) || x[0] == 123.456f)) *reinterpret_cast<f32x4*>(a.out + (size_t)row * a.ldo + col) = x;
      }
    }
    // (3b) weight gradient: accW[n][k] (lane (p, j): dW[(nt0 + n) * 16 + 4j + r][(kt0 + k) * 16 + p]) += sum over the tile's rows of dY[row][.] X[row][.]
#pragma unroll
    for (int kb = 0; kb < ((ABL & 2) ? 0 : TR / 32); ++kb) {
      bf16x8 y[WNT][3];
#pragma unroll
      for (int n = 0; n < WNT; ++n) {
#pragma unroll
        for (int q = 0; q < 3; ++q) y[n][q] = tr_frag<LDP, true>(pY + q * PLANE, kb, nt0 + n, p, j);
      }
      bf16x8 x[2][3];
#pragma unroll
      for (int q = 0; q < 3; ++q) x[0][q] = tr_frag<LDP, true>(pX + q * PLANE, kb, kt0, p, j);
      if (kt0 == 0) {
#pragma unroll
        for (int n = 0; n < WNT; ++n) dbp[n] += (sum8(y[n][2]) + sum8(y[n][1])) + sum8(y[n][0]);
      }
#pragma unroll
      for (int k = 0; k < WKT; ++k) {
        if (k + 1 < WKT) {
#pragma unroll
          for (int q = 0; q < 3; ++q) x[(k + 1) & 1][q] = tr_frag<LDP, true>(pX + q * PLANE, kb, kt0 + k + 1, p, j);
        }
        auto& g = x[k & 1];
#pragma unroll
        for (int n = 0; n < WNT; ++n) accW[n][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y[n][1], g[1], accW[n][k], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < WNT; ++n) accW[n][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y[n][0], g[2], accW[n][k], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < WNT; ++n) accW[n][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y[n][2], g[0], accW[n][k], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < WNT; ++n) accW[n][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y[n][0], g[1], accW[n][k], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < WNT; ++n) accW[n][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y[n][1], g[0], accW[n][k], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < WNT; ++n) accW[n][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y[n][0], g[0], accW[n][k], 0, 0, 0);
      }
    }
    lds_barrier();
  }
  // ---- the workgroup's slab: dW [D, D] | db [D]
  float* slab = a.slabs + (size_t)blockIdx.x * C::SLAB;
#pragma unroll
  for (int n = 0; n < WNT; ++n) {
#pragma unroll
    for (int k = 0; k < WKT; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(size_t)((nt0 + n) * 16 + 4 * j + r) * D + (kt0 + k) * 16 + p] = accW[n][k][r];
    const float s = gsum16(dbp[n]);
    if (kt0 == 0 && j == 0) slab[(size_t)D * D + (nt0 + n) * 16 + p] = s;
  }
}

// ---- the fused q/k/v projection's backward in one pass: dX = dQKV [Wq; Wk; Wv] (+ residual), dWq/k/v += dQKV^T X (+ their bias gradients) ----
// The same idea one size up: the gradient rows are NB * D wide (NB = 3: [dQ | dK | dV]; NB = 2: the pruned last encoder block's [dK | dV]).  The kernel-per-op
// backward read them twice (wgrad_b3 over three column blocks, re-reading and re-splitting X for each; gemm_rows_b3k for dX) and split them twice.  Here a
// TR-row tile of dQKV (76.8 KB as three planes at D = 128) and of X sit in LDS ONCE: the data gradient sweeps the 12 k-blocks with the transposed weight's
// fragments streamed from the pre-split image in L2 three k-blocks ahead (144 registers of stationary weights do not fit; ONE block ahead left every MFMA group waiting for its fragment: 426 -> 280 us at D = 128), the weight gradient keeps all NB * D x D
// accumulators in registers (96 per lane at D = 128: 3 x 8 tiles per wave), the residual rows are requested at the top of the tile.  One LDS stage
// (104 - 111 KB): the next tile's rows travel in registers under the products, two LDS-only barriers per tile.
struct QkvArgs {
  const float* dY;      // [M, NB * D]
  const float* X;       // [M, D]
  const float* res;     // [M, D] or NULL
  const uint4* W;       // three-plane image (launch_pack_b3) of the packed stacked transposed weights: k extent NB * D, n extent D
  float* out;           // [M, D]
  float* slabs;         // per workgroup: dW [NB * D, D] | db [NB * D]
  int M, ldy, ldx, ldr, ldo;
  int ntiles;
};

template <int D, int NB>
struct QkvCfg {
  static constexpr int NW = 8, NT = 512;
  static constexpr int N = NB * D;
  static constexpr int TR = D == 128 ? 32 : 64;
  static constexpr int KB = N / 32;                        // k-blocks of the data gradient
  static constexpr int KBT = 4 * ((N + 127) / 128);        // k-blocks per column tile in the image
  static constexpr int CTW = D / 16, RS = NW / CTW, RT = (TR / 16) / RS;
  static constexpr int LDY = N + 16, PLY = TR * LDY;       // pitches = 8 (mod 64) dwords at D = 128, 40 at D = 64: row reads and P8 transposing reads conflict-free
  static constexpr int LDX = D + 16, PLX = TR * LDX;
  static constexpr size_t SMEM = (size_t)3 * PLY * 2 + (size_t)3 * PLX * 2;
  static constexpr int WKT = D == 128 ? 8 : 2, WPG = CTW / WKT;        // weight gradient: column tiles per wave, waves side by side
  static constexpr int WNT = (N / 16) / (NW / WPG);                    // ... row tiles (of dW) per wave
  static constexpr int NJY = TR * (N / 4) / NT, NJX = TR * (D / 4) / NT;
  static constexpr size_t SLAB = (size_t)N * (D + 1);
  static_assert(RT * RS * 16 == TR && WNT * (NW / WPG) * 16 == N && WKT * WPG == CTW, "tiling");
  static_assert(NJY * NT == TR * (N / 4) && NJX * NT == TR * (D / 4), "staging");
};

template <int D, int NB, bool RES>
__global__ __launch_bounds__(512, 2) void linear_bwd_qkv_kernel(QkvArgs a) {
  using C = QkvCfg<D, NB>;
  constexpr int NT = C::NT, N = C::N, TR = C::TR, KB = C::KB, KBT = C::KBT, CTW = C::CTW, RT = C::RT, LDY = C::LDY, PLY = C::PLY, LDX = C::LDX, PLX = C::PLX;
  constexpr int WNT = C::WNT, WKT = C::WKT, WPG = C::WPG, NJY = C::NJY, NJX = C::NJX;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* pY = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* pX = pY + 3 * PLY;
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int ct = wave % CTW, rt0 = (wave / CTW) * RT;
  const int nt0 = (wave / WPG) * WNT, kt0 = (wave % WPG) * WKT;
  const int G = gridDim.x;
  // Every per-lane address of a tile (8 global rows, 8 LDS staging slots, the fragment bases of both images, the weight image) is loop-invariant: hoisted,
  // they are ~40 registers next to 96 accumulators and the compiler spills them (42 scratch stores in the prologue, 38 reloads per tile).  The thread id is
  // laundered per tile instead, so each address is a couple of VALU instructions next to its use (the trick of tower_bwd.hip).
  int tid = tid0, lane = tid0 & 63, j = lane >> 4, p = lane & 15, col = ct * 16 + 4 * j;
  const uint4* img = a.W + ((size_t)ct * KBT * 3) * 64 + lane;
  auto relaunder = [&]() {
    tid = tid0;
    asm volatile("" : "+v"(tid));
    lane = tid & 63;
    j = lane >> 4;
    p = lane & 15;
    col = ct * 16 + 4 * j;
    img = a.W + ((size_t)ct * KBT * 3) * 64 + lane;
  };

  f32x4 accW[WNT][WKT];
#pragma unroll
  for (int n = 0; n < WNT; ++n)
#pragma unroll
    for (int k = 0; k < WKT; ++k) accW[n][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbp[WNT];
#pragma unroll
  for (int n = 0; n < WNT; ++n) dbp[n] = 0.f;

  f32x4 vy[NJY], vx[NJX], rv[RT];      // rv: the residual rows of the tile's output, in the data gradient's accumulator layout
  auto load_tile = [&](int t) {
    const int tt = min(t, a.ntiles - 1);
    if (RES) {      // (with the tile: requested inside the data gradient, these HBM loads would sit in front of its L2 weight loads -- results return in issue order)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const long long row = min((long long)tt * TR + (rt0 + rt) * 16 + p, (long long)a.M - 1);
        rv[rt] = *reinterpret_cast<const f32x4*>(a.res + (size_t)row * a.ldr + col);
      }
    }
#pragma unroll
    for (int jj = 0; jj < NJY; ++jj) {
      const int i = tid + NT * jj;
      const int tr = i / (N / 4), tc = (i - tr * (N / 4)) * 4;
      vy[jj] = *reinterpret_cast<const f32x4*>(a.dY + (size_t)min(tt * TR + tr, a.M - 1) * a.ldy + tc);
    }
#pragma unroll
    for (int jj = 0; jj < NJX; ++jj) {
      const int i = tid + NT * jj;
      const int tr = i / (D / 4), tc = (i - tr * (D / 4)) * 4;
      vx[jj] = *reinterpret_cast<const f32x4*>(a.X + (size_t)min(tt * TR + tr, a.M - 1) * a.ldx + tc);
    }
  };
  auto store_tile = [&](int t) {
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < NJY; ++jj) {
      const int i = tid + NT * jj;
      const int tr = i / (N / 4), tc = (i - tr * (N / 4)) * 4;
      store4<3, PLY>(pY + tr * LDY + tc, t * TR + tr < a.M ? vy[jj] : z);
    }
#pragma unroll
    for (int jj = 0; jj < NJX; ++jj) {
      const int i = tid + NT * jj;
      const int tr = i / (D / 4), tc = (i - tr * (D / 4)) * 4;
      store4<3, PLX>(pX + tr * LDX + tc, t * TR + tr < a.M ? vx[jj] : z);
    }
  };

  int t = blockIdx.x;
  load_tile(t);
  for (; t < a.ntiles; t += G) {
    relaunder();
    store_tile(t);
    lds_barrier();
    f32x4 rcur[RT];      // (this tile's residual rows: the registers are reloaded for the next tile below)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) rcur[rt] = RES ? rv[rt] : f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- data gradient: acc[rt] (lane (p, j): row (rt0 + rt) * 16 + p, columns col .. col + 3) = dY[row][0 .. N) . W[.][column]
    {
      f32x4 acc[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const __bf16* frag = pY + (rt0 * 16 + p) * LDY + 8 * j;
      // the transposed weight's fragments come from L2 (500 - 700 cycles) and a k-block's twelve products take 200: they are requested LA k-blocks ahead
      constexpr int LA = 3;
      uint4 bw[KB][3];
#pragma unroll
      for (int kk = 0; kk < LA && kk < KB; ++kk)
#pragma unroll
        for (int q = 0; q < 3; ++q) bw[kk][q] = img[(kk * 3 + q) * 64];
      bf16x8 f[2][RT][3];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int q = 0; q < 3; ++q) f[0][rt][q] = *reinterpret_cast<const bf16x8*>(frag + rt * 16 * LDY + q * PLY);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        if (kb + LA < KB) {
#pragma unroll
          for (int q = 0; q < 3; ++q) bw[kb + LA][q] = img[((kb + LA) * 3 + q) * 64];
          __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise sinks these requests to one k-block ahead of their use)
        }
        if (kb + 1 < KB) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int q = 0; q < 3; ++q) f[(kb + 1) & 1][rt][q] = *reinterpret_cast<const bf16x8*>(frag + rt * 16 * LDY + (kb + 1) * 32 + q * PLY);
        }
        auto& g = f[kb & 1];
        const bf16x8 wh = __builtin_bit_cast(bf16x8, bw[kb][0]), wm = __builtin_bit_cast(bf16x8, bw[kb][1]), wl = __builtin_bit_cast(bf16x8, bw[kb][2]);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, g[rt][1], acc[rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, g[rt][2], acc[rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, g[rt][0], acc[rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, g[rt][1], acc[rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, g[rt][0], acc[rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, g[rt][0], acc[rt], 0, 0, 0);
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const long long row = (long long)t * TR + (rt0 + rt) * 16 + p;
        f32x4 x = acc[rt];
        if (RES) x += rcur[rt];
        if (row < a.M) *reinterpret_cast<f32x4*>(a.out + (size_t)row * a.ldo + col) = x;
      }
    }
    // the next tile's rows travel under the weight gradient's products (requested here, not at the top of the tile: next to the data gradient's
    // fragments and the 96 accumulator registers they would not fit -- 73 spilled registers at D = 128; past the last tile: re-reads the last rows)
    load_tile(t + G);
    // ---- weight gradient: accW[n][k] (lane (p, j): dW[(nt0 + n) * 16 + 4j + r][(kt0 + k) * 16 + p]) += sum over the tile's rows of dY[row][.] X[row][.]
#pragma unroll
    for (int kb = 0; kb < TR / 32; ++kb) {
      bf16x8 y[WNT][3];
#pragma unroll
      for (int n = 0; n < WNT; ++n)
#pragma unroll
        for (int q = 0; q < 3; ++q) y[n][q] = tr_frag<LDY, true>(pY + q * PLY, kb, nt0 + n, p, j);
      bf16x8 x[2][3];
#pragma unroll
      for (int q = 0; q < 3; ++q) x[0][q] = tr_frag<LDX, true>(pX + q * PLX, kb, kt0, p, j);
      if (kt0 == 0) {
#pragma unroll
        for (int n = 0; n < WNT; ++n) dbp[n] += (sum8(y[n][2]) + sum8(y[n][1])) + sum8(y[n][0]);
      }
#pragma unroll
      for (int k = 0; k < WKT; ++k) {
        if (k + 1 < WKT) {
#pragma unroll
          for (int q = 0; q < 3; ++q) x[(k + 1) & 1][q] = tr_frag<LDX, true>(pX + q * PLX, kb, kt0 + k + 1, p, j);
        }
        auto& g = x[k & 1];
#pragma unroll
        for (int n = 0; n < WNT; ++n) accW[n][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y[n][1], g[1], accW[n][k], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < WNT; ++n) accW[n][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y[n][0], g[2], accW[n][k], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < WNT; ++n) accW[n][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y[n][2], g[0], accW[n][k], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < WNT; ++n) accW[n][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y[n][0], g[1], accW[n][k], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < WNT; ++n) accW[n][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y[n][1], g[0], accW[n][k], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < WNT; ++n) accW[n][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y[n][0], g[0], accW[n][k], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise hoists every column tile's fragments above the first product: 96 registers)
      }
    }
    lds_barrier();      // the next tile's planes go over these
  }
  relaunder();
  float* slab = a.slabs + (size_t)blockIdx.x * C::SLAB;
#pragma unroll
  for (int n = 0; n < WNT; ++n) {
#pragma unroll
    for (int k = 0; k < WKT; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(size_t)((nt0 + n) * 16 + 4 * j + r) * D + (kt0 + k) * 16 + p] = accW[n][k][r];
    const float sdb = gsum16(dbp[n]);
    if (kt0 == 0 && j == 0) slab[(size_t)N * D + (nt0 + n) * 16 + p] = sdb;
  }
}

// workgroups of one launch: one per CU (110 KB of LDS).  (Fewer -- 128, 192 -- to leave CUs to the backward's other branches, as the separate weight-gradient
// kernel does: 3.27 - 3.32 ms per step either way, inside the spread; INTEL_PAIR_CUS in debug builds)
int pair_grid(int ntiles) {
  static const int cap = [] { const int v = INTEL_DEBUG_ENV("INTEL_PAIR_CUS", 0); return v > 0 ? v : 1 << 30; }();
  int g = num_cus();
  if (g > cap) g = cap;
  return ntiles < g ? ntiles : g;
}

template <int D, bool MASK, int ABL = 0>
int launch_one(PairArgs& a, int grid, hipStream_t st) {
  using C = PairCfg<D>;
  static_assert(C::SMEM <= 160 * 1024, "LDS budget");
  allow_lds((linear_bwd_pair_kernel<D, MASK, ABL>), C::SMEM);
  const double flops = 4.0 * a.M * D * D;
  const double bytes = 12.0 * a.M * D;      // dY, X in; dXout out
  LAUNCH_S(a.M, D, D, flops, bytes, (linear_bwd_pair_kernel<D, MASK, ABL>), dim3(grid), dim3(C::NT), C::SMEM, st, a);
  INTEL_CHECK_LAUNCH();
  return 0;
}

int pair_mode() {
  static const int m = [] { const char* e = getenv("INTEL_PAIR_BWD"); return !e || !e[0] || e[0] == 'a' ? 2 : (e[0] == '0' ? 0 : 1); }();
  return m;
}

template <int D, int NB>
int launch_qkv(QkvArgs& a, int grid, hipStream_t st) {
  using C = QkvCfg<D, NB>;
  static_assert(C::SMEM <= 160 * 1024, "LDS budget");
  const double flops = 4.0 * a.M * (double)C::N * D;
  const double bytes = 4.0 * a.M * ((double)C::N + D + D + (a.res ? D : 0));      // dQKV, X (, residual) in; dX out
  if (a.res) {
    allow_lds((linear_bwd_qkv_kernel<D, NB, true>), C::SMEM);
    LAUNCH_S(a.M, C::N, D, flops, bytes, (linear_bwd_qkv_kernel<D, NB, true>), dim3(grid), dim3(C::NT), C::SMEM, st, a);
  } else {
    allow_lds((linear_bwd_qkv_kernel<D, NB, false>), C::SMEM);
    LAUNCH_S(a.M, C::N, D, flops, bytes, (linear_bwd_qkv_kernel<D, NB, false>), dim3(grid), dim3(C::NT), C::SMEM, st, a);
  }
  INTEL_CHECK_LAUNCH();
  return 0;
}

}  // namespace

bool linear_bwd_pair_supported(int M, int d) { return pair_mode() != 0 && M > 0 && (d == 64 || d == 128) && gemm_planes() == 3; }

size_t linear_bwd_pair_slab_floats(int M, int d) {
  if (!(d == 64 || d == 128) || M <= 0) return 0;
  const int tr = d == 128 ? 32 : 64;
  return (size_t)pair_grid(cdiv(M, tr)) * ((size_t)d * (d + 1));
}

int launch_linear_bwd_pair(const float* dY, int lddy, const float* X, int ldx, int M, int d, const void* WT_b3, int relu_mask, float* dXout, int ldo,
                           float* dW, float* db, int acc_w, int acc_b, ReduceQueue* q, hipStream_t st) {
  if (M <= 0) return 0;
  INTEL_CHECK_ARG(d == 64 || d == 128, "linear_bwd_pair: width %d (64 / 128 only)", d);
  INTEL_CHECK_ARG(gemm_planes() == 3, "linear_bwd_pair: fp32 mode only");
  INTEL_CHECK_ARG(q != nullptr, "linear_bwd_pair: needs the reduce queue");
  INTEL_CHECK_ARG(lddy % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)dY | (uintptr_t)X | (uintptr_t)dXout) % 16 == 0, "linear_bwd_pair: unaligned operand");
  const int tr = d == 128 ? 32 : 64;
  PairArgs a;
  a.dY = dY; a.X = X; a.W = reinterpret_cast<const uint4*>(WT_b3); a.out = dXout;
  a.M = M; a.ldy = lddy; a.ldx = ldx; a.ldo = ldo;
  a.ntiles = cdiv(M, tr);
  const int grid = pair_grid(a.ntiles);
  const size_t slab = (size_t)d * (d + 1);
  a.slabs = redq_alloc(q, (size_t)grid * slab);
  INTEL_CHECK_ARG(a.slabs != nullptr, "linear_bwd_pair: reduce arena exhausted");
  int rc;
#ifdef INTEL_DEBUG
  static const int abl = INTEL_DEBUG_ENV("INTEL_PAIR_ABL", 0);
  if (abl && d == 128) {
    switch (abl) {
      case 1: return launch_one<128, true, 1>(a, grid, st);
      case 2: return launch_one<128, true, 2>(a, grid, st);
      case 3: return launch_one<128, true, 3>(a, grid, st);
      case 4: return launch_one<128, true, 4>(a, grid, st);
      case 8: return launch_one<128, true, 8>(a, grid, st);
      case 16: return launch_one<128, true, 16>(a, grid, st);
      case 24: return launch_one<128, true, 24>(a, grid, st);
      case 28: return launch_one<128, true, 28>(a, grid, st);
      case 31: return launch_one<128, true, 31>(a, grid, st);
      case 27: return launch_one<128, true, 27>(a, grid, st);
      default: break;
    }
  }
#endif
  if (d == 128) rc = relu_mask ? launch_one<128, true>(a, grid, st) : launch_one<128, false>(a, grid, st);
  else rc = relu_mask ? launch_one<64, true>(a, grid, st) : launch_one<64, false>(a, grid, st);
  if (rc) return rc;
  if (dW) redq_push(q, a.slabs, slab, grid, d, d, dW, d, acc_w);
  if (db) redq_push(q, a.slabs + (size_t)d * d, slab, grid, 1, d, db, d, acc_b);
  return 0;
}

// (same switch as the feed-forward pairs: INTEL_PAIR_BWD=0.  Same-box A/B at the headline, 100 steps, three rounds: all q/k/v products -- both towers, both encoders -- 3.28 - 3.29 ms
// per step against 3.33 - 3.35 without; the 64-wide tower alone 3.31; the 128-wide products alone 3.42 - 3.44: they are worth taking together or not at all)
bool linear_bwd_qkv_supported(int M, int d, int nb) {
  return pair_mode() != 0 && M > 0 && (d == 64 || d == 128) && (nb == 3 || (nb == 2 && d == 128)) && gemm_planes() == 3;
}

size_t linear_bwd_qkv_slab_floats(int M, int d, int nb) {
  if (!(d == 64 || d == 128) || M <= 0 || nb < 2 || nb > 3) return 0;
  const int tr = d == 128 ? 32 : 64;
  return (size_t)pair_grid(cdiv(M, tr)) * ((size_t)nb * d * (d + 1));
}

int launch_linear_bwd_qkv(const float* dY, int lddy, const float* X, int ldx, const float* res, int ldr, int M, int d, int nb, const void* WT_b3, float* dXout, int ldo,
                          float* const* dW, float* const* db, const int* acc, ReduceQueue* q, hipStream_t st) {
  if (M <= 0) return 0;
  INTEL_CHECK_ARG((d == 64 || d == 128) && (nb == 3 || (nb == 2 && d == 128)), "linear_bwd_qkv: width %d x %d blocks unsupported", d, nb);
  INTEL_CHECK_ARG(gemm_planes() == 3, "linear_bwd_qkv: fp32 mode only");
  INTEL_CHECK_ARG(q != nullptr, "linear_bwd_qkv: needs the reduce queue");
  INTEL_CHECK_ARG(lddy % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && (!res || ldr % 4 == 0) &&
                  ((uintptr_t)dY | (uintptr_t)X | (uintptr_t)dXout | (uintptr_t)res) % 16 == 0, "linear_bwd_qkv: unaligned operand");
  const int tr = d == 128 ? 32 : 64;
  QkvArgs a;
  a.dY = dY; a.X = X; a.res = res; a.W = reinterpret_cast<const uint4*>(WT_b3); a.out = dXout;
  a.M = M; a.ldy = lddy; a.ldx = ldx; a.ldr = ldr; a.ldo = ldo;
  a.ntiles = cdiv(M, tr);
  const int grid = pair_grid(a.ntiles);
  const size_t slab = (size_t)nb * d * (d + 1);
  a.slabs = redq_alloc(q, (size_t)grid * slab);
  INTEL_CHECK_ARG(a.slabs != nullptr, "linear_bwd_qkv: reduce arena exhausted");
  int rc;
  if (d == 128) rc = nb == 3 ? launch_qkv<128, 3>(a, grid, st) : launch_qkv<128, 2>(a, grid, st);
  else rc = launch_qkv<64, 3>(a, grid, st);
  if (rc) return rc;
  for (int c = 0; c < nb; ++c) {
    if (dW && dW[c]) redq_push(q, a.slabs + (size_t)c * d * d, slab, grid, d, d, dW[c], d, acc ? acc[c] : 0);
    if (db && db[c]) redq_push(q, a.slabs + (size_t)nb * d * d + (size_t)c * d, slab, grid, 1, d, db[c], d, acc ? acc[c] : 0);
  }
  return 0;
}
