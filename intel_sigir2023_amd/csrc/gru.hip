// GRU4Rec encoder on MI355X (models/GeneralSeq.py:58-78): the input projection of all history rows is one GEMM on the bf16 matrix
// pipe; the RECURRENCE is one kernel per direction (gru_seq_fwd_kernel / gru_seq_bwd_kernel below): a workgroup owns 16 sessions for
// the whole time loop, W_hh fragments in registers, the state rows in LDS, packed and length-ordered histories.  The per-step form
// (hidden GEMM + gate kernel per step, INTEL_GRU_SEQ=0) is kept for cross-checking (tests/test_gru_seq_gpu.py).  Sessions shorter
// than t keep their state (the reference packs sequences by length, GeneralSeq.py:66-71).
// Gate order r, z, n and the update h' = (1-z) n + z h follow torch.nn.GRU.
#include "gru.h"
#include "kernels.h"
#include "session.h"

int gru_fwd_steps(GruBufs& g, const float* E0, int B, int T, int dm, int Hd, const int* len, const float* bih, const float* bhh,
                  float* out, int ldo, int col0, hipStream_t st);

static float* carve(char* base, size_t& off, size_t n) {
  off = rup_sz(off, 256);
  float* p = reinterpret_cast<float*>(base + off);
  off += n * sizeof(float);
  return p;
}

void gru_layout_packed(GruBufs& g, int dm, int Hd, char* base, size_t& off) {
  g.pWih = carve(base, off, packed_floats(dm, 3 * Hd));
  g.pWhh = carve(base, off, packed_floats(Hd, 3 * Hd));
  g.pWihT = carve(base, off, packed_floats(3 * Hd, dm));
  g.pWhhT = carve(base, off, packed_floats(3 * Hd, Hd));
  g.pWout = carve(base, off, packed_floats(Hd, dm));
  g.pWoutT = carve(base, off, packed_floats(dm, Hd));
}
void gru_layout_act(GruBufs& g, int B, int T, int dm, int Hd, char* base, size_t& off) {
  const size_t bt = (size_t)B * T;
  g.GI = carve(base, off, bt * 3 * Hd);
  g.HP = carve(base, off, bt * Hd);
  g.GATES = carve(base, off, bt * 3 * Hd);
  g.GHN = carve(base, off, bt * Hd);
  g.GH = carve(base, off, (size_t)B * 3 * Hd);
  g.HCUR = carve(base, off, (size_t)B * Hd);
  g.dGI = carve(base, off, bt * 3 * Hd);
  g.dGH = carve(base, off, bt * 3 * Hd);
  g.dHa = carve(base, off, (size_t)B * Hd);
  g.dHb = carve(base, off, (size_t)B * Hd);
  g.dVEC = carve(base, off, (size_t)B * dm);
}

int gru_pack(GruBufs& g, const float* Wih, const float* Whh, const float* Wout, int dm, int Hd, hipStream_t st) {
  int rc;
  if ((rc = launch_pack_b(Wih, dm, dm, 3 * Hd, 0, g.pWih, 0, st))) return rc;
  if ((rc = launch_pack_b(Whh, Hd, Hd, 3 * Hd, 0, g.pWhh, 0, st))) return rc;
  if ((rc = launch_pack_b(Wih, dm, 3 * Hd, dm, 1, g.pWihT, 0, st))) return rc;
  if ((rc = launch_pack_b(Whh, Hd, 3 * Hd, Hd, 1, g.pWhhT, 0, st))) return rc;
  if ((rc = launch_pack_b(Wout, Hd, Hd, dm, 0, g.pWout, 0, st))) return rc;
  return launch_pack_b(Wout, Hd, dm, Hd, 1, g.pWoutT, 0, st);
}

// Gate non-linearities on the hardware exp / rcp (v_exp_f32, v_rcp_f32: ~1 ulp each) instead of libm's expf / tanhf, whose
// range reductions and branches were most of a recurrence step's instructions (absolute error <= 2e-7, both forms of the recurrence)
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float gtanh(float x) {
  const float ax = fabsf(x);
  const float e = __expf(-2.f * ax);                                   // (0, 1]
  const float big = (1.f - e) * __builtin_amdgcn_rcpf(1.f + e);
  const float x2 = ax * ax;                                            // |x| < 0.1: odd series (the quotient cancels there)
  const float small = ax * (1.f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * -0.053968254f)));
  return copysignf(ax < 0.1f ? small : big, x);
}

// step t: gates from GI[:,t] + GH, new state -> HCUR and HP[:,t+1]
__global__ void gru_gate_fwd_kernel(const float* __restrict__ GI, const float* __restrict__ GH, float* __restrict__ HP,
                                    float* __restrict__ HCUR, float* __restrict__ GATES, float* __restrict__ GHN,
                                    const int* __restrict__ len, int B, int T, int Hd, int t) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Hd) return;
  const int b = i / Hd, c = i - b * Hd;
  const size_t row = (size_t)b * T + t;
  const float hp = HP[row * Hd + c];
  float hn = hp;
  if (t < len[b]) {
    const float* gi = GI + row * 3 * Hd;
    const float* gh = GH + (size_t)b * 3 * Hd;
    const float r = sigm(gi[c] + gh[c]);
    const float z = sigm(gi[Hd + c] + gh[Hd + c]);
    const float ghn = gh[2 * Hd + c];
    const float n = gtanh(gi[2 * Hd + c] + r * ghn);
    hn = (1.f - z) * n + z * hp;
    float* ga = GATES + row * 3 * Hd;
    ga[c] = r; ga[Hd + c] = z; ga[2 * Hd + c] = n;
    GHN[row * Hd + c] = ghn;
  }
  HCUR[i] = hn;
  if (t + 1 < T) HP[(row + 1) * Hd + c] = hn;
}

// step t backward: dh (in) -> dGI[:,t], dGH[:,t], dh_prev partial (the z*dh and pass-through part)
__global__ void gru_gate_bwd_kernel(const float* __restrict__ dH, const float* __restrict__ HP, const float* __restrict__ GATES,
                                    const float* __restrict__ GHN, const int* __restrict__ len, int B, int T, int Hd, int t,
                                    float* __restrict__ dGI, float* __restrict__ dGH, float* __restrict__ dHprev) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Hd) return;
  const int b = i / Hd, c = i - b * Hd;
  const size_t row = (size_t)b * T + t;
  const float dh = dH[i];
  float* dgi = dGI + row * 3 * Hd;
  float* dgh = dGH + row * 3 * Hd;
  if (t < len[b]) {
    const float* ga = GATES + row * 3 * Hd;
    const float r = ga[c], z = ga[Hd + c], n = ga[2 * Hd + c];
    const float hp = HP[row * Hd + c];
    const float dn = dh * (1.f - z);
    const float dz = dh * (hp - n);
    const float dnp = dn * (1.f - n * n);
    const float dzp = dz * z * (1.f - z);
    const float dr = dnp * GHN[row * Hd + c];
    const float drp = dr * r * (1.f - r);
    dgi[c] = drp; dgi[Hd + c] = dzp; dgi[2 * Hd + c] = dnp;
    dgh[c] = drp; dgh[Hd + c] = dzp; dgh[2 * Hd + c] = dnp * r;
    dHprev[i] = dh * z;
  } else {
    dgi[c] = 0.f; dgi[Hd + c] = 0.f; dgi[2 * Hd + c] = 0.f;
    dgh[c] = 0.f; dgh[Hd + c] = 0.f; dgh[2 * Hd + c] = 0.f;
    dHprev[i] = dh;
  }
}

// ------------------------------------------------------------------------------------------
// The recurrence as ONE kernel per direction (hidden size 128, the reference's constant: IntEL.py:105-106).  Sessions are
// independent, so a workgroup owns 16 of them for the whole time loop: no launch, no [B, 3H] round trip through HBM and no
// grid-wide dependency per step (the per-step form is two launches per step and direction: 80 dependent launches per encoder
// and training step -- at batch 512 the whole step was that chain).
//   * 512 threads = 8 waves; wave w owns hidden units 16w .. 16w+15 and, forward, their three gate columns (r, z, n tiles of
//     h W_hh^T): its B fragments -- 3 x 128 x 16 floats of W_hh -- stay in 96 registers for the whole loop; backward, the
//     16 columns of dGH W_hh (K = 384): again 96 registers;
//   * the A operand (h_{t-1} [16, 128], resp. dGH_t [16, 384]) lives in LDS, rewritten by the gate phase of every step (row
//     pitch +4 floats: the 16-byte fragment reads of 16 rows fall into disjoint banks);
//   * exact fp32 MFMAs (v_mfma_f32_16x16x4_f32: lane (i, j) supplies k = 16 g + 4 j + s to MFMA s of group g); the accumulator
//     tile puts rows 4j .. 4j+3 of unit i on lane (i, j), so the gate arithmetic is lane-local and the new state goes back to
//     LDS with one 4-byte store per row;
//   * the loop runs to the longest history of the workgroup's 16 sessions; shorter ones keep their state (forward) / pass the
//     gradient through and write zero gate gradients (backward), as the per-step kernels do.
// The same stashes as the per-step form (HP, GATES, GHN, HCUR; dGI, dGH), so the two forms mix freely (INTEL_GRU_SEQ=0: per step).
// ------------------------------------------------------------------------------------------
#define GS_ROWS 16
#define GS_H 128
#define GS_LDH (GS_H + 4)
#define GS_LDG (3 * GS_H + 4)

__device__ __forceinline__ f32x4 gs_mma4(const f32x4& a, const f32x4& b, f32x4 c) {
#pragma unroll
  for (int s = 0; s < 4; ++s) c = mfma16(a[s], b[s], c);
  return c;
}

// Workgroup barrier over the LDS rows only: __syncthreads() also waits for every global access of the wave (s_waitcnt vmcnt(0)),
// i.e. for the step's stash stores to be acknowledged by HBM -- microseconds per step of a 20-step loop
__device__ __forceinline__ void gs_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef __bf16 gs_bf16x8 __attribute__((ext_vector_type(8)));
#define GS_LDP (GS_H + 8)            // bf16 plane pitch of the state rows (272 B: 16 rows' 16-byte fragments fall into disjoint banks)
#define GS_LDQ (3 * GS_H + 8)        // ... of the gate-gradient rows

__device__ __forceinline__ void gs_split(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}
// eight consecutive floats -> three bf16 planes
__device__ __forceinline__ void gs_split8(const float (&x)[8], gs_bf16x8& h, gs_bf16x8& m, gs_bf16x8& l) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    __bf16 a, b, c;
    gs_split(x[i], a, b, c);
    h[i] = a; m[i] = b; l[i] = c;
  }
}
// the six plane products of weight >= 2^-16, smallest first (as gemm_rows_b3 / tower.hip): fp32 accuracy on the bf16 pipe
__device__ __forceinline__ f32x4 gs_mma6(const gs_bf16x8& ah, const gs_bf16x8& am, const gs_bf16x8& al, const gs_bf16x8& bh,
                                         const gs_bf16x8& bm, const gs_bf16x8& bl, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
  return c;
}

// PL: the step's product on the bf16 pipe as three-plane splits (six plane products: 72 MFMAs of 16 cycles per wave and step
// instead of 96 of 32; the state / gate-gradient rows are kept in LDS as three bf16 planes, W_hh is split once into registers)
// STASH: training (the backward reads GATES, GHN, HP); inference keeps only the final state
template <bool PL, bool STASH = true>
__global__ __launch_bounds__(512, 1) void gru_seq_fwd_kernel(const float* __restrict__ GI, const float* __restrict__ Whh,
                                                             const float* __restrict__ bhh, const int* __restrict__ len, int B, int T,
                                                             float* __restrict__ HP, float* __restrict__ HCUR,
                                                             float* __restrict__ GATES, float* __restrict__ GHN,
                                                             const int* __restrict__ off, const int* __restrict__ order) {
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[PL ? 3 * GS_ROWS * GS_LDP * 2 : GS_ROWS * GS_LDH * 4];
  float* hs = reinterpret_cast<float*>(smem_raw);
  __bf16* hp3 = reinterpret_cast<__bf16*>(smem_raw);
  constexpr int PLANE = GS_ROWS * GS_LDP;
  __shared__ int slen[GS_ROWS], sses[GS_ROWS];
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b0 = blockIdx.x * GS_ROWS;
  const int unit = 16 * w + p;
  if (PL) {
    for (int i = tid; i < 3 * PLANE; i += 512) hp3[i] = (__bf16)0.f;      // h_0 = 0
  } else {
    for (int i = tid; i < GS_ROWS * GS_LDH; i += 512) hs[i] = 0.f;
  }
  // workgroup slot i is session order[b0 + i] (sessions of similar length together) or b0 + i
  if (tid < GS_ROWS) {
    const int bsess = (b0 + tid < B) ? (order ? order[b0 + tid] : b0 + tid) : -1;
    sses[tid] = bsess;
    slen[tid] = bsess >= 0 ? min(len[bsess], T) : 0;
  }
  // B fragments: gate q, W_hh[q*128 + unit][k]: fp32 MFMAs take k = 16 j + 4 g + s, the bf16 ones k = 32 j + 8 g + s
  f32x4 wb[PL ? 1 : 3][PL ? 1 : 8];
  gs_bf16x8 wq[PL ? 3 : 1][PL ? 4 : 1][3];
  float bh[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    bh[q] = bhh[q * GS_H + unit];
    const float* wrow = Whh + (size_t)(q * GS_H + unit) * GS_H;
    if (PL) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(wrow + 32 * j + 8 * g);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(wrow + 32 * j + 8 * g + 4);
        const float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        gs_split8(x, wq[PL ? q : 0][PL ? j : 0][0], wq[PL ? q : 0][PL ? j : 0][1], wq[PL ? q : 0][PL ? j : 0][2]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) wb[PL ? 0 : q][PL ? 0 : j] = *reinterpret_cast<const f32x4*>(wrow + 16 * j + 4 * g);
    }
  }
  __syncthreads();
  int tmax = 0, lr[4];
  size_t rb[4];           // first row of the session: b*T (padded [B, T] rows) or off[b] (packed: only the len[b] valid rows exist)
#pragma unroll
  for (int i = 0; i < GS_ROWS; ++i) tmax = max(tmax, slen[i]);
  int bs[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    lr[r] = slen[4 * g + r];
    bs[r] = sses[4 * g + r];
    rb[r] = bs[r] >= 0 ? (off ? (size_t)off[bs[r]] : (size_t)bs[r] * T) : 0;
  }
  float h[4] = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < tmax; ++t) {
    // gate inputs of this step: in flight under the MFMAs
    float gi[4][3];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t row = rb[r] + t;
#pragma unroll
      for (int q = 0; q < 3; ++q) gi[r][q] = (t < lr[r]) ? GI[row * (3 * GS_H) + q * GS_H + unit] : 0.f;
    }
    f32x4 acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[q] = f32x4{bh[q], bh[q], bh[q], bh[q]};
    if (PL) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const __bf16* ap = hp3 + p * GS_LDP + 32 * j + 8 * g;
        const gs_bf16x8 ah = *reinterpret_cast<const gs_bf16x8*>(ap);
        const gs_bf16x8 am = *reinterpret_cast<const gs_bf16x8*>(ap + PLANE);
        const gs_bf16x8 al = *reinterpret_cast<const gs_bf16x8*>(ap + 2 * PLANE);
#pragma unroll
        for (int q = 0; q < 3; ++q)
          acc[q] = gs_mma6(ah, am, al, wq[PL ? q : 0][PL ? j : 0][0], wq[PL ? q : 0][PL ? j : 0][1], wq[PL ? q : 0][PL ? j : 0][2], acc[q]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(hs + p * GS_LDH + 16 * j + 4 * g);
#pragma unroll
        for (int q = 0; q < 3; ++q) acc[q] = gs_mma4(a, wb[PL ? 0 : q][PL ? 0 : j], acc[q]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = bs[r] >= 0 ? bs[r] : B;
      const size_t row = rb[r] + t;
      if (t < lr[r]) {
        const float rg = sigm(gi[r][0] + acc[0][r]);
        const float zg = sigm(gi[r][1] + acc[1][r]);
        const float ghn = acc[2][r];
        const float ng = gtanh(gi[r][2] + rg * ghn);
        h[r] = (1.f - zg) * ng + zg * h[r];
        if (STASH) {
          float* ga = GATES + row * (3 * GS_H);
          ga[unit] = rg; ga[GS_H + unit] = zg; ga[2 * GS_H + unit] = ng;
          GHN[row * GS_H + unit] = ghn;
        }
      }
      if (STASH && b < B && t + 1 < (off ? lr[r] : T)) HP[(row + 1) * GS_H + unit] = h[r];      // packed: row t+1 exists only below len
      if (STASH && t == 0 && b < B && (off ? lr[r] > 0 : true)) HP[row * GS_H + unit] = 0.f;      // h_0 = 0: the first row of every session (no separate fill launch)
    }
    gs_lds_barrier();                     // every wave has read h_{t-1}
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (PL) {
        __bf16 a, bq, c;
        gs_split(h[r], a, bq, c);
        __bf16* d = hp3 + (4 * g + r) * GS_LDP + unit;
        d[0] = a; d[PLANE] = bq; d[2 * PLANE] = c;
      } else {
        hs[(4 * g + r) * GS_LDH + unit] = h[r];
      }
    }
    gs_lds_barrier();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = bs[r] >= 0 ? bs[r] : B;
    if (b < B) {
      HCUR[(size_t)b * GS_H + unit] = h[r];
      // the steps the loop did not run (t >= the workgroup's longest history) keep the state: h_{t-1} stash for the weight gradient
      // (padded rows only: packed histories have no rows past len)
      for (int t = max(tmax, 0); t + 1 < T && !off && STASH; ++t) HP[((size_t)b * T + t + 1) * GS_H + unit] = h[r];
      if (STASH && !off && tmax <= 0) HP[(size_t)b * T * GS_H + unit] = 0.f;      // (the time loop, which writes h_0 = 0, did not run at all)
    }
  }
}

template <bool PL>
__global__ __launch_bounds__(512, 1) void gru_seq_bwd_kernel(const float* __restrict__ dH0, const float* __restrict__ HP,
                                                             const float* __restrict__ GATES, const float* __restrict__ GHN,
                                                             const float* __restrict__ Whh, const int* __restrict__ len, int B, int T,
                                                             float* __restrict__ dGI, float* __restrict__ dGH, const int* __restrict__ off,
                                                             const int* __restrict__ order) {
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[PL ? 3 * GS_ROWS * GS_LDQ * 2 : GS_ROWS * GS_LDG * 4];
  float* ds = reinterpret_cast<float*>(smem_raw);
  __bf16* dq3 = reinterpret_cast<__bf16*>(smem_raw);
  constexpr int PLANE = GS_ROWS * GS_LDQ;
  __shared__ int slen[GS_ROWS], sses[GS_ROWS];
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b0 = blockIdx.x * GS_ROWS;
  const int unit = 16 * w + p;
  if (tid < GS_ROWS) {
    const int bsess = (b0 + tid < B) ? (order ? order[b0 + tid] : b0 + tid) : -1;
    sses[tid] = bsess;
    slen[tid] = bsess >= 0 ? min(len[bsess], T) : 0;
  }
  if (PL) {
    for (int i = tid; i < 3 * PLANE; i += 512) dq3[i] = (__bf16)0.f;      // the pad columns are never written again
  }
  // B fragments of dh_{t-1} += dGH_t W_hh: W_hh[k][unit], k = 16 j + 4 g + s (fp32 MFMAs) / 32 j + 8 g + s (bf16 ones)
  f32x4 wb[PL ? 1 : 24];
  gs_bf16x8 wq[PL ? 12 : 1][3];
  if (PL) {
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      float x[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) x[s] = Whh[(size_t)(32 * j + 8 * g + s) * GS_H + unit];
      gs_split8(x, wq[PL ? j : 0][0], wq[PL ? j : 0][1], wq[PL ? j : 0][2]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 24; ++j)
#pragma unroll
      for (int s = 0; s < 4; ++s) wb[PL ? 0 : j][s] = Whh[(size_t)(16 * j + 4 * g + s) * GS_H + unit];
  }
  __syncthreads();
  int tmax = 0, lr[4];
#pragma unroll
  for (int i = 0; i < GS_ROWS; ++i) tmax = max(tmax, slen[i]);
  float dh[4];
  size_t rb[4];
  int bs[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    lr[r] = slen[4 * g + r];
    bs[r] = sses[4 * g + r];
    const int b = bs[r] >= 0 ? bs[r] : B;
    dh[r] = b < B ? dH0[(size_t)b * GS_H + unit] : 0.f;
    rb[r] = b < B ? (off ? (size_t)off[b] : (size_t)b * T) : 0;
  }
  // steps nobody in this workgroup reached: zero gate gradients (they feed the weight-gradient products over all B*T rows;
  // packed histories have no such rows)
  for (int t = T - 1; t >= tmax && !off; --t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = bs[r] >= 0 ? bs[r] : B;
      if (b < B) {
        const size_t row = (size_t)b * T + t;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          dGI[row * (3 * GS_H) + q * GS_H + unit] = 0.f;
          dGH[row * (3 * GS_H) + q * GS_H + unit] = 0.f;
        }
      }
    }
  // the step's stash values (r, z, n, h_{t-1}, gh_n of four rows): loaded one step ahead, under the previous step's MFMAs
  float sv[4][5];
  auto load_stash = [&](int t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t row = rb[r] + t;
      const bool on = t >= 0 && t < lr[r];
      const float* ga = GATES + row * (3 * GS_H);
      sv[r][0] = on ? ga[unit] : 0.f;
      sv[r][1] = on ? ga[GS_H + unit] : 0.f;
      sv[r][2] = on ? ga[2 * GS_H + unit] : 0.f;
      sv[r][3] = on ? HP[row * GS_H + unit] : 0.f;
      sv[r][4] = on ? GHN[row * GS_H + unit] : 0.f;
    }
  };
  load_stash(tmax - 1);
  for (int t = tmax - 1; t >= 0; --t) {
    float dprev[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = bs[r] >= 0 ? bs[r] : B;
      const size_t row = rb[r] + t;
      float drp = 0.f, dzp = 0.f, dnp = 0.f, dnr = 0.f;
      dprev[r] = dh[r];
      if (t < lr[r]) {
        const float rg = sv[r][0], zg = sv[r][1], ng = sv[r][2];
        const float hp = sv[r][3];
        const float dn = dh[r] * (1.f - zg);
        const float dz = dh[r] * (hp - ng);
        dnp = dn * (1.f - ng * ng);
        dzp = dz * zg * (1.f - zg);
        const float dr = dnp * sv[r][4];
        drp = dr * rg * (1.f - rg);
        dnr = dnp * rg;
        dprev[r] = dh[r] * zg;
      }
      if (b < B && (!off || t < lr[r])) {       // packed: the row exists only below len
        float* dgi = dGI + row * (3 * GS_H);
        float* dgh = dGH + row * (3 * GS_H);
        dgi[unit] = drp; dgi[GS_H + unit] = dzp; dgi[2 * GS_H + unit] = dnp;
        dgh[unit] = drp; dgh[GS_H + unit] = dzp; dgh[2 * GS_H + unit] = dnr;
      }
      if (PL) {
        const float v3[3] = {drp, dzp, dnr};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          __bf16 a, bq, c;
          gs_split(v3[q], a, bq, c);
          __bf16* d = dq3 + (4 * g + r) * GS_LDQ + q * GS_H + unit;
          d[0] = a; d[PLANE] = bq; d[2 * PLANE] = c;
        }
      } else {
        float* dl = ds + (4 * g + r) * GS_LDG;
        dl[unit] = drp; dl[GS_H + unit] = dzp; dl[2 * GS_H + unit] = dnr;
      }
    }
    gs_lds_barrier();
    load_stash(t - 1);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (PL) {
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const __bf16* ap = dq3 + p * GS_LDQ + 32 * j + 8 * g;
        const gs_bf16x8 ah = *reinterpret_cast<const gs_bf16x8*>(ap);
        const gs_bf16x8 am = *reinterpret_cast<const gs_bf16x8*>(ap + PLANE);
        const gs_bf16x8 al = *reinterpret_cast<const gs_bf16x8*>(ap + 2 * PLANE);
        acc = gs_mma6(ah, am, al, wq[PL ? j : 0][0], wq[PL ? j : 0][1], wq[PL ? j : 0][2], acc);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 24; ++j) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(ds + p * GS_LDG + 16 * j + 4 * g);
        acc = gs_mma4(a, wb[PL ? 0 : j], acc);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) dh[r] = dprev[r] + acc[r];
    gs_lds_barrier();                     // the next step rewrites the gate-gradient rows
  }
}

// INTEL_GRU_SEQ: 0 the per-step form, 1 the one-kernel recurrence with exact fp32 MFMAs, 2 (default) with three-plane bf16 products
static int gru_seq_mode() {
  static const int m = [] { const char* e = getenv("INTEL_GRU_SEQ"); return e ? atoi(e) : 2; }();
  return m;
}
static bool gru_seq_on(int Hd, const float* Whh) {
  return gru_seq_mode() != 0 && Hd == GS_H && Whh != nullptr && (reinterpret_cast<uintptr_t>(Whh) & 15) == 0;
}

bool gru_packed_supported(int Hd) { return gru_seq_mode() != 0 && Hd == GS_H; }
bool gru_ext_proj_supported(int dm, int Hd) { return gru_seq_mode() != 0 && Hd == GS_H && dm % 16 == 0; }

int gru_fwd(GruBufs& g, const float* E0, int B, int T, int dm, int Hd, const int* len, const float* bih, const float* bhh,
            float* out, int ldo, int col0, hipStream_t st, const float* Whh, const int* off, int rows, const int* order, bool stash) {
  if (!off) rows = B * T;
  INTEL_CHECK_ARG(!off || gru_seq_on(Hd, Whh), "gru: packed history rows need the one-kernel recurrence (hidden size 128, aligned W_hh)");
  if (gru_seq_on(Hd, Whh)) {
    int rc;
    GemmEpilogue ei;
    ei.bias = bih;
    if ((rc = launch_gemm_rows(E0, dm, rows, dm, g.pWih, 3 * Hd, g.GI, 3 * Hd, ei, st))) return rc;
    if (gru_seq_mode() == 1)
      LAUNCH(gru_seq_fwd_kernel<false>, dim3(cdiv(B, GS_ROWS)), dim3(512), 0, st, g.GI, Whh, bhh, len, B, T, g.HP, g.HCUR, g.GATES, g.GHN, off, order);
    else if (stash)
      LAUNCH(gru_seq_fwd_kernel<true>, dim3(cdiv(B, GS_ROWS)), dim3(512), 0, st, g.GI, Whh, bhh, len, B, T, g.HP, g.HCUR, g.GATES, g.GHN, off, order);
    else
      LAUNCH((gru_seq_fwd_kernel<true, false>), dim3(cdiv(B, GS_ROWS)), dim3(512), 0, st, g.GI, Whh, bhh, len, B, T, g.HP, g.HCUR, g.GATES, g.GHN, off, order);
    INTEL_CHECK_LAUNCH();
    if (g.ext_proj) return 0;
    GemmEpilogue e0;
    return launch_gemm_rows(g.HCUR, Hd, B, Hd, g.pWout, dm, out + col0, ldo, e0, st);
  }
  return gru_fwd_steps(g, E0, B, T, dm, Hd, len, bih, bhh, out, ldo, col0, st);
}

int gru_fwd_steps(GruBufs& g, const float* E0, int B, int T, int dm, int Hd, const int* len, const float* bih, const float* bhh,
                  float* out, int ldo, int col0, hipStream_t st) {
  int rc;
  GemmEpilogue ei;
  ei.bias = bih;
  if ((rc = launch_gemm_rows(E0, dm, B * T, dm, g.pWih, 3 * Hd, g.GI, 3 * Hd, ei, st))) return rc;
  // h_0 = 0: HP[:,0]
  for (int t = 0; t < T; ++t) {
    if (t == 0) {
      // zero HP[:,0,:] rows (stride T*Hd): use copy-free fill via a strided kernel = fill whole HP once
      if ((rc = launch_fill(g.HP, (long long)B * T * Hd, 0.f, st))) return rc;
    }
    GemmEpilogue eh;
    eh.bias = bhh;
    if ((rc = launch_gemm_rows(g.HP + (size_t)t * Hd, T * Hd, B, Hd, g.pWhh, 3 * Hd, g.GH, 3 * Hd, eh, st))) return rc;
    LAUNCH(gru_gate_fwd_kernel, dim3(cdiv(B * Hd, 256)), dim3(256), 0, st, g.GI, g.GH, g.HP, g.HCUR, g.GATES, g.GHN, len,
                       B, T, Hd, t);
    INTEL_CHECK_LAUNCH();
  }
  GemmEpilogue e0;
  return launch_gemm_rows(g.HCUR, Hd, B, Hd, g.pWout, dm, out + col0, ldo, e0, st);
}

int gru_bwd(GruBufs& g, const float* E0, int B, int T, int dm, int Hd, const int* len, const float* Whh, const float* bhh,
            const float* dout, int ldo, int col0, const GruGrads& gg, float* dE0, float* scratch, float* slabs,
            hipStream_t st, const int* off, int prows, const int* order, ReduceQueue* q) {
  (void)bhh; (void)scratch;
  INTEL_CHECK_ARG(!off || gru_seq_on(Hd, Whh), "gru: packed history rows need the one-kernel recurrence (hidden size 128, aligned W_hh)");
  int rc;
  // vec = HCUR Wout^T
  if (!g.ext_proj && gg.dWout && (rc = launch_wgrad(dout + col0, ldo, g.HCUR, Hd, B, dm, Hd, gg.dWout, Hd, nullptr, 0, slabs, st, q))) return rc;
  GemmEpilogue e0;
  float *dH = g.dHa, *dHn = g.dHb;
  if (!g.ext_proj && (rc = launch_gemm_rows(dout + col0, ldo, B, dm, g.pWoutT, Hd, dH, Hd, e0, st))) return rc;
  const bool seq = gru_seq_on(Hd, Whh);
  if (seq) {
    if (gru_seq_mode() == 1)
      LAUNCH(gru_seq_bwd_kernel<false>, dim3(cdiv(B, GS_ROWS)), dim3(512), 0, st, dH, g.HP, g.GATES, g.GHN, Whh, len, B, T, g.dGI, g.dGH, off, order);
    else
      LAUNCH(gru_seq_bwd_kernel<true>, dim3(cdiv(B, GS_ROWS)), dim3(512), 0, st, dH, g.HP, g.GATES, g.GHN, Whh, len, B, T, g.dGI, g.dGH, off, order);
    INTEL_CHECK_LAUNCH();
  }
  for (int t = T - 1; t >= 0 && !seq; --t) {
    LAUNCH(gru_gate_bwd_kernel, dim3(cdiv(B * Hd, 256)), dim3(256), 0, st, dH, g.HP, g.GATES, g.GHN, len, B, T, Hd, t,
                       g.dGI, g.dGH, dHn);
    INTEL_CHECK_LAUNCH();
    // dh_{t-1} += dGH_t Whh
    GemmEpilogue ea;
    ea.accumulate = 1;
    if ((rc = launch_gemm_rows(g.dGH + (size_t)t * 3 * Hd, T * 3 * Hd, B, 3 * Hd, g.pWhhT, Hd, dHn, Hd, ea, st))) return rc;
    float* tmp = dH; dH = dHn; dHn = tmp;
  }
  const int rows = off ? prows : B * T;
  if (gg.dWih && (rc = launch_wgrad(g.dGI, 3 * Hd, E0, dm, rows, 3 * Hd, dm, gg.dWih, dm, gg.dbih, 0, slabs, st, q))) return rc;
  if (gg.dWhh && (rc = launch_wgrad(g.dGH, 3 * Hd, g.HP, Hd, rows, 3 * Hd, Hd, gg.dWhh, Hd, gg.dbhh, 0, slabs, st, q))) return rc;
  return launch_gemm_rows(g.dGI, 3 * Hd, rows, 3 * Hd, g.pWihT, dm, dE0, dm, e0, st);
}
