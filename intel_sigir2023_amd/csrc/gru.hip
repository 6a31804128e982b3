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
// Round 5, three-plane form (gru_seq_fwd_pl / gru_seq_bwd_pl): the weight fragments are the A operand and the state / gate-gradient rows the B
// operand, so the accumulator tile puts hidden units 16w+4g .. +3 of SESSION p on lane (p, g): every row access of a step is one 16-byte
// vector per lane and tensor (3 loads + 5 stores forward, 5 loads + 6 stores backward, instead of 12 + 20 / 20 + 24 dwords) and the new rows go
// to LDS as one 8-byte store per plane (instead of 12 / 36 two-byte stores); the LDS rows are double-buffered (one barrier per step); the
// step's loads are issued at clamped row indices a step ahead (no select on a freshly loaded value).  The split planes of W_hh are pinned in
// their 144 registers (gs_keep).  Per-step shader clocks at 512 sessions x 20 steps (debug build, INTEL_GRU_DBG=1): forward ~6 100 =
// 3 000 LDS reads + MFMAs (the floor: 2 waves x 72 MFMAs x 16 cycles = 2 304 per SIMD), 1 300 gate arithmetic, 500 stash stores, 500 LDS writes,
// 270 barrier; backward ~7 600 with ~2 500 of it waiting for the stash rows (with every request pointed at one hot row: 5 850).
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

typedef __bf16 gs_bf16x4 __attribute__((ext_vector_type(4)));
#define GS_HBUF (3 * GS_ROWS * GS_LDP)      // bf16 elements of one state buffer (three planes)
#define GS_QBUF (3 * GS_ROWS * GS_LDQ)      // ... of one gate-gradient buffer
__device__ __forceinline__ void gs_split4(const f32x4& x, gs_bf16x4& h, gs_bf16x4& m, gs_bf16x4& l) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    __bf16 a, b, c;
    gs_split(x[i], a, b, c);
    h[i] = a; m[i] = b; l[i] = c;
  }
}

// The split planes of W_hh must STAY in their 144 registers: left to itself the compiler keeps the 96 fp32 values instead and re-derives the planes
// in every step (rematerialisation: ~460 conversion / mask / subtract instructions per wave and step -- more VALU time than the step's MFMAs)
__device__ __forceinline__ void gs_keep(gs_bf16x8& v) { asm volatile("" : "+v"(v)); }

#ifdef INTEL_DEBUG
__device__ int g_gru_abl;                       // INTEL_GRU_ABL (debug builds): 1 = every stash / gate-input request goes to row 0 (cache-hot: what is memory latency?), 2 = no global stores
__device__ unsigned long long g_gru_dbg[16];      // INTEL_GRU_DBG=1 (debug builds): per-phase shader clocks of workgroup 0's thread 0, summed over the steps
#define GS_MARK(ph) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long now__ = clock64(); g_gru_dbg[ph] += now__ - tstamp__; tstamp__ = now__; } } while (0)
#define GS_MARK0() unsigned long long tstamp__ = clock64()
#else
#define GS_MARK(ph) do { } while (0)
#define GS_MARK0() do { } while (0)
#endif

// Forward recurrence, three-plane products.  Lane (p, g) of wave w: session slot p, hidden units 16w + 4g .. + 3.
template <bool STASH>
__device__ __forceinline__ void gru_seq_fwd_pl(const float* __restrict__ GI, const float* __restrict__ Whh, const float* __restrict__ bhh,
                                               const int* __restrict__ len, int B, int T, float* __restrict__ HP, float* __restrict__ HCUR,
                                               float* __restrict__ GATES, float* __restrict__ GHN, const int* __restrict__ off,
                                               const int* __restrict__ order) {
  __shared__ __attribute__((aligned(16))) __bf16 hp3[2 * GS_HBUF];
  __shared__ int slen[GS_ROWS], sses[GS_ROWS];
  constexpr int PLANE = GS_ROWS * GS_LDP;
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b0 = blockIdx.x * GS_ROWS;
  const int u0 = 16 * w + 4 * g;
  for (int i = tid; i < GS_HBUF; i += 512) hp3[i] = (__bf16)0.f;      // buffer 0: h_0 = 0
  if (tid < GS_ROWS) {
    const int bsess = (b0 + tid < B) ? (order ? order[b0 + tid] : b0 + tid) : -1;
    sses[tid] = bsess;
    slen[tid] = bsess >= 0 ? min(len[bsess], T) : 0;
  }
  // A fragments: gate q, W_hh[q*128 + 16w + p][32 j + 8 g + s], split once
  gs_bf16x8 wq[3][4][3];
  f32x4 bh[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    bh[q] = *reinterpret_cast<const f32x4*>(bhh + q * GS_H + u0);
    const float* wrow = Whh + (size_t)(q * GS_H + 16 * w + p) * GS_H;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(wrow + 32 * j + 8 * g);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(wrow + 32 * j + 8 * g + 4);
      const float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      gs_split8(x, wq[q][j][0], wq[q][j][1], wq[q][j][2]);
      gs_keep(wq[q][j][0]); gs_keep(wq[q][j][1]); gs_keep(wq[q][j][2]);
    }
  }
  __syncthreads();
  int tmax = 0;
#pragma unroll
  for (int i = 0; i < GS_ROWS; ++i) tmax = max(tmax, slen[i]);
  const int ls = slen[p], bsess = sses[p];
  const bool live = bsess >= 0;
  const size_t rb = live ? (off ? (size_t)off[bsess] : (size_t)bsess * T) : 0;      // first row: b*T (padded [B, T] rows) or off[b] (packed)
  // gate inputs of step t at a clamped row (a session without history reads row 0, which exists whenever the loop runs): the values of steps
  // past the session's length are loaded and never looked at
  auto ldgi = [&](int t, f32x4 (&v)[3]) {
    const int tc = min(t, ls - 1);
    const float* src = GI + (tc >= 0 ? rb + tc : 0) * (3 * GS_H) + u0;
#pragma unroll
    for (int q = 0; q < 3; ++q) v[q] = *reinterpret_cast<const f32x4*>(src + q * GS_H);
  };
  f32x4 h4 = f32x4{0.f, 0.f, 0.f, 0.f};
  int cur = 0;
  GS_MARK0();
  // One register set in flight: the top of step t takes the values requested at the top of step t-1 (the only wait of the step, for requests a whole
  // step old) and requests step t+1's.  (Rotating two or three sets through an unrolled loop costs more than it hides: the compiler's wait-count
  // bookkeeping merges the copies' states at the loop header and then waits for EVERYTHING in flight -- stores included -- at the top of each step.)
  f32x4 gnext[3];
  auto step = [&](int t) {
    GS_MARK(0);
    f32x4 gi[3] = {gnext[0], gnext[1], gnext[2]};
    ldgi(t + 1, gnext);
    __builtin_amdgcn_sched_barrier(0);      // the requests stay HERE: the scheduler sinks them below the MFMAs otherwise
    f32x4 acc[3] = {bh[0], bh[1], bh[2]};
    const __bf16* rd = hp3 + cur * GS_HBUF + p * GS_LDP + 8 * g;
    // two passes over the state rows: the r and z columns first, the n columns second -- the sigmoids of r and z (half of the gate arithmetic) are
    // issued between the k-blocks of the second pass and run on the vector ALU while the matrix pipe works (the fragments are read twice: LDS
    // has the time, the registers for keeping all twelve do not exist).  They are computed for finished sessions too (never looked at).
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const gs_bf16x8 xh = *reinterpret_cast<const gs_bf16x8*>(rd + 32 * j);
      const gs_bf16x8 xm = *reinterpret_cast<const gs_bf16x8*>(rd + 32 * j + PLANE);
      const gs_bf16x8 xl = *reinterpret_cast<const gs_bf16x8*>(rd + 32 * j + 2 * PLANE);
#pragma unroll
      for (int q = 0; q < 2; ++q) acc[q] = gs_mma6(wq[q][j][0], wq[q][j][1], wq[q][j][2], xh, xm, xl, acc[q]);      // [unit][session]
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 rg, zg;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const gs_bf16x8 xh = *reinterpret_cast<const gs_bf16x8*>(rd + 32 * j);
      const gs_bf16x8 xm = *reinterpret_cast<const gs_bf16x8*>(rd + 32 * j + PLANE);
      const gs_bf16x8 xl = *reinterpret_cast<const gs_bf16x8*>(rd + 32 * j + 2 * PLANE);
      acc[2] = gs_mma6(wq[2][j][0], wq[2][j][1], wq[2][j][2], xh, xm, xl, acc[2]);
      rg[j] = sigm(gi[0][j] + acc[0][j]);
      zg[j] = sigm(gi[1][j] + acc[1][j]);
      asm volatile("" : "+v"(rg[j]), "+v"(zg[j]));      // (computed HERE: otherwise sunk into the branch below, behind the last MFMA)
      __builtin_amdgcn_sched_barrier(0);
    }
    const size_t row = rb + t;
#ifdef INTEL_DEBUG
    asm volatile("" :: "v"(acc[2]));
    GS_MARK(1);
#endif
    if (t < ls) {
      f32x4 ng;
      const f32x4 ghn = acc[2];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ng[r] = gtanh(gi[2][r] + rg[r] * ghn[r]);
        h4[r] = (1.f - zg[r]) * ng[r] + zg[r] * h4[r];
      }
      if (STASH) {
        float* ga = GATES + row * (3 * GS_H) + u0;
        *reinterpret_cast<f32x4*>(ga) = rg;
        *reinterpret_cast<f32x4*>(ga + GS_H) = zg;
        *reinterpret_cast<f32x4*>(ga + 2 * GS_H) = ng;
        *reinterpret_cast<f32x4*>(GHN + row * GS_H + u0) = ghn;
      }
    }
    GS_MARK(2);
    if (STASH && live) {
      if (t + 1 < (off ? ls : T)) *reinterpret_cast<f32x4*>(HP + (row + 1) * GS_H + u0) = h4;      // packed: row t+1 exists only below len
      if (t == 0 && (off ? ls > 0 : true)) *reinterpret_cast<f32x4*>(HP + row * GS_H + u0) = f32x4{0.f, 0.f, 0.f, 0.f};      // h_0 = 0 (no fill launch)
    }
    GS_MARK(3);
    gs_bf16x4 a, bq, c;
    gs_split4(h4, a, bq, c);
    __bf16* wr = hp3 + (cur ^ 1) * GS_HBUF + p * GS_LDP + u0;      // the other buffer: nobody reads it during this step
    *reinterpret_cast<gs_bf16x4*>(wr) = a;
    *reinterpret_cast<gs_bf16x4*>(wr + PLANE) = bq;
    *reinterpret_cast<gs_bf16x4*>(wr + 2 * PLANE) = c;
    cur ^= 1;
    GS_MARK(4);
    gs_lds_barrier();
    GS_MARK(5);
  };
  if (tmax > 0) {
    ldgi(0, gnext);
    for (int t = 0; t < tmax; ++t) step(t);
  }
  if (live) {
    *reinterpret_cast<f32x4*>(HCUR + (size_t)bsess * GS_H + u0) = h4;
    // the steps the loop did not run (t >= the workgroup's longest history) keep the state: h_{t-1} stash for the weight gradient
    // (padded rows only: packed histories have no rows past len)
    for (int t = tmax; t + 1 < T && !off && STASH; ++t) *reinterpret_cast<f32x4*>(HP + ((size_t)bsess * T + t + 1) * GS_H + u0) = h4;
    if (STASH && !off && tmax <= 0) *reinterpret_cast<f32x4*>(HP + (size_t)bsess * T * GS_H + u0) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// Backward recurrence, three-plane products: dh_{t-1}[unit][session] = sum_k W_hh[k][unit] dGH_t[session][k], same lane roles.
__device__ __forceinline__ void gru_seq_bwd_pl(const float* __restrict__ dH0, const float* __restrict__ HP, const float* __restrict__ GATES,
                                               const float* __restrict__ GHN, const float* __restrict__ Whh, const int* __restrict__ len, int B,
                                               int T, float* __restrict__ dGI, float* __restrict__ dGH, const int* __restrict__ off,
                                               const int* __restrict__ order) {
  __shared__ __attribute__((aligned(16))) __bf16 dq3[2 * GS_QBUF];
  __shared__ int slen[GS_ROWS], sses[GS_ROWS];
  constexpr int PLANE = GS_ROWS * GS_LDQ;
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b0 = blockIdx.x * GS_ROWS;
  const int u0 = 16 * w + 4 * g;
  if (tid < GS_ROWS) {
    const int bsess = (b0 + tid < B) ? (order ? order[b0 + tid] : b0 + tid) : -1;
    sses[tid] = bsess;
    slen[tid] = bsess >= 0 ? min(len[bsess], T) : 0;
  }
  // A fragments: W_hh[32 j + 8 g + s][16w + p]
  gs_bf16x8 wq[12][3];
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    float x[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) x[s] = Whh[(size_t)(32 * j + 8 * g + s) * GS_H + 16 * w + p];
    gs_split8(x, wq[j][0], wq[j][1], wq[j][2]);
    gs_keep(wq[j][0]); gs_keep(wq[j][1]); gs_keep(wq[j][2]);
  }
  __syncthreads();
  int tmax = 0;
#pragma unroll
  for (int i = 0; i < GS_ROWS; ++i) tmax = max(tmax, slen[i]);
  const int ls = slen[p], bsess = sses[p];
  const bool live = bsess >= 0;
  const size_t rb = live ? (off ? (size_t)off[bsess] : (size_t)bsess * T) : 0;
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 dh4 = live ? *reinterpret_cast<const f32x4*>(dH0 + (size_t)bsess * GS_H + u0) : zero4;
  // steps nobody in this workgroup reached: zero gate gradients (they feed the weight-gradient products over all B*T rows;
  // packed histories have no such rows)
  for (int t = T - 1; t >= tmax && !off && live; --t) {
    const size_t row = (size_t)bsess * T + t;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      *reinterpret_cast<f32x4*>(dGI + row * (3 * GS_H) + q * GS_H + u0) = zero4;
      *reinterpret_cast<f32x4*>(dGH + row * (3 * GS_H) + q * GS_H + u0) = zero4;
    }
  }
  // the step's stash values (r, z, n, h_{t-1}, gh_n) at a clamped row, requested a whole step ahead and looked at only below the session's length
  auto ldst = [&](int t, f32x4 (&v)[5]) {
    const int tc = min(max(t, 0), ls - 1);
#ifdef INTEL_DEBUG
    const size_t row = (g_gru_abl & 1) ? 0 : (tc >= 0 ? rb + tc : 0);
#else
    const size_t row = tc >= 0 ? rb + tc : 0;
#endif
    const float* ga = GATES + row * (3 * GS_H) + u0;
    v[0] = *reinterpret_cast<const f32x4*>(ga);
    v[1] = *reinterpret_cast<const f32x4*>(ga + GS_H);
    v[2] = *reinterpret_cast<const f32x4*>(ga + 2 * GS_H);
    v[3] = *reinterpret_cast<const f32x4*>(HP + row * GS_H + u0);
    v[4] = *reinterpret_cast<const f32x4*>(GHN + row * GS_H + u0);
  };
  int cur = 0;
  GS_MARK0();
  f32x4 snext[5];
  auto step = [&](int t) {
    GS_MARK(8);
    f32x4 sv[5] = {snext[0], snext[1], snext[2], snext[3], snext[4]};      // (one set in flight: see the forward kernel)
    f32x4 drp = zero4, dzp = zero4, dnp = zero4, dnr = zero4, dprev = dh4;
    if (t < ls) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float rg = sv[0][r], zg = sv[1][r], ng = sv[2][r], hp = sv[3][r];
        const float dn = dh4[r] * (1.f - zg);
        const float dz = dh4[r] * (hp - ng);
        dnp[r] = dn * (1.f - ng * ng);
        dzp[r] = dz * zg * (1.f - zg);
        const float dr = dnp[r] * sv[4][r];
        drp[r] = dr * rg * (1.f - rg);
        dnr[r] = dnp[r] * rg;
        dprev[r] = dh4[r] * zg;
      }
    }
#ifdef INTEL_DEBUG
    asm volatile("" :: "v"(drp), "v"(dzp), "v"(dnp), "v"(dnr));
#endif
    GS_MARK(9);
#ifdef INTEL_DEBUG
    if (live && (!off || t < ls) && !(g_gru_abl & 2)) {
#else
    if (live && (!off || t < ls)) {       // packed: the row exists only below len
#endif
      float* dgi = dGI + (rb + t) * (3 * GS_H) + u0;
      float* dgh = dGH + (rb + t) * (3 * GS_H) + u0;
      *reinterpret_cast<f32x4*>(dgi) = drp;
      *reinterpret_cast<f32x4*>(dgi + GS_H) = dzp;
      *reinterpret_cast<f32x4*>(dgi + 2 * GS_H) = dnp;
      *reinterpret_cast<f32x4*>(dgh) = drp;
      *reinterpret_cast<f32x4*>(dgh + GS_H) = dzp;
      *reinterpret_cast<f32x4*>(dgh + 2 * GS_H) = dnr;
    }
    GS_MARK(10);
    __bf16* wr = dq3 + cur * GS_QBUF + p * GS_LDQ + u0;
    {
      gs_bf16x4 a, bq, c;
      gs_split4(drp, a, bq, c);
      *reinterpret_cast<gs_bf16x4*>(wr) = a; *reinterpret_cast<gs_bf16x4*>(wr + PLANE) = bq; *reinterpret_cast<gs_bf16x4*>(wr + 2 * PLANE) = c;
      gs_split4(dzp, a, bq, c);
      *reinterpret_cast<gs_bf16x4*>(wr + GS_H) = a; *reinterpret_cast<gs_bf16x4*>(wr + GS_H + PLANE) = bq; *reinterpret_cast<gs_bf16x4*>(wr + GS_H + 2 * PLANE) = c;
      gs_split4(dnr, a, bq, c);
      *reinterpret_cast<gs_bf16x4*>(wr + 2 * GS_H) = a; *reinterpret_cast<gs_bf16x4*>(wr + 2 * GS_H + PLANE) = bq; *reinterpret_cast<gs_bf16x4*>(wr + 2 * GS_H + 2 * PLANE) = c;
    }
    GS_MARK(11);
    gs_lds_barrier();      // the only barrier of the step: the next step writes the other buffer
    GS_MARK(12);
    ldst(t - 1, snext);      // in flight under the step's MFMAs (requested at the top of the step instead, next to the step's stores: 64 -> 76 us per launch)
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[3] = {zero4, zero4, zero4};      // three chains of four k-blocks: a dependent MFMA does not wait for its predecessor
    const __bf16* rd = dq3 + cur * GS_QBUF + p * GS_LDQ + 8 * g;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const gs_bf16x8 xh = *reinterpret_cast<const gs_bf16x8*>(rd + 32 * j);
      const gs_bf16x8 xm = *reinterpret_cast<const gs_bf16x8*>(rd + 32 * j + PLANE);
      const gs_bf16x8 xl = *reinterpret_cast<const gs_bf16x8*>(rd + 32 * j + 2 * PLANE);
      acc[j % 3] = gs_mma6(wq[j][0], wq[j][1], wq[j][2], xh, xm, xl, acc[j % 3]);
    }
    dh4 = dprev + ((acc[0] + acc[1]) + acc[2]);
#ifdef INTEL_DEBUG
    asm volatile("" :: "v"(dh4));
#endif
    GS_MARK(13);
    cur ^= 1;
  };
  if (tmax > 0) {
    ldst(tmax - 1, snext);
    for (int t = tmax - 1; t >= 0; --t) step(t);
  }
}

// Exact-fp32 form (INTEL_GRU_SEQ=1: v_mfma_f32_16x16x4_f32, 96 MFMAs of 32 cycles per wave and step): lane (p, g) of wave w holds hidden unit
// 16w + p of sessions 4g .. 4g+3; two barriers per step.  Kept as the cross-check of the three-plane form.
// STASH: training (the backward reads GATES, GHN, HP); inference keeps only the final state
template <bool STASH>
__device__ __forceinline__ void gru_seq_fwd_f32(const float* __restrict__ GI, const float* __restrict__ Whh, const float* __restrict__ bhh,
                                                const int* __restrict__ len, int B, int T, float* __restrict__ HP, float* __restrict__ HCUR,
                                                float* __restrict__ GATES, float* __restrict__ GHN, const int* __restrict__ off,
                                                const int* __restrict__ order) {
  __shared__ __attribute__((aligned(16))) float hs[GS_ROWS * GS_LDH];
  __shared__ int slen[GS_ROWS], sses[GS_ROWS];
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b0 = blockIdx.x * GS_ROWS;
  const int unit = 16 * w + p;
  for (int i = tid; i < GS_ROWS * GS_LDH; i += 512) hs[i] = 0.f;      // h_0 = 0
  // workgroup slot i is session order[b0 + i] (sessions of similar length together) or b0 + i
  if (tid < GS_ROWS) {
    const int bsess = (b0 + tid < B) ? (order ? order[b0 + tid] : b0 + tid) : -1;
    sses[tid] = bsess;
    slen[tid] = bsess >= 0 ? min(len[bsess], T) : 0;
  }
  // B fragments: gate q, W_hh[q*128 + unit][k], k = 16 j + 4 g + s
  f32x4 wb[3][8];
  float bh[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    bh[q] = bhh[q * GS_H + unit];
    const float* wrow = Whh + (size_t)(q * GS_H + unit) * GS_H;
#pragma unroll
    for (int j = 0; j < 8; ++j) wb[q][j] = *reinterpret_cast<const f32x4*>(wrow + 16 * j + 4 * g);
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
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(hs + p * GS_LDH + 16 * j + 4 * g);
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[q] = gs_mma4(a, wb[q][j], acc[q]);
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
    for (int r = 0; r < 4; ++r) hs[(4 * g + r) * GS_LDH + unit] = h[r];
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

// PL: the three-plane form (default); otherwise the exact-fp32 form
template <bool PL, bool STASH = true>
__global__ __launch_bounds__(512, 1) void gru_seq_fwd_kernel(const float* __restrict__ GI, const float* __restrict__ Whh,
                                                             const float* __restrict__ bhh, const int* __restrict__ len, int B, int T,
                                                             float* __restrict__ HP, float* __restrict__ HCUR,
                                                             float* __restrict__ GATES, float* __restrict__ GHN,
                                                             const int* __restrict__ off, const int* __restrict__ order) {
  if constexpr (PL) gru_seq_fwd_pl<STASH>(GI, Whh, bhh, len, B, T, HP, HCUR, GATES, GHN, off, order);
  else gru_seq_fwd_f32<STASH>(GI, Whh, bhh, len, B, T, HP, HCUR, GATES, GHN, off, order);
}

__device__ __forceinline__ void gru_seq_bwd_f32(const float* __restrict__ dH0, const float* __restrict__ HP, const float* __restrict__ GATES,
                                                const float* __restrict__ GHN, const float* __restrict__ Whh, const int* __restrict__ len, int B,
                                                int T, float* __restrict__ dGI, float* __restrict__ dGH, const int* __restrict__ off,
                                                const int* __restrict__ order) {
  __shared__ __attribute__((aligned(16))) float ds[GS_ROWS * GS_LDG];
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
  // B fragments of dh_{t-1} += dGH_t W_hh: W_hh[k][unit], k = 16 j + 4 g + s
  f32x4 wb[24];
#pragma unroll
  for (int j = 0; j < 24; ++j)
#pragma unroll
    for (int s = 0; s < 4; ++s) wb[j][s] = Whh[(size_t)(16 * j + 4 * g + s) * GS_H + unit];
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
      float* dl = ds + (4 * g + r) * GS_LDG;
      dl[unit] = drp; dl[GS_H + unit] = dzp; dl[2 * GS_H + unit] = dnr;
    }
    gs_lds_barrier();
    load_stash(t - 1);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 24; ++j) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ds + p * GS_LDG + 16 * j + 4 * g);
      acc = gs_mma4(a, wb[j], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) dh[r] = dprev[r] + acc[r];
    gs_lds_barrier();                     // the next step rewrites the gate-gradient rows
  }
}

template <bool PL>
__global__ __launch_bounds__(512, 1) void gru_seq_bwd_kernel(const float* __restrict__ dH0, const float* __restrict__ HP,
                                                             const float* __restrict__ GATES, const float* __restrict__ GHN,
                                                             const float* __restrict__ Whh, const int* __restrict__ len, int B, int T,
                                                             float* __restrict__ dGI, float* __restrict__ dGH, const int* __restrict__ off,
                                                             const int* __restrict__ order) {
  if constexpr (PL) gru_seq_bwd_pl(dH0, HP, GATES, GHN, Whh, len, B, T, dGI, dGH, off, order);
  else gru_seq_bwd_f32(dH0, HP, GATES, GHN, Whh, len, B, T, dGI, dGH, off, order);
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
#ifdef INTEL_DEBUG
    if (INTEL_DEBUG_ENV("INTEL_GRU_DBG", 0)) {
      const int abl = INTEL_DEBUG_ENV("INTEL_GRU_ABL", 0);
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gru_abl), &abl, sizeof(abl));
      unsigned long long h[16];
      (void)hipStreamSynchronize(st);
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gru_dbg), sizeof(h));
      fprintf(stderr, "gru fwd [0-5] / previous bwd [8-13] phase clocks (sum over steps, B=%d T=%d):", B, T);
      for (int i = 0; i < 16; ++i) fprintf(stderr, " %llu", h[i]);
      fprintf(stderr, "\n");
      unsigned long long z[16] = {0};
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gru_dbg), z, sizeof(z));
    }
#endif
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
