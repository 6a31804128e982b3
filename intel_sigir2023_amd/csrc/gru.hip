// GRU4Rec encoder on MI355X: the input projection of all B*T rows is one fp32-MFMA GEMM; the
// recurrence runs T steps of (h W_hh^T GEMM over the batch + fused gate kernel); sessions shorter
// than t keep their state (the reference packs sequences by length, GeneralSeq.py:66-71).
// Gate order r, z, n and the update h' = (1-z) n + z h follow torch.nn.GRU.
#include "gru.h"
#include "kernels.h"
#include "session.h"

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

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

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
    const float n = tanhf(gi[2 * Hd + c] + r * ghn);
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

int gru_fwd(GruBufs& g, const float* E0, int B, int T, int dm, int Hd, const int* len, const float* bih, const float* bhh,
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
            hipStream_t st) {
  (void)Whh; (void)bhh; (void)scratch;
  int rc;
  // vec = HCUR Wout^T
  if (gg.dWout && (rc = launch_wgrad(dout + col0, ldo, g.HCUR, Hd, B, dm, Hd, gg.dWout, Hd, nullptr, 0, slabs, st))) return rc;
  GemmEpilogue e0;
  float *dH = g.dHa, *dHn = g.dHb;
  if ((rc = launch_gemm_rows(dout + col0, ldo, B, dm, g.pWoutT, Hd, dH, Hd, e0, st))) return rc;
  for (int t = T - 1; t >= 0; --t) {
    LAUNCH(gru_gate_bwd_kernel, dim3(cdiv(B * Hd, 256)), dim3(256), 0, st, dH, g.HP, g.GATES, g.GHN, len, B, T, Hd, t,
                       g.dGI, g.dGH, dHn);
    INTEL_CHECK_LAUNCH();
    // dh_{t-1} += dGH_t Whh
    GemmEpilogue ea;
    ea.accumulate = 1;
    if ((rc = launch_gemm_rows(g.dGH + (size_t)t * 3 * Hd, T * 3 * Hd, B, 3 * Hd, g.pWhhT, Hd, dHn, Hd, ea, st))) return rc;
    float* tmp = dH; dH = dHn; dHn = tmp;
  }
  const int rows = B * T;
  if (gg.dWih && (rc = launch_wgrad(g.dGI, 3 * Hd, E0, dm, rows, 3 * Hd, dm, gg.dWih, dm, gg.dbih, 0, slabs, st))) return rc;
  if (gg.dWhh && (rc = launch_wgrad(g.dGH, 3 * Hd, g.HP, Hd, rows, 3 * Hd, Hd, gg.dWhh, Hd, gg.dbhh, 0, slabs, st))) return rc;
  return launch_gemm_rows(g.dGI, 3 * Hd, rows, 3 * Hd, g.pWihT, dm, dE0, dm, e0, st);
}
