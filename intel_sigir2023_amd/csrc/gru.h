// GRU4Rec encoder (models/GeneralSeq.py:64-78): one-layer GRU over the session history followed by a
// bias-free projection of each session's own last hidden state.
#pragma once
#include "common.h"

struct GruBufs {
  // packed weights
  float *pWih, *pWhh, *pWihT, *pWhhT, *pWout, *pWoutT;
  // activations (batch-major): GI [B,T,3H]  HP [B,T,H] (h_{t-1})  GATES [B,T,3H] (r,z,n)  GHN [B,T,H]
  float *GI, *HP, *GATES, *GHN, *GH, *HCUR;
  // backward
  float *dGI, *dGH, *dHa, *dHb, *dVEC;
  // the bias-free output projection vec = h_last Wout^T (GeneralSeq.py:76) is done by the caller (model.cpp: the session-head chains read
  // HCUR [B, H] and write dHa [B, H]): gru_fwd stops at HCUR, gru_bwd starts from dHa and leaves dWout alone
  bool ext_proj = false;
};
bool gru_ext_proj_supported(int dm, int Hd);
struct GruGrads { float *dWih, *dWhh, *dbih, *dbhh, *dWout; };

void gru_layout_packed(GruBufs& g, int dm, int Hd, char* base, size_t& off);
void gru_layout_act(GruBufs& g, int B, int T, int dm, int Hd, char* base, size_t& off);
int gru_pack(GruBufs& g, const float* Wih, const float* Whh, const float* Wout, int dm, int Hd, hipStream_t st);
// E0 [B*T, dm] -> vec written to out[b, col0:col0+dm]
int gru_fwd(GruBufs& g, const float* E0, int B, int T, int dm, int Hd, const int* len, const float* bih, const float* bhh,
            float* out, int ldo, int col0, hipStream_t st, const float* Whh = nullptr,       // Whh (raw [3H, H]): the one-kernel recurrence
            const int* off = nullptr, int rows = 0,      // off / rows: E0 holds only the valid history rows (session b: rows off[b] .. off[b] + len[b])
            const int* order = nullptr,                  // sessions ordered by length (IntelBatch.his_order)
            bool stash = true);                          // false (inference): the recurrence keeps no gate / state stash
bool gru_packed_supported(int Hd);
// dvec = dout[b, col0:col0+dm]; writes parameter grads (overwrite) and dE0 [B*T, dm]
int gru_bwd(GruBufs& g, const float* E0, int B, int T, int dm, int Hd, const int* len, const float* Whh, const float* bhh,
            const float* dout, int ldo, int col0, const GruGrads& gg, float* dE0, float* scratch, float* slabs,
            hipStream_t st, const int* off = nullptr, int rows = 0, const int* order = nullptr,
            struct ReduceQueue* q = nullptr);      // q: the weight gradients' slabs join the backward's batched reduction instead of three immediate ones
