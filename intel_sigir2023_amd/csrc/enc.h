// Launchers of the fused BERT4Rec encoder kernels (enc.hip forward, enc_bwd.hip backward).  Packed history rows only
// (IntelBatch.his_off / hisitem_off); T = the batch's longest history (<= 32), dm = 128, two heads of 64.
#pragma once
#include "common.h"

struct ReduceQueue;

bool enc_fused_supported(int T, int dm, int heads);
// rows per tile window: session b belongs to tile off[b] / enc_tile_rows(T); a tile spans at most 64 rows
int enc_tile_rows(int T);
// tile_s[0 .. ntiles] (ntiles = ceil(rows / enc_tile_rows(T))): first session of every tile, B behind the last; once per batch
int launch_enc_tiles(const int* off, int B, int T, int rows, int* tile_s, hipStream_t st);

// one full transformer block over all packed rows (+ the next, last block's key / value projection when Wkv is given)
struct EncBlockFwd {
  const float* X = nullptr;            // [rows, dm]
  int rows = 0, B = 0, T = 0, dm = 0, heads = 0, train = 0;
  const int* off = nullptr;            // [B]
  const int* tile_s = nullptr;         // launch_enc_tiles
  const void *Wqkv = nullptr, *W1 = nullptr, *W2 = nullptr, *Wkv = nullptr;      // bf16 three-plane images (launch_pack_b3)
  const float *bqkv = nullptr, *b1 = nullptr, *b2 = nullptr, *bkv = nullptr;
  const float *g1 = nullptr, *be1 = nullptr, *g2 = nullptr, *be2 = nullptr;
  float* C = nullptr;                  // [rows, dm] LayerNorm1 output (required)
  float* out = nullptr;                // [rows, dm] block output (optional)
  float* xlast = nullptr;              // [B, dm] its row len-1 per session (optional)
  float* KV = nullptr;                 // [rows, 2 dm] (with Wkv)
  float *QKV = nullptr, *LSE = nullptr, *XH1 = nullptr, *RSTD1 = nullptr, *F1 = nullptr, *XH2 = nullptr, *RSTD2 = nullptr;   // training stash
};
int launch_enc_block_fwd(const EncBlockFwd& f, hipStream_t st);

// the pruned last block: one query row per session.  A workgroup takes ENC_LAST_SPB sessions: its chain of small dependent steps is
// latency-bound, so two half-filled 16-row tiles per CU beat one full one (B / 8 workgroups instead of B / 16)
#define ENC_LAST_SPB 8
struct EncLastFwd {
  const float* xlast = nullptr;        // [B, dm]
  const float* KV = nullptr;           // [rows, 2 dm]
  const int* off = nullptr; const int* len = nullptr;
  int B = 0, T = 0, dm = 0, heads = 0, train = 0;
  const void *Wq = nullptr, *W1 = nullptr, *W2 = nullptr;
  const float *bq = nullptr, *b1 = nullptr, *b2 = nullptr;
  const float *g1 = nullptr, *be1 = nullptr, *g2 = nullptr, *be2 = nullptr;
  float* out = nullptr; int ldo = 0;
  float *QL = nullptr, *PL = nullptr, *CL = nullptr, *XH1 = nullptr, *RSTD1 = nullptr, *F1 = nullptr, *XH2 = nullptr, *RSTD2 = nullptr;
};
int launch_enc_last_fwd(const EncLastFwd& f, hipStream_t st);

// ---- backward (enc_bwd.hip; the arithmetic mode follows gemm_planes() like the forward) ----------------------------------------
// the pruned last block: d(output vector) -> B-row gradients for the weight-gradient products, d(x_last), d[K' | V']
struct EncLastBwd {
  const float* dvec = nullptr; int ldv = 0;      // [B, ldv]
  const float* KV = nullptr; const int* off = nullptr; const int* len = nullptr;
  int B = 0, T = 0, dm = 0, heads = 0;
  const void *W2T = nullptr, *W1T = nullptr, *WqT = nullptr;      // images of the TRANSPOSED weights (dX = dY W)
  const float *g1 = nullptr, *g2 = nullptr;
  const float *XH2 = nullptr, *RSTD2 = nullptr, *F1 = nullptr, *XH1 = nullptr, *RSTD1 = nullptr, *PL = nullptr, *QL = nullptr;   // forward stash
  float *DZ2 = nullptr, *DF1 = nullptr, *DQ = nullptr, *DXL = nullptr;      // [B, dm]
  float* DKV = nullptr;                                                        // [rows, 2 dm]
  float *dg2 = nullptr, *db2 = nullptr, *dg1 = nullptr, *db1 = nullptr;      // LayerNorm parameter gradients (through the reduce queue)
  int acc_g2 = 0, acc_b2 = 0, acc_g1 = 0, acc_b1 = 0;
};
int launch_enc_last_bwd(const EncLastBwd& f, hipStream_t st, ReduceQueue* q);

// one full block: d(block output) (+ d(x_last) at the last rows) -> dZ2, dF1, dZ1, dQKV
struct EncBlockBwd {
  const float* dE = nullptr; const float* dxl = nullptr;
  int rows = 0, B = 0, T = 0, dm = 0, heads = 0;
  const int* off = nullptr; const int* tile_s = nullptr;
  const void *W2T = nullptr, *W1T = nullptr;
  const float *g1 = nullptr, *g2 = nullptr;
  const float *XH2 = nullptr, *RSTD2 = nullptr, *F1 = nullptr, *XH1 = nullptr, *RSTD1 = nullptr, *QKV = nullptr;
  float *DZ2 = nullptr, *DF1 = nullptr, *DZ1 = nullptr, *DQKV = nullptr;
  float *dg2 = nullptr, *db2 = nullptr, *dg1 = nullptr, *db1 = nullptr;
  int acc_g2 = 0, acc_b2 = 0, acc_g1 = 0, acc_b1 = 0;
};
int launch_enc_block_bwd(const EncBlockBwd& f, hipStream_t st, ReduceQueue* q);
// floats of reduction arena the two launches of one encoder take (LayerNorm partials)
size_t enc_bwd_slab_floats(int rows, int B, int T, int dm);
