// Host-side plan of the IntEL forward / backward: which kernels run, on which buffers, in which
// order.  Restates models/IntEL/IntEL.py:117-217 (forward) and derives its backward by hand (the
// reference relies on torch autograd, helpers/BaseRunner.py:288).  Everything is enqueued on one
// HIP stream without synchronisation or allocation, so a whole step is graph-capturable.
//
// Workspace layout (one caller-provided allocation, carved by `Layout`):
//   [ packed weights | forward activations (stash) | backward temporaries | reduction slabs ]
// Activations are fp32 row-major [rows, width]; "rows" are B*L candidate rows or B*T history rows.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <new>
#include <vector>

#include "../../include/intel_hip.h"
#include "kernels.h"
#include "session.h"
#include "gru.h"
#include "enc.h"
#include "chain.h"

#define MAX_TOWER_LAYERS 8

namespace {

struct Arena {
  char* base;
  size_t off;
  float* f(size_t n) {
    off = rup_sz(off, 256);
    float* p = reinterpret_cast<float*>(base + off);
    off += n * sizeof(float);
    return p;
  }
};

// tower parameter offsets relative to INTEL_P_I_WQ / INTEL_P_S_WQ
enum { T_WQ = 0, T_WK, T_WV, T_W1, T_B1, T_W2, T_B2, T_LNG, T_LNB };

struct TowerLayerBufs { float *QKV, *A, *LSE, *R1, *XH, *RSTD, *Xout, *DM; };   // DM: dropout mask / (1-p), only with dropout
struct TowerBufs {
  int d, pbase, xbase, feat_off;
  float* X0;
  TowerLayerBufs layer[MAX_TOWER_LAYERS];
  float *QV, *QK, *XBAR, *ATTW;
  float *pWqkv, *pW1, *pW2, *pWqkvT, *pW1T, *pW2T;
  float* b3WqkvT;           // bf16 three-plane image of pWqkvT (K = 3d > 128: GEMM on the bf16 pipe)
  float *b3Wqkv, *b3W1, *b3W2;   // images of the forward weights for the one-kernel tower layer (tower.hip)
  float *b3W1T, *b3W2T;          // ... and of the transposed feed-forward weights for the one-kernel middle of its backward (tower_bwd.hip)
  float *pXq, *pXqT, *pXk, *pXkT, *pXv, *pXvT;
  // --cross_attention 0
  float *MH, *MV, *pM0, *pM2, *pM2T, *pM0T;
};
struct EncBlockBufs {
  float *QKV, *A, *LSE, *C, *XH1, *RSTD1, *F1, *Eout, *XH2, *RSTD2;
  float *pWqkv, *pW1, *pW2, *pWqkvT, *pW1T, *pW2T, *bQKV;
  float* b3WqkvT;
  float *b3Wqkv, *b3W1, *b3W2;     // images of the forward weights for the fused encoder kernels (enc.hip)
  float *b3W1T, *b3W2T;            // ... and of the transposed feed-forward weights for their backward (enc_bwd.hip)
};
// the LAST BERT4Rec block only feeds row len-1 of its output forward (GeneralSeq.py:103-105): it is run
// "pruned" -- K/V for all rows, everything else for one row per session
struct EncLastBufs {
  float *KV, *XLAST, *QLAST, *PL, *OL, *CL, *XH1, *RSTD1, *F1, *XH2, *RSTD2;
  float *pWkvT, *pWqT;
  float* b3WkvT;
  float* b3WqT;
};
struct EncBufs {
  int T, dm, d_tab, pbase, predin_off;
  float* E0;
  // packed history (IntelBatch.his_off): ids / intent indices / intent rows of the valid positions only, position of each row
  int *pkIds, *pkIdx2, *rowT;
  int* tileS;               // fused encoder kernels: first session of every row tile (launch_enc_tiles)
  float* pkVec;
  bool pos_done;            // this forward folded the position embedding into the kernels that produce the input rows
  EncBlockBufs blk[INTEL_ENC_MAX_BLOCKS];
  EncLastBufs last;
  GruBufs gru;
};

// backward temporaries; two sets so that two independent branches (item tower || score tower, encoder ||
// item encoder) can run concurrently on two streams
struct Temps {
  float *dXa, *dXb, *dZ, *dF1, *dA, *dQKV, *DSUM, *SLABS, *dVB1, *dVB2, *dVB3, *ONEHOT, *dINT;
  float* dLB[5];            // [B, dmax] temporaries of the pruned last encoder block
};

struct Layout {
  int B, L, H, Hi, M;
  Temps tmp[4];            // 0 / 1: tower-sized sets of two concurrent branches; 2: session-history-encoder-sized; 3: score-tower-sized
  TowerBufs tw[2];          // 0 = item tower, 1 = score tower
  EncBufs enc[2];           // 0 = "encoder" (session history), 1 = "item_encoder"
  int F, Pin;               // width of the fusion feature / pred_layer input
  float *FEAT, *WV, *WPAD, *FEATFULL, *PREDIN, *LOGITS, *INTENTS;
  // weight_norm: WVN[s] / WPADN[s] = the valid-row / pad-row weight vectors after softmax s+1 (per-session weights);
  // WTN[s] = the per-item weights after softmax s+1 (cross_attention = 0 without pool_mean)
  float *WVN[2], *WPADN[2], *WTN[2];
  float *pInt, *pIntT, *pScore, *pWe, *pWePad, *pWeT, *pWePadT, *pPred, *pPredT;
  float *dFEAT, *dWV, *dWPAD, *dWT, *dFEATFULL, *dINTENT, *dLOGITS, *dPREDIN;
  float *dHINT, *dHB[2][3];   // B-row gradients of the session head kept until its (deferred) weight-gradient products have read them
  float *dXS;       // gradient w.r.t. the score tower output, parked between the two backward phases
  float *ONEHOT2;   // one-hot of the item-history intent indices (used on the main stream after a join)
  float *ARENA;     // slabs of the deferred reductions (ReduceQueue)
  size_t arena_floats;
  size_t total;
};

}  // namespace

struct IntelCtx {
  IntelDesc d;
  Layout lay;
  bool have_layout;
  bool fwd_done;
  int fB, fL, fH, fHi;
  const void* f_ws;
  unsigned char touched[INTEL_P_COUNT];
  // side streams: independent branches of the step (the two towers, the two sequence encoders) run
  // concurrently -- MFMA-bound GEMMs of one branch overlap the HBM-bound row kernels of another
  hipStream_t side[3];         // THREE side streams + the caller's = the four hardware queues the runtime multiplexes streams onto by default:
                               // a fifth active stream shares a queue with another one (its launches then serialise behind that stream's), and
                               // GPU_MAX_HW_QUEUES > 4 makes the whole step 1.5x slower (measured)
  hipEvent_t ev_fork, ev_join[3];
  hipEvent_t ev_x[4];          // the wide backward schedule: cross-attention backward of tower 0 / 1 done, d(intent) chain done, item-id table gradient complete
  hipStream_t table_stream = nullptr;      // intel_set_table_stream
  hipEvent_t table_wait_ev = nullptr;      // intel_set_table_wait_event: the NEXT forward's item-id gathers wait for it (one shot)
  hipEvent_t ev_tab = nullptr;             // ... made to wait on this event where the four-branch schedule is not taken (see backward_entry)
  int streams;      // 0 = not created, 1 = ready, -1 = disabled (INTEL_STREAMS=0)
  // weight-gradient / LayerNorm partial sums of a backward phase, reduced together when the phase ends
  ReduceQueue* rq;
  // nn.Dropout of the tower layers (IntEL.py:187,196) for the next training forward: p = 0 disables
  float drop_p;
  unsigned long long drop_seed;
  const float* drop_ext;       // optional 0/1 keep flags, item-tower layers then score-tower layers
  bool fwd_dropout;            // the stashed forward ran with dropout
  IntelLazyTable lazy;          // intel_set_lazy_table: the item-id table's Adam state (lazy.p == nullptr: off)
  int lazy_upto = 0;            // ... and the step the rows of a batch are brought up to ahead of the gathers
  unsigned char* iid_row_flags; // optional [item_num]: set to 1 for every item-id gradient row the backward adds into
  bool fused_tail[2];          // the stashed forward folded the last LayerNorm of tower t into the cross-attention pooling
  bool enc_packed[2];          // this forward ran encoder e on the valid history rows only (IntelBatch.his_off / hisitem_off)
  bool enc_fused[2];           // ... through the fused BERT4Rec kernels (enc.hip): packed rows, history <= 32, width 128
  // intel_set_params_unchanged: the packed weight images of the previous forward (same workspace, same batch shape) are reused
  bool params_unchanged = false, pack_ok = false, pack_train = false;
  const void* pack_ws = nullptr;
  int pack_shape[5] = {0, 0, 0, 0, 0};      // B, L, H, Hi, dropout layout
  bool enc32[2] = {false, false};      // this forward ran encoder e as the one-kernel 32-wide BERT4Rec encoder (tower32.hip: enc32_*): no stash
  bool tw32[2] = {false, false};       // this forward ran tower t as the one-kernel 32-wide tower (tower32.hip): no stash, the backward recomputes
  bool tw_bwdf[2] = {false, false};    // this forward left tower t's backward to the one-kernel middle (tower_bwd.hip): only A and the log-sum-exp are stashed
  bool tw_qkv16[2] = {false, false};   // this forward stored tower t's q/k/v stash as bf16 (bf16 mode; the backward reads it and writes dQKV the same way)
  int enc_rows[2];             // rows of encoder e: B * T, or the packed total
};

namespace {

// one weight vector per session (valid rows; pad rows have their own unless pool_mean) instead of one per item
inline bool per_session_weights(const IntelDesc& D) { return D.cross_attention || D.pool_mean; }
inline int enc_slot(int e, int off) { return INTEL_P_ENC0 + e * INTEL_ENC_STRIDE + off; }
inline int enc_blk_slot(int e, int l, int off) { return enc_slot(e, INTEL_ENC_BLOCK0 + l * INTEL_ENC_BLOCK_STRIDE + off); }

void make_layout(const IntelDesc& D, int B, int L, int H, int Hi, char* base, Layout& y, bool dropout = false) {
  Arena ar{base, 0};
  y.B = B; y.L = L; y.H = H; y.Hi = Hi; y.M = B * L;
  const int M = y.M;
  const int d_i = D.d_id + D.d_im, d_s = D.d_s;
  y.F = d_i + d_s + D.d_u + D.d_int;
  const int dm0 = D.d_c + D.d_int, dm1 = D.d_id + D.d_int;
  y.Pin = D.d_c + D.d_u + dm1 + dm0;
  const int I = D.intent_num, K = D.model_num;
  // ---- packed weights
  for (int t = 0; t < 2; ++t) {
    TowerBufs& w = y.tw[t];
    w.d = t == 0 ? d_i : d_s;
    w.pbase = t == 0 ? INTEL_P_I_WQ : INTEL_P_S_WQ;
    w.xbase = t == 0 ? INTEL_P_XI_WQ : INTEL_P_XS_WQ;
    w.feat_off = t == 0 ? 0 : d_i;
    const int d = w.d;
    w.pWqkv = ar.f(packed_floats(d, 3 * d));
    w.pW1 = ar.f(packed_floats(d, d));
    w.pW2 = ar.f(packed_floats(d, d));
    w.pWqkvT = ar.f(packed_floats(3 * rup(d, 16), d));
    w.b3WqkvT = ar.f(packed_b3_bytes(3 * rup(d, 16), d) / 4);
    w.b3Wqkv = ar.f(packed_b3_bytes(d, 3 * d) / 4);
    w.b3W1 = ar.f(packed_b3_bytes(d, d) / 4);
    w.b3W2 = ar.f(packed_b3_bytes(d, d) / 4);
    w.b3W1T = ar.f(packed_b3_bytes(d, d) / 4);
    w.b3W2T = ar.f(packed_b3_bytes(d, d) / 4);
    w.pW1T = ar.f(packed_floats(d, d));
    w.pW2T = ar.f(packed_floats(d, d));
    if (D.cross_attention) {
      w.pXq = ar.f(packed_floats(I, d));
      w.pXqT = ar.f(packed_floats(d, I));
      w.pXk = ar.f(packed_floats(d, d));
      w.pXkT = ar.f(packed_floats(d, d));
      w.pXv = ar.f(packed_floats(d, d));
      w.pXvT = ar.f(packed_floats(d, d));
    } else {
      w.pM0 = ar.f(packed_floats(I, D.q_size));
      w.pM0T = ar.f(packed_floats(D.q_size, I));
      w.pM2 = ar.f(packed_floats(D.q_size, d));
      w.pM2T = ar.f(packed_floats(d, D.q_size));
    }
  }
  y.pInt = ar.f(packed_floats(I, D.d_int));
  y.pIntT = ar.f(packed_floats(D.d_int, I));
  y.pScore = ar.f(packed_floats(K, d_s));
  y.pWe = ar.f(packed_floats(y.F, K));
  y.pWePad = ar.f(packed_floats(D.d_u + D.d_int, K));
  y.pWeT = ar.f(packed_floats(K, y.F));
  y.pWePadT = ar.f(packed_floats(K, D.d_u + D.d_int));
  y.pPred = ar.f(packed_floats(y.Pin, I));
  y.pPredT = ar.f(packed_floats(I, y.Pin));
  for (int e = 0; e < 2; ++e) {
    EncBufs& n = y.enc[e];
    n.T = e == 0 ? H : Hi;
    n.dm = e == 0 ? dm0 : dm1;
    n.d_tab = e == 0 ? D.d_c : D.d_id;
    n.pbase = enc_slot(e, 0);
    n.predin_off = e == 0 ? (D.d_c + D.d_u + dm1) : (D.d_c + D.d_u);
    const int dm = n.dm;
    if (D.encoder == INTEL_ENC_BERT4REC) {
      for (int l = 0; l < D.enc_layers; ++l) {
        EncBlockBufs& k = n.blk[l];
        k.pWqkv = ar.f(packed_floats(dm, 3 * dm));
        k.pW1 = ar.f(packed_floats(dm, dm));
        k.pW2 = ar.f(packed_floats(dm, dm));
        k.pWqkvT = ar.f(packed_floats(3 * rup(dm, 16), dm));
        k.b3WqkvT = ar.f(packed_b3_bytes(3 * rup(dm, 16), dm) / 4);
        k.pW1T = ar.f(packed_floats(dm, dm));
        k.pW2T = ar.f(packed_floats(dm, dm));
        k.bQKV = ar.f(3 * dm);
        k.b3Wqkv = ar.f(packed_b3_bytes(dm, 3 * dm) / 4);
        k.b3W1 = ar.f(packed_b3_bytes(dm, dm) / 4);
        k.b3W2 = ar.f(packed_b3_bytes(dm, dm) / 4);
        k.b3W1T = ar.f(packed_b3_bytes(dm, dm) / 4);
        k.b3W2T = ar.f(packed_b3_bytes(dm, dm) / 4);
      }
      n.last.pWkvT = ar.f(packed_floats(2 * rup(dm, 16), dm));
      n.last.b3WkvT = ar.f(packed_b3_bytes(2 * rup(dm, 16), dm) / 4);
      n.last.pWqT = ar.f(packed_floats(dm, dm));
      n.last.b3WqT = ar.f(packed_b3_bytes(dm, dm) / 4);
    } else {
      gru_layout_packed(n.gru, dm, D.gru_hidden, ar.base, ar.off);
    }
  }
  // ---- forward activations
  for (int t = 0; t < 2; ++t) {
    TowerBufs& w = y.tw[t];
    const size_t md = (size_t)M * w.d;
    w.X0 = ar.f(md);
    for (int l = 0; l < D.layers; ++l) {
      TowerLayerBufs& b = w.layer[l];
      b.QKV = ar.f(3 * md);
      b.A = ar.f(md);
      b.LSE = ar.f((size_t)B * D.heads * L);
      b.R1 = ar.f(md);
      b.XH = ar.f(md);
      b.RSTD = ar.f(M);
      b.Xout = ar.f(md);
      b.DM = dropout ? ar.f(md) : nullptr;
    }
    if (D.cross_attention) {
      w.QV = ar.f((size_t)B * w.d);
      w.QK = ar.f((size_t)B * w.d);
      w.XBAR = ar.f((size_t)B * w.d);
      w.ATTW = ar.f((size_t)B * L);
    } else {
      w.MH = ar.f((size_t)B * D.q_size);
      w.MV = ar.f((size_t)B * w.d);
      w.XBAR = D.pool_mean ? ar.f((size_t)B * w.d) : nullptr;
    }
  }
  const bool psw = per_session_weights(D);
  y.FEAT = ar.f((size_t)B * y.F);
  y.WV = ar.f((size_t)B * K);
  y.WPAD = ar.f((size_t)B * K);
  for (int s = 0; s < 2; ++s) {
    const bool on = s < D.weight_norm;
    y.WVN[s] = (on && psw) ? ar.f((size_t)B * K) : nullptr;
    y.WPADN[s] = (on && psw && !D.pool_mean) ? ar.f((size_t)B * K) : nullptr;
    y.WTN[s] = (on && !psw) ? ar.f((size_t)M * K) : nullptr;
  }
  y.FEATFULL = psw ? nullptr : ar.f((size_t)M * y.F);
  y.PREDIN = ar.f((size_t)B * y.Pin);
  y.LOGITS = ar.f((size_t)B * I);
  y.INTENTS = ar.f((size_t)B * I);
  size_t maxMD = (size_t)M * (d_i > d_s ? d_i : d_s);
  size_t maxLSE = attn_bwd_scratch_floats(B, L, d_i, D.heads);      // attention-backward scratch of the largest user
  if (attn_bwd_scratch_floats(B, L, d_s, D.heads) > maxLSE) maxLSE = attn_bwd_scratch_floats(B, L, d_s, D.heads);
  for (int e = 0; e < 2; ++e) {
    EncBufs& n = y.enc[e];
    const size_t rows = (size_t)B * n.T, md = rows * n.dm;
    n.E0 = ar.f(md);
    n.pkIds = reinterpret_cast<int*>(ar.f(rows));
    n.pkIdx2 = reinterpret_cast<int*>(ar.f(rows));
    n.rowT = reinterpret_cast<int*>(ar.f(rows));
    n.tileS = reinterpret_cast<int*>(ar.f(rows + 2));
    n.pkVec = ar.f(rows * I);
    if (md > maxMD) maxMD = md;
    if (D.encoder == INTEL_ENC_BERT4REC) {
      {
        EncLastBufs& q = n.last;
        const size_t bd = (size_t)B * n.dm;
        q.KV = ar.f(2 * md);
        q.XLAST = ar.f(bd); q.QLAST = ar.f(bd); q.OL = ar.f(bd); q.CL = ar.f(bd); q.XH1 = ar.f(bd); q.F1 = ar.f(bd); q.XH2 = ar.f(bd);
        q.RSTD1 = ar.f(B); q.RSTD2 = ar.f(B);
        q.PL = ar.f((size_t)B * D.enc_heads * n.T);
      }
      for (int l = 0; l + 1 < D.enc_layers; ++l) {
        EncBlockBufs& k = n.blk[l];
        k.QKV = ar.f(3 * md);
        k.A = ar.f(md);
        k.LSE = ar.f((size_t)B * D.enc_heads * n.T);
        k.C = ar.f(md);
        k.XH1 = ar.f(md);
        k.RSTD1 = ar.f(rows);
        k.F1 = ar.f(md);
        k.Eout = ar.f(md);
        k.XH2 = ar.f(md);
        k.RSTD2 = ar.f(rows);
      }
      if (attn_bwd_scratch_floats(B, n.T, n.dm, D.enc_heads) > maxLSE) maxLSE = attn_bwd_scratch_floats(B, n.T, n.dm, D.enc_heads);
    } else {
      gru_layout_act(n.gru, B, n.T, n.dm, D.gru_hidden, ar.base, ar.off);
    }
  }
  // ---- backward temporaries
  y.dFEAT = ar.f((size_t)B * y.F);
  y.dWV = ar.f((size_t)B * K);
  y.dWPAD = ar.f((size_t)B * K);
  y.dWT = psw ? nullptr : ar.f((size_t)M * K);
  y.dFEATFULL = psw ? nullptr : ar.f((size_t)M * y.F);
  y.dINTENT = ar.f((size_t)B * I);
  y.dLOGITS = ar.f((size_t)B * I);
  y.dPREDIN = ar.f((size_t)B * y.Pin);
  y.dXS = ar.f((size_t)M * d_s);
  y.ONEHOT2 = ar.f((size_t)B * Hi * I);
  const int dmax = d_i > d_s ? (d_i > dm0 ? (d_i > dm1 ? d_i : dm1) : (dm0 > dm1 ? dm0 : dm1))
                             : (d_s > dm0 ? (d_s > dm1 ? d_s : dm1) : (dm0 > dm1 ? dm0 : dm1));
  const int vmax = dmax > I ? (dmax > D.q_size ? dmax : D.q_size) : (I > D.q_size ? I : D.q_size);
  // slabs: the largest weight-gradient / LayerNorm / column-sum reduction
  size_t maxNK = (size_t)dmax * dmax;
  auto upd = [&](size_t v) { if (v > maxNK) maxNK = v; };
  upd((size_t)I * y.Pin); upd((size_t)K * y.F); upd((size_t)D.d_int * I); upd((size_t)d_s * K); upd((size_t)D.q_size * I);
  upd((size_t)dmax * D.q_size); upd((size_t)dmax * I);
  const int gh = D.gru_hidden;
  if (D.encoder == INTEL_ENC_GRU4REC) { upd((size_t)3 * gh * dmax); upd((size_t)3 * gh * gh); upd((size_t)dmax * gh); }
  size_t maxN = (size_t)(dmax > I ? dmax : I);
  if ((size_t)3 * gh > maxN && D.encoder == INTEL_ENC_GRU4REC) maxN = 3 * gh;
  if ((size_t)y.Pin > maxN) maxN = y.Pin;
  if ((size_t)y.F > maxN) maxN = y.F;
  size_t rowsmax = (size_t)M;
  if ((size_t)B * H > rowsmax) rowsmax = (size_t)B * H;
  if ((size_t)B * Hi > rowsmax) rowsmax = (size_t)B * Hi;
  size_t slab = 512 * (maxNK + maxN);
  size_t lnslab = (size_t)cdiv((int)rowsmax, 64) * 2 * maxN;      // >= the LayerNorm-backward slab count
  if (lnslab > slab) slab = lnslab;
  const int Tm = H > Hi ? H : Hi;
  const int Rm = I > Tm ? I : Tm;
  y.dHINT = ar.f((size_t)B * vmax);
  for (int t = 0; t < 2; ++t)
    for (int j = 0; j < 3; ++j) y.dHB[t][j] = ar.f((size_t)B * vmax);
  for (int i = 0; i < 4; ++i) {
    Temps& t = y.tmp[i];
    // set 2 only ever holds the session-history encoder's backward (rows B*H, width dm0), set 3 the score tower's (rows M, width d_s)
    const size_t mds = i < 2 ? maxMD : (i == 2 ? rup_sz((size_t)B * H * dm0, 64) : rup_sz((size_t)M * d_s, 64));
    t.dXa = ar.f(mds);
    t.dXb = ar.f(mds);
    t.dZ = ar.f(mds);
    t.dF1 = ar.f(mds);
    t.dA = ar.f(mds);
    t.dQKV = ar.f(3 * mds);
    t.DSUM = ar.f(maxLSE);
    t.dVB1 = ar.f((size_t)B * vmax);
    t.dVB2 = ar.f((size_t)B * vmax);
    t.dVB3 = ar.f((size_t)B * vmax);
    for (int j = 0; j < 5; ++j) t.dLB[j] = ar.f((size_t)B * vmax);
    t.ONEHOT = ar.f((size_t)B * Tm * Rm);
    t.dINT = ar.f((size_t)B * I);
    t.SLABS = ar.f(slab + 1024);
  }
  // arena of the deferred reductions: every weight gradient of one backward keeps its slabs until the flush
  {
    auto Wf = [&](size_t rows, size_t N, size_t Kk) { return rup_sz(wgrad_slab_floats((int)rows, (int)N, (int)Kk), 64); };
    auto Lf = [&](size_t rows, size_t N) { return 2 * rup_sz(ln_bwd_slab_floats((int)rows, (int)N), 64); };
    const size_t mx = (size_t)(dmax > I ? dmax : I), qs = (size_t)D.q_size;
    size_t a = 0;
    a += (size_t)D.layers * (5 * Wf(M, d_i, d_i) + Lf(M, d_i));
    a += (size_t)D.layers * (5 * Wf(M, d_s, d_s) + Lf(M, d_s));
    if (D.encoder == INTEL_ENC_BERT4REC) {
      a += (size_t)D.enc_layers * (6 * Wf((size_t)B * H, dm0, dm0) + Lf((size_t)B * H, dm0)) + Lf((size_t)B * H, dm0) + Wf((size_t)B * H, H, dm0);
      a += rup_sz((size_t)512 * H * dm0, 64) + rup_sz((size_t)512 * Hi * dm1, 64);      // position-embedding gradient tables
      a += 2 * D.enc_layers * (rup_sz(enc_bwd_slab_floats(B * H, B, H > 32 ? 32 : H, dm0), 64) + rup_sz(enc_bwd_slab_floats(B * Hi, B, Hi > 32 ? 32 : Hi, dm1), 64));   // fused encoder backward: LayerNorm partials
      a += (size_t)D.enc_layers * (6 * Wf((size_t)B * Hi, dm1, dm1) + Lf((size_t)B * Hi, dm1)) + Lf((size_t)B * Hi, dm1) + Wf((size_t)B * Hi, Hi, dm1);
    }
    if (D.encoder == INTEL_ENC_GRU4REC) {       // the three weight gradients of each GRU encoder (input, hidden, output projection)
      a += Wf((size_t)B * H, 3 * (size_t)gh, dm0) + Wf((size_t)B * H, 3 * (size_t)gh, gh) + Wf(B, dm0, gh);
      a += Wf((size_t)B * Hi, 3 * (size_t)gh, dm1) + Wf((size_t)B * Hi, 3 * (size_t)gh, gh) + Wf(B, dm1, gh);
    }
    a += 2 * Wf(psw ? B : M, K, y.F);                                     // fusion weights (+ pad rows)
    a += Wf(B, D.d_int, I) + Wf(B, I, y.Pin) + Wf(M, d_s, K);             // intent embedding, predictor, score embedding
    a += Wf((size_t)B * H, D.d_int, I) + Wf((size_t)B * Hi, D.d_int, I);  // shared intent embedding from the histories
    a += 6 * Wf(B, mx, mx);                                               // cross attention q / k / v of both towers
    a += 2 * rup_sz(xatt_ln_bwd_slab_floats(B, (int)dmax), 64);           // LayerNorm partials of the fused tower tails
    a += 2 * rup_sz(tower32_slab_floats(B), 64);                          // parameter-gradient slabs of the one-kernel 32-wide towers
    if (tower_bwd_fused_supported(L, d_i, D.heads)) a += (size_t)D.layers * rup_sz(tower_bwd_slab_floats(B, d_i), 64);      // ... of the one-kernel backward middles (tower_bwd.hip)
    if (tower_bwd_fused_supported(L, d_s, D.heads)) a += (size_t)D.layers * rup_sz(tower_bwd_slab_floats(B, d_s), 64);
    a += (size_t)D.layers * 2 * (rup_sz(linear_bwd_pair_slab_floats(M, d_i), 64) + rup_sz(linear_bwd_pair_slab_floats(M, d_s), 64));      // ... of the one-pass linear backwards (pair.hip)
    a += (size_t)D.layers * (rup_sz(linear_bwd_qkv_slab_floats(M, d_i, 3), 64) + rup_sz(linear_bwd_qkv_slab_floats(M, d_s, 3), 64));
    if (D.encoder == INTEL_ENC_BERT4REC)
      a += (size_t)D.enc_layers * (rup_sz(linear_bwd_qkv_slab_floats(B * H, dm0, 3), 64) + rup_sz(linear_bwd_qkv_slab_floats(B * Hi, dm1, 3), 64));
    if (D.encoder == INTEL_ENC_BERT4REC && D.enc_layers <= 2 && (dm0 == 32 || dm1 == 32)) a += 2 * rup_sz(enc32_slab_floats(B, D.enc_layers), 64);      // ... and encoders
    a += (size_t)cdiv(B, 16) * (rup_sz((size_t)K * y.F + K + (size_t)D.d_int * I + D.d_int + (size_t)d_i * d_i + (size_t)d_s * d_s, 64) + 64 +
                                rup_sz((size_t)d_i * d_i + (size_t)d_s * d_s + (size_t)(d_i + d_s) * I + (size_t)I * y.Pin + I, 64) + 64 +
                                rup_sz((size_t)(dm0 + dm1) * (size_t)gh, 64));      // session-head chains (chain.hip)
    a += 2 * (Wf(B, mx, qs) + Wf(B, qs, mx));                             // gate MLPs (cross_attention = 0)
    y.arena_floats = a + 4096;
    y.ARENA = ar.f(y.arena_floats);
  }
  y.total = rup_sz(ar.off, 256) + 256;
}

// ------------------------------------------------------------------------------------------
struct Run {
  IntelCtx* ctx;
  const IntelDesc& D;
  Layout& y;
  const void* const* params;
  void* const* grads;
  const IntelBatch* bt;
  hipStream_t st;
  int rc;
  Temps* T;
  int train;     // 0: inference forward -- the pure-stash outputs (x-hat, rstd) are not written
  const float* P(int slot) const { return static_cast<const float*>(params[slot]); }
  float* G(int slot) const { return static_cast<float*>(grads[slot]); }
  // 0 the first time a gradient slot is written in this backward, 1 afterwards (accumulate)
  int acc(int slot) {
    int a = ctx->touched[slot];
    ctx->touched[slot] = 1;
    return a;
  }
  bool ok(int r) {
    if (r != 0 && rc == 0) rc = r;
    return rc == 0;
  }
};

bool ensure_streams(IntelCtx* c) {
  if (c->streams == 0) {
    const char* e = getenv("INTEL_STREAMS");
    if (e && e[0] == '0') { c->streams = -1; return false; }
    bool ok = true;
    // (tried for the encoder / tower branches and dropped: streams restricted to disjoint CU sets with
    // hipExtStreamCreateWithCUMask and a higher stream priority for the encoder branches -- both open extra hardware queues,
    // and more than four active queues cost 1.5-4x)
    // (round 6, the same three side streams created WITH priorities -- encoder branch high and / or tower branches low, no extra queue: 3.26 ms per
    // step -> 3.29 - 3.41, every combination loses)
    for (int i = 0; i < 3; ++i) {
      ok = ok && hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking) == hipSuccess;
      ok = ok && hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming) == hipSuccess;
    }
    ok = ok && hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 4; ++i) ok = ok && hipEventCreateWithFlags(&c->ev_x[i], hipEventDisableTiming) == hipSuccess;
    c->streams = ok ? 1 : -1;
  }
  return c->streams == 1;
}
// side streams 0..n-1 wait for everything enqueued so far on the main stream
void fork_streams(Run& r, int n) {
  IntelCtx* c = r.ctx;
  if (!ensure_streams(c)) return;
  r.ok((int)hipEventRecord(c->ev_fork, r.st));
  for (int i = 0; i < n; ++i) r.ok((int)hipStreamWaitEvent(c->side[i], c->ev_fork, 0));
}
// the main stream waits for side streams 0..n-1
void join_streams(Run& r, int n) {
  IntelCtx* c = r.ctx;
  if (c->streams != 1) return;
  for (int i = 0; i < n; ++i) {
    r.ok((int)hipEventRecord(c->ev_join[i], c->side[i]));
    r.ok((int)hipStreamWaitEvent(r.st, c->ev_join[i], 0));
  }
}
// stream `to` waits for everything enqueued so far on side stream i (no-op when concurrency is off: one stream)
void wait_side(Run& r, int i, hipStream_t to) {
  IntelCtx* c = r.ctx;
  if (c->streams != 1) return;
  r.ok((int)hipEventRecord(c->ev_join[i], c->side[i]));
  r.ok((int)hipStreamWaitEvent(to, c->ev_join[i], 0));
}
// a Run on side stream i (or on the main stream when concurrency is off) with temporaries set t
Run branch(Run& r, int side, int t) {
  Run b = r;
  if (r.ctx->streams == 1 && side >= 0) b.st = r.ctx->side[side];
  b.T = &r.y.tmp[t];
  b.rc = 0;
  return b;
}

#define RUN(expr)             \
  do {                        \
    if (!r.ok(expr)) return;  \
  } while (0)

bool need(Run& r, int slot, const char* name) {
  if (r.params[slot] == nullptr) {
    intel_set_error("missing parameter %s (slot %d)", name, slot);
    r.rc = INTEL_E_ARG;
    return false;
  }
  return true;
}

void lin(Run& r, const float* A, int lda, int M, int K, const float* Bp, int N, float* C, int ldc, const GemmEpilogue& ep) {
  RUN(launch_gemm_rows(A, lda, M, K, Bp, N, C, ldc, ep, r.st));
}

// dW (+)= dY^T X for parameter slots (w_slot, b_slot or -1)
void wgrad(Run& r, const float* dY, int lddy, const float* X, int ldx, int M, int N, int K, int w_slot, int b_slot, int io16 = 0) {
  float* dW = r.G(w_slot);
  float* db = b_slot >= 0 ? r.G(b_slot) : nullptr;
  if (!dW) return;
  int a = r.acc(w_slot);
  if (b_slot >= 0) { int ab = r.acc(b_slot); (void)ab; }
  if (!io16 && smallk_supported(N, K) && K <= 32 && M >= 4096 && (K % 16 != 0)) {     // tiny input width: VALU kernel (smallk.hip)
    RUN(launch_wgrad_smallk(dY, lddy, X, ldx, M, N, K, dW, K, db, a, nullptr, r.st, r.ctx->rq));
    return;
  }
  RUN(launch_wgrad(dY, lddy, X, ldx, M, N, K, dW, K, db, a, nullptr, r.st, r.ctx->rq, nullptr, io16));
}

// the weight gradients of n linears that share the input X and whose output gradients sit side by side in dY
// (fused q/k/v): one product over the stacked columns
void wgrad_split(Run& r, const float* dY, int lddy, const float* X, int ldx, int M, int Nsub, int K, int n, const int* w_slots,
                 const int* b_slots, int dy_bf16 = 0) {
  WgradSplit sp;
  sp.n = n;
  sp.dy_bf16 = dy_bf16;
  bool any = false, all = true;
  for (int p = 0; p < n; ++p) {
    sp.dW[p] = r.G(w_slots[p]);
    sp.db[p] = b_slots[p] >= 0 ? r.G(b_slots[p]) : nullptr;
    any = any || sp.dW[p];
    all = all && sp.dW[p];
  }
  bool bias_uniform = true;
  for (int p = 1; p < n; ++p) bias_uniform = bias_uniform && ((sp.db[p] != nullptr) == (sp.db[0] != nullptr));
  if (!any) return;
  if (dy_bf16 && (!all || !bias_uniform || smallk_supported(Nsub, K))) {
    r.ok(INTEL_E_ARG);
    intel_set_error("wgrad_split: a bf16-stored dY needs all the stacked weights trainable");
    return;
  }
  if (!all || !bias_uniform || smallk_supported(Nsub, K)) {      // mixed cases: one product per weight
    for (int p = 0; p < n; ++p) wgrad(r, dY + p * Nsub, lddy, X, ldx, M, Nsub, K, w_slots[p], b_slots[p]);
    return;
  }
  for (int p = 0; p < n; ++p) {
    sp.acc[p] = r.acc(w_slots[p]);
    if (b_slots[p] >= 0) r.acc(b_slots[p]);
  }
  RUN(launch_wgrad(dY, lddy, X, ldx, M, n * Nsub, K, nullptr, K, nullptr, 0, nullptr, r.st, r.ctx->rq, &sp));
}

// the backward of a fused q/k/v (or k/v) projection in ONE pass where pair.hip covers the shape: dXout = dY WT (+ res), dW[p] / db[p] (+)= ...; returns false
// when the shape is not covered (the caller then runs wgrad_split + the K > 128 row GEMM)
bool qkv_bwd_one_pass(Run& r, const float* dY, const float* X, const float* res, int M, int d, int nb, const void* WT_b3, float* dXout, const int* w_slots,
                      const int* b_slots) {
  if (gemm_planes() != 3 || !linear_bwd_qkv_supported(M, d, nb)) return false;
  float *gw[3] = {nullptr, nullptr, nullptr}, *gb[3] = {nullptr, nullptr, nullptr};
  int ac[3] = {0, 0, 0};
  bool any_b = false;
  for (int p = 0; p < nb; ++p) {
    gw[p] = r.G(w_slots[p]);
    gb[p] = b_slots[p] >= 0 ? r.G(b_slots[p]) : nullptr;
    if (!gw[p]) return false;      // (a frozen weight: the separate kernels handle the mixed case)
    any_b = any_b || gb[p];
  }
  for (int p = 0; p < nb; ++p) {
    ac[p] = r.acc(w_slots[p]);
    if (b_slots[p] >= 0) r.acc(b_slots[p]);
  }
  r.ok(launch_linear_bwd_qkv(dY, nb * d, X, d, res, d, M, d, nb, WT_b3, dXout, d, gw, any_b ? gb : nullptr, ac, r.ctx->rq, r.st));
  return true;
}

// ---- weight packing -------------------------------------------------------------------------
void pack_tower(Run& r, TowerBufs& w) {
  const int d = w.d, pb = w.pbase;
  const int nt = rup(d, 16) / 16;
  for (int j = 0; j < 3; ++j) {
    RUN(launch_pack_b(r.P(pb + T_WQ + j), d, d, d, 0, w.pWqkv, j * nt, r.st));
    // dX = dQKV @ [Wq;Wk;Wv]: reduction index = 3 stacked output dims (each padded to 16)
    RUN(launch_pack_b(r.P(pb + T_WQ + j), d, d, d, 1, w.pWqkvT, 0, r.st, j * nt, 3 * nt));
  }
  RUN(launch_pack_b(r.P(pb + T_W1), d, d, d, 0, w.pW1, 0, r.st));
  RUN(launch_pack_b(r.P(pb + T_W2), d, d, d, 0, w.pW2, 0, r.st));
  RUN(launch_pack_b(r.P(pb + T_W1), d, d, d, 1, w.pW1T, 0, r.st));
  RUN(launch_pack_b(r.P(pb + T_W2), d, d, d, 1, w.pW2T, 0, r.st));
  const int I = r.D.intent_num;
  if (r.D.cross_attention) {
    const int xb = w.xbase;
    RUN(launch_pack_b(r.P(xb + 0), I, I, d, 0, w.pXq, 0, r.st));    // QV = intent Wq^T
    RUN(launch_pack_b(r.P(xb + 0), I, d, I, 1, w.pXqT, 0, r.st));   // dintent = dQV Wq
    RUN(launch_pack_b(r.P(xb + 1), d, d, d, 1, w.pXkT, 0, r.st));   // QK = QV Wk
    RUN(launch_pack_b(r.P(xb + 1), d, d, d, 0, w.pXk, 0, r.st));    // dQV = dQK Wk^T
    RUN(launch_pack_b(r.P(xb + 2), d, d, d, 0, w.pXv, 0, r.st));    // pooled = xbar Wv^T
    RUN(launch_pack_b(r.P(xb + 2), d, d, d, 1, w.pXvT, 0, r.st));   // dxbar = dpooled Wv
  } else {
    const int mb = (w.pbase == INTEL_P_I_WQ) ? INTEL_P_MI_W0 : INTEL_P_MS_W0;
    const int q = r.D.q_size;
    RUN(launch_pack_b(r.P(mb + 0), I, I, q, 0, w.pM0, 0, r.st));
    RUN(launch_pack_b(r.P(mb + 0), I, q, I, 1, w.pM0T, 0, r.st));
    RUN(launch_pack_b(r.P(mb + 2), q, q, d, 0, w.pM2, 0, r.st));
    RUN(launch_pack_b(r.P(mb + 2), q, d, q, 1, w.pM2T, 0, r.st));
  }
}

void pack_fp32(Run& r);

void pack_all(Run& r) {
  const IntelDesc& D = r.D;
  Layout& y = r.y;
  pack_jobs_reset();
  pack_fp32(r);
  if (r.rc) return;
  // bf16 three-plane images of the K > 128 weights (data gradients through the fused q/k/v weights)
  for (int t = 0; t < 2; ++t) {
    TowerBufs& w = y.tw[t];
    if (!(D.layers > 0 && tower32_supported(y.L, w.d, D.heads, D.layers, r.train)))      // (tower32.hip reads the raw weights)
      RUN(launch_pack_b3(w.pWqkvT, 3 * rup(w.d, 16), w.d, w.b3WqkvT, r.st));
    const bool bwdf = r.train && D.layers > 0 && tower_bwd_fused_supported(y.L, w.d, D.heads) && tower_bwd_fused_wanted(w.d);      // (tower_fwd decides with the same predicate + no dropout)
    const bool fwdf = tower_fused_supported(y.L, w.d, D.heads) && tower_fused_wanted(r.train, w.d);
    if (fwdf || bwdf) {
      RUN(launch_pack_b3(w.pWqkv, w.d, 3 * w.d, w.b3Wqkv, r.st));
      RUN(launch_pack_b3(w.pW1, w.d, w.d, w.b3W1, r.st));
    }
    if (fwdf) RUN(launch_pack_b3(w.pW2, w.d, w.d, w.b3W2, r.st));
    // the one-pass linear backward (pair.hip) streams the transposed feed-forward weights as images too (tower_bwd decides with the same predicate)
    const bool pairf = r.train && D.layers > 0 && !bwdf && gemm_planes() == 3 && linear_bwd_pair_supported(y.M, w.d) &&
                       !tower32_supported(y.L, w.d, D.heads, D.layers, r.train);
    if (bwdf || pairf) {
      RUN(launch_pack_b3(w.pW1T, w.d, w.d, w.b3W1T, r.st));
      RUN(launch_pack_b3(w.pW2T, w.d, w.d, w.b3W2T, r.st));
    }
  }
  if (D.encoder == INTEL_ENC_BERT4REC) {
    for (int e = 0; e < 2; ++e) {
      EncBufs& n = y.enc[e];
      for (int l = 0; l + 1 < D.enc_layers; ++l) RUN(launch_pack_b3(n.blk[l].pWqkvT, 3 * rup(n.dm, 16), n.dm, n.blk[l].b3WqkvT, r.st));
      RUN(launch_pack_b3(n.last.pWkvT, 2 * rup(n.dm, 16), n.dm, n.last.b3WkvT, r.st));
      if (D.enc_layers >= 2 && enc_fused_supported(n.T, n.dm, D.enc_heads)) {      // the fused encoder kernels stream the forward weights as images
        for (int l = 0; l < D.enc_layers; ++l) {
          RUN(launch_pack_b3(n.blk[l].pWqkv, n.dm, 3 * n.dm, n.blk[l].b3Wqkv, r.st));
          RUN(launch_pack_b3(n.blk[l].pW1, n.dm, n.dm, n.blk[l].b3W1, r.st));
          RUN(launch_pack_b3(n.blk[l].pW2, n.dm, n.dm, n.blk[l].b3W2, r.st));
          if (r.train) {
            RUN(launch_pack_b3(n.blk[l].pW1T, n.dm, n.dm, n.blk[l].b3W1T, r.st));
            RUN(launch_pack_b3(n.blk[l].pW2T, n.dm, n.dm, n.blk[l].b3W2T, r.st));
          }
        }
        if (r.train) RUN(launch_pack_b3(n.last.pWqT, n.dm, n.dm, n.last.b3WqT, r.st));
      }
    }
  }
  RUN(pack_b3_flush(r.st));
  RUN(vec_copy_flush(r.st));
}

void pack_fp32(Run& r) {
  const IntelDesc& D = r.D;
  Layout& y = r.y;
  const int I = D.intent_num, K = D.model_num;
  struct BatchGuard {     // every launch_pack_b below is recorded and issued as one job table
    Run& r;
    explicit BatchGuard(Run& rr) : r(rr) { pack_batch_begin(); }
    ~BatchGuard() { r.ok(pack_batch_end(r.st)); }
  } guard(r);
  for (int t = 0; t < 2; ++t) {
    pack_tower(r, y.tw[t]);
    if (r.rc) return;
  }
  const int off = y.F - (D.d_u + D.d_int);
  RUN(launch_pack_b(r.P(INTEL_P_INTENT_W), I, I, D.d_int, 0, y.pInt, 0, r.st));
  RUN(launch_pack_b(r.P(INTEL_P_INTENT_W), I, D.d_int, I, 1, y.pIntT, 0, r.st));
  RUN(launch_pack_b(r.P(INTEL_P_SCORE_W), K, K, D.d_s, 0, y.pScore, 0, r.st));
  RUN(launch_pack_b(r.P(INTEL_P_WE_W), y.F, y.F, K, 0, y.pWe, 0, r.st));
  RUN(launch_pack_b(r.P(INTEL_P_WE_W) + off, y.F, D.d_u + D.d_int, K, 0, y.pWePad, 0, r.st));
  RUN(launch_pack_b(r.P(INTEL_P_WE_W), y.F, K, y.F, 1, y.pWeT, 0, r.st));
  RUN(launch_pack_b(r.P(INTEL_P_WE_W) + off, y.F, K, D.d_u + D.d_int, 1, y.pWePadT, 0, r.st));
  RUN(launch_pack_b(r.P(INTEL_P_PRED_W), y.Pin, y.Pin, I, 0, y.pPred, 0, r.st));
  RUN(launch_pack_b(r.P(INTEL_P_PRED_W), y.Pin, I, y.Pin, 1, y.pPredT, 0, r.st));
  for (int e = 0; e < 2; ++e) {
    EncBufs& n = y.enc[e];
    const int dm = n.dm, nt = rup(dm, 16) / 16;
    if (D.encoder == INTEL_ENC_BERT4REC) {
      for (int l = 0; l < D.enc_layers; ++l) {
        EncBlockBufs& k = n.blk[l];
        for (int j = 0; j < 3; ++j) {
          const float* W = r.P(enc_blk_slot(e, l, INTEL_ENC_WQ + 2 * j));
          RUN(launch_pack_b(W, dm, dm, dm, 0, k.pWqkv, j * nt, r.st));
          RUN(launch_pack_b(W, dm, dm, dm, 1, k.pWqkvT, 0, r.st, j * nt, 3 * nt));
          RUN(launch_vec_copy(r.P(enc_blk_slot(e, l, INTEL_ENC_BQ + 2 * j)), k.bQKV + j * dm, dm, r.st));
        }
        RUN(launch_pack_b(r.P(enc_blk_slot(e, l, INTEL_ENC_W1)), dm, dm, dm, 0, k.pW1, 0, r.st));
        RUN(launch_pack_b(r.P(enc_blk_slot(e, l, INTEL_ENC_W2)), dm, dm, dm, 0, k.pW2, 0, r.st));
        RUN(launch_pack_b(r.P(enc_blk_slot(e, l, INTEL_ENC_W1)), dm, dm, dm, 1, k.pW1T, 0, r.st));
        RUN(launch_pack_b(r.P(enc_blk_slot(e, l, INTEL_ENC_W2)), dm, dm, dm, 1, k.pW2T, 0, r.st));
      }
      {   // last block, pruned: dX = dKV @ [Wk;Wv] and dXlast = dQ @ Wq
        const int l = D.enc_layers - 1;
        RUN(launch_pack_b(r.P(enc_blk_slot(e, l, INTEL_ENC_WK)), dm, dm, dm, 1, n.last.pWkvT, 0, r.st, 0, 2 * nt));
        RUN(launch_pack_b(r.P(enc_blk_slot(e, l, INTEL_ENC_WV)), dm, dm, dm, 1, n.last.pWkvT, 0, r.st, nt, 2 * nt));
        RUN(launch_pack_b(r.P(enc_blk_slot(e, l, INTEL_ENC_WQ)), dm, dm, dm, 1, n.last.pWqT, 0, r.st));
      }
    } else {
      RUN(gru_pack(n.gru, r.P(enc_slot(e, INTEL_ENC_GRU_WIH)), r.P(enc_slot(e, INTEL_ENC_GRU_WHH)),
                   r.P(enc_slot(e, INTEL_ENC_GRU_OUT)), dm, D.gru_hidden, r.st));
    }
  }
}

// ---- towers (IntEL.py:182-197) ----------------------------------------------------------------
// Training forward with cross attention: the output of a tower's last layer is consumed only by the pooling kernel, which can
// rebuild it from the LayerNorm's x-hat stash (x = x-hat * gamma + beta).  Then the W2 GEMM does not store the output at all,
// the pooling kernels read x-hat, and the pooling backward applies the LayerNorm backward to its gradient rows in registers:
// one [B*L, d] write and three reads less per tower.
static bool tail_fusable(const IntelCtx* ctx, const IntelDesc& D, int L, int d, bool train) {
  return train && D.cross_attention && D.layers > 0 && !(ctx->drop_p > 0.f) && d <= 128 && xatt_ln_fused_supported(L, d);
}

// the training-mode dropout of tower t for the one-kernel 32-wide tower: the same draw as launch_dropout_mask below (stream id = tower *
// MAX_TOWER_LAYERS + layer; external keep flags laid out item-tower layers first, then score-tower layers)
static Tower32Dropout tower32_dropout(const Run& r, int tower) {
  Tower32Dropout d{0.f, 0ull, 0u, nullptr};
  const bool on = r.train ? r.ctx->drop_p > 0.f : false;
  if (!on) return d;
  d.p = r.ctx->drop_p;
  d.seed = r.ctx->drop_seed;
  d.stream0 = (unsigned)(tower * MAX_TOWER_LAYERS);
  if (r.ctx->drop_ext) d.ext = r.ctx->drop_ext + (tower == 0 ? 0 : (size_t)r.D.layers * r.y.M * r.y.tw[0].d);
  return d;
}

// inference: may the first layer build its input inside the one-kernel layer (tower.hip: TowerInput) instead of reading a materialised X0?
static bool tower_input_in_kernel(const Run& r, const TowerBufs& w) {
  static const int on = [] { const char* e = getenv("INTEL_TOWER_GATHER"); return (e && e[0] == '0') ? 0 : 1; }();
  const IntelDesc& D = r.D;
  return on && !r.train && gemm_planes() == 3 && D.layers > 0 && !tower32_supported(r.y.L, w.d, D.heads, D.layers, r.train) && tower_fused_supported(r.y.L, w.d, D.heads) &&
         tower_fused_wanted(0, w.d);
}

// in: the first layer's input is built in the kernel (inference; tower_input_in_kernel) -- w.X0 is then never written nor read
void tower_fwd(Run& r, TowerBufs& w, const TowerInput* in = nullptr) {
  const IntelDesc& D = r.D;
  const int M = r.y.M, d = w.d, B = r.y.B, L = r.y.L, pb = w.pbase;
  const float* X = in ? nullptr : w.X0;
  const int tw_i = &w == &r.y.tw[0] ? 0 : 1;
  // the reference's own widths (32-wide towers): ALL tied layers in one kernel, nothing stashed (tower32.hip)
  r.ctx->tw32[tw_i] = D.layers > 0 && tower32_supported(L, d, D.heads, D.layers, r.train);
  if (r.ctx->tw32[tw_i]) {
    r.ctx->tw_qkv16[tw_i] = false;
    r.ctx->tw_bwdf[tw_i] = false;
    Tower32Dropout dr = tower32_dropout(r, tw_i);
    RUN(launch_tower32_fwd(X, B, L, D.heads, D.layers, r.P(pb + T_WQ), r.P(pb + T_WK), r.P(pb + T_WV), r.P(pb + T_W1), r.P(pb + T_B1), r.P(pb + T_W2),
                           r.P(pb + T_B2), r.P(pb + T_LNG), r.P(pb + T_LNB), w.layer[D.layers - 1].Xout, r.st, &dr));
    return;
  }
  // one kernel per layer (tower.hip): the session's tile stays on chip from the q/k/v projection to the LayerNorm
  // bf16-mode training: the one-kernel layer leaves its stashes as bf16 arrays, which needs the whole-sequence attention backward
  const bool h16_ok = attn_seq_h16_supported(L, d / D.heads) && (d == 64 || d == 128);
  const bool fused = tower_fused_supported(L, d, D.heads) && tower_fused_wanted(r.train, w.d) && !(r.train && r.ctx->drop_p > 0.f) &&
                     !(r.train && gemm_planes() == 1 && !h16_ok);
  // bf16 mode: q/k/v (and, in the backward, their gradients) live in HBM as bf16 arrays -- every consumer rounds them to bf16
  // before its product anyway (attention backward, the q/k/v data- and weight-gradient products)
  r.ctx->tw_qkv16[tw_i] = fused && r.train && gemm_planes() == 1;
  // the backward's middle in one kernel (tower_bwd.hip): it recomputes q/k/v and the relu output, so the one-kernel forward stashes A and the log-sum-exp only
  r.ctx->tw_bwdf[tw_i] = r.train && D.layers > 0 && tower_bwd_fused_supported(L, d, D.heads) && tower_bwd_fused_wanted(d) && !(r.ctx->drop_p > 0.f);
  const bool slim = r.ctx->tw_bwdf[tw_i];
  for (int l = 0; fused && l < D.layers; ++l) {
    TowerLayerBufs& b = w.layer[l];
    const bool tail = l == D.layers - 1 && tail_fusable(r.ctx, D, L, d, r.train);      // x-hat / rstd only
    RUN(launch_tower_fwd_fused(X, B, L, d, D.heads, w.b3Wqkv, w.b3W1, w.b3W2, r.P(pb + T_B1), r.P(pb + T_B2), r.P(pb + T_LNG),
                               r.P(pb + T_LNB), tail ? nullptr : b.Xout, r.train, slim ? nullptr : b.QKV, b.A, b.LSE, slim ? nullptr : b.R1, b.XH, b.RSTD, r.st,
                               r.ctx->tw_qkv16[tw_i], l == 0 ? in : nullptr));
    X = b.Xout;
  }
  if (in && !fused) {
    intel_set_error("tower_fwd: in-kernel input without the one-kernel layer");
    r.ok(INTEL_E_STATE);
    return;
  }
  for (int l = 0; !fused && l < D.layers; ++l) {
    TowerLayerBufs& b = w.layer[l];
    GemmEpilogue e0;
    lin(r, X, d, M, d, w.pWqkv, 3 * d, b.QKV, 3 * d, e0);                       // q,k,v (bias=False, IntEL.py:60)
    if (r.rc) return;
    RUN(launch_attn_fwd(b.QKV, B, L, d, D.heads, nullptr, b.A, b.LSE, r.st));    // no mask (IntEL.py:184)
    GemmEpilogue e1;
    e1.bias = r.P(pb + T_B1);
    e1.relu = 1;                                                                 // stores relu(W1 h + b1)
    lin(r, b.A, d, M, d, w.pW1, d, b.R1, d, e1);
    if (r.rc) return;
    GemmEpilogue e2;
    e2.bias = r.P(pb + T_B2);
    e2.res = X; e2.ldres = d;
    e2.gamma = r.P(pb + T_LNG); e2.beta = r.P(pb + T_LNB);
    if (r.train) { e2.xhat = b.XH; e2.ldxhat = d; e2.rstd = b.RSTD; }
    if (r.train && r.ctx->drop_p > 0.f) {
      // h = LayerNorm(dropout(W2 relu(.) + b2) + residual): mask, plain-bias GEMM, then the masked add + LayerNorm
      const int tower = &w == &r.y.tw[0] ? 0 : 1;
      const float* ext = nullptr;
      if (r.ctx->drop_ext) ext = r.ctx->drop_ext + (tower == 0 ? (size_t)l * M * r.y.tw[0].d : (size_t)D.layers * M * r.y.tw[0].d + (size_t)l * M * r.y.tw[1].d);
      RUN(launch_dropout_mask(b.DM, (long long)M * d, r.ctx->drop_p, r.ctx->drop_seed, (unsigned)(tower * MAX_TOWER_LAYERS + l), ext, r.st));
      GemmEpilogue e2b;
      e2b.bias = e2.bias;
      lin(r, b.R1, d, M, d, w.pW2, d, b.Xout, d, e2b);
      if (r.rc) return;
      RUN(launch_add_layernorm(b.Xout, d, X, d, M, d, e2.gamma, e2.beta, b.Xout, d, b.XH, d, b.RSTD, r.st, b.DM));
    } else if (d <= 128) {
      if (l == D.layers - 1 && tail_fusable(r.ctx, D, L, d, r.train)) e2.no_out = 1;      // x-hat / rstd only
      lin(r, b.R1, d, M, d, w.pW2, d, b.Xout, d, e2);
    } else {
      GemmEpilogue e2b;
      e2b.bias = e2.bias;
      lin(r, b.R1, d, M, d, w.pW2, d, b.Xout, d, e2b);
      if (r.rc) return;
      RUN(launch_add_layernorm(b.Xout, d, X, d, M, d, e2.gamma, e2.beta, b.Xout, d, r.train ? b.XH : nullptr, d, r.train ? b.RSTD : nullptr, r.st));
    }
    if (r.rc) return;
    X = b.Xout;
  }
}

// dXout (in r.T->dXa) -> dX0 (returned pointer, one of dXa/dXb)
float* tower_bwd(Run& r, TowerBufs& w, float* dX, float* dXalt, bool last_ln_done) {
  const IntelDesc& D = r.D;
  Layout& y = r.y;
  const int M = y.M, d = w.d, B = y.B, L = y.L, pb = w.pbase;
  if (r.ctx->tw32[&w == &r.y.tw[0] ? 0 : 1]) {      // every tied layer, data and parameter gradients, in one kernel (tower32.hip)
    static const int slots[9] = {T_WQ, T_WK, T_WV, T_W1, T_B1, T_W2, T_B2, T_LNG, T_LNB};
    float* g[9];
    int acc[9];
    for (int p = 0; p < 9; ++p) {
      g[p] = r.G(pb + slots[p]);
      acc[p] = r.acc(pb + slots[p]);
    }
    Tower32Dropout dr = tower32_dropout(r, &w == &r.y.tw[0] ? 0 : 1);
    if (!r.ctx->fwd_dropout) dr.p = 0.f;
    static const int tw32_share = INTEL_DEBUG_ENV("INTEL_TW32_SHARE", 0);      // 1 .. 8: eighths of the CUs for the small-batch grid (A/B probe: debug builds only, common.h)
    if (!r.ok(launch_tower32_bwd(w.X0, dX, B, L, D.heads, D.layers, r.P(pb + T_WQ), r.P(pb + T_WK), r.P(pb + T_WV), r.P(pb + T_W1), r.P(pb + T_B1),
                                 r.P(pb + T_W2), r.P(pb + T_B2), r.P(pb + T_LNG), r.P(pb + T_LNB), dXalt, g, acc, r.ctx->rq, r.st, &dr,
                                 tw32_share > 0 ? tw32_share : ((D.encoder == INTEL_ENC_BERT4REC && B <= 2 * num_cus()) ? 5 : 8))))      // (tower32.hip: tower32_grid)
      return nullptr;
    return dXalt;
  }
  for (int l = D.layers - 1; l >= 0; --l) {
    TowerLayerBufs& b = w.layer[l];
    const float* Xin = l == 0 ? w.X0 : w.layer[l - 1].Xout;
    // the layer's small weight-gradient products (32-wide towers) leave as one launch at its end.  With dropout the first one's dY (dZ * mask) sits
    // in the dA buffer, which the layer rewrites below: that product is launched at once, the scope opens behind it
    if (!r.ctx->fwd_dropout) wgrad_batch_begin();
    const float* dZ = r.T->dZ;             // gradient behind this layer's LayerNorm
    if (l == D.layers - 1 && last_ln_done) {
      dZ = dX;                              // the pooling backward already applied it (fused tail)
    } else {
      int a = r.acc(pb + T_LNG);
      r.acc(pb + T_LNB);
      if (!r.ok(launch_layernorm_bwd(dX, d, b.XH, d, b.RSTD, M, d, r.P(pb + T_LNG), r.T->dZ, d, r.G(pb + T_LNG), r.G(pb + T_LNB), a,
                                     nullptr, r.st, r.ctx->rq)))
        return nullptr;
    }
    const float* dZd = dZ;                 // gradient behind the dropout: dZ * mask (the residual branch keeps dZ)
    if (r.ctx->fwd_dropout) {
      if (!r.ok(launch_mul2(dZ, b.DM, (long long)M * d, r.T->dA, r.st))) return nullptr;
      dZd = r.T->dA;
    }
    // bf16 mode: the stashes read only by matrix products (A, R1 -- R1 also as a sign test) and the gradients consumed only by
    // matrix products (dF1, dA, dQKV) are bf16 arrays; dZ stays fp32 (it is also the residual gradient)
    const int h16 = r.ctx->tw_qkv16[&w == &r.y.tw[0] ? 0 : 1] ? 1 : 0;
    if (r.ctx->tw_bwdf[&w == &r.y.tw[0] ? 0 : 1]) {
      // dZ -> dQKV in one kernel, dW2 / db2 / dW1 / db1 accumulated in its workgroups (tower_bwd.hip).  By width / mode the same launch also
      // produces dX = dQKV Wqkv + dZ (scope bit 0) and the q/k/v weight gradients (bit 1); what it leaves is done below as before
      const int scope = tower_bwd_fused_scope(d);
      static const int slots[7] = {T_W2, T_B2, T_W1, T_B1, T_WQ, T_WK, T_WV};
      float* g[7];
      int acc[7];
      for (int k = 0; k < 7; ++k) {
        const bool mine = k < 4 || (scope & 2);
        g[k] = mine ? r.G(pb + slots[k]) : nullptr;
        acc[k] = mine ? r.acc(pb + slots[k]) : 0;
      }
      if (!r.ok(launch_tower_bwd_fused(Xin, b.A, b.LSE, dZ, B, L, d, D.heads, w.b3Wqkv, w.b3W1, w.b3W2T, w.b3W1T, w.b3WqkvT, r.P(pb + T_B1), r.T->dQKV,
                                       (scope & 1) ? dXalt : nullptr, g, acc, r.ctx->rq, r.st, h16, h16)))
        return nullptr;
      if (!(scope & 2)) {
        const int ws[3] = {pb + T_WQ, pb + T_WK, pb + T_WV}, bs[3] = {-1, -1, -1};
        wgrad_split(r, r.T->dQKV, 3 * d, Xin, d, M, d, d, 3, ws, bs, h16);
      }
      if (!(scope & 1)) {
        GemmEpilogue er;
        er.res = dZ; er.ldres = d;
        er.b3 = w.b3WqkvT;
        er.a_bf16 = h16;
        lin(r, r.T->dQKV, 3 * d, M, 3 * d, w.pWqkvT, d, dXalt, d, er);
      }
      if (r.rc) return nullptr;
      if (!r.ok(wgrad_batch_flush(r.st))) return nullptr;
      float* t = dX; dX = dXalt; dXalt = t;
      continue;
    } else if (!h16 && gemm_planes() == 3 && linear_bwd_pair_supported(M, d)) {
      // each feed-forward linear's backward in ONE pass over the rows (pair.hip): dZ / dF1 are staged once for the data and the weight gradient,
      // the relu output is read once (operand + mask)
      float* gw2 = r.G(pb + T_W2); float* gb2 = r.G(pb + T_B2);
      const int aw2 = r.acc(pb + T_W2), ab2 = r.acc(pb + T_B2);
      if (!r.ok(launch_linear_bwd_pair(dZd, d, b.R1, d, M, d, w.b3W2T, 1, r.T->dF1, d, gw2, gb2, aw2, ab2, r.ctx->rq, r.st))) return nullptr;
      if (r.ctx->fwd_dropout) wgrad_batch_begin();
      float* gw1 = r.G(pb + T_W1); float* gb1 = r.G(pb + T_B1);
      const int aw1 = r.acc(pb + T_W1), ab1 = r.acc(pb + T_B1);
      if (!r.ok(launch_linear_bwd_pair(r.T->dF1, d, b.A, d, M, d, w.b3W1T, 0, r.T->dA, d, gw1, gb1, aw1, ab1, r.ctx->rq, r.st))) return nullptr;
      if (!r.ok(launch_attn_bwd(b.QKV, b.A, r.T->dA, b.LSE, B, L, d, D.heads, nullptr, r.T->dQKV, r.T->DSUM, r.st, nullptr, h16))) return nullptr;
    } else {
    wgrad(r, dZd, d, b.R1, d, M, d, d, pb + T_W2, pb + T_B2, h16 ? 2 : 0);
    if (r.ctx->fwd_dropout) wgrad_batch_begin();
    GemmEpilogue em;
    em.mask = b.R1; em.ldmask = d;
    em.mask_bf16 = h16; em.c_bf16 = h16;
    lin(r, dZd, d, M, d, w.pW2T, d, r.T->dF1, d, em);                     // d(pre-relu) = (dZ W2) * [R1 > 0]
    wgrad(r, r.T->dF1, d, b.A, d, M, d, d, pb + T_W1, pb + T_B1, h16 ? 3 : 0);
    GemmEpilogue e0;
    e0.a_bf16 = h16; e0.c_bf16 = h16;
    lin(r, r.T->dF1, d, M, d, w.pW1T, d, r.T->dA, d, e0);
    if (r.rc) return nullptr;
    if (!r.ok(launch_attn_bwd(b.QKV, b.A, r.T->dA, b.LSE, B, L, d, D.heads, nullptr, r.T->dQKV, r.T->DSUM, r.st, nullptr, h16))) return nullptr;
    }
    const int ws[3] = {pb + T_WQ, pb + T_WK, pb + T_WV}, bs[3] = {-1, -1, -1};
    // dXin = dQKV @ [Wq;Wk;Wv] + dZ (residual) and the q/k/v weight gradients: one pass over the rows where pair.hip covers the shape (fp32 mode, widths 64 / 128)
    if (h16 || !qkv_bwd_one_pass(r, r.T->dQKV, Xin, dZ, M, d, 3, w.b3WqkvT, dXalt, ws, bs)) {
      wgrad_split(r, r.T->dQKV, 3 * d, Xin, d, M, d, d, 3, ws, bs, h16);
      // A has row stride 3d; the packed k extent is 3*rup(d,16).
      GemmEpilogue er;
      er.res = dZ; er.ldres = d;
      er.b3 = w.b3WqkvT;
      er.a_bf16 = h16;
      lin(r, r.T->dQKV, 3 * d, M, 3 * d, w.pWqkvT, d, dXalt, d, er);   // d % 16 == 0 (check_desc)
    }
    if (r.rc) return nullptr;
    if (!r.ok(wgrad_batch_flush(r.st))) return nullptr;      // (dZ, dF1, dQKV and the stashes are still this layer's)
    float* t = dX; dX = dXalt; dXalt = t;
  }
  return dX;
}

// ---- BERT4Rec encoder (GeneralSeq.py:89-106) -----------------------------------------------------
static void enc32_blocks(const Run& r, int e, Enc32Block* blk);
void bert_fwd(Run& r, int e) {
  const IntelDesc& D = r.D;
  EncBufs& n = r.y.enc[e];
  const int B = r.y.B, T = n.T, dm = n.dm, rows = r.ctx->enc_rows[e];
  const int* len = e == 0 ? r.bt->history_len : r.bt->history_item_len;
  const int* off = r.ctx->enc_packed[e] ? (e == 0 ? r.bt->his_off : r.bt->hisitem_off) : nullptr;     // packed rows
  if (off && n.pos_done)
    ;                                            // added by the gather / intent-embedding kernels (forward_impl)
  else if (off)
    RUN(launch_add_pos_rows(n.E0, dm, r.P(enc_slot(e, INTEL_ENC_POS)), n.rowT, rows, r.st));
  else
    RUN(launch_add_pos(n.E0, dm, r.P(enc_slot(e, INTEL_ENC_POS)), len, B, T, r.st));
  const float* X = n.E0;
  if (r.ctx->enc32[e]) {      // the reference's default widths: the whole encoder in one kernel, nothing stashed (tower32.hip)
    Enc32Block blk[INTEL_ENC_MAX_BLOCKS];
    enc32_blocks(r, e, blk);
    RUN(launch_enc32_fwd(X, off, len, B, T, D.enc_heads, D.enc_layers, blk, r.y.PREDIN + n.predin_off, r.y.Pin, r.st));
    return;
  }
  if (r.ctx->enc_fused[e]) {
    // two kernels for the whole encoder (enc.hip): every full block as one kernel over tiles of whole sessions -- the last full
    // block also projects the pruned last block's keys / values --, then the pruned last block, 16 sessions per workgroup
    const int L = D.enc_layers;
    RUN(launch_enc_tiles(off, B, T, rows, n.tileS, r.st));
    for (int l = 0; l + 1 < L; ++l) {
      EncBlockBufs& k = n.blk[l];
      EncBlockFwd f;
      f.X = X; f.rows = rows; f.B = B; f.T = T; f.dm = dm; f.heads = D.enc_heads; f.train = r.train;
      f.off = off; f.tile_s = n.tileS;
      f.Wqkv = k.b3Wqkv; f.W1 = k.b3W1; f.W2 = k.b3W2;
      f.bqkv = k.bQKV; f.b1 = r.P(enc_blk_slot(e, l, INTEL_ENC_B1)); f.b2 = r.P(enc_blk_slot(e, l, INTEL_ENC_B2));
      f.g1 = r.P(enc_blk_slot(e, l, INTEL_ENC_LN1G)); f.be1 = r.P(enc_blk_slot(e, l, INTEL_ENC_LN1B));
      f.g2 = r.P(enc_blk_slot(e, l, INTEL_ENC_LN2G)); f.be2 = r.P(enc_blk_slot(e, l, INTEL_ENC_LN2B));
      f.C = k.C;
      f.QKV = k.QKV; f.LSE = k.LSE; f.XH1 = k.XH1; f.RSTD1 = k.RSTD1; f.F1 = k.F1; f.XH2 = k.XH2; f.RSTD2 = k.RSTD2;
      const bool feeds_last = l + 2 == L;
      f.out = (r.train || !feeds_last) ? k.Eout : nullptr;      // inference: the last full block's output lives on as K' / V' and x_last only
      if (feeds_last) {
        const size_t kv_off = (size_t)(dm / 16) * 4 * 3 * 64 * 4;      // floats: the image's column tiles dm/16 .. are [k | v]
        f.Wkv = n.blk[L - 1].b3Wqkv + kv_off;
        f.bkv = n.blk[L - 1].bQKV + dm;
        f.KV = n.last.KV;
        f.xlast = n.last.XLAST;
      }
      RUN(launch_enc_block_fwd(f, r.st));
      X = k.Eout;
    }
    {
      const int l = L - 1;
      EncBlockBufs& k = n.blk[l];
      EncLastBufs& q = n.last;
      EncLastFwd f;
      f.xlast = q.XLAST; f.KV = q.KV; f.off = off; f.len = len; f.B = B; f.T = T; f.dm = dm; f.heads = D.enc_heads; f.train = r.train;
      f.Wq = k.b3Wqkv; f.W1 = k.b3W1; f.W2 = k.b3W2;
      f.bq = k.bQKV; f.b1 = r.P(enc_blk_slot(e, l, INTEL_ENC_B1)); f.b2 = r.P(enc_blk_slot(e, l, INTEL_ENC_B2));
      f.g1 = r.P(enc_blk_slot(e, l, INTEL_ENC_LN1G)); f.be1 = r.P(enc_blk_slot(e, l, INTEL_ENC_LN1B));
      f.g2 = r.P(enc_blk_slot(e, l, INTEL_ENC_LN2G)); f.be2 = r.P(enc_blk_slot(e, l, INTEL_ENC_LN2B));
      f.out = r.y.PREDIN + n.predin_off; f.ldo = r.y.Pin;
      f.QL = q.QLAST; f.PL = q.PL; f.CL = q.CL; f.XH1 = q.XH1; f.RSTD1 = q.RSTD1; f.F1 = q.F1; f.XH2 = q.XH2; f.RSTD2 = q.RSTD2;
      RUN(launch_enc_last_fwd(f, r.st));
    }
    return;
  }
  for (int l = 0; l + 1 < D.enc_layers; ++l) {
    EncBlockBufs& k = n.blk[l];
    {   // fused q/k/v projection (bias=True in TransformerLayer, layers.py:70)
      GemmEpilogue eb;
      eb.bias = k.bQKV;
      lin(r, X, dm, rows, dm, k.pWqkv, 3 * dm, k.QKV, 3 * dm, eb);
      if (r.rc) return;
    }
    RUN(launch_attn_fwd(k.QKV, B, T, dm, D.enc_heads, len, k.A, k.LSE, r.st, off));
    RUN(launch_add_layernorm(k.A, dm, X, dm, rows, dm, r.P(enc_blk_slot(e, l, INTEL_ENC_LN1G)),
                             r.P(enc_blk_slot(e, l, INTEL_ENC_LN1B)), k.C, dm, r.train ? k.XH1 : nullptr, dm, r.train ? k.RSTD1 : nullptr, r.st));
    GemmEpilogue e1;
    e1.bias = r.P(enc_blk_slot(e, l, INTEL_ENC_B1));
    e1.relu = 1;
    lin(r, k.C, dm, rows, dm, k.pW1, dm, k.F1, dm, e1);
    if (r.rc) return;
    GemmEpilogue e2;
    e2.bias = r.P(enc_blk_slot(e, l, INTEL_ENC_B2));
    e2.res = k.C; e2.ldres = dm;
    if (dm <= 128) {
      e2.gamma = r.P(enc_blk_slot(e, l, INTEL_ENC_LN2G)); e2.beta = r.P(enc_blk_slot(e, l, INTEL_ENC_LN2B));
      if (r.train) { e2.xhat = k.XH2; e2.ldxhat = dm; e2.rstd = k.RSTD2; }
      lin(r, k.F1, dm, rows, dm, k.pW2, dm, k.Eout, dm, e2);
    } else {
      GemmEpilogue e2b;
      e2b.bias = e2.bias;
      lin(r, k.F1, dm, rows, dm, k.pW2, dm, k.Eout, dm, e2b);
      if (r.rc) return;
      RUN(launch_add_layernorm(k.Eout, dm, k.C, dm, rows, dm, r.P(enc_blk_slot(e, l, INTEL_ENC_LN2G)),
                               r.P(enc_blk_slot(e, l, INTEL_ENC_LN2B)), k.Eout, dm, k.XH2, dm, k.RSTD2, r.st));
    }
    if (r.rc) return;
    X = k.Eout;
  }
  {   // ---- last block, pruned to the one output row that is used (his_vector = seq[b, len-1])
    const int l = D.enc_layers - 1;
    EncBlockBufs& k = n.blk[l];
    EncLastBufs& q = n.last;
    const size_t third = (size_t)rup(dm, 16) * rup(dm, 16);
    GemmEpilogue ekv;
    ekv.bias = k.bQKV + dm;
    lin(r, X, dm, rows, dm, k.pWqkv + third, 2 * dm, q.KV, 2 * dm, ekv);            // [k | v] for every row
    if (r.rc) return;
    RUN(launch_select_last(X, dm, len, B, T, q.XLAST, dm, 0, r.st, off));
    GemmEpilogue eq;
    eq.bias = k.bQKV;
    lin(r, q.XLAST, dm, B, dm, k.pWqkv, dm, q.QLAST, dm, eq);
    if (r.rc) return;
    RUN(launch_attn_lastq_fwd(q.KV, q.QLAST, len, B, T, dm, D.enc_heads, q.OL, q.PL, r.st, off));
    RUN(launch_add_layernorm(q.OL, dm, q.XLAST, dm, B, dm, r.P(enc_blk_slot(e, l, INTEL_ENC_LN1G)),
                             r.P(enc_blk_slot(e, l, INTEL_ENC_LN1B)), q.CL, dm, q.XH1, dm, q.RSTD1, r.st));
    GemmEpilogue e1;
    e1.bias = r.P(enc_blk_slot(e, l, INTEL_ENC_B1));
    e1.relu = 1;
    lin(r, q.CL, dm, B, dm, k.pW1, dm, q.F1, dm, e1);
    if (r.rc) return;
    GemmEpilogue e2;
    e2.bias = r.P(enc_blk_slot(e, l, INTEL_ENC_B2));
    e2.res = q.CL; e2.ldres = dm;
    float* vec = r.y.PREDIN + n.predin_off;
    if (dm <= 128) {
      e2.gamma = r.P(enc_blk_slot(e, l, INTEL_ENC_LN2G)); e2.beta = r.P(enc_blk_slot(e, l, INTEL_ENC_LN2B));
      e2.xhat = q.XH2; e2.ldxhat = dm; e2.rstd = q.RSTD2;
      lin(r, q.F1, dm, B, dm, k.pW2, dm, vec, r.y.Pin, e2);
    } else {
      GemmEpilogue e2b;
      e2b.bias = e2.bias;
      lin(r, q.F1, dm, B, dm, k.pW2, dm, q.OL, dm, e2b);
      if (r.rc) return;
      RUN(launch_add_layernorm(q.OL, dm, q.CL, dm, B, dm, r.P(enc_blk_slot(e, l, INTEL_ENC_LN2G)),
                               r.P(enc_blk_slot(e, l, INTEL_ENC_LN2B)), vec, r.y.Pin, q.XH2, dm, q.RSTD2, r.st));
    }
  }
}

// returns dE0 (gradient w.r.t. the encoder input rows, pos-emb already handled)
// the parameters of encoder e's blocks in the order of Enc32Block (raw reference tensors)
static void enc32_blocks(const Run& r, int e, Enc32Block* blk) {
  for (int l = 0; l < r.D.enc_layers; ++l) {
    auto P = [&](int o) { return r.P(enc_blk_slot(e, l, o)); };
    blk[l] = Enc32Block{P(INTEL_ENC_WQ), P(INTEL_ENC_BQ), P(INTEL_ENC_WK), P(INTEL_ENC_BK), P(INTEL_ENC_WV), P(INTEL_ENC_BV), P(INTEL_ENC_LN1G), P(INTEL_ENC_LN1B),
                        P(INTEL_ENC_W1), P(INTEL_ENC_B1), P(INTEL_ENC_W2), P(INTEL_ENC_B2), P(INTEL_ENC_LN2G), P(INTEL_ENC_LN2B)};
  }
}

// the blocks' backward: returns the gradient of the encoder's input rows (position-embedding gradient: bert_bwd below)
float* bert_bwd_blocks(Run& r, int e) {
  const IntelDesc& D = r.D;
  Layout& y = r.y;
  EncBufs& n = y.enc[e];
  const int B = y.B, T = n.T, dm = n.dm, rows = r.ctx->enc_rows[e];
  const int* len = e == 0 ? r.bt->history_len : r.bt->history_item_len;
  const int* off = r.ctx->enc_packed[e] ? (e == 0 ? r.bt->his_off : r.bt->hisitem_off) : nullptr;     // packed rows
  float *dX = r.T->dXa, *dXalt = r.T->dXb;
  if (r.ctx->enc32[e]) {      // every block, data and parameter gradients, in one kernel (tower32.hip: enc32_bwd_kernel)
    Enc32Block blk[INTEL_ENC_MAX_BLOCKS];
    enc32_blocks(r, e, blk);
    float* g[INTEL_ENC_MAX_BLOCKS][14];
    int acc[INTEL_ENC_MAX_BLOCKS][14];
    for (int l = 0; l < D.enc_layers; ++l)
      for (int o = 0; o < 14; ++o) {
        g[l][o] = r.G(enc_blk_slot(e, l, o));
        acc[l][o] = r.acc(enc_blk_slot(e, l, o));
      }
    if (!r.ok(launch_enc32_bwd(n.E0, off, len, B, T, D.enc_heads, D.enc_layers, blk, y.dPREDIN + n.predin_off, y.Pin, dX, g, acc, r.ctx->rq, r.st))) return nullptr;
    return dX;
  }
  static const bool fused_bwd_on = [] { const char* e = getenv("INTEL_ENC_FUSED_BWD"); return !(e && e[0] == '0'); }();      // 0: kernel-per-op backward on the fused forward's stash
  const bool fused_bwd = fused_bwd_on && r.ctx->enc_fused[e];
  if (fused_bwd) {
    // the data-gradient chain of every block as one kernel (enc_bwd.hip); the weight gradients (reductions over all rows) and
    // the two K > 128 data-gradient products stay on the GEMM kernels
    const int L = D.enc_layers;
    float *dZl = r.T->dLB[0], *dF1l = r.T->dLB[1], *dXl = r.T->dLB[2], *dQl = r.T->dLB[4];
    auto ln_slots = [&](int l, float** outs, int* accs) {
      const int sl[4] = {enc_blk_slot(e, l, INTEL_ENC_LN2G), enc_blk_slot(e, l, INTEL_ENC_LN2B), enc_blk_slot(e, l, INTEL_ENC_LN1G), enc_blk_slot(e, l, INTEL_ENC_LN1B)};
      for (int i = 0; i < 4; ++i) {
        outs[i] = r.G(sl[i]);
        accs[i] = outs[i] ? r.acc(sl[i]) : 0;
      }
    };
    {
      const int l = L - 1;
      EncBlockBufs& k = n.blk[l];
      EncLastBufs& q = n.last;
      const float* Xin = n.blk[l - 1].Eout;
      EncLastBwd f;
      f.dvec = y.dPREDIN + n.predin_off; f.ldv = y.Pin; f.KV = q.KV; f.off = off; f.len = len; f.B = B; f.T = T; f.dm = dm; f.heads = D.enc_heads;
      f.W2T = k.b3W2T; f.W1T = k.b3W1T; f.WqT = q.b3WqT;
      f.g1 = r.P(enc_blk_slot(e, l, INTEL_ENC_LN1G)); f.g2 = r.P(enc_blk_slot(e, l, INTEL_ENC_LN2G));
      f.XH2 = q.XH2; f.RSTD2 = q.RSTD2; f.F1 = q.F1; f.XH1 = q.XH1; f.RSTD1 = q.RSTD1; f.PL = q.PL; f.QL = q.QLAST;
      f.DZ2 = dZl; f.DF1 = dF1l; f.DQ = dQl; f.DXL = dXl; f.DKV = r.T->dQKV;
      float* outs[4]; int accs[4];
      ln_slots(l, outs, accs);
      f.dg2 = outs[0]; f.db2 = outs[1]; f.dg1 = outs[2]; f.db1 = outs[3];
      f.acc_g2 = accs[0]; f.acc_b2 = accs[1]; f.acc_g1 = accs[2]; f.acc_b1 = accs[3];
      if (!r.ok(launch_enc_last_bwd(f, r.st, r.ctx->rq))) return nullptr;
      wgrad(r, dZl, dm, q.F1, dm, B, dm, dm, enc_blk_slot(e, l, INTEL_ENC_W2), enc_blk_slot(e, l, INTEL_ENC_B2));
      wgrad(r, dF1l, dm, q.CL, dm, B, dm, dm, enc_blk_slot(e, l, INTEL_ENC_W1), enc_blk_slot(e, l, INTEL_ENC_B1));
      wgrad(r, dQl, dm, q.XLAST, dm, B, dm, dm, enc_blk_slot(e, l, INTEL_ENC_WQ), enc_blk_slot(e, l, INTEL_ENC_BQ));
      {
        const int ws[2] = {enc_blk_slot(e, l, INTEL_ENC_WK), enc_blk_slot(e, l, INTEL_ENC_WV)};
        const int bs[2] = {enc_blk_slot(e, l, INTEL_ENC_BK), enc_blk_slot(e, l, INTEL_ENC_BV)};
        // dE = dKV [Wk;Wv] (d(x_last) joins inside the block kernel) and the k / v weight + bias gradients: one pass where pair.hip covers the shape
        if (!qkv_bwd_one_pass(r, r.T->dQKV, Xin, nullptr, rows, dm, 2, q.b3WkvT, dX, ws, bs)) {
          wgrad_split(r, r.T->dQKV, 2 * dm, Xin, dm, rows, dm, dm, 2, ws, bs);
          GemmEpilogue e0;
          e0.b3 = q.b3WkvT;
          lin(r, r.T->dQKV, 2 * dm, rows, 2 * dm, q.pWkvT, dm, dX, dm, e0);
        }
      }
      if (r.rc) return nullptr;
    }
    for (int l = L - 2; l >= 0; --l) {
      EncBlockBufs& k = n.blk[l];
      const float* Xin = l == 0 ? n.E0 : n.blk[l - 1].Eout;
      EncBlockBwd f;
      f.dE = dX; f.dxl = l == L - 2 ? dXl : nullptr;
      f.rows = rows; f.B = B; f.T = T; f.dm = dm; f.heads = D.enc_heads; f.off = off; f.tile_s = n.tileS;
      f.W2T = k.b3W2T; f.W1T = k.b3W1T;
      f.g1 = r.P(enc_blk_slot(e, l, INTEL_ENC_LN1G)); f.g2 = r.P(enc_blk_slot(e, l, INTEL_ENC_LN2G));
      f.XH2 = k.XH2; f.RSTD2 = k.RSTD2; f.F1 = k.F1; f.XH1 = k.XH1; f.RSTD1 = k.RSTD1; f.QKV = k.QKV;
      f.DZ2 = r.T->dZ; f.DF1 = r.T->dF1; f.DZ1 = r.T->dA; f.DQKV = r.T->dQKV;
      float* outs[4]; int accs[4];
      ln_slots(l, outs, accs);
      f.dg2 = outs[0]; f.db2 = outs[1]; f.dg1 = outs[2]; f.db1 = outs[3];
      f.acc_g2 = accs[0]; f.acc_b2 = accs[1]; f.acc_g1 = accs[2]; f.acc_b1 = accs[3];
      if (!r.ok(launch_enc_block_bwd(f, r.st, r.ctx->rq))) return nullptr;
      wgrad(r, r.T->dZ, dm, k.F1, dm, rows, dm, dm, enc_blk_slot(e, l, INTEL_ENC_W2), enc_blk_slot(e, l, INTEL_ENC_B2));
      wgrad(r, r.T->dF1, dm, k.C, dm, rows, dm, dm, enc_blk_slot(e, l, INTEL_ENC_W1), enc_blk_slot(e, l, INTEL_ENC_B1));
      {
        const int ws[3] = {enc_blk_slot(e, l, INTEL_ENC_WQ), enc_blk_slot(e, l, INTEL_ENC_WK), enc_blk_slot(e, l, INTEL_ENC_WV)};
        const int bs[3] = {enc_blk_slot(e, l, INTEL_ENC_BQ), enc_blk_slot(e, l, INTEL_ENC_BK), enc_blk_slot(e, l, INTEL_ENC_BV)};
        // (dZ1 = the residual into the block input)
        if (!qkv_bwd_one_pass(r, r.T->dQKV, Xin, r.T->dA, rows, dm, 3, k.b3WqkvT, dXalt, ws, bs)) {
          wgrad_split(r, r.T->dQKV, 3 * dm, Xin, dm, rows, dm, dm, 3, ws, bs);
          GemmEpilogue er;
          er.res = r.T->dA; er.ldres = dm;
          er.b3 = k.b3WqkvT;
          lin(r, r.T->dQKV, 3 * dm, rows, 3 * dm, k.pWqkvT, dm, dXalt, dm, er);
        }
      }
      if (r.rc) return nullptr;
      float* t = dX; dX = dXalt; dXalt = t;
    }
  }
  if (!fused_bwd) {   // ---- last block, pruned (see bert_fwd): gradient of one output row per session
    const int l = D.enc_layers - 1;
    EncBlockBufs& k = n.blk[l];
    EncLastBufs& q = n.last;
    const float* Xin = l == 0 ? n.E0 : n.blk[l - 1].Eout;
    float *dZl = r.T->dLB[0], *dF1l = r.T->dLB[1], *dCl = r.T->dLB[2], *dSl = r.T->dLB[3], *dQl = r.T->dLB[4];
    const float* dvec = y.dPREDIN + n.predin_off;
    {
      const int sg = enc_blk_slot(e, l, INTEL_ENC_LN2G), sb = enc_blk_slot(e, l, INTEL_ENC_LN2B);
      int a = r.acc(sg);
      r.acc(sb);
      if (!r.ok(launch_layernorm_bwd(dvec, y.Pin, q.XH2, dm, q.RSTD2, B, dm, r.P(sg), dZl, dm, r.G(sg), r.G(sb), a, nullptr, r.st, r.ctx->rq)))
        return nullptr;
    }
    wgrad(r, dZl, dm, q.F1, dm, B, dm, dm, enc_blk_slot(e, l, INTEL_ENC_W2), enc_blk_slot(e, l, INTEL_ENC_B2));
    GemmEpilogue em;
    em.mask = q.F1; em.ldmask = dm;
    lin(r, dZl, dm, B, dm, k.pW2T, dm, dF1l, dm, em);
    wgrad(r, dF1l, dm, q.CL, dm, B, dm, dm, enc_blk_slot(e, l, INTEL_ENC_W1), enc_blk_slot(e, l, INTEL_ENC_B1));
    GemmEpilogue ec;
    ec.res = dZl; ec.ldres = dm;
    lin(r, dF1l, dm, B, dm, k.pW1T, dm, dCl, dm, ec);
    if (r.rc) return nullptr;
    {
      const int sg = enc_blk_slot(e, l, INTEL_ENC_LN1G), sb = enc_blk_slot(e, l, INTEL_ENC_LN1B);
      int a = r.acc(sg);
      r.acc(sb);
      if (!r.ok(launch_layernorm_bwd(dCl, dm, q.XH1, dm, q.RSTD1, B, dm, r.P(sg), dSl, dm, r.G(sg), r.G(sb), a, nullptr, r.st, r.ctx->rq)))
        return nullptr;
    }
    // attention of the single query row: dS is both d(attention output) and the residual into Xlast
    if (!r.ok(launch_attn_lastq_bwd(q.KV, q.QLAST, q.PL, dSl, len, B, T, dm, D.enc_heads, dQl, r.T->dQKV, r.st, off))) return nullptr;
    wgrad(r, dQl, dm, q.XLAST, dm, B, dm, dm, enc_blk_slot(e, l, INTEL_ENC_WQ), enc_blk_slot(e, l, INTEL_ENC_BQ));
    GemmEpilogue exl;
    exl.res = dSl; exl.ldres = dm;
    lin(r, dQl, dm, B, dm, q.pWqT, dm, dZl, dm, exl);                       // dXlast = dQ Wq + dS
    {
      const int ws[2] = {enc_blk_slot(e, l, INTEL_ENC_WK), enc_blk_slot(e, l, INTEL_ENC_WV)};
      const int bs[2] = {enc_blk_slot(e, l, INTEL_ENC_BK), enc_blk_slot(e, l, INTEL_ENC_BV)};
      wgrad_split(r, r.T->dQKV, 2 * dm, Xin, dm, rows, dm, dm, 2, ws, bs);
    }
    GemmEpilogue e0;
    e0.b3 = q.b3WkvT;
    lin(r, r.T->dQKV, 2 * dm, rows, 2 * dm, q.pWkvT, dm, dX, dm, e0);          // dX = dKV [Wk;Wv]
    if (r.rc) return nullptr;
    if (!r.ok(launch_add_at_last(dZl, dm, dm, len, B, T, dX, r.st, off))) return nullptr;
  }
  for (int l = D.enc_layers - 2; l >= 0 && !fused_bwd; --l) {
    EncBlockBufs& k = n.blk[l];
    const float* Xin = l == 0 ? n.E0 : n.blk[l - 1].Eout;
    // LN2: Eout = LN2(F2 + C)
    {
      const int sg = enc_blk_slot(e, l, INTEL_ENC_LN2G), sb = enc_blk_slot(e, l, INTEL_ENC_LN2B);
      int a = r.acc(sg);
      r.acc(sb);
      if (!r.ok(launch_layernorm_bwd(dX, dm, k.XH2, dm, k.RSTD2, rows, dm, r.P(sg), r.T->dZ, dm, r.G(sg), r.G(sb), a, nullptr, r.st, r.ctx->rq)))
        return nullptr;
    }
    wgrad(r, r.T->dZ, dm, k.F1, dm, rows, dm, dm, enc_blk_slot(e, l, INTEL_ENC_W2), enc_blk_slot(e, l, INTEL_ENC_B2));
    GemmEpilogue em;
    em.mask = k.F1; em.ldmask = dm;
    lin(r, r.T->dZ, dm, rows, dm, k.pW2T, dm, r.T->dF1, dm, em);
    wgrad(r, r.T->dF1, dm, k.C, dm, rows, dm, dm, enc_blk_slot(e, l, INTEL_ENC_W1), enc_blk_slot(e, l, INTEL_ENC_B1));
    // dC = dF1 W1 + dZ (residual into LN2)
    GemmEpilogue ec;
    ec.res = r.T->dZ; ec.ldres = dm;
    lin(r, r.T->dF1, dm, rows, dm, k.pW1T, dm, r.T->dA, dm, ec);
    if (r.rc) return nullptr;
    // LN1: C = LN1(A + Xin): dS = LN1bwd(dC)  -> dA_attn = dS, residual dXin += dS
    {
      const int sg = enc_blk_slot(e, l, INTEL_ENC_LN1G), sb = enc_blk_slot(e, l, INTEL_ENC_LN1B);
      int a = r.acc(sg);
      r.acc(sb);
      if (!r.ok(launch_layernorm_bwd(r.T->dA, dm, k.XH1, dm, k.RSTD1, rows, dm, r.P(sg), r.T->dZ, dm, r.G(sg), r.G(sb), a, nullptr, r.st, r.ctx->rq)))
        return nullptr;
    }
    if (!r.ok(launch_attn_bwd(k.QKV, k.A, r.T->dZ, k.LSE, B, T, dm, D.enc_heads, len, r.T->dQKV, r.T->DSUM, r.st, off))) return nullptr;
    {
      const int ws[3] = {enc_blk_slot(e, l, INTEL_ENC_WQ), enc_blk_slot(e, l, INTEL_ENC_WK), enc_blk_slot(e, l, INTEL_ENC_WV)};
      const int bs[3] = {enc_blk_slot(e, l, INTEL_ENC_BQ), enc_blk_slot(e, l, INTEL_ENC_BK), enc_blk_slot(e, l, INTEL_ENC_BV)};
      wgrad_split(r, r.T->dQKV, 3 * dm, Xin, dm, rows, dm, dm, 3, ws, bs);
    }
    {
      GemmEpilogue er;
      er.res = r.T->dZ; er.ldres = dm;
      er.b3 = k.b3WqkvT;
      lin(r, r.T->dQKV, 3 * dm, rows, 3 * dm, k.pWqkvT, dm, dXalt, dm, er);   // dm % 16 == 0 (check_desc)
    }
    if (r.rc) return nullptr;
    float* t = dX; dX = dXalt; dXalt = t;
  }
  return dX;
}

// returns dE0 (gradient w.r.t. the encoder input rows, pos-emb already handled)
float* bert_bwd(Run& r, int e) {
  const IntelDesc& D = r.D;
  EncBufs& n = r.y.enc[e];
  const int T = n.T, dm = n.dm, rows = r.ctx->enc_rows[e];
  const int* len = e == 0 ? r.bt->history_len : r.bt->history_item_len;
  const int* off = r.ctx->enc_packed[e] ? (e == 0 ? r.bt->his_off : r.bt->hisitem_off) : nullptr;     // packed rows
  float* dX = bert_bwd_blocks(r, e);
  if (!dX || r.rc) return nullptr;
  // position embedding gradient: dpos[p,:] = sum of dE over the rows at position p (per-workgroup LDS tables + one atomic per entry)
  if (r.G(enc_slot(e, INTEL_ENC_POS)) && pos_grad_supported(T, dm)) {
    const int ps = enc_slot(e, INTEL_ENC_POS);
    if (D.history_max + 1 > T && !r.ok(launch_fill(r.G(ps) + (size_t)T * dm, (long long)(D.history_max + 1 - T) * dm, 0.f, r.st))) return nullptr;
    r.acc(ps);
    if (!r.ok(launch_pos_grad(dX, dm, off ? n.rowT : nullptr, len, T, rows, r.G(ps), r.st, r.ctx->rq, off, r.y.B))) return nullptr;
  } else if (r.G(enc_slot(e, INTEL_ENC_POS))) {      // table too large for LDS: onehot^T dE through the weight-gradient kernel
    const int ps = enc_slot(e, INTEL_ENC_POS);
    if (!r.ok(launch_make_onehot(off ? n.rowT : nullptr, len, T, rows, T, r.T->ONEHOT, r.st))) return nullptr;
    if (D.history_max + 1 > T && !r.ok(launch_fill(r.G(ps) + (size_t)T * dm, (long long)(D.history_max + 1 - T) * dm, 0.f, r.st))) return nullptr;
    r.acc(ps);
    if (!r.ok(launch_wgrad(r.T->ONEHOT, T, dX, dm, rows, T, dm, r.G(ps), dm, nullptr, 0, nullptr, r.st, r.ctx->rq))) return nullptr;
  }
  return dX;
}


// ---- the session head as B-row chains (chain.hip) -------------------------------------------------------------------------------
// IntEL.py:147-153 (intent prediction) and :201-215 (pooling queries, fusion weights, score aggregation) are ~13 dependent small
// launches forward and ~20 backward on one row per session.  With per-session fusion weights (the reference's default:
// --cross_attention 1, no weight normalisation) they run as two chain launches per direction around the pooling kernels; every
// buffer a later launch reads (the stash, the operands of the deferred weight-gradient products) is written as before.
// INTEL_HEAD_FUSED=0 keeps the kernel-per-op head.
static bool head_fused_ok(const IntelDesc& D, const Layout& y, int train) {
  static const int mode = [] { const char* e = getenv("INTEL_HEAD_FUSED"); return !e ? 1 : (e[0] == '0' ? 0 : (e[0] == '2' ? 2 : 1)); }();      // parsed once: 0 off, 2 backward chains forced on
  const int on = mode != 0;
  if (!on || !D.cross_attention || D.pool_mean || D.weight_norm != 0 || D.model_num > 16) return false;
  // (bf16 mode: the chains' links of 64 / 128 input features round their operands like the kernel-per-op head's products do -- chain.h: CH_BF16)
  // train: 0 = inference forward, 1 = training BACKWARD, 2 = training FORWARD -- the forward chains write the kernel-per-op path's stash, so
  // the two directions decide independently.  The BACKWARD chains pay where the step is launch-bound (the reference's hyper-parameters at
  // its batch of 512: +8.5 % sessions/s same-box).  With more sessions per step the kernel-per-op head's launches hide under the towers'
  // kernels while a chain launch holds whole CUs (measured same-box, round 4, both directions on: Tmall shape 1024 / 2048 / 4096 sessions
  // -0.5 / -2.7 / -1.8 %, LifeData 2048 -2.5 %, stress 1024 +-0): they stopped at 768 sessions per step.  Round 5: part of that loss was the two
  // weight-gradient chain launches queued behind the ITEM tower -- the backward's longest branch, with the table's Adam sweep behind it; they now go
  // behind the score tower (intel_backward: leaves_behind_score).  Forced on against off after that, two boxes (profiles/r05_threshold_sweep.txt and
  // the run before it): 1024 sessions +2.6 / +0.3 %, 2048 +1.0 / -0.5 %, 3072 -0.4 %, 4096 -1.6 / -1.4 %: the limit moves to 1024, no further.
  // The FORWARD chains run at any batch
  // size (evaluation +2 ... +7 %, the training forward at the headline +0.5 %).  INTEL_HEAD_FUSED=2 forces the backward chains on too.
  // With BOTH towers on the one-kernel 32-wide path (tower32.hip) there are no tower launches to hide the head's under: the backward chains pay
  // up to 4096 sessions (published hyper-parameters, same-box: GRU4Rec encoders 1024 sessions +4 %, 4096 +-0, 8192 -5 %; BERT4Rec encoders
  // 1024 / 2048 / 4096: +2 / +4 / +5.5 %).
  const int force = mode == 2;
  const bool tw32_both = D.layers > 0 && tower32_supported(y.L, y.tw[0].d, D.heads, D.layers, 1) && tower32_supported(y.L, y.tw[1].d, D.heads, D.layers, 1);
  const int bwd_limit = !tw32_both ? 1024 : (D.encoder == INTEL_ENC_BERT4REC ? (1 << 30) : 4096);      // (BERT4Rec encoders: still +5 / +4 / +1.3 % at 6144 / 8192 / 16 384 sessions)
  if (train == 1 && y.B > bwd_limit && !force) return false;
  if ((D.d_u % 16) || (D.d_int % 16) || (D.d_c % 4)) return false;
  // LDS tiles of the largest of the four chains (16 sessions x (width + 4) floats per tile)
  const size_t Ip = rup(D.intent_num, 16) + 4, Pp = rup(y.Pin, 16) + 4, Fp = rup(y.F, 16) + 4, dd = y.tw[0].d + y.tw[1].d + 8;
  size_t w = Pp + 2 * Ip + D.d_u + D.d_int + 8 + 2 * dd;                                   // forward a
  w = std::max(w, 2 * 20 + 2 * Fp + 2 * ((size_t)D.d_int + 4) + 2 * Ip + 2 * dd);          // backward a
  w = std::max(w, 5 * Ip + 2 * Pp + 3 * dd);                                                // backward b
  return 16 * w * sizeof(float) < 150 * 1024;
}

static void chain_run(Run& r, ChainPlan& p) {
  if (!p.ok) {
    intel_set_error("session-head chain: op table / tile layout overflow");
    r.ok(INTEL_E_ARG);
    return;
  }
  p.a.B = r.y.B;
  std::stable_sort(p.a.ops, p.a.ops + p.a.nops, [](const ChainOp& x, const ChainOp& z) { return x.level < z.level; });      // the kernel walks level by level
  r.ok(launch_chain(p.a, (size_t)p.lds, r.st));
}

// forward a: [ctx | user | encoder outputs] -> intent logits -> softmax -> relu(h_intent), relu(h_u), pooling queries QV, QK of both towers
static void head_fwd_a(Run& r, const IntelOut* out) {
  const IntelDesc& D = r.D;
  Layout& y = r.y;
  const IntelBatch& bt = *r.bt;
  const int I = D.intent_num, Pin = y.Pin, F = y.F;
  TowerBufs &ti = y.tw[0], &ts = y.tw[1];
  const int off_u = ti.d + ts.d, off_int = off_u + D.d_u, enc0 = D.d_c + D.d_u;
  ChainPlan p;
  p.bf16 = r.D.dtype == INTEL_DTYPE_BF16;
  ChainTile PRED = p.tile(Pin), LOG = p.tile(I), INT = p.tile(I), HU = p.tile(D.d_u), HI = p.tile(D.d_int);
  ChainTile QV[2] = {p.tile(ti.d), p.tile(ts.d)}, QK[2] = {p.tile(ti.d), p.tile(ts.d)};
  p.load(0, r.P(INTEL_P_CTX_EMB), D.d_c, 0, D.d_c, PRED, 0, 0, bt.context_mh, false, y.PREDIN, Pin, 0);
  p.load(0, r.P(INTEL_P_UID_EMB), D.d_u, 0, D.d_u, PRED, D.d_c, 0, bt.u_id_c, false, y.PREDIN, Pin, D.d_c);
  int lv = 1;
  if (D.encoder == INTEL_ENC_GRU4REC && y.enc[0].gru.ext_proj && y.enc[1].gru.ext_proj) {
    // GRU4Rec: the encoders stop at their last hidden state; vec = h_last Wout^T (GeneralSeq.py:76) is the chain's first link
    if (Pin > enc0 + y.enc[0].dm + y.enc[1].dm || PRED.width > Pin) p.load(0, y.PREDIN, Pin, Pin, 0, PRED, Pin, PRED.width - Pin);      // (zero padding)
    for (int e = 0; e < 2; ++e) {
      EncBufs& n = y.enc[e];
      ChainTile HC = p.tile(D.gru_hidden);
      p.load(0, n.gru.HCUR, D.gru_hidden, 0, D.gru_hidden, HC, 0);
      p.lin(1, HC, 0, D.gru_hidden, n.gru.pWout, n.dm, nullptr, PRED, n.predin_off, p.bfl(D.gru_hidden, n.dm), y.PREDIN, Pin, n.predin_off);
    }
    lv = 2;
  } else {
    p.load(0, y.PREDIN, Pin, enc0, Pin - enc0, PRED, enc0, PRED.width - enc0);
  }
  p.load(0, r.P(INTEL_P_UID_EMB), D.d_u, 0, D.d_u, HU, 0, 0, bt.u_id_c, true, y.FEAT, F, off_u);      // relu(h_u), IntEL.py:178
  p.lin(lv, PRED, 0, Pin, y.pPred, I, r.P(INTEL_P_PRED_B), LOG, 0, p.bfl(Pin, I));
  {
    ChainOp& o = p.add(CH_SOFTMAX, lv + 1);
    o.in_off = LOG.off; o.in_ld = LOG.ld; o.out_off = INT.off; o.out_ld = INT.ld; o.N = I; o.NP = INT.width;
    o.gout = y.INTENTS; o.gld = I; o.gcol = 0; o.gout2 = out->intents;
  }
  p.lin(lv + 2, INT, 0, I, y.pInt, D.d_int, r.P(INTEL_P_INTENT_B), HI, 0, CH_RELU | p.bfl(I, D.d_int), y.FEAT, F, off_int);     // relu(h_intent), IntEL.py:212
  for (int t = 0; t < 2; ++t) {
    TowerBufs& w = y.tw[t];
    p.lin(lv + 2, INT, 0, I, w.pXq, w.d, nullptr, QV[t], 0, 0, w.QV, w.d, 0);
    p.lin(lv + 3, QV[t], 0, w.d, w.pXkT, w.d, nullptr, QK[t], 0, p.bfl(w.d, w.d), w.QK, w.d, 0);
  }
  chain_run(r, p);
}

// forward b: pooled value projections -> fusion feature -> weight_embeddings (valid rows / pad rows) -> weights, ens_score
static void head_fwd_b(Run& r, const IntelOut* out) {
  const IntelDesc& D = r.D;
  Layout& y = r.y;
  const IntelBatch& bt = *r.bt;
  const int K = D.model_num, F = y.F;
  TowerBufs &ti = y.tw[0], &ts = y.tw[1];
  const int off_u = ti.d + ts.d, npad = D.d_u + D.d_int;
  ChainPlan p;
  p.bf16 = r.D.dtype == INTEL_DTYPE_BF16;
  ChainTile XB[2] = {p.tile(ti.d), p.tile(ts.d)}, FEAT = p.tile(F), WV = p.tile(16), WP = p.tile(16);
  for (int t = 0; t < 2; ++t) {
    TowerBufs& w = y.tw[t];
    p.load(0, w.XBAR, w.d, 0, w.d, XB[t], 0);
    p.lin(1, XB[t], 0, w.d, w.pXv, w.d, nullptr, FEAT, w.feat_off, p.bfl(w.d, w.d), y.FEAT, F, w.feat_off);
  }
  p.load(0, y.FEAT, F, off_u, npad, FEAT, off_u, FEAT.width - off_u);
  p.lin(2, FEAT, 0, F, y.pWe, K, r.P(INTEL_P_WE_B), WV, 0, p.bfl(F, K), y.WV, K, 0);
  p.lin(2, FEAT, off_u, npad, y.pWePad, K, r.P(INTEL_P_WE_B), WP, 0, p.bfl(npad, K), y.WPAD, K, 0);
  {
    ChainOp& o = p.add(CH_ENS_FWD, 3);
    o.in_off = WV.off; o.in_ld = WV.ld; o.aux_off = WP.off; o.aux_ld = WP.ld;
    p.a.ens.scores = bt.scores; p.a.ens.slen = bt.session_len; p.a.ens.L = y.L; p.a.ens.K = K;
    p.a.ens.weights = out->weights; p.a.ens.ens = out->ens_score;
  }
  chain_run(r, p);
}

// backward a: d(weights), d(ens_score) -> d(fusion weights) -> dFEAT -> d(h_intent) -> first share of d(intent); d(pooled) -> dxbar of both towers
static void head_bwd_a(Run& r, const float* d_weights, const float* d_ens) {
  const IntelDesc& D = r.D;
  Layout& y = r.y;
  const IntelBatch& bt = *r.bt;
  const int I = D.intent_num, K = D.model_num, F = y.F;
  TowerBufs &ti = y.tw[0], &ts = y.tw[1];
  const int off_u = ti.d + ts.d, off_int = off_u + D.d_u, npad = D.d_u + D.d_int;
  ChainPlan p;
  p.bf16 = r.D.dtype == INTEL_DTYPE_BF16;
  ChainTile DWV = p.tile(16), DWP = p.tile(16), DF = p.tile(F), MASK = p.tile(D.d_int), DH = p.tile(D.d_int), DI = p.tile(I);
  ChainTile G1[2] = {p.tile(ti.d), p.tile(ts.d)};
  {
    ChainOp& o = p.add(CH_ENS_BWD, 0);
    o.out_off = DWV.off; o.out_ld = DWV.ld; o.aux_off = DWP.off; o.aux_ld = DWP.ld;
    p.a.ens.scores = bt.scores; p.a.ens.slen = bt.session_len; p.a.ens.L = y.L; p.a.ens.K = K;
    p.a.ens.d_weights = d_weights; p.a.ens.d_ens = d_ens; p.a.ens.dwv = y.dWV; p.a.ens.dwpad = y.dWPAD;
  }
  p.load(0, y.FEAT, F, off_int, D.d_int, MASK, 0);
  p.lin(1, DWV, 0, K, y.pWeT, F, nullptr, DF, 0, p.bfl(F, K), y.dFEAT, F, 0);
  p.lin(2, DWP, 0, K, y.pWePadT, npad, nullptr, DF, off_u, CH_ACCUM | p.bfl(npad, K), y.dFEAT, F, off_u);      // padded rows only see [h_u | h_intent]
  {
    ChainOp& o = p.add(CH_MASKCOPY, 3);
    o.in_off = DF.off + off_int; o.in_ld = DF.ld; o.aux_off = MASK.off; o.aux_ld = MASK.ld; o.out_off = DH.off; o.out_ld = DH.ld;
    o.N = D.d_int; o.NP = DH.width; o.gout = y.dHINT; o.gld = D.d_int; o.gcol = 0;
  }
  for (int t = 0; t < 2; ++t) {
    TowerBufs& w = y.tw[t];
    p.lin(3, DF, w.feat_off, w.d, w.pXvT, w.d, nullptr, G1[t], 0, p.bfl(w.d, w.d), y.dHB[t][0], w.d, 0);      // dxbar
  }
  p.lin(4, DH, 0, D.d_int, y.pIntT, I, nullptr, DI, 0, p.bfl(I, D.d_int), y.dINTENT, I, 0);
  chain_run(r, p);
}

// the weight gradients of head_bwd_a's links -- weight_embeddings (valid + pad rows), intent_embeddings (h_intent's share), the two
// value projections of the pooling -- as their own chain launch: LEAVES of the backward (nothing reads them before the optimizer),
// issued on a side stream so that they do not lengthen the critical chain.  One partial per workgroup -> the reduce queue (final flush).
static void head_bwd_a_leaves(Run& r, int leaf_tag) {
  const IntelDesc& D = r.D;
  Layout& y = r.y;
  const int I = D.intent_num, K = D.model_num, F = y.F;
  TowerBufs &ti = y.tw[0], &ts = y.tw[1];
  const int off_u = ti.d + ts.d, npad = D.d_u + D.d_int;
  ChainPlan p;
  p.bf16 = r.D.dtype == INTEL_DTYPE_BF16;
  const int S = cdiv(y.B, 16);
  const int o_we = 0, o_be = o_we + K * F, o_wi = o_be + K, o_bi = o_wi + D.d_int * I, o_v0 = o_bi + D.d_int, o_v1 = o_v0 + ti.d * ti.d,
            stride = (int)rup_sz((size_t)o_v1 + ts.d * ts.d, 64);
  float* slab = redq_alloc(r.ctx->rq, (size_t)S * stride);
  if (!slab) { intel_set_error("head_bwd_a: reduction arena exhausted"); r.ok(INTEL_E_WORKSPACE); return; }
  const bool g_we = r.G(INTEL_P_WE_W) != nullptr, g_wi = r.G(INTEL_P_INTENT_W) != nullptr;
  if (g_we) {
    ChainTile FE = p.tile(F), DWV = p.tile(16), DWP = p.tile(16);
    p.load(0, y.FEAT, F, 0, F, FE, 0, FE.width);
    p.load(0, y.dWV, K, 0, K, DWV, 0, DWV.width);
    p.load(0, y.dWPAD, K, 0, K, DWP, 0, DWP.width);
    p.wgrad(1, DWV, 0, K, FE, 0, F, slab + o_we, F, 0, stride, slab + o_be, false, p.bfl(F, K));
    p.wgrad(2, DWP, 0, K, FE, off_u, npad, slab + o_we, F, off_u, stride, slab + o_be, true, p.bfl(npad, K));      // padded rows only see [h_u | h_intent]
    const int a = r.acc(INTEL_P_WE_W), ab = r.acc(INTEL_P_WE_B);
    redq_push(r.ctx->rq, slab + o_we, stride, S, K, F, r.G(INTEL_P_WE_W), F, a);
    redq_push(r.ctx->rq, slab + o_be, stride, S, 1, K, r.G(INTEL_P_WE_B), K, ab);
  }
  if (g_wi) {
    ChainTile IN = p.tile(I), DH = p.tile(D.d_int);
    p.load(0, y.INTENTS, I, 0, I, IN, 0, IN.width);
    p.load(0, y.dHINT, D.d_int, 0, D.d_int, DH, 0, DH.width);
    p.wgrad(1, DH, 0, D.d_int, IN, 0, I, slab + o_wi, I, 0, stride, slab + o_bi, false, p.bfl(I, D.d_int));
    const int a = r.acc(INTEL_P_INTENT_W), ab = r.acc(INTEL_P_INTENT_B);
    redq_set_tag(r.ctx->rq, 0);            // the intent-embedding slot is shared with the encoders' branches: final flush
    redq_push(r.ctx->rq, slab + o_wi, stride, S, D.d_int, I, r.G(INTEL_P_INTENT_W), I, a);
    redq_push(r.ctx->rq, slab + o_bi, stride, S, 1, D.d_int, r.G(INTEL_P_INTENT_B), D.d_int, ab);
    redq_set_tag(r.ctx->rq, leaf_tag);
  }
  for (int t = 0; t < 2; ++t) {
    TowerBufs& w = y.tw[t];
    if (!r.G(w.xbase + 2)) continue;
    ChainTile XB = p.tile(w.d), DFt = p.tile(w.d);
    p.load(0, w.XBAR, w.d, 0, w.d, XB, 0);
    p.load(0, y.dFEAT, F, w.feat_off, w.d, DFt, 0);
    p.wgrad(1, DFt, 0, w.d, XB, 0, w.d, slab + (t == 0 ? o_v0 : o_v1), w.d, 0, stride, nullptr, false, p.bfl(w.d, w.d));      // pooled = xbar Wv^T
    redq_push(r.ctx->rq, slab + (t == 0 ? o_v0 : o_v1), stride, S, w.d, w.d, r.G(w.xbase + 2), w.d, r.acc(w.xbase + 2));
  }
  if (p.a.nops) chain_run(r, p);
}

// backward b: dQK of both towers -> dQV -> their shares of d(intent); + the intent loss's own gradient; softmax backward; d(pred_layer input)
static void head_bwd_b(Run& r, const float* d_intents) {
  const IntelDesc& D = r.D;
  Layout& y = r.y;
  const int I = D.intent_num, Pin = y.Pin;
  ChainPlan p;
  p.bf16 = r.D.dtype == INTEL_DTYPE_BF16;
  ChainTile DA = p.tile(I), DB = p.tile(I), DC = p.tile(I), Y = p.tile(I), DL = p.tile(I), DP = p.tile(Pin);
  ChainTile G2[2] = {p.tile(y.tw[0].d), p.tile(y.tw[1].d)}, G3[2] = {p.tile(y.tw[0].d), p.tile(y.tw[1].d)};
  p.load(0, y.dINTENT, I, 0, I, DA, 0, DA.width);
  p.load(0, y.INTENTS, I, 0, I, Y, 0, Y.width);
  for (int t = 0; t < 2; ++t) {
    TowerBufs& w = y.tw[t];
    p.load(0, y.dHB[t][1], w.d, 0, w.d, G2[t], 0);
    p.lin(1, G2[t], 0, w.d, w.pXk, w.d, nullptr, G3[t], 0, p.bfl(w.d, w.d), y.dHB[t][2], w.d, 0);            // dQV
    p.lin(2, G3[t], 0, w.d, w.pXqT, I, nullptr, t == 0 ? DB : DC, 0);
  }
  {
    ChainOp& o = p.add(CH_SOFTMAX_BWD, 3);
    o.in_off = DA.off; o.in_ld = DA.ld; o.in2_off = DB.off; o.in3_off = DC.off; o.aux_off = Y.off; o.aux_ld = Y.ld;
    o.out_off = DL.off; o.out_ld = DL.ld; o.N = I; o.NP = DL.width; o.gadd = d_intents;
    o.gout = y.dLOGITS; o.gld = I; o.gcol = 0;
  }
  p.lin(4, DL, 0, I, y.pPredT, Pin, nullptr, DP, 0, p.bfl(Pin, I), y.dPREDIN, Pin, 0);
  const bool gru_ext = D.encoder == INTEL_ENC_GRU4REC && y.enc[0].gru.ext_proj && y.enc[1].gru.ext_proj;
  if (gru_ext) {
    // GRU4Rec: d(h_last) = d(vec) Wout, d(vec) = the encoder's columns of d(pred_layer input)
    for (int e = 0; e < 2; ++e) {
      EncBufs& n = y.enc[e];
      ChainTile DH = p.tile(D.gru_hidden);
      p.lin(5, DP, n.predin_off, n.dm, n.gru.pWoutT, D.gru_hidden, nullptr, DH, 0, p.bfl(D.gru_hidden, n.dm), n.gru.dHa, D.gru_hidden, 0);
    }
  }
  chain_run(r, p);
}

// the weight gradients of head_bwd_b's links (key / query projections of both poolings, pred_layer, the GRU output projections) as a
// leaf chain launch (see head_bwd_a_leaves)
static void head_bwd_b_leaves(Run& r) {
  const IntelDesc& D = r.D;
  Layout& y = r.y;
  const int I = D.intent_num, Pin = y.Pin;
  ChainPlan p;
  p.bf16 = r.D.dtype == INTEL_DTYPE_BF16;
  const int S = cdiv(y.B, 16);
  int off = 0;
  int o_k[2], o_q[2];
  for (int t = 0; t < 2; ++t) { o_k[t] = off; off += y.tw[t].d * y.tw[t].d; o_q[t] = off; off += y.tw[t].d * I; }
  const int o_wp = off, o_bp = o_wp + I * Pin;
  const bool gru_ext = D.encoder == INTEL_ENC_GRU4REC && y.enc[0].gru.ext_proj && y.enc[1].gru.ext_proj;
  int o_go[2] = {0, 0};
  o_go[0] = (int)rup_sz((size_t)o_bp + I, 64);
  o_go[1] = o_go[0] + (gru_ext ? y.enc[0].dm * D.gru_hidden : 0);
  const int stride = o_go[1] + (gru_ext ? (int)rup_sz((size_t)y.enc[1].dm * D.gru_hidden, 64) : 0);
  float* slab = redq_alloc(r.ctx->rq, (size_t)S * stride);
  if (!slab) { intel_set_error("head_bwd_b: reduction arena exhausted"); r.ok(INTEL_E_WORKSPACE); return; }
  ChainTile Y{0, 0, 0};
  bool haveY = false;
  for (int t = 0; t < 2; ++t) {
    TowerBufs& w = y.tw[t];
    if (r.G(w.xbase + 1)) {          // QK = QV Wk: dWk[i][j] = sum_b QV[b][i] dQK[b][j]
      ChainTile QVt = p.tile(w.d), G2 = p.tile(w.d);
      p.load(0, w.QV, w.d, 0, w.d, QVt, 0);
      p.load(0, y.dHB[t][1], w.d, 0, w.d, G2, 0);
      p.wgrad(1, QVt, 0, w.d, G2, 0, w.d, slab + o_k[t], w.d, 0, stride, nullptr, false, p.bfl(w.d, w.d));
      redq_push(r.ctx->rq, slab + o_k[t], stride, S, w.d, w.d, r.G(w.xbase + 1), w.d, r.acc(w.xbase + 1));
    }
    if (r.G(w.xbase + 0)) {          // QV = intent Wq^T
      if (!haveY) { Y = p.tile(I); p.load(0, y.INTENTS, I, 0, I, Y, 0, Y.width); haveY = true; }
      ChainTile G3 = p.tile(w.d);
      p.load(0, y.dHB[t][2], w.d, 0, w.d, G3, 0);
      p.wgrad(1, G3, 0, w.d, Y, 0, I, slab + o_q[t], I, 0, stride);
      redq_push(r.ctx->rq, slab + o_q[t], stride, S, w.d, I, r.G(w.xbase + 0), I, r.acc(w.xbase + 0));
    }
  }
  ChainTile DPt{0, 0, 0};
  if (r.G(INTEL_P_PRED_W)) {
    ChainTile PR = p.tile(Pin), DL = p.tile(I);
    p.load(0, y.PREDIN, Pin, 0, Pin, PR, 0, PR.width);
    p.load(0, y.dLOGITS, I, 0, I, DL, 0, DL.width);
    p.wgrad(1, DL, 0, I, PR, 0, Pin, slab + o_wp, Pin, 0, stride, slab + o_bp, false, p.bfl(Pin, I));
    const int a = r.acc(INTEL_P_PRED_W), ab = r.acc(INTEL_P_PRED_B);
    redq_push(r.ctx->rq, slab + o_wp, stride, S, I, Pin, r.G(INTEL_P_PRED_W), Pin, a);
    redq_push(r.ctx->rq, slab + o_bp, stride, S, 1, I, r.G(INTEL_P_PRED_B), I, ab);
  }
  if (gru_ext) {      // dWout = d(vec)^T h_last
    for (int e = 0; e < 2; ++e) {
      EncBufs& n = y.enc[e];
      const int Hd = D.gru_hidden, ws = enc_slot(e, INTEL_ENC_GRU_OUT);
      if (!r.G(ws)) continue;
      ChainTile HC = p.tile(Hd), DV = p.tile(n.dm);
      p.load(0, n.gru.HCUR, Hd, 0, Hd, HC, 0);
      p.load(0, y.dPREDIN, Pin, n.predin_off, n.dm, DV, 0, DV.width);
      p.wgrad(1, DV, 0, n.dm, HC, 0, Hd, slab + o_go[e], Hd, 0, stride, nullptr, false, p.bfl(Hd, n.dm));
      redq_push(r.ctx->rq, slab + o_go[e], stride, S, n.dm, Hd, r.G(ws), Hd, r.acc(ws));
    }
  }
  (void)DPt;
  if (p.a.nops) chain_run(r, p);
}

// ---- forward ------------------------------------------------------------------------------------
void forward_impl(Run& r, const IntelOut* out) {
  const IntelDesc& D = r.D;
  Layout& y = r.y;
  const IntelBatch& bt = *r.bt;
  const int B = y.B, L = y.L, M = y.M, I = D.intent_num, K = D.model_num;
  // the side branches start with work that needs no packed weights (history packing, embedding gathers): they fork BEFORE the
  // packing launches of the main stream and wait for them (ev_pack) in front of their first matrix product
  {   // GRU4Rec with the fused session head: the output projections are links of the head's chains -- of BOTH encoders or of neither (the
      // chains take them over only as a pair: an encoder left alone with the flag would have its projection computed by nobody)
    const bool both = D.encoder == INTEL_ENC_GRU4REC && head_fused_ok(D, y, r.train ? 2 : 0) && gru_ext_proj_supported(y.enc[0].dm, D.gru_hidden) &&
                      gru_ext_proj_supported(y.enc[1].dm, D.gru_hidden);
    for (int e = 0; e < 2; ++e) y.enc[e].gru.ext_proj = both;
  }
  fork_streams(r, 3);
  hipEvent_t ev_pack = r.ctx->ev_x[0];
  bool raw_only = false;
  {
    IntelCtx* c = r.ctx;
    const int shape[5] = {y.B, y.L, y.H, y.Hi, (r.train && c->drop_p > 0.f) ? 1 : 0};
    const bool reuse = c->params_unchanged && c->pack_ok && c->pack_ws == (const void*)y.ARENA && memcmp(shape, c->pack_shape, sizeof(shape)) == 0 &&
                       c->pack_train == (r.train != 0);      // the set of images depends on the mode (one-kernel tower layer)
    c->pack_ok = false;
    // When NONE of the four branches reads a packed image -- both towers on the one-kernel 32-wide path, both encoders on the one-kernel BERT4Rec
    // path, the small input linears on their raw-weight kernels: the reference's default widths -- only the session head needs the images: the packing
    // launches go to the score tower's stream (the shortest branch) and the caller's stream waits for them in front of the head instead of in front
    // of its own encoder branch (published IntEL-MSE configuration: the images were ~60 us at the head of every branch).  INTEL_PACK_SIDE=0: off.
    static const int pack_side_on = [] { const char* e = getenv("INTEL_PACK_SIDE"); return (e && e[0] == '0') ? 0 : 1; }();
    const bool towers_raw = pack_side_on && c->streams == 1 && D.layers > 0 && D.cross_attention && !reuse && smallk_supported(D.d_s, K) &&
                            tower32_supported(L, y.tw[0].d, D.heads, D.layers, r.train) && tower32_supported(L, y.tw[1].d, D.heads, D.layers, r.train);
    raw_only = towers_raw && D.encoder == INTEL_ENC_BERT4REC && c->enc32[0] && c->enc32[1] && smallk_supported(D.d_int, I) && bt.his_item_idx != nullptr;
    // (measured and dropped: splitting the packing where only the ENCODERS need images -- GRU4Rec at the published hyper-parameters: no change)
    Run pk = branch(r, raw_only ? 2 : -1, 0);
    if (!reuse) {
      pack_all(pk);
      if (!r.ok(pk.rc)) return;
      c->pack_train = r.train != 0;
    }
    c->pack_ws = (const void*)y.ARENA;       // the layout is a pure function of the shape and the workspace base
    memcpy(c->pack_shape, shape, sizeof(shape));
    c->pack_ok = true;
    if (c->streams == 1) r.ok((int)hipEventRecord(ev_pack, pk.st));
  }
  hipStream_t main_st = r.st;
  auto wait_pack = [&](Run& b) {
    if (r.ctx->streams == 1 && b.st != main_st && !raw_only) b.ok((int)hipStreamWaitEvent(b.st, ev_pack, 0));
  };
  // ===== four independent branches: the two sequence encoders (predict_intent, IntEL.py:126-155) and the
  // two tied self-attention towers (IntEL.py:170-197) run concurrently on four streams
  const bool lazy_gather = r.ctx->lazy.p != nullptr && r.ctx->lazy_upto > r.ctx->lazy.base && r.P(INTEL_P_IID_EMB) == r.ctx->lazy.p;
  // The caller's optimizer may still be sweeping the item-id table on another stream (intel_set_table_wait_event): only the two branches that READ
  // the table wait for it -- the session-history encoder and the score tower start at once, under the sweep (HBM-bound; they are matrix work)
  auto wait_table = [&](Run& b) {
    if (r.ctx->table_wait_ev) b.ok((int)hipStreamWaitEvent(b.st, r.ctx->table_wait_ev, 0));
  };
  auto encoder_branch = [&](Run& r, int e) {
    EncBufs& n = y.enc[e];
    const int rows = r.ctx->enc_rows[e], dm = n.dm;
    const bool pk = r.ctx->enc_packed[e];
    GemmEpilogue eb;
    eb.bias = r.P(INTEL_P_INTENT_B);
    if (e == 0) {
      if (pk) RUN(launch_his_pack(bt.history_len, bt.his_off, B, n.T, bt.his_context_mh, n.pkIds, nullptr, nullptr, bt.his_intents, I, n.pkVec, n.rowT, r.st));
      const float* hint = pk ? n.pkVec : bt.his_intents;
      // packed BERT4Rec rows: the position embedding rides on the kernels that write the input row (one [rows, dm] round trip less)
      n.pos_done = pk && D.encoder == INTEL_ENC_BERT4REC && smallk_supported(D.d_int, I);
      const float* pos = n.pos_done ? r.P(enc_slot(e, INTEL_ENC_POS)) : nullptr;
      RUN(launch_gather_rows(r.P(INTEL_P_CTX_EMB), D.d_c, pk ? n.pkIds : bt.his_context_mh, rows, n.E0, dm, 0, 0, r.st, pos, n.rowT));
      if (smallk_supported(D.d_int, I))
        RUN(launch_linear_smallk(hint, I, rows, I, r.P(INTEL_P_INTENT_W), eb.bias, D.d_int, n.E0 + D.d_c, dm, 0, r.st, pos ? pos + D.d_c : nullptr, n.rowT));
      else
        lin(r, hint, I, rows, I, y.pInt, D.d_int, n.E0 + D.d_c, dm, eb);
    } else {
      if (pk) RUN(launch_his_pack(bt.history_item_len, bt.hisitem_off, B, n.T, bt.his_item_id, n.pkIds, bt.his_item_idx, n.pkIdx2,
                                  bt.his_item_idx ? nullptr : bt.his_item_int, I, n.pkVec, n.rowT, r.st));
      n.pos_done = pk && D.encoder == INTEL_ENC_BERT4REC && bt.his_item_idx != nullptr;
      const float* pos = n.pos_done ? r.P(enc_slot(e, INTEL_ENC_POS)) : nullptr;
      wait_table(r);
      // lazy table Adam (intel_set_lazy_table): rows that are behind are replayed inside the gather
      if (lazy_gather)
        RUN(launch_gather_rows_lazy(r.ctx->lazy, r.ctx->lazy_upto, pk ? n.pkIds : bt.his_item_id, rows, n.E0, dm, 0, r.st, pos, n.rowT));
      else
        RUN(launch_gather_rows(r.P(INTEL_P_IID_EMB), D.d_id, pk ? n.pkIds : bt.his_item_id, rows, n.E0, dm, 0, 0, r.st, pos, n.rowT));
      wait_pack(r);
      if (bt.his_item_idx)
        RUN(launch_onehot_linear(r.P(INTEL_P_INTENT_W), r.P(INTEL_P_INTENT_B), D.d_int, I, pk ? n.pkIdx2 : bt.his_item_idx, rows, n.E0, dm, D.d_id, r.st, pos, n.rowT));
      else
        lin(r, pk ? n.pkVec : bt.his_item_int, I, rows, I, y.pInt, D.d_int, n.E0 + D.d_id, dm, eb);
    }
    if (r.rc) return;
    if (D.encoder == INTEL_ENC_BERT4REC) {
      bert_fwd(r, e);
    } else {
      const int* len = e == 0 ? bt.history_len : bt.history_item_len;
      RUN(gru_fwd(n.gru, n.E0, B, n.T, dm, D.gru_hidden, len, r.P(enc_slot(e, INTEL_ENC_GRU_BIH)),
                  r.P(enc_slot(e, INTEL_ENC_GRU_BHH)), y.PREDIN, y.Pin, n.predin_off, r.st, r.P(enc_slot(e, INTEL_ENC_GRU_WHH)),
                  pk ? (e == 0 ? bt.his_off : bt.hisitem_off) : nullptr, rows, e == 0 ? bt.his_order : bt.hisitem_order, r.train != 0));
    }
  };
  TowerBufs& ti = y.tw[0];
  TowerBufs& ts = y.tw[1];
  {
    Run b1 = branch(r, 0, 1), b2 = branch(r, 1, 0), b3 = branch(r, 2, 1);
    encoder_branch(r, 0);                        // session-history encoder on the caller's stream
    encoder_branch(b1, 1);
    // item tower
    wait_table(b2);
    if (!lazy_gather && tower_input_in_kernel(r, ti) && D.d_id % 4 == 0) {
      // inference: the candidate rows go from the tables straight into the first layer's LDS tile (IntEL.py:170-173 inside tower.hip)
      TowerInput in;
      in.tab0 = r.P(INTEL_P_IID_EMB); in.idx0 = bt.i_id_s; in.d0 = D.d_id;
      if (D.d_im > 0) { in.tab1 = r.P(INTEL_P_ITEM_EMB); in.idx1 = bt.i_class_c; }
      wait_pack(b2);
      tower_fwd(b2, ti, &in);
    } else
    if (b2.ok(lazy_gather ? launch_gather_rows_lazy(r.ctx->lazy, r.ctx->lazy_upto, bt.i_id_s, M, ti.X0, ti.d, 0, b2.st, nullptr, nullptr)
                          : launch_gather_rows(r.P(INTEL_P_IID_EMB), D.d_id, bt.i_id_s, M, ti.X0, ti.d, 0, 0, b2.st)) &&
        (D.d_im == 0 || b2.ok(launch_gather_rows(r.P(INTEL_P_ITEM_EMB), D.d_im, bt.i_class_c, M, ti.X0, ti.d, D.d_id, 0, b2.st)))) {
      wait_pack(b2);
      tower_fwd(b2, ti);
    }
    // score tower (its K-wide input linear stays a launch of its own: computed inside the one-kernel layer, the layer ran 137 -> 183 us per launch at the
    // headline shape for the 21 us the linear takes -- measured and dropped)
    {
      GemmEpilogue es;
      es.bias = r.P(INTEL_P_SCORE_B);
      if (smallk_supported(D.d_s, K)) {
        b3.ok(launch_linear_smallk(bt.scores, K, M, K, r.P(INTEL_P_SCORE_W), es.bias, D.d_s, ts.X0, D.d_s, 0, b3.st));
        wait_pack(b3);
      } else {
        wait_pack(b3);
        lin(b3, bt.scores, K, M, K, y.pScore, D.d_s, ts.X0, D.d_s, es);
      }
      if (!b3.rc) tower_fwd(b3, ts);
    }
    r.ok(b1.rc); r.ok(b2.rc); r.ok(b3.rc);
  }
  // the intent prediction needs the two encoders only (main + side 0): it and the intent-side projections of the pooling run
  // while the towers (side 1, side 2) are still busy; each pooling waits for its own tower
  wait_side(r, 0, r.st);
  if (raw_only) r.ok((int)hipStreamWaitEvent(r.st, ev_pack, 0));      // the head's chains / products read the packed images
  if (r.rc) return;
  if (head_fused_ok(D, y, r.train ? 2 : 0)) {
    // the session head as two chain launches around the two pooling kernels (chain.hip)
    const float scale = 1.0f / sqrtf((float)D.q_size);
    head_fwd_a(r, out);
    if (r.rc) return;
    fork_streams(r, 1);
    {
      Run b1 = branch(r, 0, 1);
      for (int t = 1; t >= 0; --t) {
        Run& q = t == 1 ? b1 : r;
        TowerBufs& w = y.tw[t];
        wait_side(q, 1 + t, q.st);                 // tower t ran on side stream 1 + t
        if (tail_fusable(r.ctx, D, L, w.d, r.train)) {
          q.ok(launch_xatt_pool_fwd(w.layer[D.layers - 1].XH, B, L, w.d, w.QK, bt.session_len, scale, w.XBAR, w.ATTW, q.st,
                                    r.P(w.pbase + T_LNG), r.P(w.pbase + T_LNB)));
        } else {
          q.ok(launch_xatt_pool_fwd(D.layers > 0 ? w.layer[D.layers - 1].Xout : w.X0, B, L, w.d, w.QK, bt.session_len, scale, w.XBAR, w.ATTW, q.st));
        }
      }
      r.ok(b1.rc);
    }
    join_streams(r, 1);
    if (r.rc) return;
    head_fwd_b(r, out);
    return;
  }
  RUN(launch_gather_rows(r.P(INTEL_P_CTX_EMB), D.d_c, bt.context_mh, B, y.PREDIN, y.Pin, 0, 0, r.st));
  RUN(launch_gather_rows(r.P(INTEL_P_UID_EMB), D.d_u, bt.u_id_c, B, y.PREDIN, y.Pin, D.d_c, 0, r.st));
  {
    GemmEpilogue ep;
    ep.bias = r.P(INTEL_P_PRED_B);
    lin(r, y.PREDIN, y.Pin, B, y.Pin, y.pPred, I, y.LOGITS, I, ep);
    if (r.rc) return;
    RUN(launch_softmax_rows(y.LOGITS, B, I, y.INTENTS, r.st));
  }
  // ===== predict_ensemble, after the towers (IntEL.py:199-217)
  const int off_u = ti.d + ts.d, off_int = off_u + D.d_u;
  const float scale = 1.0f / sqrtf((float)D.q_size);
  GemmEpilogue e0;
  // the two towers' intent-conditioned pooling chains (five small launches each) are independent: item tower on the
  // main stream, score tower on a side stream
  auto pool_tower = [&](Run& r, int t) {
    TowerBufs& w = y.tw[t];
    const float* Xf = D.layers > 0 ? w.layer[D.layers - 1].Xout : w.X0;
    if (D.cross_attention) {
      lin(r, y.INTENTS, I, B, I, w.pXq, w.d, w.QV, w.d, e0);
      lin(r, w.QV, w.d, B, w.d, w.pXkT, w.d, w.QK, w.d, e0);
      if (r.rc) return;
      wait_side(r, 1 + t, r.st);                 // tower t ran on side stream 1 + t
      if (tail_fusable(r.ctx, D, L, w.d, r.train)) {
        const int pb = w.pbase;
        RUN(launch_xatt_pool_fwd(w.layer[D.layers - 1].XH, B, L, w.d, w.QK, bt.session_len, scale, w.XBAR, w.ATTW, r.st,
                                 r.P(pb + T_LNG), r.P(pb + T_LNB)));
      } else {
        RUN(launch_xatt_pool_fwd(Xf, B, L, w.d, w.QK, bt.session_len, scale, w.XBAR, w.ATTW, r.st));
      }
      lin(r, w.XBAR, w.d, B, w.d, w.pXv, w.d, y.FEAT + w.feat_off, y.F, e0);
    } else {
      const int mb = t == 0 ? INTEL_P_MI_W0 : INTEL_P_MS_W0;
      GemmEpilogue eh;
      eh.bias = r.P(mb + 1);
      eh.relu = 1;
      lin(r, y.INTENTS, I, B, I, w.pM0, D.q_size, w.MH, D.q_size, eh);
      lin(r, w.MH, D.q_size, B, D.q_size, w.pM2, w.d, w.MV, w.d, e0);
      if (r.rc) return;
      wait_side(r, 1 + t, r.st);                 // tower t ran on side stream 1 + t
      if (D.pool_mean)      // aWELv_IntEL.py:195-198: (h * g(intent)).mean(dim=1), unmasked
        RUN(launch_gate_mean_fwd(Xf, w.d, w.MV, B, L, w.XBAR, y.FEAT, y.F, w.feat_off, r.st));
      else
        RUN(launch_gate_fwd(Xf, w.d, w.MV, B, L, y.FEATFULL, y.F, w.feat_off, r.st));
    }
  };
  fork_streams(r, 1);
  // the user / intent columns of the fusion feature do not depend on the towers either
  RUN(launch_gather_rows(r.P(INTEL_P_UID_EMB), D.d_u, bt.u_id_c, B, y.FEAT, y.F, off_u, 1, r.st));     // relu(h_u)
  {
    GemmEpilogue eh;
    eh.bias = r.P(INTEL_P_INTENT_B);
    eh.relu = 1;
    lin(r, y.INTENTS, I, B, I, y.pInt, D.d_int, y.FEAT + off_int, y.F, eh);                             // relu(h_intent)
    if (r.rc) return;
  }
  {
    Run b1 = branch(r, 0, 1);
    pool_tower(b1, 1);
    pool_tower(r, 0);
    r.ok(b1.rc);
  }
  join_streams(r, 1);
  if (r.rc) return;
  GemmEpilogue ew;
  ew.bias = r.P(INTEL_P_WE_B);
  if (per_session_weights(D)) {
    lin(r, y.FEAT, y.F, B, y.F, y.pWe, K, y.WV, K, ew);
    if (!D.pool_mean) lin(r, y.FEAT + off_u, y.F, B, D.d_u + D.d_int, y.pWePad, K, y.WPAD, K, ew);
    if (r.rc) return;
    // weight_norm: softmax over the K weights, once or twice (aWELv_IntEL.py:199-200); with pool_mean
    // every row of the list, pads included, carries the session's vector (aWELv_IntEL.py:200: .repeat over the list)
    const float *wv = y.WV, *wpad = D.pool_mean ? y.WV : y.WPAD;
    for (int s = 0; s < D.weight_norm; ++s) {
      RUN(launch_softmax_rows(wv, B, K, y.WVN[s], r.st));
      if (!D.pool_mean) RUN(launch_softmax_rows(wpad, B, K, y.WPADN[s], r.st));
      wv = y.WVN[s];
      wpad = D.pool_mean ? y.WVN[s] : y.WPADN[s];
    }
    RUN(launch_ens_fwd(wv, wpad, bt.scores, bt.session_len, B, L, K, 0, out->weights, out->ens_score, r.st));
  } else {
    RUN(launch_bcast_rows(y.FEAT + off_u, y.F, D.d_u + D.d_int, B, L, y.FEATFULL, y.F, off_u, r.st));
    lin(r, y.FEATFULL, y.F, M, y.F, y.pWe, K, out->weights, K, ew);
    if (r.rc) return;
    for (int s = 0; s < D.weight_norm; ++s) {
      RUN(launch_softmax_rows(s == 0 ? out->weights : y.WTN[s - 1], M, K, y.WTN[s], r.st));
      if (s == D.weight_norm - 1) {
        hipError_t e = hipMemcpyAsync(out->weights, y.WTN[s], (size_t)M * K * sizeof(float), hipMemcpyDeviceToDevice, r.st);
        if (e != hipSuccess) { intel_set_error("weights copy failed: %s", hipGetErrorString(e)); r.rc = (int)e; return; }
      }
    }
    RUN(launch_ens_fwd(nullptr, nullptr, bt.scores, bt.session_len, B, L, K, 1, out->weights, out->ens_score, r.st));
  }
  hipError_t e = hipMemcpyAsync(out->intents, y.INTENTS, (size_t)B * I * sizeof(float), hipMemcpyDeviceToDevice, r.st);
  if (e != hipSuccess) {
    intel_set_error("intents copy failed: %s", hipGetErrorString(e));
    r.rc = (int)e;
  }
}

// ---- backward -----------------------------------------------------------------------------------
// phase 0 = whole backward; phase 1 = everything the item-id table gradient depends on (so that its
// data-parallel all-reduce can start) ; phase 2 = the rest (score tower layers, session-history encoder)
void backward_impl(Run& r, const float* d_weights, const float* d_ens, const float* d_intents, int phase) {
  const IntelDesc& D = r.D;
  Layout& y = r.y;
  const IntelBatch& bt = *r.bt;
  const int B = y.B, L = y.L, M = y.M, I = D.intent_num, K = D.model_num;
  TowerBufs& ti = y.tw[0];
  TowerBufs& ts = y.tw[1];
  const int off_u = ti.d + ts.d, off_int = off_u + D.d_u, npad = D.d_u + D.d_int;
  const float scale = 1.0f / sqrtf((float)D.q_size);
  GemmEpilogue e0;
  GemmEpilogue eacc;
  eacc.accumulate = 1;
  redq_reset(r.ctx->rq, y.ARENA, y.arena_floats);
  // The weight-gradient products and table scatters of the session head are LEAVES of the dependency graph: nothing in this
  // backward reads their results.  The one-call schedule keeps them off the critical chain (loss -> fusion weights -> cross
  // attention backward -> d(intent) -> encoders): they are collected here and launched once the chain has been enqueued.
  const bool wide = phase == 0 && ensure_streams(r.ctx);
  struct Leaf { std::function<void(Run&)> f; bool shared; };      // shared: writes a slot other branches write too
  std::vector<Leaf> lv_main, lv_score;
  std::vector<Leaf>* defer = wide ? &lv_main : nullptr;
  auto leaf = [&](Run& q, std::function<void(Run&)> f, bool shared = false) {
    if (defer) defer->push_back(Leaf{f, shared});
    else f(q);
  };
  // tag > 0: the leaves' reduction jobs are pushed under it (the caller reduces them right away), shared slots stay untagged
  auto run_leaves = [&](Run& q, std::vector<Leaf>& v, int tag) {
    for (Leaf& l : v) {
      redq_set_tag(r.ctx->rq, l.shared ? 0 : tag);
      l.f(q);
    }
    redq_set_tag(r.ctx->rq, tag);
    v.clear();
  };
  // the session head's data-gradient chain as two chain launches around the pooling backward (chain.hip): one-call schedule only
  const bool hfused = wide && head_fused_ok(D, y, 1);
  if (!hfused)      // (the forward may have left the GRU output projections to its chains: this backward does them in gru_bwd)
    for (int e = 0; e < 2; ++e) y.enc[e].gru.ext_proj = false;
  if (phase != 2) {
    memset(r.ctx->touched, 0, sizeof(r.ctx->touched));
    // embedding tables accumulate with atomics into caller-zeroed buffers
    r.ctx->touched[INTEL_P_IID_EMB] = r.ctx->touched[INTEL_P_ITEM_EMB] = 1;
    r.ctx->touched[INTEL_P_UID_EMB] = r.ctx->touched[INTEL_P_CTX_EMB] = 1;

    // ===== fusion weights + aggregation (IntEL.py:212-215)
    if (hfused) {
      // dWV, dWPAD, dFEAT, dHINT, first share of d(intent), dxbar of both towers + the weight gradients of weight_embeddings,
      // intent_embeddings (h_intent's share) and the pooling value projections
      head_bwd_a(r, d_weights, d_ens);
      if (r.rc) return;
    } else if (per_session_weights(D)) {
      RUN(launch_ens_bwd(d_weights, d_ens, bt.scores, bt.session_len, B, L, K, 0, y.dWV, y.dWPAD, nullptr, r.st));
      if (D.pool_mean) RUN(launch_add2(y.dWV, y.dWPAD, (long long)B * K, y.dWV, r.st));      // pad rows carry the same vector
      for (int s = D.weight_norm - 1; s >= 0; --s) {                                           // softmax stages, last first
        RUN(launch_softmax_rows_bwd(y.WVN[s], y.dWV, B, K, y.dWV, r.st));
        if (!D.pool_mean) RUN(launch_softmax_rows_bwd(y.WPADN[s], y.dWPAD, B, K, y.dWPAD, r.st));
      }
      leaf(r, [=, &y](Run& r) {
        wgrad(r, y.dWV, K, y.FEAT, y.F, B, K, y.F, INTEL_P_WE_W, INTEL_P_WE_B);
        if (r.rc) return;
        if (!r.D.pool_mean && r.G(INTEL_P_WE_W)) {   // padded rows only see [h_u | h_intent]
          RUN(launch_wgrad(y.dWPAD, K, y.FEAT + off_u, y.F, B, K, npad, r.G(INTEL_P_WE_W) + off_u, y.F, r.G(INTEL_P_WE_B), 1, nullptr, r.st, r.ctx->rq));
        }
      });
      if (r.rc) return;
      lin(r, y.dWV, K, B, K, y.pWeT, y.F, y.dFEAT, y.F, e0);
      if (!D.pool_mean) lin(r, y.dWPAD, K, B, K, y.pWePadT, npad, y.dFEAT + off_u, y.F, eacc);
    } else {
      RUN(launch_ens_bwd(d_weights, d_ens, bt.scores, bt.session_len, B, L, K, 1, nullptr, nullptr, y.dWT, r.st));
      for (int s = D.weight_norm - 1; s >= 0; --s) RUN(launch_softmax_rows_bwd(y.WTN[s], y.dWT, M, K, y.dWT, r.st));
      leaf(r, [=, &y](Run& r) { wgrad(r, y.dWT, K, y.FEATFULL, y.F, M, K, y.F, INTEL_P_WE_W, INTEL_P_WE_B); });
      lin(r, y.dWT, K, M, K, y.pWeT, y.F, y.dFEATFULL, y.F, e0);
      if (r.rc) return;
      // per-session parts of the feature: h_u, h_intent are broadcast over the list
      RUN(launch_session_colsum(y.dFEATFULL, y.F, off_u, npad, B, L, y.dFEAT, y.F, off_u, 0, r.st));
    }
    if (r.rc) return;
    // h_u = relu(uid_emb[u])
    if (r.G(INTEL_P_UID_EMB))
      leaf(r, [=, &y, &bt](Run& r) {
        RUN(launch_scatter_add_rows(y.dFEAT, y.F, off_u, r.D.d_u, bt.u_id_c, B, r.G(INTEL_P_UID_EMB), y.FEAT, y.F, off_u, r.st));
      });
    // h_intent = relu(intent_embeddings(intent)): dpre -> dHINT [B, d_int]
    if (!hfused) RUN(launch_copy_cols(y.dFEAT, y.F, off_int, D.d_int, B, y.dHINT, D.d_int, 0, y.FEAT, y.F, off_int, 0, r.st));
    if (!hfused) leaf(r, [=, &y](Run& r) { wgrad(r, y.dHINT, r.D.d_int, y.INTENTS, I, B, r.D.d_int, I, INTEL_P_INTENT_W, INTEL_P_INTENT_B); }, true);
    if (!hfused) lin(r, y.dHINT, D.d_int, B, D.d_int, y.pIntT, I, y.dINTENT, I, e0);      // first contribution to d(intent)
    if (r.rc) return;
  }

  // ===== cross attention backward of one tower (cheap: B-row GEMMs + one pass over [B,L,d]); leaves the
  // gradient w.r.t. the tower output in dXout and this tower's contribution to d(intent) in dint
  auto xatt_bwd = [&](Run& r, int t, float* dXout, float* dint) {
    TowerBufs& w = y.tw[t];
    const int d = w.d;
    const float* Xf = D.layers > 0 ? w.layer[D.layers - 1].Xout : w.X0;
    float *g1 = y.dHB[t][0], *g2 = y.dHB[t][1], *g3 = y.dHB[t][2];      // B-row gradients (kept for the deferred weight gradients)
    if (D.cross_attention) {
      const int xb = w.xbase;
      // pooled = xbar Wv^T
      leaf(r, [=, &y, &w](Run& r) { wgrad(r, y.dFEAT + w.feat_off, y.F, w.XBAR, d, B, d, d, xb + 2, -1); });
      lin(r, y.dFEAT + w.feat_off, y.F, B, d, w.pXvT, d, g1, d, e0);          // dxbar
      if (r.rc) return;
      if (r.ctx->fused_tail[t]) {       // dXout receives dZ: the gradient BEHIND the last layer's LayerNorm
        const int pb = w.pbase;
        TowerLayerBufs& lb = w.layer[D.layers - 1];
        const int a = r.acc(pb + T_LNG);
        r.acc(pb + T_LNB);
        RUN(launch_xatt_pool_ln_bwd(lb.XH, lb.RSTD, r.P(pb + T_LNG), r.P(pb + T_LNB), B, L, d, w.QK, w.ATTW, g1, d, scale, dXout,
                                    g2, r.G(pb + T_LNG), r.G(pb + T_LNB), a, r.st, r.ctx->rq));
      } else {
        RUN(launch_xatt_pool_bwd(Xf, B, L, d, w.QK, w.ATTW, g1, d, scale, dXout, g2, r.st));   // dX, dQK
      }
      // QK = QV Wk  (QK[b][j] = sum_i QV[b][i] Wk[i][j])
      leaf(r, [=, &w](Run& r) { wgrad(r, w.QV, d, g2, d, B, d, d, xb + 1, -1); });
      lin(r, g2, d, B, d, w.pXk, d, g3, d, e0);                            // dQV
      leaf(r, [=, &y](Run& r) { wgrad(r, g3, d, y.INTENTS, I, B, d, I, xb + 0, -1); });
      lin(r, g3, d, B, d, w.pXqT, I, dint, I, e0);
    } else {
      const int mb = t == 0 ? INTEL_P_MI_W0 : INTEL_P_MS_W0;
      if (D.pool_mean)
        RUN(launch_gate_mean_bwd(y.dFEAT, y.F, w.feat_off, w.XBAR, d, w.MV, B, L, dXout, g1, r.st));   // dX (every row), dMV
      else
        RUN(launch_gate_bwd(y.dFEATFULL, y.F, w.feat_off, Xf, d, w.MV, B, L, dXout, g1, r.st));    // dX, dMV
      const int qs = D.q_size;
      leaf(r, [=, &w](Run& r) { wgrad(r, g1, d, w.MH, qs, B, d, qs, mb + 2, -1); });
      GemmEpilogue em;
      em.mask = w.MH; em.ldmask = D.q_size;
      lin(r, g1, d, B, d, w.pM2T, D.q_size, g2, D.q_size, em);            // d(pre-relu hidden)
      leaf(r, [=, &y](Run& r) { wgrad(r, g2, qs, y.INTENTS, I, B, qs, I, mb + 0, mb + 1); });
      lin(r, g2, D.q_size, B, D.q_size, w.pM0T, I, dint, I, e0);
    }
  };
  // the same with the chain launches around it (hfused): only the pooling kernel; dxbar (g1) comes from head_bwd_a, dQV (g3), the
  // share of d(intent) and the three projections' weight gradients from the chains
  auto pool_bwd_fused = [&](Run& r, int t, float* dXout) {
    TowerBufs& w = y.tw[t];
    const int d = w.d;
    float *g1 = y.dHB[t][0], *g2 = y.dHB[t][1];
    if (r.ctx->fused_tail[t]) {       // dXout receives dZ: the gradient BEHIND the last layer's LayerNorm
      const int pb = w.pbase;
      TowerLayerBufs& lb = w.layer[D.layers - 1];
      const int a = r.acc(pb + T_LNG);
      r.acc(pb + T_LNB);
      RUN(launch_xatt_pool_ln_bwd(lb.XH, lb.RSTD, r.P(pb + T_LNG), r.P(pb + T_LNB), B, L, d, w.QK, w.ATTW, g1, d, scale, dXout,
                                  g2, r.G(pb + T_LNG), r.G(pb + T_LNB), a, r.st, r.ctx->rq));
    } else {
      RUN(launch_xatt_pool_bwd(D.layers > 0 ? w.layer[D.layers - 1].Xout : w.X0, B, L, d, w.QK, w.ATTW, g1, d, scale, dXout, g2, r.st));
    }
  };
  // ===== tied self-attention layers of the item tower + its embedding-table gradients
  auto item_tower_bwd = [&](Run& r, float* dXout) {
    TowerBufs& w = y.tw[0];
    const int d = w.d;
    float* dX0 = tower_bwd(r, w, dXout, dXout == r.T->dXa ? r.T->dXb : r.T->dXa, r.ctx->fused_tail[0]);
    if (r.rc || !dX0) return;
    const bool vec = (D.d_id % 16 == 0) && D.d_id <= 128 && (D.d_id & (D.d_id - 1)) == 0;
    const bool vecm = (D.d_im % 16 == 0) && D.d_im <= 128 && (D.d_im & (D.d_im - 1)) == 0;
    if (r.G(INTEL_P_IID_EMB)) {
      if (bt.iid_sort_ids && bt.iid_sort_rows && vec)
        RUN(launch_scatter_add_sorted(dX0, d, 0, D.d_id, bt.iid_sort_ids, bt.iid_sort_rows, M, r.G(INTEL_P_IID_EMB), r.st, r.ctx->iid_row_flags));
      else
        RUN(launch_scatter_add_rows(dX0, d, 0, D.d_id, bt.i_id_s, M, r.G(INTEL_P_IID_EMB), nullptr, 0, 0, r.st, r.ctx->iid_row_flags));
    }
    if (D.d_im > 0 && r.G(INTEL_P_ITEM_EMB)) {
      if (bt.cls_sort_ids && bt.cls_sort_rows && vecm)
        RUN(launch_scatter_add_sorted(dX0, d, D.d_id, D.d_im, bt.cls_sort_ids, bt.cls_sort_rows, M, r.G(INTEL_P_ITEM_EMB), r.st));
      else
        RUN(launch_scatter_add_rows(dX0, d, D.d_id, D.d_im, bt.i_class_c, M, r.G(INTEL_P_ITEM_EMB), nullptr, 0, 0, r.st));
    }
  };
  // ===== one sequence encoder: returns dE (gradient w.r.t. its input rows), scatters the table part
  auto encoder_branch = [&](Run& r, int e) -> float* {
    EncBufs& n = y.enc[e];
    const int rows = r.ctx->enc_rows[e], dm = n.dm;
    const bool pk = r.ctx->enc_packed[e];
    float* dE = nullptr;
    if (D.encoder == INTEL_ENC_BERT4REC) {
      dE = bert_bwd(r, e);
    } else {
      const int* len = e == 0 ? bt.history_len : bt.history_item_len;
      GruGrads gg;
      gg.dWih = r.G(enc_slot(e, INTEL_ENC_GRU_WIH)); gg.dWhh = r.G(enc_slot(e, INTEL_ENC_GRU_WHH));
      gg.dbih = r.G(enc_slot(e, INTEL_ENC_GRU_BIH)); gg.dbhh = r.G(enc_slot(e, INTEL_ENC_GRU_BHH));
      gg.dWout = r.G(enc_slot(e, INTEL_ENC_GRU_OUT));
      dE = r.T->dXa;
      r.ok(gru_bwd(n.gru, n.E0, B, n.T, dm, D.gru_hidden, len, r.P(enc_slot(e, INTEL_ENC_GRU_WHH)),
                   r.P(enc_slot(e, INTEL_ENC_GRU_BHH)), y.dPREDIN, y.Pin, n.predin_off, gg, dE, r.T->dXb, r.T->SLABS, r.st,
                   pk ? (e == 0 ? bt.his_off : bt.hisitem_off) : nullptr, rows, e == 0 ? bt.his_order : bt.hisitem_order, r.ctx->rq));
    }
    if (r.rc || !dE) return nullptr;
    if (e == 0) {
      if (r.G(INTEL_P_CTX_EMB))
        r.ok(launch_scatter_add_rows(dE, dm, 0, D.d_c, pk ? n.pkIds : bt.his_context_mh, rows, r.G(INTEL_P_CTX_EMB), nullptr, 0, 0, r.st));
    } else {
      const bool vec = (D.d_id % 16 == 0) && D.d_id <= 128 && (D.d_id & (D.d_id - 1)) == 0;
      if (r.G(INTEL_P_IID_EMB)) {
        if (bt.hisitem_sort_ids && bt.hisitem_sort_rows && vec)
          r.ok(launch_scatter_add_sorted(dE, dm, 0, D.d_id, bt.hisitem_sort_ids, bt.hisitem_sort_rows, B * n.T, r.G(INTEL_P_IID_EMB), r.st,
                                         r.ctx->iid_row_flags, pk ? bt.hisitem_off : nullptr, bt.history_item_len, n.T));
        else
          r.ok(launch_scatter_add_rows(dE, dm, 0, D.d_id, pk ? n.pkIds : bt.his_item_id, rows, r.G(INTEL_P_IID_EMB), nullptr, 0, 0, r.st, r.ctx->iid_row_flags));
      }
    }
    return dE;
  };
  // gradient of the SHARED intent_embeddings weight from one encoder's input rows (main stream, fixed order)
  auto intent_wgrad = [&](Run& r, int e, float* dE) {
    const EncBufs& n = y.enc[e];
    const int rows = r.ctx->enc_rows[e];
    const bool pk = r.ctx->enc_packed[e];
    if (e == 0) {
      wgrad(r, dE + D.d_c, n.dm, pk ? n.pkVec : bt.his_intents, I, rows, D.d_int, I, INTEL_P_INTENT_W, INTEL_P_INTENT_B);
    } else if (bt.his_item_idx) {
      if (r.G(INTEL_P_INTENT_W)) {   // dW[c][j] = sum_m dE[m][c] onehot[m][j]: the dense wgrad on a materialised one-hot
        RUN(launch_make_onehot(pk ? n.pkIdx2 : bt.his_item_idx, nullptr, 0, rows, I, y.ONEHOT2, r.st));
        wgrad(r, dE + D.d_id, n.dm, y.ONEHOT2, I, rows, D.d_int, I, INTEL_P_INTENT_W, INTEL_P_INTENT_B);
      }
    } else {
      wgrad(r, dE + D.d_id, n.dm, pk ? n.pkVec : bt.his_item_int, I, rows, D.d_int, I, INTEL_P_INTENT_W, INTEL_P_INTENT_B);
    }
  };

  // ===== the whole backward in one call (phase 0) with the branches on four streams: nothing heavy waits for a chain of small
  // launches it does not depend on.
  //   side 2:  cross-attention backward of the score tower -> [x1] -> score tower layers (set 3) -> wait c -> the head's leaves
  //   main:    cross-attention backward of the item tower -> [x0] -> wait x1 -> d(intent) chain -> [c] -> item-history encoder (set 1)
  //   side 1:  wait x0 -> item tower layers + item-id / class table gradients (set 0)
  //   side 0:  wait c -> session-history encoder (set 2)
  //   main:    join; [iid] (the caller's table stream waits for it: intel_set_table_stream); shared intent-embedding gradients, reductions
  // The two-call form (phases 1 and 2) keeps its order: there the caller overlaps the table's all-reduce with phase 2.
  if (wide && hfused) {
    // The same four branches with the session head as chain launches.  The d(intent) chain (head_bwd_b) is the critical link -- both
    // encoders wait for it -- and it is a SMALL launch (B / 16 workgroups) that must not queue behind the towers' kernels for LDS:
    //   main:    [head_bwd_a above] -> pooling backward of the item tower -> [x0] -> wait x1 -> head_bwd_b -> [c] -> item-history encoder
    //   side 2:  pooling backward of the score tower -> [x1] -> wait c -> score tower (set 3) -> the head's table scatters
    //   side 1:  wait x0, c -> item tower + item-id / class table gradients (set 0)
    //   side 0:  wait c -> session-history encoder (set 2)
    IntelCtx* c = r.ctx;
    r.T = &y.tmp[0];
    Run m = r;
    Run s2 = branch(r, 2, 1), s1 = branch(r, 1, 0), s0 = branch(r, 0, 2);
    // every branch reduces its own weight-gradient slabs on its own stream: the final flush (on the critical tail, in front of the dense
    // groups' Adam) is left with the shared intent-embedding slot
    defer = &lv_main;
    r.ok((int)hipEventRecord(c->ev_fork, r.st));
    r.ok((int)hipStreamWaitEvent(c->side[2], c->ev_fork, 0));
    redq_set_tag(c->rq, 1);
    pool_bwd_fused(s2, 1, y.dXS);                      // (its LayerNorm partials, when the tail is fused, belong to the score tower's tag)
    r.ok((int)hipEventRecord(c->ev_x[1], s2.st));
    redq_set_tag(c->rq, 2);
    pool_bwd_fused(m, 0, y.tmp[0].dXa);
    r.ok(m.rc); r.ok(s2.rc);
    if (r.rc) return;
    r.ok((int)hipEventRecord(c->ev_x[0], r.st));
    redq_set_tag(c->rq, 0);
    r.ok((int)hipStreamWaitEvent(r.st, c->ev_x[1], 0));
    head_bwd_b(r, d_intents);                          // dQV of both towers, d(intent), softmax backward, d(pred_layer input) + their weight gradients
    if (r.rc) return;
    r.ok((int)hipEventRecord(c->ev_x[2], r.st));
    r.ok((int)hipStreamWaitEvent(c->side[0], c->ev_x[2], 0));
    r.ok((int)hipStreamWaitEvent(c->side[1], c->ev_x[0], 0));
    // The weight gradients of the head's links (two chain launches, nothing in this backward reads them; reduced right there: tag 5, the shared
    // intent-embedding slot stays for the final flush) go behind the SHORTER tower branch.  With both towers on the one-kernel 32-wide path that is
    // the item tower's stream (round 4); with 64 / 128-wide towers the item tower's branch is the backward's longest and carries the table's Adam
    // sweep behind it -- the two launches and their reduction there pushed the sweep out by ~150 us (the whole of the -1.8 % that kept the backward
    // chains off at the headline's batch): behind the score tower they end a millisecond before anything waits for them.
    const bool leaves_behind_score = !(D.layers > 0 && tower32_supported(L, y.tw[0].d, D.heads, D.layers, 1) && tower32_supported(L, y.tw[1].d, D.heads, D.layers, 1));
    auto chain_leaves = [&](Run& q) {
      redq_set_tag(c->rq, 5);
      if (!q.rc) head_bwd_a_leaves(q, 5);
      q.ok((int)hipStreamWaitEvent(q.st, c->ev_x[2], 0));
      if (!q.rc) head_bwd_b_leaves(q);
      if (!q.rc) q.ok(redq_flush_tag(c->rq, 5, q.st));
      redq_set_tag(c->rq, 0);
    };
    {   // score tower: it needs its pooling backward only (the tower kernels leave room for the chain launches: tower32.hip)
      redq_set_tag(c->rq, 1);
      Run t3 = s2;
      t3.T = &y.tmp[3];
      t3.rc = 0;
      TowerBufs& w = y.tw[1];
      float* dX0 = tower_bwd(t3, w, y.dXS, y.tmp[3].dXb, c->fused_tail[1]);
      if (!t3.rc && dX0) wgrad(t3, dX0, w.d, bt.scores, K, M, w.d, K, INTEL_P_SCORE_W, INTEL_P_SCORE_B);
      if (!t3.rc) t3.ok(redq_flush_tag(c->rq, 1, t3.st));
      // the head's leaves (nothing in this backward reads them): table scatters and the weight gradients of head_bwd_b's links
      defer = nullptr;
      redq_set_tag(c->rq, 0);
      t3.ok((int)hipStreamWaitEvent(t3.st, c->ev_x[2], 0));
      if (!t3.rc) run_leaves(t3, lv_main, 0);
      if (!t3.rc && r.G(INTEL_P_CTX_EMB))
        t3.ok(launch_scatter_add_rows(y.dPREDIN, y.Pin, 0, D.d_c, bt.context_mh, B, r.G(INTEL_P_CTX_EMB), nullptr, 0, 0, t3.st));
      if (!t3.rc && r.G(INTEL_P_UID_EMB))
        t3.ok(launch_scatter_add_rows(y.dPREDIN, y.Pin, D.d_c, D.d_u, bt.u_id_c, B, r.G(INTEL_P_UID_EMB), nullptr, 0, 0, t3.st));
      if (leaves_behind_score) chain_leaves(t3);
      r.ok(t3.rc);
    }
    redq_set_tag(c->rq, 2);
    item_tower_bwd(s1, y.tmp[0].dXa);
    if (!s1.rc) s1.ok(redq_flush_tag(c->rq, 2, s1.st));
    redq_set_tag(c->rq, 0);
    if (!leaves_behind_score) chain_leaves(s1);
    r.ok(s1.rc);
    if (r.rc) return;
    redq_set_tag(c->rq, 3);
    float* dE0 = encoder_branch(s0, 0);
    if (!s0.rc) s0.ok(redq_flush_tag(c->rq, 3, s0.st));      // (the encoders' slabs are large whatever the batch: always branch-local)
    redq_set_tag(c->rq, 0);
    if (!s0.rc && dE0) intent_wgrad(s0, 0, dE0);
    Run e1 = r;
    e1.T = &y.tmp[1];
    e1.rc = 0;
    redq_set_tag(c->rq, 4);
    float* dE1 = encoder_branch(e1, 1);
    if (!e1.rc) e1.ok(redq_flush_tag(c->rq, 4, e1.st));
    redq_set_tag(c->rq, 0);
    if (!e1.rc && dE1) intent_wgrad(e1, 1, dE1);
    r.ok(e1.rc); r.ok(s0.rc);
    if (c->table_stream) {
      r.ok((int)hipEventRecord(c->ev_x[3], r.st));
      r.ok((int)hipStreamWaitEvent(c->table_stream, c->ev_x[3], 0));
      if (c->table_stream != c->side[1]) wait_side(r, 1, c->table_stream);
    }
    join_streams(r, 3);
    if (r.rc || !dE1 || !dE0) return;
    RUN(redq_flush(c->rq, r.st));
    return;
  }
  if (wide) {
    IntelCtx* c = r.ctx;
    r.T = &y.tmp[0];
    Run m = r;                                   // main stream, set 0 for the B-row temporaries of the chain
    Run s2 = branch(r, 2, 1), s1 = branch(r, 1, 0), s0 = branch(r, 0, 2);
    r.ok((int)hipEventRecord(c->ev_fork, r.st));
    r.ok((int)hipStreamWaitEvent(c->side[2], c->ev_fork, 0));
    // score tower (side 2): cross-attention backward, then its layers with set 3, then the pooling's weight gradients
    // (every branch reduces the slabs of ITS weights on its own stream when it is done -- tags 1..4 -- instead of leaving
    // 0.5 GB of reduction to the tail; the shared intent-embedding slot and the session head stay in the final flush)
    defer = hfused ? &lv_main : &lv_score;      // (hfused: dQV is produced later, by head_bwd_b on the main stream -- every leaf goes behind it)
    redq_set_tag(c->rq, 1);
    if (hfused) pool_bwd_fused(s2, 1, y.dXS);
    else xatt_bwd(s2, 1, y.dXS, y.tmp[1].dINT);
    defer = &lv_main;
    r.ok((int)hipEventRecord(c->ev_x[1], s2.st));
    {
      Run t3 = s2;
      t3.T = &y.tmp[3];
      TowerBufs& w = y.tw[1];
      float* dX0 = tower_bwd(t3, w, y.dXS, y.tmp[3].dXb, c->fused_tail[1]);
      if (!t3.rc && dX0) wgrad(t3, dX0, w.d, bt.scores, K, M, w.d, K, INTEL_P_SCORE_W, INTEL_P_SCORE_B);
      if (!t3.rc) run_leaves(t3, lv_score, 1);
      if (!t3.rc) t3.ok(redq_flush_tag(c->rq, 1, t3.st));
      r.ok(t3.rc);
    }
    redq_set_tag(c->rq, 2);
    r.ok(s2.rc);
    // item tower: cross-attention backward on the main stream, layers on side 1
    if (hfused) pool_bwd_fused(m, 0, y.tmp[0].dXa);
    else xatt_bwd(m, 0, y.tmp[0].dXa, y.tmp[0].dINT);
    r.ok(m.rc);
    if (r.rc) return;
    r.ok((int)hipEventRecord(c->ev_x[0], r.st));
    r.ok((int)hipStreamWaitEvent(c->side[1], c->ev_x[0], 0));
    item_tower_bwd(s1, y.tmp[0].dXa);
    if (!s1.rc) s1.ok(redq_flush_tag(c->rq, 2, s1.st));
    r.ok(s1.rc);
    redq_set_tag(c->rq, 0);
    // d(intent) chain (needs both cross-attention backwards)
    r.ok((int)hipStreamWaitEvent(r.st, c->ev_x[1], 0));
    if (hfused) {
      head_bwd_b(r, d_intents);      // dQV of both towers, d(intent) summed, softmax backward, d(pred_layer input)
    } else {
      RUN(launch_add2(y.dINTENT, y.tmp[0].dINT, (long long)B * I, y.dINTENT, r.st));
      RUN(launch_add2(y.dINTENT, y.tmp[1].dINT, (long long)B * I, y.dINTENT, r.st));
      if (d_intents) RUN(launch_add2(y.dINTENT, d_intents, (long long)B * I, y.dINTENT, r.st));
      RUN(launch_softmax_rows_bwd(y.INTENTS, y.dINTENT, B, I, y.dLOGITS, r.st));
      lin(r, y.dLOGITS, I, B, I, y.pPredT, y.Pin, y.dPREDIN, y.Pin, e0);
    }
    if (r.rc) return;
    // the two encoders need the chain's d(pred_layer input): session history on side 0 (set 2), item history on the main
    // stream (set 1).  The leaves of the session head go to side 2 behind the score tower (the shortest branch): off the
    // critical chain, and in the same host order as before (ahead of the encoders' shares of the shared intent-embedding slot).
    r.ok((int)hipEventRecord(c->ev_x[2], r.st));
    r.ok((int)hipStreamWaitEvent(c->side[0], c->ev_x[2], 0));
    r.ok((int)hipStreamWaitEvent(c->side[2], c->ev_x[2], 0));
    defer = nullptr;
    {
      Run lf = s2;
      lf.rc = 0;
      // the leaves' own slots (fusion weights, pooling projections, pred_layer) are reduced right here (tag 5); the
      // intent-embedding slot, shared with the encoders, stays untagged for the final flush
      run_leaves(lf, lv_main, 5);
      if (!hfused) wgrad(lf, y.dLOGITS, I, y.PREDIN, y.Pin, B, I, y.Pin, INTEL_P_PRED_W, INTEL_P_PRED_B);
      if (!lf.rc) lf.ok(redq_flush_tag(c->rq, 5, lf.st));
      redq_set_tag(c->rq, 0);
      if (!lf.rc && r.G(INTEL_P_CTX_EMB))
        lf.ok(launch_scatter_add_rows(y.dPREDIN, y.Pin, 0, D.d_c, bt.context_mh, B, r.G(INTEL_P_CTX_EMB), nullptr, 0, 0, lf.st));
      if (!lf.rc && r.G(INTEL_P_UID_EMB))
        lf.ok(launch_scatter_add_rows(y.dPREDIN, y.Pin, D.d_c, D.d_u, bt.u_id_c, B, r.G(INTEL_P_UID_EMB), nullptr, 0, 0, lf.st));
      r.ok(lf.rc);
    }
    if (r.rc) return;
    redq_set_tag(c->rq, 3);
    float* dE0 = encoder_branch(s0, 0);
    if (!s0.rc) s0.ok(redq_flush_tag(c->rq, 3, s0.st));
    redq_set_tag(c->rq, 0);
    if (!s0.rc && dE0) intent_wgrad(s0, 0, dE0);
    Run e1 = r;
    e1.T = &y.tmp[1];
    e1.rc = 0;
    redq_set_tag(c->rq, 4);
    float* dE1 = encoder_branch(e1, 1);
    if (!e1.rc) e1.ok(redq_flush_tag(c->rq, 4, e1.st));
    redq_set_tag(c->rq, 0);
    if (!e1.rc && dE1) intent_wgrad(e1, 1, dE1);
    r.ok(e1.rc); r.ok(s0.rc);
    if (c->table_stream) {
      // the item-id table gradient is complete once the item tower (side 1) and the item-history encoder (main stream, up to
      // here) are: the caller's optimizer sweep of the table may start without waiting for the other two branches
      r.ok((int)hipEventRecord(c->ev_x[3], r.st));
      r.ok((int)hipStreamWaitEvent(c->table_stream, c->ev_x[3], 0));
      if (c->table_stream != c->side[1]) wait_side(r, 1, c->table_stream);
    }
    join_streams(r, 3);
    if (r.rc || !dE1 || !dE0) return;
    RUN(redq_flush(c->rq, r.st));
    return;
  }
  constexpr bool enc0_early = true;      // the session-history encoder joins phase 1 on a third stream
  if (phase != 2) {
    // cross-attention backward of both towers first: d(intent) is then complete and the intent path can
    // start while the (heavy) tower layers are still running
    r.T = &y.tmp[0];
    fork_streams(r, 1);
    {
      Run b1 = branch(r, 0, 1);
      xatt_bwd(b1, 1, y.dXS, y.tmp[1].dINT);
      xatt_bwd(r, 0, y.tmp[0].dXa, y.tmp[0].dINT);
      r.ok(b1.rc);
    }
    join_streams(r, 1);
    if (r.rc) return;
    RUN(launch_add2(y.dINTENT, y.tmp[0].dINT, (long long)B * I, y.dINTENT, r.st));
    RUN(launch_add2(y.dINTENT, y.tmp[1].dINT, (long long)B * I, y.dINTENT, r.st));
    if (d_intents) RUN(launch_add2(y.dINTENT, d_intents, (long long)B * I, y.dINTENT, r.st));
    RUN(launch_softmax_rows_bwd(y.INTENTS, y.dINTENT, B, I, y.dLOGITS, r.st));
    wgrad(r, y.dLOGITS, I, y.PREDIN, y.Pin, B, I, y.Pin, INTEL_P_PRED_W, INTEL_P_PRED_B);
    lin(r, y.dLOGITS, I, B, I, y.pPredT, y.Pin, y.dPREDIN, y.Pin, e0);
    if (r.rc) return;
    if (r.G(INTEL_P_CTX_EMB))
      RUN(launch_scatter_add_rows(y.dPREDIN, y.Pin, 0, D.d_c, bt.context_mh, B, r.G(INTEL_P_CTX_EMB), nullptr, 0, 0, r.st));
    if (r.G(INTEL_P_UID_EMB))
      RUN(launch_scatter_add_rows(y.dPREDIN, y.Pin, D.d_c, D.d_u, bt.u_id_c, B, r.G(INTEL_P_UID_EMB), nullptr, 0, 0, r.st));
    // phase 1 branches: item tower layers (main, set 0) || item-history encoder (side 0, set 1): after the
    // join the item-id table gradient is complete
    // Both encoders' backward chains are long runs of small launches (B*H rows): the session-history encoder joins phase 1 on a
    // third stream (set 2) instead of trailing the score tower in phase 2, where it was the critical path.
    float *dE1 = nullptr, *dE0 = nullptr;
    fork_streams(r, enc0_early ? 2 : 1);
    {
      Run b0 = branch(r, -1, 0), b1 = branch(r, 0, 1), b2 = branch(r, 1, 2);
      item_tower_bwd(b0, y.tmp[0].dXa);
      dE1 = encoder_branch(b1, 1);
      if (enc0_early) dE0 = encoder_branch(b2, 0);
      r.ok(b0.rc); r.ok(b1.rc); r.ok(b2.rc);
    }
    join_streams(r, enc0_early ? 2 : 1);
    if (r.rc || !dE1 || (enc0_early && !dE0)) return;
    r.T = &y.tmp[0];
    intent_wgrad(r, 1, dE1);
    if (enc0_early) intent_wgrad(r, 0, dE0);
    if (r.rc) return;
    RUN(redq_flush(r.ctx->rq, r.st));
  }
  if (phase != 1) {
    // phase 2: score tower layers (main, set 0)
    float* dE0 = nullptr;
    if (!enc0_early) fork_streams(r, 1);
    {
      Run b0 = branch(r, -1, 0), b1 = branch(r, 0, 1);
      // the score tower's output gradient was parked in dXS by phase 1; tower_bwd ping-pongs dXS <-> dXb
      {
        TowerBufs& w = y.tw[1];
        float* dX0 = tower_bwd(b0, w, y.dXS, y.tmp[0].dXb, r.ctx->fused_tail[1]);
        if (!b0.rc && dX0) wgrad(b0, dX0, w.d, bt.scores, K, M, w.d, K, INTEL_P_SCORE_W, INTEL_P_SCORE_B);
      }
      if (!enc0_early) dE0 = encoder_branch(b1, 0);
      r.ok(b0.rc); r.ok(b1.rc);
    }
    if (!enc0_early) join_streams(r, 1);
    if (r.rc || (!enc0_early && !dE0)) return;
    r.T = &y.tmp[0];
    if (!enc0_early) intent_wgrad(r, 0, dE0);
    if (r.rc) return;
    RUN(redq_flush(r.ctx->rq, r.st));
  }
}

int check_desc(const IntelDesc& d) {
  INTEL_CHECK_ARG(d.model_num >= 1 && d.model_num <= 16, "model_num %d unsupported (1..16)", d.model_num);
  INTEL_CHECK_ARG(d.intent_num >= 1, "intent_num must be positive");
  INTEL_CHECK_ARG(d.layers >= 0 && d.layers <= MAX_TOWER_LAYERS, "num_layers %d unsupported (0..%d)", d.layers, MAX_TOWER_LAYERS);
  INTEL_CHECK_ARG(d.heads >= 1, "num_heads must be >= 1");
  const int d_i = d.d_id + d.d_im;
  INTEL_CHECK_ARG(d_i % 4 == 0 && d.d_s % 4 == 0 && d.d_id % 4 == 0 && d.d_im % 4 == 0, "embedding sizes must be multiples of 4");
  INTEL_CHECK_ARG(d.d_u % 4 == 0 && d.d_c % 4 == 0 && d.d_int % 4 == 0, "embedding sizes must be multiples of 4");
  INTEL_CHECK_ARG(d_i % d.heads == 0 && d.d_s % d.heads == 0, "tower widths must be divisible by num_heads");
  INTEL_CHECK_ARG(d_i % 16 == 0 && d.d_s % 16 == 0, "tower widths (i_emb+im_emb=%d, s_emb=%d) must be multiples of 16", d_i, d.d_s);
  INTEL_CHECK_ARG(d.encoder == INTEL_ENC_BERT4REC || d.encoder == INTEL_ENC_GRU4REC, "Invalid sequence encoder.");
  INTEL_CHECK_ARG(d.weight_norm >= 0 && d.weight_norm <= 2, "weight_norm %d unsupported (0 none, 1 softmax, 2 double softmax)", d.weight_norm);
  INTEL_CHECK_ARG(!d.pool_mean || !d.cross_attention, "pool_mean (aWELv_IntEL) uses the gate MLPs: cross_attention must be 0");
  INTEL_CHECK_ARG(d.dtype == INTEL_DTYPE_F32 || d.dtype == INTEL_DTYPE_BF16, "dtype %d unsupported", d.dtype);
  if (d.encoder == INTEL_ENC_BERT4REC) {
    INTEL_CHECK_ARG(d.enc_layers >= 1 && d.enc_layers <= INTEL_ENC_MAX_BLOCKS, "encoder blocks %d unsupported", d.enc_layers);
    INTEL_CHECK_ARG((d.d_c + d.d_int) % 16 == 0 && (d.d_id + d.d_int) % 16 == 0, "encoder widths must be multiples of 16");
    INTEL_CHECK_ARG((d.d_c + d.d_int) % d.enc_heads == 0 && (d.d_id + d.d_int) % d.enc_heads == 0, "encoder widths must be divisible by heads");
  }
  return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" IntelCtx* intel_create(const IntelDesc* desc) {
  if (!desc) {
    intel_set_error("intel_create: null descriptor");
    return nullptr;
  }
  if (check_desc(*desc) != 0) return nullptr;
  IntelCtx* c = new (std::nothrow) IntelCtx();
  if (!c) return nullptr;
  c->d = *desc;
  c->have_layout = false;
  c->fwd_done = false;
  c->streams = 0;
  c->drop_p = 0.f;
  c->drop_seed = 0;
  c->drop_ext = nullptr;
  c->fwd_dropout = false;
  c->iid_row_flags = nullptr;
  memset(&c->lazy, 0, sizeof(c->lazy));
  c->lazy_upto = 0;
  c->fused_tail[0] = c->fused_tail[1] = false;
  c->rq = redq_create();
  if (!c->rq) {
    delete c;
    return nullptr;
  }
  return c;
}

extern "C" void intel_destroy(IntelCtx* ctx) {
  if (!ctx) return;
  if (ctx->streams == 1 || ctx->streams == 2) {
    for (int i = 0; i < 3; ++i) {
      (void)hipStreamDestroy(ctx->side[i]);
      (void)hipEventDestroy(ctx->ev_join[i]);
    }
    (void)hipEventDestroy(ctx->ev_fork);
    for (int i = 0; i < 4; ++i) (void)hipEventDestroy(ctx->ev_x[i]);
  }
  if (ctx->ev_tab) (void)hipEventDestroy(ctx->ev_tab);
  redq_destroy(ctx->rq);
  delete ctx;
}

// on = 0: run every branch on the caller's stream (used while profiling single kernels); on = 1: default
extern "C" void* intel_side_stream(IntelCtx* ctx, int i) {
  if (!ctx || i < 0 || i > 2 || !ensure_streams(ctx)) return nullptr;
  return (void*)ctx->side[i];
}

extern "C" void intel_set_table_wait_event(IntelCtx* ctx, void* event) {
  if (ctx) ctx->table_wait_ev = (hipEvent_t)event;
}

extern "C" void intel_set_table_stream(IntelCtx* ctx, void* stream) {
  if (ctx) ctx->table_stream = (hipStream_t)stream;
}

extern "C" void intel_set_params_unchanged(IntelCtx* ctx, int on) {
  if (ctx) ctx->params_unchanged = on != 0;
}

extern "C" void intel_set_concurrency(IntelCtx* ctx, int on) {
  if (!ctx) return;
  if (!on) {
    if (ctx->streams == 1) ctx->streams = 2;          // keep the streams, just do not use them
    else if (ctx->streams == 0) ctx->streams = -1;
  } else {
    if (ctx->streams == 2) ctx->streams = 1;
    else if (ctx->streams == -1) ctx->streams = 0;
  }
}

extern "C" int intel_set_dropout(IntelCtx* ctx, float p, unsigned long long seed, const float* keep_flags) {
  INTEL_CHECK_ARG(ctx, "intel_set_dropout: null context");
  INTEL_CHECK_ARG(p >= 0.f && p < 1.f, "intel_set_dropout: p=%g outside [0, 1)", (double)p);
  ctx->drop_p = p;
  ctx->drop_seed = seed;
  ctx->drop_ext = keep_flags;
  return 0;
}

extern "C" int intel_set_iid_grad_row_flags(IntelCtx* ctx, unsigned char* row_flags) {
  INTEL_CHECK_ARG(ctx, "intel_set_iid_grad_row_flags: null context");
  ctx->iid_row_flags = row_flags;
  return 0;
}

extern "C" int intel_set_lazy_table(IntelCtx* ctx, const IntelLazyTable* t, int upto) {
  INTEL_CHECK_ARG(ctx, "intel_set_lazy_table: null context");
  if (t) {
    INTEL_CHECK_ARG(t->p && t->m && t->v && t->last && t->sched, "intel_set_lazy_table: null tensor");
    INTEL_CHECK_ARG(t->rows == ctx->d.item_num && t->d == ctx->d.d_id, "intel_set_lazy_table: table is [%lld, %d], iid_embeddings.weight is [%d, %d]",
                    t->rows, t->d, ctx->d.item_num, ctx->d.d_id);
    ctx->lazy = *t;
  } else {
    memset(&ctx->lazy, 0, sizeof(ctx->lazy));
  }
  ctx->lazy_upto = upto;
  return 0;
}

extern "C" size_t intel_workspace_bytes(const IntelCtx* ctx, int B, int L, int H, int Hi, int train) {
  if (!ctx || B <= 0 || L <= 0 || H <= 0 || Hi <= 0) return 0;
  Layout y;
  make_layout(ctx->d, B, L, H, Hi, nullptr, y, train && ctx->drop_p > 0.f);
  return y.total;
}

static int check_batch(const IntelCtx* ctx, const IntelBatch* b) {
  INTEL_CHECK_ARG(b && b->B > 0 && b->L > 0 && b->H > 0 && b->Hi > 0, "bad batch shape");
  INTEL_CHECK_ARG(b->i_id_s && b->i_class_c && b->scores && b->session_len && b->u_id_c && b->context_mh, "batch: null tensor");
  INTEL_CHECK_ARG(b->his_context_mh && b->his_intents && b->history_len && b->his_item_id && b->history_item_len, "batch: null history tensor");
  INTEL_CHECK_ARG(b->his_item_idx || b->his_item_int, "batch: need his_item_idx or his_item_int");
  INTEL_CHECK_ARG(b->H <= ctx->d.history_max + 1 && b->Hi <= ctx->d.history_max + 1 || ctx->d.encoder != INTEL_ENC_BERT4REC,
                  "history longer than history_max+1 position embeddings");
  return 0;
}

extern "C" int intel_forward(IntelCtx* ctx, const void* const* params, const IntelBatch* batch, void* workspace,
                             size_t workspace_bytes, const IntelOut* out, int train, void* stream) {
  INTEL_CHECK_ARG(ctx && params && out && workspace, "intel_forward: null argument");
  int rc = check_batch(ctx, batch);
  if (rc) return rc;
  INTEL_CHECK_ARG(out->weights && out->ens_score && out->intents, "intel_forward: null output");
  INTEL_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "workspace must be 256-byte aligned");
  const bool dropout = train && ctx->drop_p > 0.f;
  make_layout(ctx->d, batch->B, batch->L, batch->H, batch->Hi, static_cast<char*>(workspace), ctx->lay, dropout);
  ctx->fwd_dropout = dropout;
  {   // run the encoders on the valid history rows only when the caller supplied the row offsets (INTEL_PACK_HISTORY=0: never)
    static const int pack_on = [] { const char* e = getenv("INTEL_PACK_HISTORY"); return (e && e[0] == '0') ? 0 : 1; }();
    const IntelDesc& D = ctx->d;
    for (int e = 0; e < 2; ++e) {
      const int T = e == 0 ? batch->H : batch->Hi, dm = e == 0 ? D.d_c + D.d_int : D.d_id + D.d_int;
      const int* off = e == 0 ? batch->his_off : batch->hisitem_off;
      const int nrows = e == 0 ? batch->n_his_rows : batch->n_hisitem_rows;
      const bool pk = pack_on && off && nrows > 0 && nrows <= batch->B * T &&
                      ((D.encoder == INTEL_ENC_BERT4REC && D.enc_layers >= 1 && attn_packed_supported(T, dm / D.enc_heads)) ||
                       (D.encoder == INTEL_ENC_GRU4REC && gru_packed_supported(D.gru_hidden)));
      ctx->enc_packed[e] = pk;
      ctx->enc_rows[e] = pk ? nrows : batch->B * T;
      ctx->enc_fused[e] = pk && D.encoder == INTEL_ENC_BERT4REC && D.enc_layers >= 2 && enc_fused_supported(T, dm, D.enc_heads);
      ctx->enc32[e] = D.encoder == INTEL_ENC_BERT4REC && enc32_supported(T, dm, D.enc_heads, D.enc_layers, train) && enc32_batch_ok(batch->B, train);
    }
  }
  ctx->fused_tail[0] = tail_fusable(ctx, ctx->d, batch->L, ctx->lay.tw[0].d, train != 0);
  ctx->fused_tail[1] = tail_fusable(ctx, ctx->d, batch->L, ctx->lay.tw[1].d, train != 0);
  if (workspace_bytes < ctx->lay.total) {
    intel_set_error("intel_forward: workspace %zu < %zu bytes", workspace_bytes, ctx->lay.total);
    return INTEL_E_WORKSPACE;
  }
  ctx->have_layout = true;
  Run r{ctx, ctx->d, ctx->lay, params, nullptr, batch, (hipStream_t)stream, 0, &ctx->lay.tmp[0], train};
  gemm_set_planes(ctx->d.dtype == INTEL_DTYPE_BF16 ? 1 : 3);
  forward_impl(r, out);
  ctx->table_wait_ev = nullptr;      // (one shot: the waits are enqueued)
  gemm_set_planes(3);
  ctx->fwd_done = (r.rc == 0) && train;
  ctx->fB = batch->B; ctx->fL = batch->L; ctx->fH = batch->H; ctx->fHi = batch->Hi;
  ctx->f_ws = workspace;
  return r.rc;
}

static int backward_entry(IntelCtx* ctx, const void* const* params, const IntelBatch* batch, void* workspace,
                          size_t workspace_bytes, const float* d_weights, const float* d_ens_score,
                          const float* d_intents, void* const* grads, void* stream, int phase) {
  INTEL_CHECK_ARG(ctx && params && grads && workspace, "intel_backward: null argument");
  INTEL_CHECK_ARG(phase >= 0 && phase <= 2, "intel_backward: phase must be 0, 1 or 2");
  int rc = check_batch(ctx, batch);
  if (rc) return rc;
  if (!ctx->fwd_done || ctx->fB != batch->B || ctx->fL != batch->L || ctx->fH != batch->H || ctx->fHi != batch->Hi ||
      ctx->f_ws != workspace) {
    intel_set_error("intel_backward: no matching intel_forward(train=1) on this context/workspace");
    return INTEL_E_STATE;
  }
  if (workspace_bytes < ctx->lay.total) return INTEL_E_WORKSPACE;
  INTEL_CHECK_ARG(d_weights || d_ens_score || d_intents, "intel_backward: all output gradients are null");
  Run r{ctx, ctx->d, ctx->lay, params, grads, batch, (hipStream_t)stream, 0, &ctx->lay.tmp[0], 1};
  gemm_set_planes(ctx->d.dtype == INTEL_DTYPE_BF16 ? 1 : 3);
  wgrad_batch_reset();
  backward_impl(r, d_weights, d_ens_score, d_intents, phase);
  wgrad_batch_reset();
  gemm_set_planes(3);
  // The caller's table stream (intel_set_table_stream) is promised the finished item-id table gradient.  The four-branch schedule
  // hands it over as early as possible; every other way through a one-call backward (INTEL_STREAMS=0) does it
  // here, after everything: without this wait the caller's optimizer sweep raced the backward (found by the A/B switch tests)
  // Four-branch schedule too (round 6): the table stream is released early by the backward (item tower + item-history encoder done), but the optimizer's
  // sweep of the 1 M-row table is HBM-bound on the whole chip and starved the backward's LAST launch -- the batched slab reduction every dense Adam
  // group waits for (279 us under the sweep against ~25 alone: tools/rocprof_timeline.py) -- and with it the packing and the head of the next forward.
  // The sweep now starts behind that reduction: +0.5 ... +1.2 % sessions/s same-box at the headline.  INTEL_TABLE_AFTER_FLUSH=0: the early release.
  static const int table_after_flush = [] { const char* e = getenv("INTEL_TABLE_AFTER_FLUSH"); return e ? atoi(e) : 1; }();
  if (r.rc == 0 && phase == 0 && ctx->table_stream && (ctx->streams != 1 || table_after_flush)) {
    if (!ctx->ev_tab && hipEventCreateWithFlags(&ctx->ev_tab, hipEventDisableTiming) != hipSuccess) {
      intel_set_error("intel_backward: event creation failed");
      return INTEL_E_STATE;
    }
    hipError_t e = hipEventRecord(ctx->ev_tab, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->table_stream, ctx->ev_tab, 0);
    if (e != hipSuccess) {
      intel_set_error("intel_backward: table stream hand-over failed: %s", hipGetErrorString(e));
      return (int)e;
    }
  }
  return r.rc;
}

extern "C" int intel_backward(IntelCtx* ctx, const void* const* params, const IntelBatch* batch, void* workspace,
                              size_t workspace_bytes, const float* d_weights, const float* d_ens_score,
                              const float* d_intents, void* const* grads, void* stream) {
  return backward_entry(ctx, params, batch, workspace, workspace_bytes, d_weights, d_ens_score, d_intents, grads, stream, 0);
}

extern "C" int intel_backward_phase(IntelCtx* ctx, const void* const* params, const IntelBatch* batch, void* workspace,
                                    size_t workspace_bytes, const float* d_weights, const float* d_ens_score,
                                    const float* d_intents, void* const* grads, int phase, void* stream) {
  return backward_entry(ctx, params, batch, workspace, workspace_bytes, d_weights, d_ens_score, d_intents, grads, stream, phase);
}
