// Op table of the B-row chain kernel (chain.hip): one launch = a few dependency levels of small products over [16 sessions, width]
// tiles in LDS.  Offsets and pitches are in floats relative to the kernel's dynamic LDS; a tile of width w is laid out by ChainPlan.
#pragma once
#include "common.h"

enum { CH_LOAD = 0, CH_LIN, CH_MASKCOPY, CH_SOFTMAX, CH_SOFTMAX_BWD, CH_ENS_FWD, CH_ENS_BWD, CH_WGRAD };
enum { CH_RELU = 1, CH_ACCUM = 2, CH_GATHER = 4, CH_BF16 = 8 };      // CH_BF16 (LIN, WGRAD; bf16 mode): both operands rounded to bf16 before the product

struct ChainOp {
  int kind, level, flags;
  int in_off, in_ld;          // source tile (LIN: its first k column; must be 16-byte aligned, KG * 16 readable columns, zero padded)
  int in2_off, in3_off;       // SOFTMAX_BWD: further summands of dy (pitch in_ld), -1 = none
  int out_off, out_ld;        // destination tile (ENS_BWD: the dwv tile)
  int aux_off, aux_ld;        // MASKCOPY: the mask tile; SOFTMAX_BWD: y; ENS_FWD: wpad; ENS_BWD: the dwpad tile
  int KG, NT;                 // LIN: 16-wide k groups / 16-wide column tiles of the packed weight
  int N, NP;                  // real columns / columns written (zero padded)
  const float* P;             // LIN: packed weight (launch_pack_b);  LOAD: global source
  const float* bias;          // LIN: [N] or null
  float* gout;                // optional global copy: gout[b * gld + gcol + c]
  int gld, gcol;
  float* gout2;               // SOFTMAX: second global copy, dense [B, N]
  const int* idx;             // LOAD + CH_GATHER: source row of session b
  int src_ld, src_col;        // LOAD: pitch / first column of the global source
  const float* gadd;          // SOFTMAX_BWD: optional global summand of dy, dense [B, N]
  // WGRAD: this workgroup's share of dW[n][k] = sum over its 16 sessions of dY[b][n] X[b][k] (dY = `in` tile, N real columns, NT column
  // tiles; X = `aux` tile, K real columns, KG column tiles) -> gout[blockIdx * gstride + n * gld + gcol + k] (CH_ACCUM: added to what an
  // earlier level of this launch wrote there); gout2 (optional): column sums of dY (the bias gradient) -> gout2[blockIdx * gstride + n]
  int K, gstride;
};

struct ChainEns {
  const float* scores;        // [B, L, K]
  const int* slen;            // [B]
  int L, K;
  float *weights, *ens;       // ENS_FWD outputs [B, L, K], [B, L]
  const float *d_weights, *d_ens;   // ENS_BWD inputs (either may be null)
  float *dwv, *dwpad;         // ENS_BWD global copies [B, K]
};

#define CHAIN_MAX_OPS 28
struct ChainArgs {
  int B, nops, nlevels;
  ChainEns ens;
  ChainOp ops[CHAIN_MAX_OPS];
};

int launch_chain(const ChainArgs& a, size_t lds_floats, hipStream_t st);

// host-side builder: tiles are carved from LDS in order; every op is appended with its level
struct ChainTile { int off, ld, width; };
struct ChainPlan {
  ChainArgs a;
  int lds = 0;
  bool ok = true;
  bool bf16 = false;      // --dtype bf16: links whose FORWARD product the mode runs on the bf16 pipe round their operands (forward, data gradient, weight gradient)
  // the mode's rule for a linear of K input features and N outputs (oracle.forward_bf16: _on_bf16_pipe; csrc/gemm.hip: launch_gemm_rows)
  int bfl(int K, int N) const { return (bf16 && (K == 64 || K == 128) && N % 4 == 0) ? CH_BF16 : 0; }
  ChainPlan() { a.B = 0; a.nops = 0; a.nlevels = 0; a.ens = ChainEns{}; }
  ChainTile tile(int width) {
    const int w = rup(width, 16);
    ChainTile t{lds, w + 4, w};
    lds += 16 * (w + 4);
    return t;
  }
  ChainOp& add(int kind, int level) {
    static ChainOp dummy;
    if (a.nops >= CHAIN_MAX_OPS) { ok = false; return dummy; }
    ChainOp& o = a.ops[a.nops++];
    o = ChainOp{};
    o.kind = kind; o.level = level;
    o.in2_off = o.in3_off = -1;
    if (level + 1 > a.nlevels) a.nlevels = level + 1;
    return o;
  }
  // tile[:, col0 : col0 + n] = (relu) src[row(b), src_col : src_col + n], zero padded up to `pad_to` columns of the tile (0 = n)
  void load(int level, const float* src, int src_ld, int src_col, int n, const ChainTile& t, int col0, int pad_to = 0, const int* idx = nullptr,
            bool relu = false, float* gout = nullptr, int gld = 0, int gcol = 0) {
    ChainOp& o = add(CH_LOAD, level);
    o.P = src; o.src_ld = src_ld; o.src_col = src_col; o.N = n; o.NP = pad_to > n ? pad_to : n;
    o.out_off = t.off + col0; o.out_ld = t.ld;
    o.idx = idx; o.flags = (idx ? CH_GATHER : 0) | (relu ? CH_RELU : 0);
    o.gout = gout; o.gld = gld; o.gcol = gcol;
  }
  // out[:, ocol : ocol + N] (+)= act(in[:, icol : icol + K] W + bias);  Pk = launch_pack_b image with k extent K, n extent N
  void lin(int level, const ChainTile& in, int icol, int K, const float* Pk, int N, const float* bias, const ChainTile& out, int ocol, int flags = 0,
           float* gout = nullptr, int gld = 0, int gcol = 0) {
    ChainOp& o = add(CH_LIN, level);
    o.in_off = in.off + icol; o.in_ld = in.ld; o.KG = rup(K, 16) / 16; o.NT = rup(N, 16) / 16; o.N = N; o.NP = o.NT * 16;
    o.P = Pk; o.bias = bias; o.out_off = out.off + ocol; o.out_ld = out.ld; o.flags = flags;
    o.gout = gout; o.gld = gld; o.gcol = gcol;
    if (icol + o.KG * 16 > in.width || ocol + o.NP > out.width || (icol & 3) || (ocol & 3)) ok = false;
  }
  // per-workgroup partial of dW [N, K] (+ db [N]) from the tiles dY[:, ycol : ycol + N] and X[:, xcol : xcol + K]; slab / dbslab point at
  // workgroup 0's partial, `stride` floats apart per workgroup; the matrix has row pitch ld and starts at column col0 of the slab rows
  void wgrad(int level, const ChainTile& dy, int ycol, int N, const ChainTile& x, int xcol, int K, float* slab, int ld, int col0, int stride,
             float* dbslab = nullptr, bool accumulate = false, int bf = 0) {
    ChainOp& o = add(CH_WGRAD, level);
    o.in_off = dy.off + ycol; o.in_ld = dy.ld; o.N = N; o.NT = rup(N, 16) / 16;
    o.aux_off = x.off + xcol; o.aux_ld = x.ld; o.K = K; o.KG = rup(K, 16) / 16;
    o.gout = slab; o.gld = ld; o.gcol = col0; o.gstride = stride; o.gout2 = dbslab; o.flags = (accumulate ? CH_ACCUM : 0) | bf;
    if (ycol + o.NT * 16 > dy.width || xcol + o.KG * 16 > x.width) ok = false;
  }
};
