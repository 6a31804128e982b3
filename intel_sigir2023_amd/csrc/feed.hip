// Device-side batch assembly (include/intel_hip.h "input feed"): the reference builds every sample in Python
// (models/BaseModel.py:158-197 -> GeneralSeq.py:35-54 -> IntEL.py:220-239) and pads them in collate_batch
// (BaseModel.py:121-142) at ~1e3 sessions/s; here the corpus is a columnar store in HBM and one wave assembles
// one session: gather of the candidate list through the permutation, per-list min-max of each base score in
// fp64 (BaseModel.py:172-173, then the same fp64 -> fp32 rounding the model's .float() applies), ranking
// labels from the four counts, and the two history windows.  HBM-bound; a session moves a few KB.
#include "kernels.h"
#include "../../include/intel_hip.h"

#define FEED_MAX_L 1024      // candidates per list (LDS slots of the permutation)

namespace {

// 64-bit mix (splitmix64 finaliser): the random key of candidate j of session s under `seed`
__device__ __forceinline__ unsigned long long feed_key(unsigned long long seed, unsigned s, unsigned j) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)s * 4096ull + j + 1ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__device__ __forceinline__ double wave_min_d(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmin(v, __shfl_xor(v, m));
  return v;
}
__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m));
  return v;
}

__global__ __launch_bounds__(256) void feed_collate_kernel(IntelFeedStore S, const int* __restrict__ sess_idx, int shuffle,
                                                           const int* __restrict__ perm, unsigned long long seed, IntelFeedOut O) {
  __shared__ int src_of[4][FEED_MAX_L];       // slot i of the output list takes stored position src_of[i]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wave;
  const bool live = b < O.B;
  const int s = live ? sess_idx[b] : 0;
  const long long off = S.list_off[s];
  const int n = live ? (int)(S.list_off[s + 1] - off) : 0;
  const int L = O.L, K = S.n_scores, I = S.intent_num;
  int* so = src_of[wave];
  if (shuffle == 2) {
    for (int i = lane; i < n; i += 64) so[i] = perm[(size_t)b * L + i];
  } else if (shuffle == 1) {
    // rank of candidate j among the keys of its list = its output slot
    for (int j = lane; j < n; j += 64) {
      const unsigned long long kj = feed_key(seed, (unsigned)s, (unsigned)j);
      int r = 0;
      for (int t = 0; t < n; ++t) {
        const unsigned long long kt = feed_key(seed, (unsigned)s, (unsigned)t);
        r += (kt < kj || (kt == kj && t < j)) ? 1 : 0;
      }
      so[r] = j;
    }
  } else {
    for (int i = lane; i < n; i += 64) so[i] = i;
  }
  __syncthreads();
  if (!live) return;
  // ---- candidate list ----
  const int n3 = S.n_pay[s], n2 = n3 + S.n_fav[s], n1 = n2 + S.n_click[s], n0 = n1 + S.n_trueneg[s];
  for (int i = lane; i < L; i += 64) {
    int id = 0, cls = 0, lab = 0;
    if (i < n) {
      const int j = so[i];
      id = S.item_id[off + j];
      cls = S.item_class[off + j];
      lab = j < n3 ? 3 : (j < n2 ? 2 : (j < n1 ? 1 : (j < n0 ? 0 : -1)));
    }
    O.i_id_s[(size_t)b * L + i] = id;
    O.i_class_c[(size_t)b * L + i] = cls;
    O.ranking[(size_t)b * L + i] = lab;
  }
  for (int k = 0; k < K; ++k) {
    double mn = INFINITY, mx = -INFINITY;
    for (int j = lane; j < n; j += 64) {
      const double x = S.scores[(size_t)(off + j) * K + k];
      mn = fmin(mn, x);
      mx = fmax(mx, x);
    }
    mn = wave_min_d(mn);
    mx = wave_max_d(mx);
    const double den = (mx - mn) + 1e-6;
    for (int i = lane; i < L; i += 64) {
      float v = 0.f;
      if (i < n) v = (float)((S.scores[(size_t)(off + so[i]) * K + k] - mn) / den);
      O.scores[((size_t)b * L + i) * K + k] = v;
    }
  }
  const int uid = S.u_id[s];
  if (lane == 0) {
    O.session_len[b] = n;
    O.u_id_c[b] = uid;
    O.context_mh[b] = S.context_mh[s];
  }
  {
    const float* row = S.intent_rows + (size_t)S.intent_row[s] * I;
    for (int c = lane; c < I; c += 64) O.intents[(size_t)b * I + c] = row[c];
  }
  // ---- session history: the last max_his of the user's earlier sessions (GeneralSeq.py:38-51) ----
  {
    const int p = S.position[s];
    const int start = (S.max_his > 0 && p > S.max_his) ? p - S.max_his : 0;
    const int len = p - start;                     // 0 when the user has no history
    const long long base = S.uhis_off[uid] + start;
    const int H = O.H;
    for (int t = lane; t < H; t += 64) O.his_context_mh[(size_t)b * H + t] = t < len ? S.uhis_context_mh[base + t] : 0;
    for (int i = lane; i < H * I; i += 64) {
      const int t = i / I, c = i - t * I;
      O.his_intents[(size_t)b * H * I + i] = t < len ? S.intent_rows[(size_t)S.uhis_intent_row[base + t] * I + c] : 0.f;
    }
    if (lane == 0) O.history_len[b] = len > 0 ? len : 1;
  }
  // ---- positive-item history (IntEL.py:222-237) ----
  {
    const int p = S.item_position[s];
    const int start = (S.max_his > 0 && p > S.max_his) ? p - S.max_his : 0;
    const int len = p - start;
    const long long base = S.uitem_off[uid] + start;
    const int Hi = O.Hi;
    for (int t = lane; t < Hi; t += 64) {
      O.his_item_id[(size_t)b * Hi + t] = t < len ? S.uitem_id[base + t] : 0;
      O.his_item_idx[(size_t)b * Hi + t] = t < len ? S.uitem_intent_idx[base + t] : -1;
    }
    if (lane == 0) O.history_item_len[b] = len > 0 ? len : 1;
  }
}

}  // namespace

extern "C" void intel_feed_abi_sizes(int* out2) {
  out2[0] = (int)sizeof(IntelFeedStore);
  out2[1] = (int)sizeof(IntelFeedOut);
}

extern "C" int intel_feed_collate(const IntelFeedStore* store, const int* sess_idx, int shuffle, const int* perm,
                                  unsigned long long seed, const IntelFeedOut* out, void* stream) {
  INTEL_CHECK_ARG(store && sess_idx && out, "intel_feed_collate: null argument");
  INTEL_CHECK_ARG(out->B > 0 && out->L > 0 && out->H > 0 && out->Hi > 0, "intel_feed_collate: bad output shape");
  INTEL_CHECK_ARG(out->L <= FEED_MAX_L, "intel_feed_collate: list length %d > %d", out->L, FEED_MAX_L);
  INTEL_CHECK_ARG(shuffle >= 0 && shuffle <= 2 && (shuffle != 2 || perm), "intel_feed_collate: shuffle=%d needs perm", shuffle);
  INTEL_CHECK_ARG(out->i_id_s && out->i_class_c && out->scores && out->ranking && out->session_len && out->u_id_c && out->context_mh &&
                      out->intents && out->his_context_mh && out->his_intents && out->history_len && out->his_item_id &&
                      out->his_item_idx && out->history_item_len,
                  "intel_feed_collate: null output");
  const double bytes = (double)out->B * (16.0 * out->L + 12.0 * out->L * store->n_scores + 4.0 * store->intent_num * (2 + 2.0 * out->H) + 16.0 * out->Hi + 8.0 * out->H);
  LAUNCH_W(0.0, bytes, feed_collate_kernel, dim3(cdiv(out->B, 4)), dim3(256), 0, (hipStream_t)stream, *store, sess_idx, shuffle, perm, seed, *out);
  INTEL_CHECK_LAUNCH();
  return 0;
}
