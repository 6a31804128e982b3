// Loss kernels: BPR (loss/BPRloss.py), Plackett-Luce listwise (loss/Listloss.py) and the intent
// CE/KL loss (loss/BaseIntloss.py), forward and hand-derived backward in one launch per loss.
//
// One workgroup per session -- a single wave for lists of up to 64 candidates, four waves for longer ones; thread i owns
// candidate i (rows i, i+NT, ...), the per-list vectors live in LDS, per-list reductions are wave shuffles (+ one LDS hop).
// The [B,L,L] and [B,L,L,K] float64 intermediates of the reference are never materialised.
// The float64 "diversity" terms are evaluated in double like the reference (BPRloss.py:14-18,
// Listloss.py:18-23); everything else is fp32 in the same operation order.
#include "kernels.h"
#include "session.h"

#define LOSS_KMAX 16

__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + expf(-z)); }

// sum over the NT threads of the workgroup (NT = 256: four waves through LDS; NT = 64: one wave, shuffles only)
template <typename T, int NT = 256>
__device__ __forceinline__ T block_sum(T v, T* red /* [4] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (NT == 64) {
    __syncthreads();        // callers rely on the barrier (LDS vectors written before, read after)
    return v;
  }
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

struct LossArgs {
  int B, L, K;
  const float* ens; const int* ranking; const int* slen;
  const float* noise;                       // BPR only; NULL -> counter-based generator keyed by (seed, b, i, j)
  unsigned long long seed;
  unsigned long long session0;              // global index of this launch's first session (data-parallel shards)
  const double* sc64; const float* sc32;    // base scores (either)
  const float* weights;
  int cal_div; double alpha; float grad_scale;
  float* lossb; double* divb;               // per-session partial results [B]
  int* select;                              // BPR: [B,L]
  float* d_ens; float* d_weights;
};

__device__ __forceinline__ double score_at(const LossArgs& a, size_t idx) {
  return a.sc64 ? a.sc64[idx] : (double)a.sc32[idx];
}

// ------------------------------------------------------------------------------------------
// BPR
// ------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(NT) void bpr_loss_kernel(LossArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int L = a.L, K = a.K, b = blockIdx.x, tid = threadIdx.x;
  float* s_s = reinterpret_cast<float*>(smem_raw);       // [L] ens
  int* s_r = reinterpret_cast<int*>(s_s + L);            // [L] clamped rank
  int* s_sel = s_r + L;                                  // [L]
  float* s_g = reinterpret_cast<float*>(s_sel + L);      // [L] dL/dz_i
  __shared__ float redf[4];
  __shared__ double redd[4];
  __shared__ int redi[4];
  const int len = min(a.slen[b], L);
  int npos_l = 0;
  for (int i = tid; i < L; i += NT) {
    s_s[i] = a.ens[(size_t)b * L + i];
    const int r = max(a.ranking[(size_t)b * L + i], 0);
    s_r[i] = r;
    npos_l += r > 0;
  }
  {
    int v = wave_sum_i(npos_l);
    __syncthreads();
    if ((tid & 63) == 0) redi[tid >> 6] = v;
    __syncthreads();
  }
  const int npos = NT == 64 ? redi[0] : redi[0] + redi[1] + redi[2] + redi[3];
  const float inv_npos = 1.f / (float)npos;     // npos == 0 -> inf/NaN like the reference (0/0)
  const float c = a.grad_scale / ((float)a.B * (float)npos);
  float loss_l = 0.f;
  double div_l = 0.0;
  for (int i = tid; i < L; i += NT) {
    const int ri = s_r[i];
    const bool vi = i < len;
    // closest lower tier among valid j: smallest positive D = ri - rj
    int minD = 0x7fffffff;
    if (vi) {
      for (int j = 0; j < len; ++j) {
        const int D = ri - s_r[j];
        if (D > 0 && D < minD) minD = D;
      }
    }
    const float* nrow = a.noise ? a.noise + ((size_t)b * L + i) * L : nullptr;
    const unsigned long long ctr0 = ((a.session0 + (unsigned long long)b) * L + i) * L;
    float best = -1.f;
    int sel = 0;
    for (int j = 0; j < L; ++j) {
      const bool cand = vi && j < len && (ri - s_r[j]) == minD;
      float u;
      if (nrow) {
        u = nrow[j];
      } else {
        // 24 uniform bits from a 32-bit hash of the 64-bit element counter and the seed (two multiplies to fold the halves, then
        // the two-round "lowbias32" finaliser).  32-bit on purpose: integer multiplies are quarter rate and the splitmix64 this
        // replaces cost ~200 cycles per candidate -- 19 of the kernel's 34 us on the step's critical path
        const unsigned long long ctr = ctr0 + (unsigned long long)j + 1ull;
        unsigned h = ((unsigned)ctr * 0x9E3779B1u) ^ (((unsigned)(ctr >> 32) + (unsigned)(a.seed >> 32)) * 0x85EBCA77u) ^ (unsigned)a.seed;
        h ^= h >> 16; h *= 0x7FEB352Du;
        h ^= h >> 15; h *= 0x846CA68Bu;
        h ^= h >> 16;
        u = (float)(h >> 8) * (1.0f / 16777216.0f);
      }
      const float v = (cand ? 1.f : 0.f) + u / 10.f;            // possible_mask + rand/10 (BPRloss.py:26-28)
      if (v > best) { best = v; sel = j; }
    }
    s_sel[i] = sel;
    a.select[(size_t)b * L + i] = sel;
    const float z = s_s[i] - s_s[sel];
    const float sg = sigmoidf_(z);
    const bool pos = ri > 0;
    float g = 0.f;
    if (pos) {
      loss_l += -logf(sg);
      g = c * (sg - 1.f);
    }
    if (a.cal_div && pos) {
      const float sp = sg * (1.f - sg);                 // sigma'(z), fp32 like the reference
      const double zd = (double)z, spd = (double)sp;
      const double spp = spd * (1.0 - 2.0 * (double)sg);  // sigma''(z)
      double acc = 0.0, gz = 0.0;
      for (int k = 0; k < K; ++k) {
        const double dlt = score_at(a, ((size_t)b * L + i) * K + k) - score_at(a, ((size_t)b * L + sel) * K + k) - zd;
        const double w = (double)a.weights[((size_t)b * L + i) * K + k];
        acc += spd * dlt * dlt * w;
        gz += w * (spp * dlt * dlt - 2.0 * spd * dlt);
        if (a.d_weights) a.d_weights[((size_t)b * L + i) * K + k] = (float)(-a.alpha * (double)c * spd * dlt * dlt);
      }
      div_l += acc;
      g += (float)(-a.alpha * (double)c * gz);
    } else if (a.d_weights) {
      for (int k = 0; k < K; ++k) a.d_weights[((size_t)b * L + i) * K + k] = 0.f;
    }
    s_g[i] = g;
  }
  const float loss_b = block_sum<float, NT>(loss_l, redf) * inv_npos;
  double div_b = 0.0;
  if (a.cal_div) div_b = block_sum<double, NT>(div_l, redd) / (double)npos;
  if (tid == 0) {
    a.lossb[b] = loss_b;
    a.divb[b] = div_b;
  }
  __syncthreads();
  if (a.d_ens) {
    for (int j = tid; j < L; j += NT) {
      float acc = s_g[j];
      for (int i = 0; i < L; ++i)
        if (s_sel[i] == j) acc -= s_g[i];
      a.d_ens[(size_t)b * L + j] = acc;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Plackett-Luce listwise
// ------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(NT) void list_loss_kernel(LossArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int L = a.L, K = a.K, b = blockIdx.x, tid = threadIdx.x;
  double* s_sc = reinterpret_cast<double*>(smem_raw);            // [L][K] base scores (diversity only)
  double* s_U = s_sc + (a.cal_div ? (size_t)L * K : 0);          // [L][K]
  double* s_Aw = s_U + (a.cal_div ? (size_t)L * K : 0);          // [L]
  float* s_s = reinterpret_cast<float*>(s_Aw + (a.cal_div ? L : 0));   // [L]
  float* s_E = s_s + L;                                          // [L]
  int* s_r = reinterpret_cast<int*>(s_E + L);                    // [L]
  __shared__ float redf[4];
  __shared__ double redd[4];
  __shared__ int redi[4];
  const int len = min(a.slen[b], L);
  int npos_l = 0;
  for (int i = tid; i < L; i += NT) {
    s_s[i] = a.ens[(size_t)b * L + i];
    const int r = max(a.ranking[(size_t)b * L + i], 0);
    s_r[i] = r;
    npos_l += r > 0;
  }
  if (a.cal_div)
    for (int i = tid; i < L * K; i += NT) s_sc[i] = score_at(a, (size_t)b * L * K + i);
  {
    int v = wave_sum_i(npos_l);
    __syncthreads();
    if ((tid & 63) == 0) redi[tid >> 6] = v;
    __syncthreads();
  }
  const int npos = NT == 64 ? redi[0] : redi[0] + redi[1] + redi[2] + redi[3];
  const float c = a.grad_scale / ((float)a.B * (float)npos);
  float loss_l = 0.f;
  double div_l = 0.0;
  for (int i = tid; i < L; i += NT) {
    const int ri = s_r[i];
    const bool vi = i < len, pos = ri > 0;
    const float si = s_s[i];
    float E = 0.f;
    double U[LOSS_KMAX];
#pragma unroll
    for (int k = 0; k < LOSS_KMAX; ++k) U[k] = 0.0;
    if (vi) {
      for (int j = 0; j < len; ++j) {
        if (ri > s_r[j]) {
          const float z = si - s_s[j];
          const float e = expf(-z);
          E += e;
          if (a.cal_div) {
#pragma unroll
            for (int k = 0; k < LOSS_KMAX; ++k)
              if (k < K) U[k] += (double)e * ((s_sc[i * K + k] - s_sc[j * K + k]) - (double)z);
          }
        }
      }
    }
    s_E[i] = E;
    if (pos) loss_l += logf(fmaxf(E + 1.f, 1.f));
    if (a.cal_div) {
      double Aw = 0.0;
      const float bo = 2.f * (1.f + E) * (1.f + E);
#pragma unroll
      for (int k = 0; k < LOSS_KMAX; ++k)
        if (k < K) {
          s_U[i * K + k] = U[k];
          Aw += (double)a.weights[((size_t)b * L + i) * K + k] * U[k] * U[k];
          if (a.d_weights)
            a.d_weights[((size_t)b * L + i) * K + k] = pos ? (float)(-a.alpha * (double)c * U[k] * U[k] / (double)bo) : 0.f;
        }
      s_Aw[i] = Aw;
      if (pos) div_l += Aw / (double)bo;
    } else if (a.d_weights) {
      for (int k = 0; k < K; ++k) a.d_weights[((size_t)b * L + i) * K + k] = 0.f;
    }
  }
  const float loss_b = block_sum<float, NT>(loss_l, redf) / (float)npos;
  double div_b = 0.0;
  if (a.cal_div) div_b = block_sum<double, NT>(div_l, redd) / (double)npos;
  if (tid == 0) {
    a.lossb[b] = loss_b;
    a.divb[b] = div_b;
  }
  __syncthreads();
  if (!a.d_ens) return;
  // G(i,j) = dLoss/dz_ij for pairs with M_ij (i positive); d_ens[t] = sum_j G(t,j) - sum_i G(i,t)
  for (int t = tid; t < L; t += NT) {
    float acc = 0.f;
    if (t < len) {
      const int rt = s_r[t];
      const float st = s_s[t];
      for (int o = 0; o < len; ++o) {
        const int ro = s_r[o];
        if (rt > ro && rt > 0) {          // row t, column o
          const float z = st - s_s[o];
          const float e = expf(-z);
          const float E = s_E[t];
          float G = -c * e / (E + 1.f);
          if (a.cal_div) {
            const double bo = (double)(2.f * (1.f + E) * (1.f + E));
            double t1 = 0.0;
            for (int k = 0; k < K; ++k)
              t1 += (double)a.weights[((size_t)b * L + t) * K + k] * s_U[t * K + k] * ((s_sc[t * K + k] - s_sc[o * K + k]) - (double)z + 1.0);
            const double dT = -(double)e * (2.0 * t1 / bo - 4.0 * s_Aw[t] * (double)(1.f + E) / (bo * bo));
            G += (float)(-a.alpha * (double)c * dT);
          }
          acc += G;
        }
        if (ro > rt && ro > 0) {          // row o, column t
          const float z = s_s[o] - st;
          const float e = expf(-z);
          const float E = s_E[o];
          float G = -c * e / (E + 1.f);
          if (a.cal_div) {
            const double bo = (double)(2.f * (1.f + E) * (1.f + E));
            double t1 = 0.0;
            for (int k = 0; k < K; ++k)
              t1 += (double)a.weights[((size_t)b * L + o) * K + k] * s_U[o * K + k] * ((s_sc[o * K + k] - s_sc[t * K + k]) - (double)z + 1.0);
            const double dT = -(double)e * (2.0 * t1 / bo - 4.0 * s_Aw[o] * (double)(1.f + E) / (bo * bo));
            G += (float)(-a.alpha * (double)c * dT);
          }
          acc -= G;
        }
      }
    }
    a.d_ens[(size_t)b * L + t] = acc;
  }
}

// ------------------------------------------------------------------------------------------
// MSE (loss/MSEloss.py:12-30): one wave per session.  lossb = mean over the valid slots of (ens - max(label,0))^2;
// divb = mean over the valid slots of sum_k w_k (s_k - ens)^2 in float64 (the base scores are float64 there).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mse_loss_kernel(LossArgs a) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const int L = a.L, K = a.K;
  const int n = min(a.slen[b], L);
  const float c = a.grad_scale / ((float)a.B * (float)n);
  float acc = 0.f;
  double dacc = 0.0;
  for (int i = lane; i < L; i += 64) {
    const size_t o = (size_t)b * L + i;
    float g = 0.f;
    if (i < n) {
      const float e = a.ens[o];
      const float df = e - (float)max(a.ranking[o], 0);
      acc += df * df;
      g = 2.f * c * df;
      if (a.cal_div) {
        double gz = 0.0;
        for (int k = 0; k < K; ++k) {
          const double dl = score_at(a, o * K + k) - (double)e;
          const double w = (double)a.weights[o * K + k];
          dacc += w * dl * dl;
          gz += w * dl;
          if (a.d_weights) a.d_weights[o * K + k] = (float)(-a.alpha * (double)c * dl * dl);
        }
        g += (float)(a.alpha * (double)c * 2.0 * gz);          // d/de of -alpha * w (s - e)^2 = +2 alpha w (s - e)
      } else if (a.d_weights) {
        for (int k = 0; k < K; ++k) a.d_weights[o * K + k] = 0.f;
      }
    } else if (a.d_weights) {
      for (int k = 0; k < K; ++k) a.d_weights[o * K + k] = 0.f;
    }
    if (a.d_ens) a.d_ens[o] = g;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    acc += __shfl_xor(acc, m);
    dacc += __shfl_xor(dacc, m);
  }
  if (lane == 0) {
    a.lossb[b] = acc / (float)n;
    a.divb[b] = dacc / (double)n;
  }
}

// loss = mean_b lossb;  total = float(double(loss) + (-mean_b divb) * alpha)  (in-place += keeps fp32)
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ lossb, const double* __restrict__ divb, int B,
                                                            int cal_div, double alpha, float* __restrict__ loss) {
  __shared__ float redf[4];
  __shared__ double redd[4];
  float s = 0.f;
  double dsum = 0.0;
  for (int i = threadIdx.x; i < B; i += 256) {
    s += lossb[i];
    dsum += divb[i];
  }
  const float ls = block_sum<float>(s, redf) / (float)B;
  const double dv = block_sum<double>(dsum, redd) / (double)B;
  if (threadIdx.x == 0) loss[0] = cal_div ? (float)((double)ls + (-dv) * alpha) : ls;
}

size_t loss_ws_bytes(int B) { return rup_sz((size_t)B * sizeof(float), 16) + rup_sz((size_t)B * sizeof(double), 16) + 64; }

static int run_pair_loss(bool bpr, LossArgs& a, float* loss, void* ws, size_t ws_bytes, hipStream_t st) {
  INTEL_CHECK_ARG(a.K <= LOSS_KMAX, "loss: model_num %d > %d unsupported", a.K, LOSS_KMAX);
  INTEL_CHECK_ARG(ws_bytes >= loss_ws_bytes(a.B), "loss: workspace too small");
  INTEL_CHECK_ARG(!a.cal_div || a.weights, "loss: diversity needs the fusion weights");
  INTEL_CHECK_ARG(!a.cal_div || a.sc64 || a.sc32, "loss: diversity needs the base scores");
  a.divb = reinterpret_cast<double*>(ws);
  a.lossb = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + rup_sz((size_t)a.B * sizeof(double), 16));
  size_t smem;
  if (bpr) {
    smem = (size_t)a.L * 4 * sizeof(float);
    // lists of up to 64 candidates (Tmall shape): ONE wave per session -- lane = candidate, every per-list reduction is a
    // wave reduction -- instead of a 256-thread workgroup with 50 busy lanes; longer lists keep four waves
    if (a.L <= 64) {
      LAUNCH(bpr_loss_kernel<64>, dim3(a.B), dim3(64), smem, st, a);
    } else {
      allow_lds(bpr_loss_kernel<256>, smem);
      LAUNCH(bpr_loss_kernel<256>, dim3(a.B), dim3(256), smem, st, a);
    }
  } else {
    smem = (size_t)a.L * 3 * sizeof(float) + (a.cal_div ? ((size_t)2 * a.L * a.K + a.L) * sizeof(double) : 0);
    INTEL_CHECK_ARG(smem <= 150 * 1024, "list loss: L=%d K=%d does not fit LDS", a.L, a.K);
    if (a.L <= 64) {
      allow_lds(list_loss_kernel<64>, smem);
      LAUNCH(list_loss_kernel<64>, dim3(a.B), dim3(64), smem, st, a);
    } else {
      allow_lds(list_loss_kernel<256>, smem);
      LAUNCH(list_loss_kernel<256>, dim3(a.B), dim3(256), smem, st, a);
    }
  }
  INTEL_CHECK_LAUNCH();
  LAUNCH(loss_finalize_kernel, dim3(1), dim3(256), 0, st, a.lossb, a.divb, a.B, a.cal_div, a.alpha, loss);
  INTEL_CHECK_LAUNCH();
  return 0;
}

int launch_bpr_loss(int B, int L, int K, const float* ens, const int* ranking, const int* slen, const float* noise,
                    const double* sc64, const float* sc32, const float* weights, int cal_div, double alpha,
                    float grad_scale, float* loss, int* select, float* d_ens, float* d_weights, void* ws, size_t ws_bytes,
                    hipStream_t st, unsigned long long seed, int use_seed, unsigned long long session0) {
  LossArgs a;
  a.session0 = session0;
  a.B = B; a.L = L; a.K = K; a.ens = ens; a.ranking = ranking; a.slen = slen; a.noise = noise; a.sc64 = sc64; a.sc32 = sc32;
  a.weights = weights; a.cal_div = cal_div; a.alpha = alpha; a.grad_scale = grad_scale; a.select = select; a.d_ens = d_ens;
  a.d_weights = d_weights; a.lossb = nullptr; a.divb = nullptr;
  a.seed = seed;
  INTEL_CHECK_ARG((noise || use_seed) && select, "bpr loss: a noise tensor (or a seed) and the select buffer are required");
  return run_pair_loss(true, a, loss, ws, ws_bytes, st);
}
int launch_mse_loss(int B, int L, int K, const float* ens, const int* ranking, const int* slen, const double* sc64,
                    const float* sc32, const float* weights, int cal_div, double alpha, float grad_scale, float* loss,
                    float* d_ens, float* d_weights, void* ws, size_t ws_bytes, hipStream_t st) {
  LossArgs a;
  a.B = B; a.L = L; a.K = K; a.ens = ens; a.ranking = ranking; a.slen = slen; a.noise = nullptr; a.seed = 0; a.session0 = 0; a.sc64 = sc64; a.sc32 = sc32;
  a.weights = weights; a.cal_div = cal_div; a.alpha = alpha; a.grad_scale = grad_scale; a.select = nullptr; a.d_ens = d_ens;
  a.d_weights = d_weights;
  INTEL_CHECK_ARG(ws_bytes >= loss_ws_bytes(B), "loss: workspace too small");
  INTEL_CHECK_ARG(!cal_div || (weights && (sc64 || sc32)), "mse loss: diversity needs the fusion weights and the base scores");
  a.divb = reinterpret_cast<double*>(ws);
  a.lossb = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + rup_sz((size_t)B * sizeof(double), 16));
  LAUNCH(mse_loss_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, a);
  INTEL_CHECK_LAUNCH();
  LAUNCH(loss_finalize_kernel, dim3(1), dim3(256), 0, st, a.lossb, a.divb, B, cal_div, alpha, loss);
  INTEL_CHECK_LAUNCH();
  return 0;
}
int launch_list_loss(int B, int L, int K, const float* ens, const int* ranking, const int* slen, const double* sc64,
                     const float* sc32, const float* weights, int cal_div, double alpha, float grad_scale, float* loss,
                     float* d_ens, float* d_weights, void* ws, size_t ws_bytes, hipStream_t st) {
  LossArgs a;
  a.B = B; a.L = L; a.K = K; a.ens = ens; a.ranking = ranking; a.slen = slen; a.noise = nullptr; a.seed = 0; a.session0 = 0; a.sc64 = sc64; a.sc32 = sc32;
  a.weights = weights; a.cal_div = cal_div; a.alpha = alpha; a.grad_scale = grad_scale; a.select = nullptr; a.d_ens = d_ens;
  a.d_weights = d_weights; a.lossb = nullptr; a.divb = nullptr;
  return run_pair_loss(false, a, loss, ws, ws_bytes, st);
}

// ------------------------------------------------------------------------------------------
// intent loss (BaseIntloss.py:30-67)
// ------------------------------------------------------------------------------------------
__global__ void min_bits_kernel(const float* __restrict__ x, long long n, int* __restrict__ out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int m = 0x7fffffff;
  for (; i < n; i += (long long)gridDim.x * blockDim.x) {
    int bits = __float_as_int(x[i]);
    bits = bits >= 0 ? bits : (bits ^ 0x7fffffff);   // order-preserving map float -> int
    m = min(m, bits);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMin(out, m);
}

// one wave per session row; rowout[b] = {ce_b, kl_b}
__global__ __launch_bounds__(256) void intent_loss_kernel(const float* __restrict__ pred, const double* __restrict__ label, int B, int I,
                                                          const int* __restrict__ minbits, double kl_weight, double T2,
                                                          float grad_scale, double* __restrict__ rowout, float* __restrict__ d_pred) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const bool soften = (*minbits == 0);            // predict_labels.min() == 0 (+0.0 only; -0.0 == 0 too)
  const float* p = pred + (size_t)b * I;
  const double* t = label + (size_t)b * I;
  float S = 1.f;
  if (soften) {
    float s = 0.f;
    for (int c = lane; c < I; c += 64) s += p[c] + 1e-6f;
    S = wave_sum(s);
  }
  double ce = 0.0, kl = 0.0;
  double gp = 0.0;       // sum_j g_j p'_j for the softening backward
  for (int c = lane; c < I; c += 64) {
    const float ps = soften ? (p[c] + 1e-6f) / S : p[c];
    const double tc = t[c];
    const float t32 = (float)tc;
    const float lp = logf(ps);
    double g = 0.0;
    if (tc > 0) { ce -= tc * (double)lp; g += (1.0 - kl_weight) * (-tc / (double)ps); }
    if (tc == 0) { ce -= (double)logf(1.f - ps); g += (1.0 - kl_weight) * (1.0 / (double)(1.f - ps)); }
    if (t32 > 0.f) { kl += (double)(t32 * (logf(t32) - lp)); g += kl_weight * T2 * (-(double)t32 / (double)ps); }
    gp += g * (double)ps;
  }
  ce = wave_sum_d(ce);
  kl = wave_sum_d(kl);
  if (lane == 0) {
    rowout[2 * b] = ce;
    rowout[2 * b + 1] = kl;
  }
  if (d_pred) {
    gp = wave_sum_d(gp);
    const double cs = (double)grad_scale / (double)B;
    for (int c = lane; c < I; c += 64) {
      const float ps = soften ? (p[c] + 1e-6f) / S : p[c];
      const double tc = t[c];
      const float t32 = (float)tc;
      double g = 0.0;
      if (tc > 0) g += (1.0 - kl_weight) * (-tc / (double)ps);
      if (tc == 0) g += (1.0 - kl_weight) * (1.0 / (double)(1.f - ps));
      if (t32 > 0.f) g += kl_weight * T2 * (-(double)t32 / (double)ps);
      if (soften) g = (g - gp) / (double)S;
      d_pred[(size_t)b * I + c] = (float)(g * cs);
    }
  }
}

__global__ __launch_bounds__(256) void intent_finalize_kernel(const double* __restrict__ rowout, int B, double kl_weight, double T2,
                                                              double* __restrict__ out3) {
  __shared__ double red[4];
  double ce = 0.0, kl = 0.0;
  for (int i = threadIdx.x; i < B; i += 256) {
    ce += rowout[2 * i];
    kl += rowout[2 * i + 1];
  }
  ce = block_sum<double>(ce, red) / (double)B;
  kl = block_sum<double>(kl, red) / (double)B * T2;
  if (threadIdx.x == 0) {
    out3[0] = ce * (1.0 - kl_weight) + kl * kl_weight;
    out3[1] = ce;
    out3[2] = kl;
  }
}

size_t intent_ws_bytes(int B) { return (size_t)2 * B * sizeof(double) + 64; }

int launch_intent_loss(int B, int I, const float* pred, const double* label, double kl_weight, double kl_temp,
                       float grad_scale, double* out3, float* d_pred, void* ws, size_t ws_bytes, hipStream_t st) {
  INTEL_CHECK_ARG(ws_bytes >= intent_ws_bytes(B), "intent loss: workspace too small");
  double* rowout = reinterpret_cast<double*>(ws);
  int* minbits = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + (size_t)2 * B * sizeof(double));
  hipError_t e = hipMemsetAsync(minbits, 0x7f, sizeof(int), st);
  if (e != hipSuccess) { intel_set_error("intent loss: memset failed: %s", hipGetErrorString(e)); return (int)e; }
  const long long n = (long long)B * I;
  int blocks = (int)((n + 1023) / 1024);
  blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);
  LAUNCH(min_bits_kernel, dim3(blocks), dim3(256), 0, st, pred, n, minbits);
  INTEL_CHECK_LAUNCH();
  const double T2 = kl_temp * kl_temp;
  LAUNCH(intent_loss_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, pred, label, B, I, minbits, kl_weight, T2, grad_scale,
                     rowout, d_pred);
  INTEL_CHECK_LAUNCH();
  LAUNCH(intent_finalize_kernel, dim3(1), dim3(256), 0, st, rowout, B, kl_weight, T2, out3);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// (loss, ensemble_loss, intent_loss) of IntBPRloss / IntListloss / IntMSEloss.forward (loss/IntBPRloss.py:15-20): the float64
// total = ens * ensemble_weight + intent * intent_weight next to its two parts, in one tiny launch (instead of five torch
// elementwise kernels per step).  out3_int == NULL (BPRloss / Listloss / MSEloss alone): all three = the ensemble loss.
__global__ void loss_total_kernel(const float* __restrict__ loss_e, const double* __restrict__ out3_int, double w_e, double w_i,
                                  double* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double e = (double)loss_e[0];
    if (out3_int) {
      out[0] = e * w_e + out3_int[0] * w_i;
      out[1] = e;
      out[2] = out3_int[0];
    } else {
      out[0] = out[1] = out[2] = e;
    }
  }
}
int launch_loss_total(const float* loss_e, const double* out3_int, double w_e, double w_i, double* out, hipStream_t st) {
  LAUNCH(loss_total_kernel, dim3(1), dim3(64), 0, st, loss_e, out3_int, w_e, w_i, out);
  INTEL_CHECK_LAUNCH();
  return 0;
}
