// Per-session kernels of predict_ensemble (models/IntEL/IntEL.py:158-217) that are not GEMMs:
//   - single-query cross-attention pooling (IntEL.py:201-204 -> modules/attention.py:48-63)
//   - fusion weights broadcast + weighted-sum score aggregation (IntEL.py:212-215)
//   - the --cross_attention 0 elementwise gating (IntEL.py:206-209)
// One wave (64 lanes) per session, four sessions per 256-thread workgroup, wave shuffles for every
// per-list reduction; the candidate-list tile X[b] (L x d floats) stays in registers between the two passes
// over it when it fits (d in {64, 128}), else it is streamed twice.
#include <cstdint>
#include <initializer_list>
#include "kernels.h"
#include "session.h"

#define XP_MAXL 512

// ------------------------------------------------------------------------------------------
// Cross-attention with ONE query row per session:
//   att_l = scale * <qk_b, x_l>   (qk_b = Wk^T Wq intent_b, computed by two small GEMMs)
//   att  -= max over ALL L rows (attention.py:57, before masking)
//   w_l   = softmax over valid l (l < session_len), NaN -> 0
//   xbar  = sum_l w_l x_l         (value projection Wv applied afterwards to xbar: linear)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void xatt_pool_fwd_kernel(const float* __restrict__ X, int L, int d,
                                                            const float* __restrict__ qk, const int* __restrict__ slen,
                                                            float scale, int B, float* __restrict__ xbar,
                                                            float* __restrict__ attw) {
  __shared__ float s_att[4][XP_MAXL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  const float* Xb = X + (size_t)b * L * d;
  const float* q = qk + (size_t)b * d;
  const int len = min(slen[b], L);
  float* att = s_att[wave];
  // scores: 16 lanes per row, 4 rows per pass
  const int sub = lane & 15, grp = lane >> 4;
  for (int l0 = 0; l0 < L; l0 += 4) {
    const int l = l0 + grp;
    float s = 0.f;
    if (l < L) {
      for (int c = sub * 4; c < d; c += 64) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(Xb + (size_t)l * d + c);
        const f32x4 qv = *reinterpret_cast<const f32x4*>(q + c);
        s += xv[0] * qv[0] + xv[1] * qv[1] + xv[2] * qv[2] + xv[3] * qv[3];
      }
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    s += __shfl_xor(s, 8);
    if (l < L && sub == 0) att[l] = s * scale;
  }
  __builtin_amdgcn_wave_barrier();
  float mx = -INFINITY;
  for (int l = lane; l < L; l += 64) mx = fmaxf(mx, att[l]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int l = lane; l < len; l += 64) sum += expf(att[l] - mx);
  sum = wave_sum(sum);
  // sum == 0 (underflow or len == 0): softmax row is NaN in the reference and replaced by 0
  const float inv = sum > 0.f ? 1.f / sum : 0.f;
  for (int l = lane; l < L; l += 64) {
    const float w = l < len ? expf(att[l] - mx) * inv : 0.f;
    att[l] = w;
    attw[(size_t)b * L + l] = w;
  }
  __builtin_amdgcn_wave_barrier();
  for (int c = lane; c < d; c += 64) {
    float acc = 0.f;
    for (int l = 0; l < len; ++l) acc += att[l] * Xb[(size_t)l * d + c];
    xbar[(size_t)b * d + c] = acc;
  }
}

// Register-resident form for d = 64 * NV and L <= 4 * LP: the session's X rows are read from HBM ONCE (16 lanes per
// row, 4 rows per pass, 16-byte loads) and stay in registers between the score pass and the weighted sum.
template <int NV, int LP>
__global__ __launch_bounds__(256) void xatt_pool_fwd_reg_kernel(const float* __restrict__ X, int L, const float* __restrict__ qk,
                                                                const int* __restrict__ slen, float scale, int B,
                                                                float* __restrict__ xbar, float* __restrict__ attw,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta) {
  constexpr int d = 64 * NV;
  __shared__ float s_att[4][4 * LP];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  const float* Xb = X + (size_t)b * L * d;
  const int len = min(slen[b], L);
  float* att = s_att[wave];
  const int sub = lane & 15, grp = lane >> 4;
  f32x4 qv[NV], gm[NV], bt[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    qv[i] = *reinterpret_cast<const f32x4*>(qk + (size_t)b * d + sub * 4 + 64 * i);
    // gamma != nullptr: X holds the x-hat stash of the tower's last LayerNorm and the rows are x-hat * gamma + beta
    gm[i] = gamma ? *reinterpret_cast<const f32x4*>(gamma + sub * 4 + 64 * i) : f32x4{1.f, 1.f, 1.f, 1.f};
    bt[i] = gamma ? *reinterpret_cast<const f32x4*>(beta + sub * 4 + 64 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 x[LP][NV];
#pragma unroll
  for (int p = 0; p < LP; ++p) {
    const int l = 4 * p + grp;
    const int lc = min(l, L - 1);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      x[p][i] = *reinterpret_cast<const f32x4*>(Xb + (size_t)lc * d + sub * 4 + 64 * i);
      if (gamma) x[p][i] = x[p][i] * gm[i] + bt[i];
      const f32x4 t = x[p][i] * qv[i];
      s += (t[0] + t[1]) + (t[2] + t[3]);
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    s += __shfl_xor(s, 8);
    if (l < L && sub == 0) att[l] = s * scale;
  }
  __builtin_amdgcn_wave_barrier();
  float mx = -INFINITY;
  for (int l = lane; l < L; l += 64) mx = fmaxf(mx, att[l]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int l = lane; l < len; l += 64) sum += expf(att[l] - mx);
  sum = wave_sum(sum);
  const float inv = sum > 0.f ? 1.f / sum : 0.f;
  for (int l = lane; l < L; l += 64) {
    const float w = l < len ? expf(att[l] - mx) * inv : 0.f;
    att[l] = w;
    attw[(size_t)b * L + l] = w;
  }
  __builtin_amdgcn_wave_barrier();
  f32x4 acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < LP; ++p) {
    const int l = 4 * p + grp;
    const float w = l < len ? att[l] : 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] += w * x[p][i];
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = acc[i][e];
      a += __shfl_xor(a, 16);
      a += __shfl_xor(a, 32);
      acc[i][e] = a;
    }
    if (grp == 0) *reinterpret_cast<f32x4*>(xbar + (size_t)b * d + sub * 4 + 64 * i) = acc[i];
  }
}
static inline bool xp_aligned(std::initializer_list<const void*> ptrs) {
  for (const void* q : ptrs)
    if ((uintptr_t)q & 15) return false;
  return true;
}
#define XP_REG_DISPATCH(KERNEL, ...)                                                        \
  do {                                                                                      \
    if (d == 64) {                                                                          \
      if (L <= 20) LAUNCH((KERNEL<1, 5>), dim3(cdiv(B, 4)), dim3(256), 0, st, __VA_ARGS__); \
      else if (L <= 52) LAUNCH((KERNEL<1, 13>), dim3(cdiv(B, 4)), dim3(256), 0, st, __VA_ARGS__); \
      else LAUNCH((KERNEL<1, 25>), dim3(cdiv(B, 4)), dim3(256), 0, st, __VA_ARGS__);        \
    } else {                                                                                \
      if (L <= 20) LAUNCH((KERNEL<2, 5>), dim3(cdiv(B, 4)), dim3(256), 0, st, __VA_ARGS__); \
      else if (L <= 52) LAUNCH((KERNEL<2, 13>), dim3(cdiv(B, 4)), dim3(256), 0, st, __VA_ARGS__); \
      else LAUNCH((KERNEL<2, 25>), dim3(cdiv(B, 4)), dim3(256), 0, st, __VA_ARGS__);        \
    }                                                                                       \
  } while (0)

bool xatt_ln_fused_supported(int L, int d) { return (d == 64 || d == 128) && L >= 1 && L <= 100; }

int launch_xatt_pool_fwd(const float* X, int B, int L, int d, const float* qk, const int* slen, float scale,
                         float* xbar, float* attw, hipStream_t st, const float* gamma, const float* beta) {
  if (B <= 0) return 0;
  INTEL_CHECK_ARG(L <= XP_MAXL, "xatt_pool: list length %d > %d unsupported", L, XP_MAXL);
  INTEL_CHECK_ARG(d % 4 == 0, "xatt_pool: width %d must be a multiple of 4", d);
  if ((d == 64 || d == 128) && L <= 100 && xp_aligned({X, qk, xbar})) {
    XP_REG_DISPATCH(xatt_pool_fwd_reg_kernel, X, L, qk, slen, scale, B, xbar, attw, gamma, beta);
  } else {
    INTEL_CHECK_ARG(!gamma, "xatt_pool: the x-hat input form needs the register kernel (d=%d, L=%d)", d, L);
    LAUNCH(xatt_pool_fwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, X, L, d, qk, slen, scale, B, xbar, attw);
  }
  INTEL_CHECK_LAUNCH();
  return 0;
}

// backward: given dxbar[b,:], produces dX[b,l,:] (written, all L rows) and dqk[b,:]
//   g_l = <dxbar, x_l>;  datt_l = w_l (g_l - sum_j w_j g_j);  dx_l = w_l dxbar + datt_l*scale*qk
//   dqk = scale * sum_l datt_l x_l
__global__ __launch_bounds__(256) void xatt_pool_bwd_kernel(const float* __restrict__ X, int L, int d,
                                                            const float* __restrict__ qk, const float* __restrict__ attw,
                                                            const float* __restrict__ dxbar, int ldxb, float scale, int B,
                                                            float* __restrict__ dX, float* __restrict__ dqk) {
  __shared__ float s_g[4][XP_MAXL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  const float* Xb = X + (size_t)b * L * d;
  const float* q = qk + (size_t)b * d;
  const float* dxb = dxbar + (size_t)b * ldxb;
  const float* w = attw + (size_t)b * L;
  float* g = s_g[wave];
  const int sub = lane & 15, grp = lane >> 4;
  for (int l0 = 0; l0 < L; l0 += 4) {
    const int l = l0 + grp;
    float s = 0.f;
    if (l < L) {
      for (int c = sub * 4; c < d; c += 64) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(Xb + (size_t)l * d + c);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(dxb + c);
        s += xv[0] * gv[0] + xv[1] * gv[1] + xv[2] * gv[2] + xv[3] * gv[3];
      }
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    s += __shfl_xor(s, 8);
    if (l < L && sub == 0) g[l] = s;
  }
  __builtin_amdgcn_wave_barrier();
  float wg = 0.f;
  for (int l = lane; l < L; l += 64) wg += w[l] * g[l];
  wg = wave_sum(wg);
  __builtin_amdgcn_wave_barrier();
  for (int l = lane; l < L; l += 64) g[l] = w[l] * (g[l] - wg) * scale;   // = datt_l * scale
  __builtin_amdgcn_wave_barrier();
  for (int c = lane; c < d; c += 64) {
    const float dxc = dxb[c], qc = q[c];
    float acc = 0.f;
    for (int l = 0; l < L; ++l) {
      const float da = g[l];
      dX[((size_t)b * L + l) * d + c] = w[l] * dxc + da * qc;
      acc += da * Xb[(size_t)l * d + c];
    }
    dqk[(size_t)b * d + c] = acc;
  }
}

template <int NV, int LP>
__global__ __launch_bounds__(256) void xatt_pool_bwd_reg_kernel(const float* __restrict__ X, int L, const float* __restrict__ qk,
                                                                const float* __restrict__ attw, const float* __restrict__ dxbar,
                                                                int ldxb, float scale, int B, float* __restrict__ dX,
                                                                float* __restrict__ dqk) {
  constexpr int d = 64 * NV;
  __shared__ float s_g[4][4 * LP];
  __shared__ float s_w[4][4 * LP];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  const float* Xb = X + (size_t)b * L * d;
  float* g = s_g[wave];
  float* w = s_w[wave];
  const int sub = lane & 15, grp = lane >> 4;
  for (int l = lane; l < L; l += 64) w[l] = attw[(size_t)b * L + l];
  f32x4 qv[NV], gv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    qv[i] = *reinterpret_cast<const f32x4*>(qk + (size_t)b * d + sub * 4 + 64 * i);
    gv[i] = *reinterpret_cast<const f32x4*>(dxbar + (size_t)b * ldxb + sub * 4 + 64 * i);
  }
  f32x4 x[LP][NV];
#pragma unroll
  for (int p = 0; p < LP; ++p) {
    const int l = 4 * p + grp;
    const int lc = min(l, L - 1);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      x[p][i] = *reinterpret_cast<const f32x4*>(Xb + (size_t)lc * d + sub * 4 + 64 * i);
      const f32x4 t = x[p][i] * gv[i];
      s += (t[0] + t[1]) + (t[2] + t[3]);
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    s += __shfl_xor(s, 8);
    if (l < L && sub == 0) g[l] = s;
  }
  __builtin_amdgcn_wave_barrier();
  float wg = 0.f;
  for (int l = lane; l < L; l += 64) wg += w[l] * g[l];
  wg = wave_sum(wg);
  __builtin_amdgcn_wave_barrier();
  for (int l = lane; l < L; l += 64) g[l] = w[l] * (g[l] - wg) * scale;   // = datt_l * scale
  __builtin_amdgcn_wave_barrier();
  f32x4 acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < LP; ++p) {
    const int l = 4 * p + grp;
    if (l < L) {
      const float da = g[l], wl = w[l];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        *reinterpret_cast<f32x4*>(dX + ((size_t)b * L + l) * d + sub * 4 + 64 * i) = wl * gv[i] + da * qv[i];
        acc[i] += da * x[p][i];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = acc[i][e];
      a += __shfl_xor(a, 16);
      a += __shfl_xor(a, 32);
      acc[i][e] = a;
    }
    if (grp == 0) *reinterpret_cast<f32x4*>(dqk + (size_t)b * d + sub * 4 + 64 * i) = acc[i];
  }
}

int launch_xatt_pool_bwd(const float* X, int B, int L, int d, const float* qk, const float* attw, const float* dxbar,
                         int ldxb, float scale, float* dX, float* dqk, hipStream_t st) {
  if (B <= 0) return 0;
  INTEL_CHECK_ARG(L <= XP_MAXL, "xatt_pool_bwd: list length %d > %d unsupported", L, XP_MAXL);
  if ((d == 64 || d == 128) && L <= 100 && (ldxb & 3) == 0 && xp_aligned({X, qk, dxbar, dX, dqk})) {
    XP_REG_DISPATCH(xatt_pool_bwd_reg_kernel, X, L, qk, attw, dxbar, ldxb, scale, B, dX, dqk);
  } else {
    LAUNCH(xatt_pool_bwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, X, L, d, qk, attw, dxbar, ldxb, scale, B, dX, dqk);
  }
  INTEL_CHECK_LAUNCH();
  return 0;
}

// xatt_pool backward FUSED with the backward of the LayerNorm that produced X (the last layer of a tower, IntEL.py:187,196):
// X is given as its x-hat stash (+ gamma, beta, rstd); the gradient row dy_l = w_l dxbar + datt_l scale qk never leaves the
// registers and the kernel writes dz = rstd (g - mean(g) - xhat mean(g xhat)), g = gamma dy, for ALL L rows, plus per-
// workgroup column partials of dgamma = sum dy xhat and dbeta = sum dy (slab[blockIdx][2][d]) and dqk.
template <int NV, int LP>
__global__ __launch_bounds__(256) void xatt_pool_ln_bwd_reg_kernel(const float* __restrict__ XH, const float* __restrict__ rstd, int L,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   const float* __restrict__ qk, const float* __restrict__ attw,
                                                                   const float* __restrict__ dxbar, int ldxb, float scale, int B,
                                                                   float* __restrict__ dZ, float* __restrict__ dqk,
                                                                   float* __restrict__ slabs) {
  constexpr int d = 64 * NV;
  __shared__ float s_g[4][4 * LP];
  __shared__ float s_w[4][4 * LP];
  __shared__ float s_gb[2][4][d];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * 4 + wave;
  const bool live = b < B;
  const int sub = lane & 15, grp = lane >> 4;
  f32x4 ag[NV], ab[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) ag[i] = ab[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (live) {
    const float* Xb = XH + (size_t)b * L * d;
    float* g = s_g[wave];
    float* w = s_w[wave];
    for (int l = lane; l < L; l += 64) w[l] = attw[(size_t)b * L + l];
    // x_l = xhat_l * gamma + beta is never formed: <x_l, v> = <xhat_l, gamma v> + <beta, v> and
    // sum_l a_l x_l = gamma (sum_l a_l xhat_l) + beta sum_l a_l -- only the x-hat rows live in registers
    f32x4 qv[NV], gv[NV], gm[NV], bt[NV], ggv[NV];
    float cb = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      qv[i] = *reinterpret_cast<const f32x4*>(qk + (size_t)b * d + sub * 4 + 64 * i);
      gv[i] = *reinterpret_cast<const f32x4*>(dxbar + (size_t)b * ldxb + sub * 4 + 64 * i);
      gm[i] = *reinterpret_cast<const f32x4*>(gamma + sub * 4 + 64 * i);
      bt[i] = *reinterpret_cast<const f32x4*>(beta + sub * 4 + 64 * i);
      ggv[i] = gm[i] * gv[i];
      const f32x4 t = bt[i] * gv[i];
      cb += (t[0] + t[1]) + (t[2] + t[3]);
    }
    cb += __shfl_xor(cb, 1);
    cb += __shfl_xor(cb, 2);
    cb += __shfl_xor(cb, 4);
    cb += __shfl_xor(cb, 8);
    f32x4 xh[LP][NV];
#pragma unroll
    for (int p = 0; p < LP; ++p) {
      const int l = 4 * p + grp;
      const int lc = min(l, L - 1);
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        xh[p][i] = *reinterpret_cast<const f32x4*>(Xb + (size_t)lc * d + sub * 4 + 64 * i);
        const f32x4 t = xh[p][i] * ggv[i];
        s += (t[0] + t[1]) + (t[2] + t[3]);
      }
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      s += __shfl_xor(s, 4);
      s += __shfl_xor(s, 8);
      if (l < L && sub == 0) g[l] = s + cb;
    }
    __builtin_amdgcn_wave_barrier();
    float wg = 0.f;
    for (int l = lane; l < L; l += 64) wg += w[l] * g[l];
    wg = wave_sum(wg);
    __builtin_amdgcn_wave_barrier();
    for (int l = lane; l < L; l += 64) g[l] = w[l] * (g[l] - wg) * scale;   // = datt_l * scale
    __builtin_amdgcn_wave_barrier();
    // dy_l = w_l dxbar + datt_l qk is rank two over the list, so are its column sums: with A = sum_l datt_l xhat_l and
    // W = sum_l w_l xhat_l: dgamma = dxbar W + qk A, dbeta = dxbar sum w + qk sum datt, dqk = gamma A + beta sum datt
    f32x4 acc[NV], accw[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = accw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float inv_n = 1.f / (float)d;
    float sda = 0.f, sw = 0.f;
#pragma unroll
    for (int p = 0; p < LP; ++p) {
      const int l = 4 * p + grp;
      const bool ok = l < L;
      const int lc = min(l, L - 1);
      const float da = ok ? g[lc] : 0.f, wl = ok ? w[lc] : 0.f;
      f32x4 gg[NV];
      float s1 = 0.f, s2 = 0.f;
      sda += da;
      sw += wl;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const f32x4 dy = wl * gv[i] + da * qv[i];
        acc[i] += da * xh[p][i];
        accw[i] += wl * xh[p][i];
        gg[i] = dy * gm[i];
        const f32x4 gh = gg[i] * xh[p][i];
        s1 += (gg[i][0] + gg[i][1]) + (gg[i][2] + gg[i][3]);
        s2 += (gh[0] + gh[1]) + (gh[2] + gh[3]);
      }
      s1 += __shfl_xor(s1, 1);
      s1 += __shfl_xor(s1, 2);
      s1 += __shfl_xor(s1, 4);
      s1 += __shfl_xor(s1, 8);
      s2 += __shfl_xor(s2, 1);
      s2 += __shfl_xor(s2, 2);
      s2 += __shfl_xor(s2, 4);
      s2 += __shfl_xor(s2, 8);
      const float m1 = s1 * inv_n, m2 = s2 * inv_n;
      if (ok) {
        const float rs = rstd[(size_t)b * L + l];
#pragma unroll
        for (int i = 0; i < NV; ++i)
          *reinterpret_cast<f32x4*>(dZ + ((size_t)b * L + l) * d + sub * 4 + 64 * i) = rs * (gg[i] - m1 - xh[p][i] * m2);
      }
      __builtin_amdgcn_sched_barrier(0);       // one row group at a time: the x-hat rows already fill the register file
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      ag[i] = gv[i] * accw[i] + qv[i] * acc[i];          // this lane's rows; the row groups are summed below
      ab[i] = gv[i] * sw + qv[i] * sda;
    }
    sda += __shfl_xor(sda, 16);
    sda += __shfl_xor(sda, 32);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = acc[i][e];
        a += __shfl_xor(a, 16);
        a += __shfl_xor(a, 32);
        acc[i][e] = a;
      }
      if (grp == 0) *reinterpret_cast<f32x4*>(dqk + (size_t)b * d + sub * 4 + 64 * i) = acc[i] * gm[i] + bt[i] * sda;
    }
  }
  // column partials of dgamma / dbeta: over the row groups of the wave, then over the four sessions of the workgroup
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = ag[i][e], c = ab[i][e];
      a += __shfl_xor(a, 16);
      a += __shfl_xor(a, 32);
      c += __shfl_xor(c, 16);
      c += __shfl_xor(c, 32);
      ag[i][e] = a;
      ab[i][e] = c;
    }
    if (grp == 0) {
      *reinterpret_cast<f32x4*>(&s_gb[0][wave][sub * 4 + 64 * i]) = ag[i];
      *reinterpret_cast<f32x4*>(&s_gb[1][wave][sub * 4 + 64 * i]) = ab[i];
    }
  }
  __syncthreads();
  float* slab = slabs + (size_t)blockIdx.x * 2 * d;
  for (int c = threadIdx.x; c < d; c += 256) {
    slab[c] = (s_gb[0][0][c] + s_gb[0][1][c]) + (s_gb[0][2][c] + s_gb[0][3][c]);
    slab[d + c] = (s_gb[1][0][c] + s_gb[1][1][c]) + (s_gb[1][2][c] + s_gb[1][3][c]);
  }
}

size_t xatt_ln_bwd_slab_floats(int B, int d) { return (size_t)cdiv(B, 4) * 2 * d; }

// dgamma / dbeta go through the deferred reduction queue (slabs from its arena)
int launch_xatt_pool_ln_bwd(const float* XH, const float* rstd, const float* gamma, const float* beta, int B, int L, int d,
                            const float* qk, const float* attw, const float* dxbar, int ldxb, float scale, float* dZ, float* dqk,
                            float* dgamma, float* dbeta, int accumulate, hipStream_t st, ReduceQueue* q) {
  if (B <= 0) return 0;
  INTEL_CHECK_ARG(xatt_ln_fused_supported(L, d) && (ldxb & 3) == 0 && xp_aligned({XH, qk, dxbar, dZ, dqk, gamma, beta}) && q,
                  "xatt_pool_ln_bwd: unsupported shape or alignment (d=%d, L=%d)", d, L);
  const int nb = cdiv(B, 4);
  float* slabs = redq_alloc(q, xatt_ln_bwd_slab_floats(B, d));
  if (!slabs) {
    intel_set_error("xatt_pool_ln_bwd: reduction arena exhausted");
    return -2;   // INTEL_E_WORKSPACE
  }
  XP_REG_DISPATCH(xatt_pool_ln_bwd_reg_kernel, XH, rstd, L, gamma, beta, qk, attw, dxbar, ldxb, scale, B, dZ, dqk, slabs);
  INTEL_CHECK_LAUNCH();
  redq_push(q, slabs, (size_t)2 * d, nb, 1, d, dgamma, d, accumulate);
  redq_push(q, slabs + d, (size_t)2 * d, nb, 1, d, dbeta, d, accumulate);
  return 0;
}

// ------------------------------------------------------------------------------------------
// weights[b,l,:] = l < len ? wv[b,:] : wpad[b,:];  ens[b,l] = sum_k weights*scores (IntEL.py:214-215)
// (cross_attention=1: the fusion weights are session-constant for valid rows, SURVEY.md §0.4;
//  padded rows see [0,0,h_u,h_intent] -> wpad.)  per_item != 0: weights already hold per-item values.
// ------------------------------------------------------------------------------------------
__global__ void ens_fwd_kernel(const float* __restrict__ wv, const float* __restrict__ wpad, const float* __restrict__ scores,
                               const int* __restrict__ slen, int B, int L, int K, int per_item,
                               float* __restrict__ weights, float* __restrict__ ens) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * L) return;
  const int b = i / L, l = i - b * L;
  const float* src = per_item ? (weights + (size_t)i * K) : ((l < slen[b]) ? wv + (size_t)b * K : wpad + (size_t)b * K);
  float acc = 0.f;
  for (int k = 0; k < K; ++k) {
    const float w = src[k];
    if (!per_item) weights[(size_t)i * K + k] = w;
    acc += w * scores[(size_t)i * K + k];
  }
  ens[i] = acc;
}
int launch_ens_fwd(const float* wv, const float* wpad, const float* scores, const int* slen, int B, int L, int K,
                   int per_item, float* weights, float* ens, hipStream_t st) {
  if (B * L <= 0) return 0;
  LAUNCH(ens_fwd_kernel, dim3(cdiv(B * L, 256)), dim3(256), 0, st, wv, wpad, scores, slen, B, L, K, per_item, weights, ens);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// dwt[b,l,k] = d_weights + d_ens*scores;  dwv[b,k] = sum_{l<len} dwt, dwpad[b,k] = sum_{l>=len} dwt.
// per_item: writes dwt[M,K] instead (dwv/dwpad unused).
__global__ __launch_bounds__(256) void ens_bwd_kernel(const float* __restrict__ d_weights, const float* __restrict__ d_ens,
                                                      const float* __restrict__ scores, const int* __restrict__ slen, int B,
                                                      int L, int K, int per_item, float* __restrict__ dwv,
                                                      float* __restrict__ dwpad, float* __restrict__ dwt) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  const int len = slen[b];
  for (int k = 0; k < K; ++k) {
    float sv = 0.f, sp = 0.f;
    for (int l = lane; l < L; l += 64) {
      const size_t i = (size_t)b * L + l;
      float v = 0.f;
      if (d_weights) v += d_weights[i * K + k];
      if (d_ens) v += d_ens[i] * scores[i * K + k];
      if (per_item) dwt[i * K + k] = v;
      if (l < len) sv += v; else sp += v;
    }
    if (!per_item) {
      sv = wave_sum(sv);
      sp = wave_sum(sp);
      if (lane == 0) {
        dwv[(size_t)b * K + k] = sv;
        dwpad[(size_t)b * K + k] = sp;
      }
    }
  }
}
int launch_ens_bwd(const float* d_weights, const float* d_ens, const float* scores, const int* slen, int B, int L, int K,
                   int per_item, float* dwv, float* dwpad, float* dwt, hipStream_t st) {
  if (B <= 0) return 0;
  LAUNCH(ens_bwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, d_weights, d_ens, scores, slen, B, L, K, per_item, dwv, dwpad, dwt);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------
// --cross_attention 0 gating (IntEL.py:206-209): dst[m, col0+c] = x[m,c] * vec[b,c]
// backward: dx[m,c] (+)= dfeat[m,col0+c]*vec[b,c];  dvec[b,c] = sum_l dfeat[m,col0+c]*x[m,c]
// ------------------------------------------------------------------------------------------
__global__ void gate_fwd_kernel(const float* __restrict__ x, int d, const float* __restrict__ vec, int B, int L,
                                float* __restrict__ dst, int ldd, int col0) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * L * d) return;
  const int c = (int)(i % d);
  const long long m = i / d;
  const int b = (int)(m / L);
  dst[(size_t)m * ldd + col0 + c] = x[i] * vec[(size_t)b * d + c];
}
int launch_gate_fwd(const float* x, int d, const float* vec, int B, int L, float* dst, int ldd, int col0, hipStream_t st) {
  long long n = (long long)B * L * d;
  if (n <= 0) return 0;
  LAUNCH(gate_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, d, vec, B, L, dst, ldd, col0);
  INTEL_CHECK_LAUNCH();
  return 0;
}
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ dfeat, int ldf, int col0, const float* __restrict__ x,
                                                       int d, const float* __restrict__ vec, int B, int L,
                                                       float* __restrict__ dx, float* __restrict__ dvec) {
  // one block per session; thread c loops over rows
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < d; c += 256) {
    const float vc = vec[(size_t)b * d + c];
    float acc = 0.f;
    for (int l = 0; l < L; ++l) {
      const size_t m = (size_t)b * L + l;
      const float g = dfeat[m * ldf + col0 + c];
      dx[m * d + c] = g * vc;
      acc += g * x[m * d + c];
    }
    dvec[(size_t)b * d + c] = acc;
  }
}
int launch_gate_bwd(const float* dfeat, int ldf, int col0, const float* x, int d, const float* vec, int B, int L, float* dx,
                    float* dvec, hipStream_t st) {
  if (B <= 0) return 0;
  LAUNCH(gate_bwd_kernel, dim3(B), dim3(256), 0, st, dfeat, ldf, col0, x, d, vec, B, L, dx, dvec);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// mean-pooled gate of the aWELv_IntEL variant (models/supervise/aWELv_IntEL.py:188-197): xatt = h * g(intent), then
// .mean(dim=1) over ALL L rows (pads included -- nothing is masked there):
//   xbar[b,c] = mean_l x[b,l,c];  feat[b, col0+c] = xbar[b,c] * vec[b,c]
// backward: dx[b,l,c] = dfeat[b,col0+c] * vec[b,c] / L  (every row);  dvec[b,c] = dfeat[b,col0+c] * xbar[b,c]
__global__ __launch_bounds__(256) void gate_mean_fwd_kernel(const float* __restrict__ x, int d, const float* __restrict__ vec, int B, int L,
                                                            float* __restrict__ xbar, float* __restrict__ feat, int ldf, int col0) {
  const int b = blockIdx.x;
  const float inv = 1.f / (float)L;
  for (int c = threadIdx.x; c < d; c += 256) {
    float acc = 0.f;
    for (int l = 0; l < L; ++l) acc += x[((size_t)b * L + l) * d + c];
    const float m = acc * inv;
    xbar[(size_t)b * d + c] = m;
    feat[(size_t)b * ldf + col0 + c] = m * vec[(size_t)b * d + c];
  }
}
int launch_gate_mean_fwd(const float* x, int d, const float* vec, int B, int L, float* xbar, float* feat, int ldf, int col0, hipStream_t st) {
  if (B <= 0) return 0;
  LAUNCH(gate_mean_fwd_kernel, dim3(B), dim3(256), 0, st, x, d, vec, B, L, xbar, feat, ldf, col0);
  INTEL_CHECK_LAUNCH();
  return 0;
}
__global__ __launch_bounds__(256) void gate_mean_bwd_kernel(const float* __restrict__ dfeat, int ldf, int col0, const float* __restrict__ xbar,
                                                            int d, const float* __restrict__ vec, int B, int L, float* __restrict__ dx,
                                                            float* __restrict__ dvec) {
  const int b = blockIdx.x;
  const float inv = 1.f / (float)L;
  for (int c = threadIdx.x; c < d; c += 256) {
    const float g = dfeat[(size_t)b * ldf + col0 + c];
    dvec[(size_t)b * d + c] = g * xbar[(size_t)b * d + c];
    const float gx = g * vec[(size_t)b * d + c] * inv;
    for (int l = 0; l < L; ++l) dx[((size_t)b * L + l) * d + c] = gx;
  }
}
int launch_gate_mean_bwd(const float* dfeat, int ldf, int col0, const float* xbar, int d, const float* vec, int B, int L, float* dx,
                         float* dvec, hipStream_t st) {
  if (B <= 0) return 0;
  LAUNCH(gate_mean_bwd_kernel, dim3(B), dim3(256), 0, st, dfeat, ldf, col0, xbar, d, vec, B, L, dx, dvec);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// out[b, c] = sum_l src[(b*L + l), col0 + c]   (gradient of a per-session vector broadcast over the list)
__global__ __launch_bounds__(256) void session_colsum_kernel(const float* __restrict__ src, int lds, int col0, int d, int B, int L,
                                                             float* __restrict__ out, int ldo, int ocol0, int accumulate) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < d; c += 256) {
    float acc = 0.f;
    for (int l = 0; l < L; ++l) acc += src[((size_t)b * L + l) * lds + col0 + c];
    float* dst = out + (size_t)b * ldo + ocol0 + c;
    *dst = accumulate ? *dst + acc : acc;
  }
}
int launch_session_colsum(const float* src, int lds, int col0, int d, int B, int L, float* out, int ldo, int ocol0,
                          int accumulate, hipStream_t st) {
  if (B <= 0) return 0;
  LAUNCH(session_colsum_kernel, dim3(B), dim3(256), 0, st, src, lds, col0, d, B, L, out, ldo, ocol0, accumulate);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------
// encoder helpers (models/GeneralSeq.py:89-106, IntEL.py:135-143)
// ------------------------------------------------------------------------------------------
// E[b*T+t, :] += pos_emb[t < len_b ? t : 0, :]   (position ids: forward order, pads -> 0)
__global__ void add_pos_kernel(float* __restrict__ E, int dm, const float* __restrict__ pos, const int* __restrict__ len,
                               int B, int T) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * T * dm) return;
  const int c = (int)(i % dm);
  const long long m = i / dm;
  const int b = (int)(m / T), t = (int)(m - (long long)b * T);
  const int p = t < len[b] ? t : 0;
  E[i] += pos[(size_t)p * dm + c];
}
int launch_add_pos(float* E, int dm, const float* pos, const int* len, int B, int T, hipStream_t st) {
  long long n = (long long)B * T * dm;
  if (n <= 0) return 0;
  LAUNCH(add_pos_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, E, dm, pos, len, B, T);
  INTEL_CHECK_LAUNCH();
  return 0;
}
// the same over PACKED history rows: E[row] += pos[row_t[row]] (row_t = position of the row inside its session)
__global__ void add_pos_rows_kernel(float* __restrict__ E, int dm, const float* __restrict__ pos, const int* __restrict__ row_t, int rows) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)rows * dm) return;
  const int c = (int)(i % dm);
  const int m = (int)(i / dm);
  E[i] += pos[(size_t)row_t[m] * dm + c];
}
int launch_add_pos_rows(float* E, int dm, const float* pos, const int* row_t, int rows, hipStream_t st) {
  long long n = (long long)rows * dm;
  if (n <= 0) return 0;
  LAUNCH(add_pos_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, E, dm, pos, row_t, rows);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// Packing of one history: the padded positions t >= len[b] of a BERT4Rec input never reach a valid row (their keys are masked,
// GeneralSeq.py:100; everything else is row-wise; the output is multiplied by `valid` and only row len-1 is used, :103-105), so
// the encoder can run on the valid rows alone.  Session b's rows move from [b*T, b*T + len[b]) to [off[b], off[b] + len[b]):
//   ids_out[row] = ids[b*T + t];  idx2_out[row] = idx2[b*T + t] (optional);  vec_out[row, :] = vec[b*T + t, :] (optional, w wide);
//   row_t[row] = t.   One workgroup per session.
__global__ __launch_bounds__(256) void his_pack_kernel(const int* __restrict__ len, const int* __restrict__ off, int B, int T,
                                                       const int* __restrict__ ids, int* __restrict__ ids_out,
                                                       const int* __restrict__ idx2, int* __restrict__ idx2_out,
                                                       const float* __restrict__ vec, int w, float* __restrict__ vec_out,
                                                       int* __restrict__ row_t) {
  const int b = blockIdx.x;
  const int n = min(max(len[b], 0), T), base = off[b];
  for (int t = threadIdx.x; t < n; t += 256) {
    ids_out[base + t] = ids[(size_t)b * T + t];
    if (idx2) idx2_out[base + t] = idx2[(size_t)b * T + t];
    row_t[base + t] = t;
  }
  if (vec)
    for (int i = threadIdx.x; i < n * w; i += 256) vec_out[(size_t)base * w + i] = vec[(size_t)b * T * w + i];
}
int launch_his_pack(const int* len, const int* off, int B, int T, const int* ids, int* ids_out, const int* idx2, int* idx2_out,
                    const float* vec, int w, float* vec_out, int* row_t, hipStream_t st) {
  if (B <= 0) return 0;
  LAUNCH(his_pack_kernel, dim3(B), dim3(256), 0, st, len, off, B, T, ids, ids_out, idx2, idx2_out, vec, w, vec_out, row_t);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// gradient of the position embedding (GeneralSeq.py:95-97): dpos[t, :] += sum of dE[row, :] over the rows at position t.  Rows
// arrive in session order, so positions are spread evenly over any row range: every workgroup sums its rows into a [T, dm] table
// in LDS (ds_add_f32, no hot address) and writes it as one slab; the slabs are summed by the batched slab reduction like every
// other weight gradient (fixed order).  row_t: position of each packed row, or NULL for padded [B, T] rows (position = t if
// t < len[b] else 0, like the forward).
__global__ __launch_bounds__(256) void pos_grad_kernel(const float* __restrict__ dE, int dm, const int* __restrict__ row_t,
                                                       const int* __restrict__ len, int T, int rows, float* __restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) float s_tab[];      // [T][dm]
  const int n = T * dm;
  for (int i = threadIdx.x; i < n; i += 256) s_tab[i] = 0.f;
  __syncthreads();
  // lane = column (consecutive lanes hit consecutive LDS banks: a float4 per lane would put 32 lanes on 8 banks)
  const long long total = (long long)rows * dm, stride = (long long)gridDim.x * 256;
  for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < total; i0 += 4 * stride) {
    float v[4];
    int dst[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {          // four independent (position, value) loads in flight
      const long long i = i0 + u * stride;
      const bool ok = i < total;
      const long long ic = ok ? i : 0;
      const int row = (int)(ic / dm), c = (int)(ic - (long long)row * dm);
      int t;
      if (row_t) {
        t = row_t[row];
      } else {
        const int b = row / T;
        t = row - b * T;
        if (t >= len[b]) t = 0;
      }
      dst[u] = ok ? t * dm + c : -1;
      v[u] = dE[ic];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (dst[u] >= 0) unsafeAtomicAdd(s_tab + dst[u], v[u]);      // ds_add_f32 (plain atomicAdd on LDS floats is a compare-and-swap loop)
  }
  __syncthreads();
  float* slab = slabs + (size_t)blockIdx.x * n;
  for (int i = threadIdx.x; i < n; i += 256) slab[i] = s_tab[i];
}
// Packed rows (off[b] = first row of session b, len[b] rows, position = row - off[b]): no atomics at all.  A thread keeps one
// column; it visits the rows at ITS positions (t = tsub, tsub + 256/dm, ...) of the workgroup's sessions and sums them in a
// register -- every row is read once, by the dm lanes of one position group, and the slab is written directly.  The LDS-atomic
// form above spends its time in ds_add_f32 (5.5 M of them per encoder at Tmall shape: 34 us; deeper load unrolling and a
// division-free index did not move it).
#define PGP_SESS 8            // sessions whose rows are in flight together
__global__ __launch_bounds__(256) void pos_grad_packed_kernel(const float* __restrict__ dE, int dm, const int* __restrict__ off,
                                                              const int* __restrict__ len, int B, int T, float* __restrict__ slabs) {
  // a thread keeps FOUR columns (16-byte loads): dm / 4 lanes per position, 1024 / dm positions in flight per workgroup
  const int dm4 = dm >> 2, c = (threadIdx.x % dm4) * 4, tsub = threadIdx.x / dm4, tpp = 256 / dm4;
  const int per = (B + gridDim.x - 1) / gridDim.x;
  const int b_begin = blockIdx.x * per, b_end = min(B, b_begin + per);
  float* slab = slabs + (size_t)blockIdx.x * T * dm;
  for (int t = tsub; t < T; t += tpp) {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int b0 = b_begin; b0 < b_end; b0 += PGP_SESS) {
      int o[PGP_SESS], l[PGP_SESS];
#pragma unroll
      for (int u = 0; u < PGP_SESS; ++u) {
        const int b = min(b0 + u, b_end - 1);
        o[u] = off[b];
        l[u] = (b0 + u < b_end) ? min(len[b], T) : 0;
      }
      f32x4 v[PGP_SESS];
#pragma unroll
      for (int u = 0; u < PGP_SESS; ++u)
        v[u] = (t < l[u]) ? *reinterpret_cast<const f32x4*>(dE + (size_t)(o[u] + t) * dm + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < PGP_SESS; ++u) acc += v[u];
    }
    *reinterpret_cast<f32x4*>(slab + (size_t)t * dm + c) = acc;
  }
}

int pos_grad_slabs(int rows) {
  int s = cdiv(rows, 64);
  return s < 1 ? 1 : (s > 512 ? 512 : s);
}
bool pos_grad_supported(int T, int dm) { return dm % 4 == 0 && (size_t)T * dm * sizeof(float) <= 150 * 1024; }
// dpos[0:T, :] = the sum (valid after redq_flush); needs pos_grad_slabs(rows) * T * dm floats of the queue's arena
int launch_pos_grad(const float* dE, int dm, const int* row_t, const int* len, int T, int rows, float* dpos, hipStream_t st,
                    ReduceQueue* q, const int* off, int B) {
  if (rows <= 0) return 0;
  if (off && row_t && B > 0 && dm >= 4 && dm <= 1024 && 1024 % dm == 0 && q) {
    // four sessions per workgroup up to 512 workgroups (long histories at small batches: the loop over positions is the kernel's latency)
    const int S = (B + 3) / 4 < 1 ? 1 : ((B + 3) / 4 > 512 ? 512 : (B + 3) / 4);
    float* slabs = redq_alloc(q, (size_t)S * T * dm);
    if (!slabs) {
      intel_set_error("pos_grad: reduction arena exhausted");
      return -2;
    }
    LAUNCH_W(0.0, 4.0 * (double)rows * dm, pos_grad_packed_kernel, dim3(S), dim3(256), 0, st, dE, dm, off, len, B, T, slabs);
    INTEL_CHECK_LAUNCH();
    redq_push(q, slabs, (size_t)T * dm, S, T, dm, dpos, dm, 0);
    return 0;
  }
  INTEL_CHECK_ARG(pos_grad_supported(T, dm) && q, "pos_grad: table %d x %d does not fit LDS", T, dm);
  const size_t smem = (size_t)T * dm * sizeof(float);
  allow_lds(pos_grad_kernel, smem);
  const int S = pos_grad_slabs(rows);
  float* slabs = redq_alloc(q, (size_t)S * T * dm);
  if (!slabs) {
    intel_set_error("pos_grad: reduction arena exhausted");
    return -2;
  }
  LAUNCH_W(0.0, 4.0 * (double)rows * dm, pos_grad_kernel, dim3(S), dim3(256), smem, st, dE, dm, row_t, len, T, rows, slabs);
  INTEL_CHECK_LAUNCH();
  redq_push(q, slabs, (size_t)T * dm, S, T, dm, dpos, dm, 0);
  return 0;
}

// one-hot intent rows of the item history (IntEL.py:142 with his_item_int one-hot):
// E[m, col0:col0+d_int] = Wint[:, idx[m]] + bint   (idx < 0: bias only)
__global__ void onehot_linear_kernel(const float* __restrict__ W, const float* __restrict__ bias, int d_int, int I,
                                     const int* __restrict__ idx, int M, float* __restrict__ E, int lde, int col0,
                                     const float* __restrict__ pos, const int* __restrict__ row_t) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)M * d_int) return;
  const int c = (int)(i % d_int);
  const int m = (int)(i / d_int);
  const int j = idx[m];
  float v = bias[c];
  if (j >= 0) v += W[(size_t)c * I + j];
  if (pos) v += pos[(size_t)row_t[m] * lde + col0 + c];        // position embedding of a packed history row
  E[(size_t)m * lde + col0 + c] = v;
}
int launch_onehot_linear(const float* W, const float* bias, int d_int, int I, const int* idx, int M, float* E, int lde,
                         int col0, hipStream_t st, const float* pos, const int* row_t) {
  if (!row_t) pos = nullptr;
  long long n = (long long)M * d_int;
  if (n <= 0) return 0;
  LAUNCH(onehot_linear_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, bias, d_int, I, idx, M, E, lde, col0, pos, row_t);
  INTEL_CHECK_LAUNCH();
  return 0;
}
// oh[m, :] = one_hot(row index of m) over R columns: idx[m] when idx != null, else the BERT4Rec
// position of row m (t < len[b] ? t : 0).  The small-table gradients (position embeddings, the
// one-hot intent rows) are then plain dY^T X products on the MFMA wgrad kernel: deterministic and
// free of same-address atomics.
__global__ void make_onehot_kernel(const int* __restrict__ idx, const int* __restrict__ len, int T, int M, int R,
                                   float* __restrict__ oh) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)M * R) return;
  const int m = (int)(i / R), c = (int)(i - (long long)m * R);
  int row;
  if (idx) row = idx[m];
  else { const int b = m / T, t = m - b * T; row = t < len[b] ? t : 0; }
  oh[i] = (row == c) ? 1.f : 0.f;
}
int launch_make_onehot(const int* idx, const int* len, int T, int M, int R, float* oh, hipStream_t st) {
  long long n = (long long)M * R;
  if (n <= 0) return 0;
  LAUNCH(make_onehot_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, idx, len, T, M, R, oh);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// vec[b,:] = E[b*T + len_b - 1, :] * 1   (GeneralSeq.py:103-105; the selected row is always valid)
__global__ void select_last_kernel(const float* __restrict__ E, int dm, const int* __restrict__ len, int B, int T,
                                   float* __restrict__ out, int ldo, int col0, const int* __restrict__ row_off) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * dm) return;
  const int b = i / dm, c = i - b * dm;
  int t = len[b] - 1;
  t = t < 0 ? T + t : t;            // torch negative indexing when len == 0
  t = min(max(t, 0), T - 1);
  const float valid = (t < len[b]) ? 1.f : 0.f;
  const size_t base = row_off ? (size_t)row_off[b] : (size_t)b * T;
  out[(size_t)b * ldo + col0 + c] = valid > 0.f ? E[(base + t) * dm + c] : 0.f;
}
int launch_select_last(const float* E, int dm, const int* len, int B, int T, float* out, int ldo, int col0, hipStream_t st,
                       const int* row_off) {
  if (B * dm <= 0) return 0;
  LAUNCH(select_last_kernel, dim3(cdiv(B * dm, 256)), dim3(256), 0, st, E, dm, len, B, T, out, ldo, col0, row_off);
  INTEL_CHECK_LAUNCH();
  return 0;
}
// copy a column block: dst[m, dcol0 + c] = src[m, scol0 + c] (optionally * (mask>0))
__global__ void copy_cols_kernel(const float* __restrict__ src, int lds, int scol0, int d, long long M, float* __restrict__ dst,
                                 int ldd, int dcol0, const float* __restrict__ relu_out, int ldr, int rcol0, int accumulate) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * d) return;
  const int c = (int)(i % d);
  const long long m = i / d;
  float v = src[(size_t)m * lds + scol0 + c];
  if (relu_out && !(relu_out[(size_t)m * ldr + rcol0 + c] > 0.f)) v = 0.f;
  float* p = dst + (size_t)m * ldd + dcol0 + c;
  *p = accumulate ? *p + v : v;
}
int launch_copy_cols(const float* src, int lds, int scol0, int d, long long M, float* dst, int ldd, int dcol0,
                     const float* relu_out, int ldr, int rcol0, int accumulate, hipStream_t st) {
  long long n = M * d;
  if (n <= 0) return 0;
  LAUNCH(copy_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, lds, scol0, d, M, dst, ldd, dcol0,
                     relu_out, ldr, rcol0, accumulate);
  INTEL_CHECK_LAUNCH();
  return 0;
}


// ------------------------------------------------------------------------------------------
// Last BERT4Rec block, pruned: only row len-1 of the block's output is ever used
// (his_vector = seq[b, len_b - 1], models/GeneralSeq.py:103-105), so the last block needs K/V for all
// rows but Q, the attention output, both LayerNorms and the FFN for ONE row per session.  This kernel is
// that row's attention: one wave per (session, head); scores over the valid keys j < len.
//   kv: [B*T, 2*dm] = [k | v] rows;  q: [B, dm];  out: [B, dm];  P: [B*heads, T] (saved for backward)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_lastq_fwd_kernel(const float* __restrict__ kv, const float* __restrict__ q,
                                                             const int* __restrict__ len, int B, int T, int dm, int heads,
                                                             float scale, float* __restrict__ out, float* __restrict__ P,
                                                             const int* __restrict__ row_off) {
  __shared__ float s_att[4][XP_MAXL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.x * 4 + wave;
  if (bh >= B * heads) return;
  const int b = bh / heads, h = bh - b * heads;
  const int dk = dm / heads;
  const int n = min(max(len[b], 0), T);
  const float* qh = q + (size_t)b * dm + h * dk;
  const float* kb = kv + (row_off ? (size_t)row_off[b] : (size_t)b * T) * 2 * dm + h * dk;      // packed rows: see kernels.h
  const float* vb = kb + dm;
  float* att = s_att[wave];
  const int sub = lane & 15, grp = lane >> 4;
  for (int j0 = 0; j0 < n; j0 += 4) {
    const int j = j0 + grp;
    float s = 0.f;
    if (j < n) {
      for (int c = sub * 4; c < dk; c += 64) {
        const f32x4 kx = *reinterpret_cast<const f32x4*>(kb + (size_t)j * 2 * dm + c);
        const f32x4 qx = *reinterpret_cast<const f32x4*>(qh + c);
        s += kx[0] * qx[0] + kx[1] * qx[1] + kx[2] * qx[2] + kx[3] * qx[3];
      }
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    s += __shfl_xor(s, 8);
    if (j < n && sub == 0) att[j] = s * scale;
  }
  __builtin_amdgcn_wave_barrier();
  float mx = -INFINITY;
  for (int j = lane; j < n; j += 64) mx = fmaxf(mx, att[j]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < n; j += 64) sum += expf(att[j] - mx);
  sum = wave_sum(sum);
  const float inv = sum > 0.f ? 1.f / sum : 0.f;
  for (int j = lane; j < T; j += 64) {
    const float w = j < n ? expf(att[j] - mx) * inv : 0.f;
    att[j] = w;
    P[(size_t)bh * T + j] = w;
  }
  __builtin_amdgcn_wave_barrier();
  // out = sum_j w_j V_j: four rows per step (one per 16-lane group, 16-byte loads), the four partial sums joined at the end -- a lane per
  // column walking all rows one by one is a chain of n dependent loads (130 us per launch at histories of 200)
  for (int c = sub * 4; c < dk; c += 64) {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int j = grp; j < n; j += 4) acc += att[j] * *reinterpret_cast<const f32x4*>(vb + (size_t)j * 2 * dm + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      acc[k] += __shfl_xor(acc[k], 16);
      acc[k] += __shfl_xor(acc[k], 32);
    }
    if (grp == 0) *reinterpret_cast<f32x4*>(out + (size_t)b * dm + h * dk + c) = acc;
  }
}
int launch_attn_lastq_fwd(const float* kv, const float* q, const int* len, int B, int T, int dm, int heads, float* out,
                          float* P, hipStream_t st, const int* row_off) {
  if (B <= 0) return 0;
  INTEL_CHECK_ARG(T <= XP_MAXL, "attn_lastq: history length %d > %d unsupported", T, XP_MAXL);
  INTEL_CHECK_ARG(dm % heads == 0 && (dm / heads) % 4 == 0, "attn_lastq: head dim must be a multiple of 4");
  const float scale = 1.0f / sqrtf((float)(dm / heads));
  LAUNCH(attn_lastq_fwd_kernel, dim3(cdiv(B * heads, 4)), dim3(256), 0, st, kv, q, len, B, T, dm, heads, scale, out, P, row_off);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// backward: d_out [B, dm] -> dq [B, dm], dkv [B*T, 2*dm] (all T rows written; zero beyond len)
__global__ __launch_bounds__(256) void attn_lastq_bwd_kernel(const float* __restrict__ kv, const float* __restrict__ q,
                                                             const float* __restrict__ P, const float* __restrict__ d_out,
                                                             const int* __restrict__ len, int B, int T, int dm, int heads,
                                                             float scale, float* __restrict__ dq, float* __restrict__ dkv,
                                                             const int* __restrict__ row_off) {
  __shared__ float s_ds[4][XP_MAXL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.x * 4 + wave;
  if (bh >= B * heads) return;
  const int b = bh / heads, h = bh - b * heads;
  const int dk = dm / heads;
  const int n = min(max(len[b], 0), T);
  const float* qh = q + (size_t)b * dm + h * dk;
  const float* doh = d_out + (size_t)b * dm + h * dk;
  const size_t rbase = row_off ? (size_t)row_off[b] : (size_t)b * T;
  const int twr = row_off ? n : T;               // rows written: packed sessions own exactly n rows
  const float* kb = kv + rbase * 2 * dm + h * dk;
  const float* vb = kb + dm;
  const float* p = P + (size_t)bh * T;
  float* ds = s_ds[wave];
  const int sub = lane & 15, grp = lane >> 4;
  for (int j0 = 0; j0 < n; j0 += 4) {          // dP_j = <dO, V_j>
    const int j = j0 + grp;
    float s = 0.f;
    if (j < n) {
      for (int c = sub * 4; c < dk; c += 64) {
        const f32x4 vx = *reinterpret_cast<const f32x4*>(vb + (size_t)j * 2 * dm + c);
        const f32x4 gx = *reinterpret_cast<const f32x4*>(doh + c);
        s += vx[0] * gx[0] + vx[1] * gx[1] + vx[2] * gx[2] + vx[3] * gx[3];
      }
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    s += __shfl_xor(s, 8);
    if (j < n && sub == 0) ds[j] = s;
  }
  __builtin_amdgcn_wave_barrier();
  float dsum = 0.f;
  for (int j = lane; j < n; j += 64) dsum += p[j] * ds[j];
  dsum = wave_sum(dsum);
  __builtin_amdgcn_wave_barrier();
  for (int j = lane; j < T; j += 64) ds[j] = j < n ? p[j] * (ds[j] - dsum) * scale : 0.f;
  __builtin_amdgcn_wave_barrier();
  float* dkb = dkv + rbase * 2 * dm + h * dk;
  // dK_j = dS_j q, dV_j = p_j dO, dq = sum_j dS_j K_j: four rows per step (one per 16-lane group), 16-byte accesses (see the forward)
  for (int c = sub * 4; c < dk; c += 64) {
    const f32x4 qc = *reinterpret_cast<const f32x4*>(qh + c), gc = *reinterpret_cast<const f32x4*>(doh + c);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int j = grp; j < twr; j += 4) {
      const float dsj = ds[j];
      const float pj = j < n ? p[j] : 0.f;
      if (j < n) acc += dsj * *reinterpret_cast<const f32x4*>(kb + (size_t)j * 2 * dm + c);
      *reinterpret_cast<f32x4*>(dkb + (size_t)j * 2 * dm + c) = dsj * qc;
      *reinterpret_cast<f32x4*>(dkb + (size_t)j * 2 * dm + dm + c) = pj * gc;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      acc[k] += __shfl_xor(acc[k], 16);
      acc[k] += __shfl_xor(acc[k], 32);
    }
    if (grp == 0) *reinterpret_cast<f32x4*>(dq + (size_t)b * dm + h * dk + c) = acc;
  }
}
int launch_attn_lastq_bwd(const float* kv, const float* q, const float* P, const float* d_out, const int* len, int B, int T,
                          int dm, int heads, float* dq, float* dkv, hipStream_t st, const int* row_off) {
  if (B <= 0) return 0;
  INTEL_CHECK_ARG(T <= XP_MAXL, "attn_lastq_bwd: history length %d > %d unsupported", T, XP_MAXL);
  const float scale = 1.0f / sqrtf((float)(dm / heads));
  LAUNCH(attn_lastq_bwd_kernel, dim3(cdiv(B * heads, 4)), dim3(256), 0, st, kv, q, P, d_out, len, B, T, dm, heads, scale, dq, dkv, row_off);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// dX[b*T + len_b - 1, :] += src[b, :]
__global__ void add_at_last_kernel(const float* __restrict__ src, int lds, int dm, const int* __restrict__ len, int B, int T,
                                   float* __restrict__ dX, const int* __restrict__ row_off) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * dm) return;
  const int b = i / dm, c = i - b * dm;
  int t = len[b] - 1;
  t = t < 0 ? T + t : t;
  t = min(max(t, 0), T - 1);
  const size_t base = row_off ? (size_t)row_off[b] : (size_t)b * T;
  if (t < len[b]) dX[(base + t) * dm + c] += src[(size_t)b * lds + c];
}
int launch_add_at_last(const float* src, int lds, int dm, const int* len, int B, int T, float* dX, hipStream_t st, const int* row_off) {
  if (B * dm <= 0) return 0;
  LAUNCH(add_at_last_kernel, dim3(cdiv(B * dm, 256)), dim3(256), 0, st, src, lds, dm, len, B, T, dX, row_off);
  INTEL_CHECK_LAUNCH();
  return 0;
}
