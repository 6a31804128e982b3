// Row-wise HBM-bound kernels: embedding row gather / scatter-add, LayerNorm fwd/bwd, row softmax,
// small reductions.  One wave per row wherever a row reduction is needed (wave shuffles only).
#include <cstdint>
#include <initializer_list>
#include "kernels.h"

int launch_slab_reduce(const float* slabs, size_t stride, int S, int rows, int cols, float* out, int ldo,
                       int accumulate, hipStream_t st);

// ------------------------------------------------------------------------------------------
// gather: dst[m, col0:col0+d] = table[idx[m]]   (nn.Embedding forward, IntEL.py:135,141,147-148,170-172)
// A row of d floats is read by d/4 consecutive lanes with 16-byte loads (d=64 -> 256 B per row).
// ------------------------------------------------------------------------------------------
// pos / row_t (optional): dst[m, col0 + c] += pos[row_t[m] * ldd + col0 + c] -- the BERT4Rec position embedding of packed history
// rows folded into the gather (pos rows have the width of dst rows)
__global__ void gather_rows_kernel(const float* __restrict__ table, int d, const int* __restrict__ idx, int M,
                                   float* __restrict__ dst, int ldd, int col0, int relu, const float* __restrict__ pos,
                                   const int* __restrict__ row_t) {
  const int d4 = d >> 2;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)M * d4) return;
  const int m = (int)(i / d4), c = (int)(i - (long long)m * d4);
  const int row = idx[m];
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
  if (row >= 0) v = *reinterpret_cast<const f32x4*>(table + (size_t)row * d + c * 4);
  if (relu) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
  }
  if (pos) v += *reinterpret_cast<const f32x4*>(pos + (size_t)row_t[m] * ldd + col0 + c * 4);
  *reinterpret_cast<f32x4*>(dst + (size_t)m * ldd + col0 + c * 4) = v;
}
__global__ void gather_rows_scalar_kernel(const float* __restrict__ table, int d, const int* __restrict__ idx, int M,
                                          float* __restrict__ dst, int ldd, int col0, int relu, const float* __restrict__ pos,
                                          const int* __restrict__ row_t) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)M * d) return;
  const int m = (int)(i / d), c = (int)(i - (long long)m * d);
  const int row = idx[m];
  float v = row >= 0 ? table[(size_t)row * d + c] : 0.f;
  if (relu) v = fmaxf(v, 0.f);
  if (pos) v += pos[(size_t)row_t[m] * ldd + col0 + c];
  dst[(size_t)m * ldd + col0 + c] = v;
}

int launch_gather_rows(const float* table, int d, const int* idx, int M, float* dst, int ldd, int col0, int relu,
                       hipStream_t st, const float* pos, const int* row_t) {
  if (!row_t) pos = nullptr;
  if (M <= 0 || d <= 0) return 0;
  const bool vec = (d % 4 == 0) && (ldd % 4 == 0) && (col0 % 4 == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(table) & 15) == 0) && ((reinterpret_cast<uintptr_t>(pos) & 15) == 0);
  if (vec) {
    long long n = (long long)M * (d / 4);
    LAUNCH_W(0.0, 8.0 * (double)M * d + 4.0 * M, gather_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, table, d, idx, M, dst, ldd, col0, relu, pos, row_t);
  } else {
    long long n = (long long)M * d;
    LAUNCH(gather_rows_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, table, d, idx, M, dst, ldd, col0, relu, pos, row_t);
  }
  INTEL_CHECK_LAUNCH();
  return 0;
}

__global__ void bcast_rows_kernel(const float* __restrict__ src, int lds, int d, int B, int T, float* __restrict__ dst,
                                  int ldd, int col0) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * T * d) return;
  const int c = (int)(i % d);
  const long long m = i / d;
  const int b = (int)(m / T);
  dst[(size_t)m * ldd + col0 + c] = src[(size_t)b * lds + c];
}
int launch_bcast_rows(const float* src, int lds, int d, int B, int T, float* dst, int ldd, int col0, hipStream_t st) {
  long long n = (long long)B * T * d;
  if (n <= 0) return 0;
  LAUNCH(bcast_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, lds, d, B, T, dst, ldd, col0);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------
// scatter-add: grad_table[idx[m]] += src[m, col0:col0+d]   (dense embedding backward)
// One lane per element: a wave adds 64 consecutive floats = whole 256-B rows, the shape global
// float atomics run fastest at (MI355X guide, "Global float atomics").
// ------------------------------------------------------------------------------------------
__global__ void scatter_add_rows_kernel(const float* __restrict__ src, int lds, int col0, int d,
                                        const int* __restrict__ idx, int M, float* __restrict__ grad_table,
                                        const float* __restrict__ relu_out, int ldr, int rcol0,
                                        unsigned char* __restrict__ row_flags) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)M * d) return;
  const int m = (int)(i / d), c = (int)(i - (long long)m * d);
  const int row = idx[m];
  if (row < 0) return;
  if (row_flags && c == 0) row_flags[row] = 1;       // "this gradient row is not all zero" for launch_adam_rows
  float v = src[(size_t)m * lds + col0 + c];
  if (relu_out && !(relu_out[(size_t)m * ldr + rcol0 + c] > 0.f)) v = 0.f;
  if (v != 0.f) atomicAdd(grad_table + (size_t)row * d + c, v);
}
int launch_scatter_add_rows(const float* src, int lds, int col0, int d, const int* idx, int M, float* grad_table,
                            const float* relu_out, int ldr, int rcol0, hipStream_t st, unsigned char* row_flags) {
  long long n = (long long)M * d;
  if (n <= 0) return 0;
  LAUNCH_W(0.0, 8.0 * (double)M * d + 4.0 * M, scatter_add_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, lds, col0, d, idx, M,
                     grad_table, relu_out, ldr, rcol0, row_flags);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// The same with the (id, source row) pairs SORTED by id (the caller sorts the batch's ids once per step, off the critical
// path): a group of d/4 lanes walks SS_CH consecutive pairs, sums runs of equal ids in registers and issues one row of float
// atomics per run.  Distinct ids cost what they cost above; a popular id (Zipf item popularity: the top item of a 286 720-row
// step is hit ~20 000 times) costs one atomic row per SS_CH hits instead of one per hit -- the serialised same-address
// atomics were 2.4 ms of a 6.6 ms step at Zipf(1.05).  row_off / len / T: the source rows are packed history rows
// (pair row r = b*T + t reads source row row_off[b] + t; positions t >= len[b] are padding and are skipped).
#define SS_CH 16
template <int LPR>      // lanes per row = d / 4
__global__ __launch_bounds__(256) void scatter_add_sorted_kernel(const float* __restrict__ src, int lds, int col0, int d,
                                                                 const int* __restrict__ sorted_ids, const int* __restrict__ sorted_rows,
                                                                 int n, float* __restrict__ grad_table, unsigned char* __restrict__ row_flags,
                                                                 const int* __restrict__ row_off, const int* __restrict__ len, int T) {
  // A workgroup owns 256 / LPR groups x SS_CH consecutive pairs.  Runs that lie inside one group's chunk are flushed by that group;
  // the first and the last run of every chunk may continue in the neighbouring chunks, so they meet in LDS and one group merges
  // them across the workgroup: a hot id costs one atomic row per WORKGROUP span (256 pairs at d = 64), not one per 16 pairs.
  constexpr int NG = 256 / LPR;
  __shared__ int s_id[NG][2];
  __shared__ __attribute__((aligned(16))) float s_sum[NG][2][LPR * 4];
  const int g = threadIdx.x / LPR, sub = threadIdx.x - g * LPR;
  const int e0 = (blockIdx.x * NG + g) * SS_CH;
  int ids[SS_CH];
  f32x4 val[SS_CH];
#pragma unroll
  for (int e = 0; e < SS_CH; ++e) {
    const int i = min(e0 + e, n - 1);
    int id = sorted_ids[i], r = sorted_rows[i];
    if (e0 + e >= n) id = -1;
    if (row_off && id >= 0) {
      const int b = r / T, t = r - b * T;
      if (t >= len[b]) id = -1;
      r = row_off[b] + t;
    }
    ids[e] = id;
    val[e] = id >= 0 ? *reinterpret_cast<const f32x4*>(src + (size_t)r * lds + col0 + 4 * sub) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto flush = [&](int id, const f32x4& acc) {
    if (id < 0) return;
    float* dst = grad_table + (size_t)id * d + 4 * sub;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (acc[c] != 0.f) atomicAdd(dst + c, acc[c]);
    if (row_flags && sub == 0) row_flags[id] = 1;
  };
  // runs of this chunk: the first goes to LDS slot 0, the last (if different from the first) to slot 1, the ones between are flushed
  int cur = -1, nruns = 0;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  int first_id = -1;
  f32x4 first_sum = acc;
#pragma unroll
  for (int e = 0; e < SS_CH; ++e) {
    if (ids[e] < 0) continue;
    if (ids[e] != cur) {
      if (nruns == 1) { first_id = cur; first_sum = acc; }
      else if (nruns > 1) flush(cur, acc);
      cur = ids[e];
      acc = val[e];
      ++nruns;
    } else {
      acc += val[e];
    }
  }
  if (nruns == 1) { first_id = cur; first_sum = acc; cur = -1; }      // a single run: it is both first and last
  if (sub == 0) { s_id[g][0] = first_id; s_id[g][1] = cur; }
  *reinterpret_cast<f32x4*>(&s_sum[g][0][4 * sub]) = first_sum;
  *reinterpret_cast<f32x4*>(&s_sum[g][1][4 * sub]) = acc;
  __syncthreads();
  if (g == 0) {               // merge the boundary runs of the workgroup in order
    int mid = -1;
    f32x4 macc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int gg = 0; gg < NG; ++gg)
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const int id = s_id[gg][w];
        if (id < 0) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(&s_sum[gg][w][4 * sub]);
        if (id == mid) {
          macc += v;
        } else {
          flush(mid, macc);
          mid = id;
          macc = v;
        }
      }
    flush(mid, macc);
  }
}
int launch_scatter_add_sorted(const float* src, int lds, int col0, int d, const int* sorted_ids, const int* sorted_rows, int n,
                              float* grad_table, hipStream_t st, unsigned char* row_flags, const int* row_off, const int* len, int T) {
  if (n <= 0) return 0;
  INTEL_CHECK_ARG(d == 16 || d == 32 || d == 64 || d == 128, "scatter_add_sorted: row width %d unsupported", d);
  INTEL_CHECK_ARG(((lds | col0) & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0, "scatter_add_sorted: unaligned source");
  const int lpr = d / 4;
  const int per_block = (256 / lpr) * SS_CH;            // pairs per workgroup
  const dim3 grid((unsigned)cdiv(n, per_block));
  const double bytes = 8.0 * (double)n * d + 8.0 * n;
  switch (lpr) {
    case 4: LAUNCH_W(0.0, bytes, scatter_add_sorted_kernel<4>, grid, dim3(256), 0, st, src, lds, col0, d, sorted_ids, sorted_rows, n, grad_table, row_flags, row_off, len, T); break;
    case 8: LAUNCH_W(0.0, bytes, scatter_add_sorted_kernel<8>, grid, dim3(256), 0, st, src, lds, col0, d, sorted_ids, sorted_rows, n, grad_table, row_flags, row_off, len, T); break;
    case 16: LAUNCH_W(0.0, bytes, scatter_add_sorted_kernel<16>, grid, dim3(256), 0, st, src, lds, col0, d, sorted_ids, sorted_rows, n, grad_table, row_flags, row_off, len, T); break;
    default: LAUNCH_W(0.0, bytes, scatter_add_sorted_kernel<32>, grid, dim3(256), 0, st, src, lds, col0, d, sorted_ids, sorted_rows, n, grad_table, row_flags, row_off, len, T); break;
  }
  INTEL_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------
// y = LayerNorm(x + r), eps = 1e-5, biased variance (torch.nn.LayerNorm).  One wave per row.
// ------------------------------------------------------------------------------------------
#define LN_MAXPL 8   // columns per lane: N <= 512
__global__ __launch_bounds__(256) void add_layernorm_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ r,
                                                            int ldr, int M, int N, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y, int ldy,
                                                            float* __restrict__ xhat, int ldxh, float* __restrict__ rstd,
                                                            const float* __restrict__ xscale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float v[LN_MAXPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXPL; ++i) {
    const int c = lane + 64 * i;
    float t = 0.f;
    if (c < N) {
      t = x[(size_t)row * ldx + c];
      if (xscale) t *= xscale[(size_t)row * ldx + c];          // dropout keep mask / (1 - p), same layout as x
      if (r) t += r[(size_t)row * ldr + c];
    }
    v[i] = t;
    s += t;
  }
  const float mean = wave_sum(s) / (float)N;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXPL; ++i) {
    const int c = lane + 64 * i;
    const float dlt = c < N ? v[i] - mean : 0.f;
    v[i] = dlt;
    q += dlt * dlt;
  }
  const float rs = 1.f / sqrtf(wave_sum(q) / (float)N + 1e-5f);
  if (rstd && lane == 0) rstd[row] = rs;
#pragma unroll
  for (int i = 0; i < LN_MAXPL; ++i) {
    const int c = lane + 64 * i;
    if (c < N) {
      const float xh = v[i] * rs;
      if (xhat) xhat[(size_t)row * ldxh + c] = xh;
      y[(size_t)row * ldy + c] = xh * gamma[c] + beta[c];
    }
  }
}
// 16 lanes per row, 4 rows per wave, 16-byte accesses: N = 64*NV exactly, all pitches multiples of 4, 16-byte aligned bases
static inline bool ln_v4_ok(int N, std::initializer_list<int> lds, std::initializer_list<const void*> ptrs) {
  if (N != 64 && N != 128 && N != 256) return false;
  for (int l : lds)
    if (l & 3) return false;
  for (const void* q : ptrs)
    if ((uintptr_t)q & 15) return false;
  return true;
}
__device__ __forceinline__ float sum16(float v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  return v;
}
template <int NV>
__global__ __launch_bounds__(256) void add_layernorm_v4_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ r,
                                                               int ldr, int M, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ y, int ldy,
                                                               float* __restrict__ xhat, int ldxh, float* __restrict__ rstd,
                                                               const float* __restrict__ xscale) {
  constexpr int N = 64 * NV;
  const int lane = threadIdx.x & 63, sub = lane & 15;
  const int row = blockIdx.x * 16 + (threadIdx.x >> 6) * 4 + (lane >> 4);
  const bool valid = row < M;
  const size_t rc = valid ? row : M - 1;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = sub * 4 + 64 * i;
    f32x4 t = *reinterpret_cast<const f32x4*>(x + rc * ldx + c);
    if (xscale) t *= *reinterpret_cast<const f32x4*>(xscale + rc * ldx + c);
    if (r) t += *reinterpret_cast<const f32x4*>(r + rc * ldr + c);
    v[i] = t;
    s += (t[0] + t[1]) + (t[2] + t[3]);
  }
  const float mean = sum16(s) / (float)N;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] -= mean;
    q += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
  }
  const float rs = 1.f / sqrtf(sum16(q) / (float)N + 1e-5f);
  if (!valid) return;
  if (rstd && sub == 0) rstd[row] = rs;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = sub * 4 + 64 * i;
    const f32x4 xh = v[i] * rs;
    if (xhat) *reinterpret_cast<f32x4*>(xhat + rc * ldxh + c) = xh;
    *reinterpret_cast<f32x4*>(y + rc * ldy + c) =
        xh * *reinterpret_cast<const f32x4*>(gamma + c) + *reinterpret_cast<const f32x4*>(beta + c);
  }
}
int launch_add_layernorm(const float* x, int ldx, const float* r, int ldr, int M, int N, const float* gamma,
                         const float* beta, float* y, int ldy, float* xhat, int ldxh, float* rstd, hipStream_t st,
                         const float* xscale) {
  if (M <= 0) return 0;
  INTEL_CHECK_ARG(N <= 64 * LN_MAXPL, "layernorm: N=%d > %d unsupported", N, 64 * LN_MAXPL);
  if (ln_v4_ok(N, {ldx, r ? ldr : 0, ldy, xhat ? ldxh : 0}, {x, r, y, xhat, gamma, beta, xscale})) {
#define LN_V4(NV)                                                                                                       \
  LAUNCH(add_layernorm_v4_kernel<NV>, dim3(cdiv(M, 16)), dim3(256), 0, st, x, ldx, r, ldr, M, gamma, beta, y, ldy, xhat, \
         ldxh, rstd, xscale)
    if (N == 64) LN_V4(1);
    else if (N == 128) LN_V4(2);
    else LN_V4(4);
#undef LN_V4
    INTEL_CHECK_LAUNCH();
    return 0;
  }
  LAUNCH(add_layernorm_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, x, ldx, r, ldr, M, N, gamma, beta, y, ldy, xhat,
                     ldxh, rstd, xscale);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------
// LayerNorm backward.  dz = rstd * (g - mean(g) - xhat*mean(g*xhat)), g = gamma*dy.
// dgamma/dbeta: per-block column partials (each block owns LNB_ROWS rows) -> slabs -> reduce.
// ------------------------------------------------------------------------------------------
#define LNB_ROWS 64
#define LNB_SMALL_M 16384          // at most this many rows: 16 rows per block so that the launch still covers the chip
static inline int ln_bwd_blocks(int M) { return M <= LNB_SMALL_M ? cdiv(M, 16) : cdiv(M, LNB_ROWS); }
size_t ln_bwd_slab_floats(int M, int N) { return (size_t)ln_bwd_blocks(M) * 2 * N; }

__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ xhat,
                                                            int ldxh, const float* __restrict__ rstd, int M, int N, int rows_per_block,
                                                            const float* __restrict__ gamma, float* __restrict__ dz, int lddz,
                                                            float* __restrict__ slabs) {
  __shared__ float sg[4][64 * LN_MAXPL];
  __shared__ float sb[4][64 * LN_MAXPL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ag[LN_MAXPL], ab[LN_MAXPL];
#pragma unroll
  for (int i = 0; i < LN_MAXPL; ++i) ag[i] = ab[i] = 0.f;
  const int r0 = blockIdx.x * rows_per_block;
  for (int rr = wave; rr < rows_per_block; rr += 4) {
    const int row = r0 + rr;
    if (row >= M) break;
    float g[LN_MAXPL], xh[LN_MAXPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXPL; ++i) {
      const int c = lane + 64 * i;
      float d = 0.f, h = 0.f, gm = 0.f;
      if (c < N) {
        d = dy[(size_t)row * lddy + c];
        h = xhat[(size_t)row * ldxh + c];
        gm = gamma[c];
      }
      ag[i] += d * h;
      ab[i] += d;
      g[i] = d * gm;
      xh[i] = h;
      s1 += g[i];
      s2 += g[i] * h;
    }
    const float m1 = wave_sum(s1) / (float)N, m2 = wave_sum(s2) / (float)N;
    const float rs = rstd[row];
#pragma unroll
    for (int i = 0; i < LN_MAXPL; ++i) {
      const int c = lane + 64 * i;
      if (c < N) dz[(size_t)row * lddz + c] = rs * (g[i] - m1 - xh[i] * m2);
    }
  }
#pragma unroll
  for (int i = 0; i < LN_MAXPL; ++i) {
    sg[wave][lane + 64 * i] = ag[i];
    sb[wave][lane + 64 * i] = ab[i];
  }
  __syncthreads();
  float* slab = slabs + (size_t)blockIdx.x * 2 * N;
  for (int c = threadIdx.x; c < N; c += 256) {
    slab[c] = (sg[0][c] + sg[1][c]) + (sg[2][c] + sg[3][c]);
    slab[N + c] = (sb[0][c] + sb[1][c]) + (sb[2][c] + sb[3][c]);
  }
}
// N <= 32 (the 32-wide towers of the published hyper-parameters): a row per HALF wave, two rows per wave and trip -- the kernel
// above keeps one row on 64 lanes, half of them idle, and walks its 16 rows one after the other (23 us for 25 600 rows)
__global__ __launch_bounds__(256) void layernorm_bwd_n32_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ xhat,
                                                                int ldxh, const float* __restrict__ rstd, int M, int N, int rows_per_block,
                                                                const float* __restrict__ gamma, float* __restrict__ dz, int lddz,
                                                                float* __restrict__ slabs) {
  __shared__ float sg[8][32];
  __shared__ float sb[8][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, half = lane >> 5;
  const bool on = c < N;
  const float gm = on ? gamma[c] : 0.f;
  float ag = 0.f, ab = 0.f;
  const int r0 = blockIdx.x * rows_per_block;
  for (int rr = 2 * wave + half; rr < rows_per_block; rr += 8) {
    const int row = r0 + rr;
    const bool live = row < M && on;
    const int rc = row < M ? row : M - 1;
    const float d = live ? dy[(size_t)rc * lddy + c] : 0.f;
    const float h = live ? xhat[(size_t)rc * ldxh + c] : 0.f;
    ag += d * h;
    ab += d;
    const float g = d * gm;
    float s1 = g, s2 = g * h;
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) {        // sums over the 32 lanes of this half
      s1 += __shfl_xor(s1, o);
      s2 += __shfl_xor(s2, o);
    }
    const float m1 = s1 / (float)N, m2 = s2 / (float)N;
    if (live) dz[(size_t)row * lddz + c] = rstd[rc] * (g - m1 - h * m2);
  }
  sg[2 * wave + half][c] = ag;
  sb[2 * wave + half][c] = ab;
  __syncthreads();
  float* slab = slabs + (size_t)blockIdx.x * 2 * N;
  if (threadIdx.x < N) {
    const int k = threadIdx.x;
    slab[k] = ((sg[0][k] + sg[1][k]) + (sg[2][k] + sg[3][k])) + ((sg[4][k] + sg[5][k]) + (sg[6][k] + sg[7][k]));
    slab[N + k] = ((sb[0][k] + sb[1][k]) + (sb[2][k] + sb[3][k])) + ((sb[4][k] + sb[5][k]) + (sb[6][k] + sb[7][k]));
  }
}
// same row mapping as add_layernorm_v4_kernel; TR trips of 16 rows per block
template <int NV, int TR>
__global__ __launch_bounds__(256) void layernorm_bwd_v4_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ xhat,
                                                               int ldxh, const float* __restrict__ rstd, int M,
                                                               const float* __restrict__ gamma, float* __restrict__ dz, int lddz,
                                                               float* __restrict__ slabs) {
  constexpr int N = 64 * NV;
  __shared__ float sg[4][N];
  __shared__ float sb[4][N];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane & 15, grp = lane >> 4;
  f32x4 gm[NV], ag[NV], ab[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    gm[i] = *reinterpret_cast<const f32x4*>(gamma + sub * 4 + 64 * i);
    ag[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    ab[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int r0 = blockIdx.x * (16 * TR) + wave * 4 + grp;
#pragma unroll
  for (int t = 0; t < TR; ++t) {
    const int row = r0 + 16 * t;
    const bool valid = row < M;
    const size_t rc = valid ? row : M - 1;
    f32x4 g[NV], h[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = sub * 4 + 64 * i;
      f32x4 d = *reinterpret_cast<const f32x4*>(dy + rc * lddy + c);
      h[i] = *reinterpret_cast<const f32x4*>(xhat + rc * ldxh + c);
      if (!valid) d = f32x4{0.f, 0.f, 0.f, 0.f};
      ag[i] += d * h[i];
      ab[i] += d;
      g[i] = d * gm[i];
      const f32x4 gh = g[i] * h[i];
      s1 += (g[i][0] + g[i][1]) + (g[i][2] + g[i][3]);
      s2 += (gh[0] + gh[1]) + (gh[2] + gh[3]);
    }
    const float m1 = sum16(s1) / (float)N, m2 = sum16(s2) / (float)N;
    const float rs = rstd[rc];
    if (valid) {
#pragma unroll
      for (int i = 0; i < NV; ++i)
        *reinterpret_cast<f32x4*>(dz + rc * lddz + sub * 4 + 64 * i) = rs * (g[i] - m1 - h[i] * m2);
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = ag[i][e], b = ab[i][e];
      a += __shfl_xor(a, 16);
      a += __shfl_xor(a, 32);
      b += __shfl_xor(b, 16);
      b += __shfl_xor(b, 32);
      ag[i][e] = a;
      ab[i][e] = b;
    }
    if (grp == 0) {
      *reinterpret_cast<f32x4*>(&sg[wave][sub * 4 + 64 * i]) = ag[i];
      *reinterpret_cast<f32x4*>(&sb[wave][sub * 4 + 64 * i]) = ab[i];
    }
  }
  __syncthreads();
  float* slab = slabs + (size_t)blockIdx.x * 2 * N;
  for (int c = threadIdx.x; c < N; c += 256) {
    slab[c] = (sg[0][c] + sg[1][c]) + (sg[2][c] + sg[3][c]);
    slab[N + c] = (sb[0][c] + sb[1][c]) + (sb[2][c] + sb[3][c]);
  }
}
// out1[c] (+)= sum_s slabs[s][c], out2[c] (+)= sum_s slabs[s][N + c]: LayerNorm dgamma / dbeta in one launch
__global__ __launch_bounds__(256) void slab_reduce_pair_kernel(const float* __restrict__ slabs, int S, int N,
                                                               float* __restrict__ out1, float* __restrict__ out2, int accumulate) {
  __shared__ float red[64][5];
  const int o = threadIdx.x & 3, q = threadIdx.x >> 2;
  const int i = blockIdx.x * 4 + o;          // 0 .. 2N-1
  float s0 = 0.f, s1 = 0.f;
  if (i < 2 * N) {
    int s = q;
    for (; s + 64 < S; s += 128) {
      s0 += slabs[(size_t)s * 2 * N + i];
      s1 += slabs[(size_t)(s + 64) * 2 * N + i];
    }
    for (; s < S; s += 64) s0 += slabs[(size_t)s * 2 * N + i];
  }
  red[q][o] = s0 + s1;
  __syncthreads();
  if (q == 0 && i < 2 * N) {
    float acc = 0.f;
    for (int k = 0; k < 64; ++k) acc += red[k][o];
    float* dst = i < N ? out1 + i : out2 + (i - N);
    *dst = accumulate ? (*dst + acc) : acc;
  }
}

int launch_layernorm_bwd(const float* dy, int lddy, const float* xhat, int ldxh, const float* rstd, int M, int N,
                         const float* gamma, float* dz, int lddz, float* dgamma, float* dbeta, int accumulate,
                         float* slabs, hipStream_t st, ReduceQueue* q) {
  if (M <= 0) return 0;
  INTEL_CHECK_ARG(N <= 64 * LN_MAXPL, "layernorm_bwd: N=%d unsupported", N);
  const int nb = ln_bwd_blocks(M);
  if (q) {
    slabs = redq_alloc(q, ln_bwd_slab_floats(M, N));
    if (!slabs) {
      intel_set_error("layernorm_bwd: reduction arena exhausted");
      return -2;   // INTEL_E_WORKSPACE
    }
  }
  const bool small = M <= LNB_SMALL_M;
  if (ln_v4_ok(N, {lddy, ldxh, lddz}, {dy, xhat, dz, gamma})) {
#define LNB_V4(NV)                                                                                                          \
  do {                                                                                                                      \
    if (small) LAUNCH((layernorm_bwd_v4_kernel<NV, 1>), dim3(nb), dim3(256), 0, st, dy, lddy, xhat, ldxh, rstd, M, gamma, dz, lddz, slabs); \
    else LAUNCH((layernorm_bwd_v4_kernel<NV, 4>), dim3(nb), dim3(256), 0, st, dy, lddy, xhat, ldxh, rstd, M, gamma, dz, lddz, slabs);     \
  } while (0)
    if (N == 64) LNB_V4(1);
    else if (N == 128) LNB_V4(2);
    else LNB_V4(4);
#undef LNB_V4
  } else {
    if (N <= 32)
      LAUNCH(layernorm_bwd_n32_kernel, dim3(nb), dim3(256), 0, st, dy, lddy, xhat, ldxh, rstd, M, N, small ? 16 : LNB_ROWS, gamma, dz, lddz, slabs);
    else
      LAUNCH(layernorm_bwd_kernel, dim3(nb), dim3(256), 0, st, dy, lddy, xhat, ldxh, rstd, M, N, small ? 16 : LNB_ROWS, gamma, dz, lddz, slabs);
  }
  INTEL_CHECK_LAUNCH();
  if (q) {
    redq_push(q, slabs, (size_t)2 * N, nb, 1, N, dgamma, N, accumulate);
    redq_push(q, slabs + N, (size_t)2 * N, nb, 1, N, dbeta, N, accumulate);
    return 0;
  }
  LAUNCH_W(0.0, 8.0 * (double)nb * N, slab_reduce_pair_kernel, dim3(cdiv(2 * N, 4)), dim3(256), 0, st, slabs, nb, N, dgamma, dbeta, accumulate);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------
// row softmax (pred_layer(...).softmax(-1), IntEL.py:153) and its backward.  One wave per row.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, int M, int N, float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (size_t)row * N;
  float mx = -INFINITY;
  for (int c = lane; c < N; c += 64) mx = fmaxf(mx, xr[c]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int c = lane; c < N; c += 64) s += expf(xr[c] - mx);
  s = wave_sum(s);
  const float inv = 1.f / s;
  for (int c = lane; c < N; c += 64) y[(size_t)row * N + c] = expf(xr[c] - mx) * inv;
}
int launch_softmax_rows(const float* x, int M, int N, float* y, hipStream_t st) {
  if (M <= 0) return 0;
  LAUNCH(softmax_rows_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, x, M, N, y);
  INTEL_CHECK_LAUNCH();
  return 0;
}
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, int M,
                                                               int N, float* __restrict__ dx) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float s = 0.f;
  for (int c = lane; c < N; c += 64) s += y[(size_t)row * N + c] * dy[(size_t)row * N + c];
  s = wave_sum(s);
  for (int c = lane; c < N; c += 64) dx[(size_t)row * N + c] = y[(size_t)row * N + c] * (dy[(size_t)row * N + c] - s);
}
int launch_softmax_rows_bwd(const float* y, const float* dy, int M, int N, float* dx, hipStream_t st) {
  if (M <= 0) return 0;
  LAUNCH(softmax_rows_bwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, y, dy, M, N, dx);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__global__ void add2_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, float* __restrict__ y) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] = a[i] + (b ? b[i] : 0.f);
}
int launch_add2(const float* a, const float* b, long long n, float* y, hipStream_t st) {
  if (n <= 0) return 0;
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  LAUNCH(add2_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, b, n, y);
  INTEL_CHECK_LAUNCH();
  return 0;
}
__global__ void fill_kernel(float* __restrict__ p, long long n, float v) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}
int launch_fill(float* p, long long n, float v, hipStream_t st) {
  if (n <= 0) return 0;
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  LAUNCH(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, n, v);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------
// nn.Dropout (IntEL.py:63,187,196): mask[i] = keep_i / (1 - p).  keep comes from `ext` (0/1 floats, parity tests
// feed the reference's own draw) or from a counter-based generator keyed by (seed, stream, i).
// ------------------------------------------------------------------------------------------
__global__ void dropout_mask_kernel(float* __restrict__ mask, long long n, float p, unsigned long long seed, unsigned stream_id,
                                    const float* __restrict__ ext) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float scale = 1.f / (1.f - p);
  float keep;
  if (ext) {
    keep = ext[i];
  } else {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)stream_id * 0x100000000ull + (unsigned long long)i + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);      // 24 uniform bits in [0, 1)
    keep = u >= p ? 1.f : 0.f;
  }
  mask[i] = keep * scale;
}
int launch_dropout_mask(float* mask, long long n, float p, unsigned long long seed, unsigned stream_id, const float* ext, hipStream_t st) {
  if (n <= 0) return 0;
  LAUNCH(dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mask, n, p, seed, stream_id, ext);
  INTEL_CHECK_LAUNCH();
  return 0;
}
__global__ void mul2_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, float* __restrict__ y) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const f32x4 u = *reinterpret_cast<const f32x4*>(a + i), v = *reinterpret_cast<const f32x4*>(b + i);
    *reinterpret_cast<f32x4*>(y + i) = u * v;
  } else {
    for (long long k = i; k < n; ++k) y[k] = a[k] * b[k];
  }
}
int launch_mul2(const float* a, const float* b, long long n, float* y, hipStream_t st) {
  if (n <= 0) return 0;
  LAUNCH(mul2_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, st, a, b, n, y);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------
// Data-parallel exchange of a gradient table's touched rows: rows_take moves rows idx[i] of
// the table into a dense buffer (and clears them in the table), rows_add adds a buffer of rows back.  Within one
// call idx holds no repeats (a rank's compacted, unique row list; -1 = padding), so there are no atomics and the
// sum over ranks is taken in rank order on every rank: replicas stay bit-identical.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rows_take_kernel(float* __restrict__ table, int d, const int* __restrict__ idx, int n,
                                                        float* __restrict__ out, int zero_rows) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int r = idx[i];
  for (int c = lane * 4; c < d; c += 256) {
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (r >= 0) {
      f32x4* src = reinterpret_cast<f32x4*>(table + (size_t)r * d + c);
      v = *src;
      if (zero_rows) *src = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    *reinterpret_cast<f32x4*>(out + (size_t)i * d + c) = v;
  }
}
__global__ __launch_bounds__(256) void rows_add_kernel(float* __restrict__ table, int d, const int* __restrict__ idx, int n,
                                                       const float* __restrict__ rows) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int r = idx[i];
  if (r < 0) return;
  for (int c = lane * 4; c < d; c += 256) {
    f32x4* dst = reinterpret_cast<f32x4*>(table + (size_t)r * d + c);
    *dst = *dst + *reinterpret_cast<const f32x4*>(rows + (size_t)i * d + c);
  }
}
// The touched rows of a table from the row marks the embedding scatter leaves (one byte per row, intel_set_iid_grad_row_flags): idx[0 .. count) = the
// marked rows in ascending order, idx[count .. cap) = -1.  Two launches over 4096-row chunks -- count (+ the -1 fill), then every chunk sums the
// counts in front of it and compacts its own marks in order; no sort, no host-sized result (the data-parallel exchange needs a static shape: cap =
// the batch's id count, the same on every rank).  Rows beyond cap (cannot happen for marks set by that batch) are dropped.
#define RC_CHUNK 4096
__global__ __launch_bounds__(256) void rows_compact_count_kernel(const unsigned char* __restrict__ flags, long long rows, int* __restrict__ counts,
                                                                 int* __restrict__ idx, int cap) {
  __shared__ int red[4];
  const long long r0 = (long long)blockIdx.x * RC_CHUNK + threadIdx.x * 16;
  int c = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) c += (r0 + k < rows && flags[r0 + k] != 0) ? 1 : 0;
  c = wave_sum_i(c);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < cap; i += gridDim.x * 256) idx[i] = -1;
}
__global__ __launch_bounds__(256) void rows_compact_write_kernel(const unsigned char* __restrict__ flags, long long rows, const int* __restrict__ counts,
                                                                 int* __restrict__ idx, int cap) {
  __shared__ int red[4], wbase[4], base_s;
  int before = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) before += counts[b];
  before = wave_sum_i(before);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = before;
  __syncthreads();
  if (threadIdx.x == 0) base_s = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  const long long r0 = (long long)blockIdx.x * RC_CHUNK + threadIdx.x * 16;
  unsigned m = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) m |= (r0 + k < rows && flags[r0 + k] != 0) ? (1u << k) : 0u;
  const int c = __popc(m);
  // exclusive scan of c over the 256 threads: inside the wave by shuffles, across the four waves through LDS
  int incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if ((int)(threadIdx.x & 63) >= o) incl += v;
  }
  if ((threadIdx.x & 63) == 63) wbase[threadIdx.x >> 6] = incl;
  __syncthreads();
  int pos = base_s + incl - c;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) pos += wbase[w];
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (m & (1u << k)) {
      if (pos < cap) idx[pos] = (int)(r0 + k);
      ++pos;
    }
}
size_t rows_compact_scratch_ints(long long rows) { return (size_t)((rows + RC_CHUNK - 1) / RC_CHUNK) + 1; }
int launch_rows_compact(const unsigned char* flags, long long rows, int* idx, int cap, int* scratch, hipStream_t st) {
  if (cap <= 0) return 0;
  INTEL_CHECK_ARG(rows > 0 && rows < (1LL << 31), "rows_compact: table of %lld rows unsupported", rows);
  const int nblk = (int)((rows + RC_CHUNK - 1) / RC_CHUNK);
  LAUNCH_W(0.0, (double)rows + 4.0 * cap, rows_compact_count_kernel, dim3(nblk), dim3(256), 0, st, flags, rows, scratch, idx, cap);
  INTEL_CHECK_LAUNCH();
  LAUNCH_W(0.0, (double)rows + 4.0 * cap, rows_compact_write_kernel, dim3(nblk), dim3(256), 0, st, flags, rows, scratch, idx, cap);
  INTEL_CHECK_LAUNCH();
  return 0;
}
// flags[idx[i]] = 1 for idx[i] >= 0 (the rows another rank's exchange buffer adds into: the table sweep must visit them too)
__global__ __launch_bounds__(256) void rows_mark_kernel(unsigned char* __restrict__ flags, const int* __restrict__ idx, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n && idx[i] >= 0) flags[idx[i]] = 1;
}
int launch_rows_mark(unsigned char* flags, const int* idx, int n, hipStream_t st) {
  if (n <= 0) return 0;
  LAUNCH(rows_mark_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, flags, idx, n);
  INTEL_CHECK_LAUNCH();
  return 0;
}

int launch_rows_take(float* table, int d, const int* idx, int n, float* out, int zero_rows, hipStream_t st) {
  if (n <= 0) return 0;
  INTEL_CHECK_ARG(d % 4 == 0, "rows_take: row width %d must be a multiple of 4", d);
  LAUNCH_W(0.0, 8.0 * (double)n * d, rows_take_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, table, d, idx, n, out, zero_rows);
  INTEL_CHECK_LAUNCH();
  return 0;
}
int launch_rows_add(float* table, int d, const int* idx, int n, const float* rows, hipStream_t st) {
  if (n <= 0) return 0;
  INTEL_CHECK_ARG(d % 4 == 0, "rows_add: row width %d must be a multiple of 4", d);
  LAUNCH_W(0.0, 12.0 * (double)n * d, rows_add_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, table, d, idx, n, rows);
  INTEL_CHECK_LAUNCH();
  return 0;
}
