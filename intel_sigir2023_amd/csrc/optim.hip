// Fused Adam (dense, coupled L2) and on-device NDCG@k.
#include <stdio.h>

#include "../../include/intel_hip.h"
#include "kernels.h"
#include "session.h"

// torch.optim.Adam single-tensor update (helpers/BaseRunner.py:182-188 builds it with
// weight_decay = --l2 on non-bias parameters, models/BaseModel.py:53-62):
//   g += wd*p;  m = m + (g-m)*(1-b1);  v = b2*v + (1-b2)*g*g;
//   p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// Dense over the whole tensor: rows with zero gradient still decay.
// Pure HBM stream: 4 reads + 3 writes (+1 write when the gradient is cleared in the same pass).
struct AdamArgs {
  float* p; float* g; float* m; float* v; long long n;
  float step_size, beta1, beta2, eps, wd, inv_bc2_sqrt, grad_scale; int zero_grad;
  float omb1, omb2;      // 1 - beta1, 1 - beta2 as torch forms them: in DOUBLE from the decimal betas, then rounded to float (adam_scalars)
};

// The step's scalars the way torch.optim.Adam's single-tensor path forms them (python doubles: bias_correction = 1 - beta ** step, step_size =
// lr / bias_correction1, lerp weight 1 - beta1, addcmul value 1 - beta2; each becomes a float only where a tensor op consumes it).  The C ABI
// carries lr and the betas as floats: (double)0.999f is 0.99900001287..., and 1.f - 0.999f is 1.3e-5 off float(0.001) -- a systematic 6e-6 on
// sqrt(v) that tests/test_trajectory_gpu.py sees against the reference at t = 1.  The decimal value the caller meant is recovered by printing
// the float with 7 significant digits -- accepted ONLY when that decimal rounds back to the very float (true of every hyper-parameter written with <= 7
// digits); any other float (a scheduled lr with more digits) is taken as the double it is.  The last value per call site is cached: three
// snprintf / strtod pairs per Adam launch were measurable host time at the published batch of 512.
static double adam_decimal(float x) {
  static thread_local float last_x[4] = {0.f, 0.f, 0.f, 0.f};
  static thread_local double last_v[4] = {0.0, 0.0, 0.0, 0.0};
  static thread_local int next = 0;
  for (int i = 0; i < 4; ++i)
    if (last_x[i] == x && last_v[i] != 0.0) return last_v[i];
  char buf[48];
  snprintf(buf, sizeof(buf), "%.7g", (double)x);
  double v = strtod(buf, nullptr);
  if ((float)v != x) v = (double)x;
  last_x[next] = x;
  last_v[next] = v;
  next = (next + 1) & 3;
  return v;
}
static void adam_scalars(float lr, float beta1, float beta2, int step, float* step_size, float* inv_bc2_sqrt, float* omb1, float* omb2) {
  const double b1 = adam_decimal(beta1), b2 = adam_decimal(beta2);
  const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
  *step_size = (float)(adam_decimal(lr) / bc1);
  *inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  *omb1 = (float)(1.0 - b1);
  *omb2 = (float)(1.0 - b2);
}

// The roundings are pinned (explicit fused multiply-adds, no contraction left to the compiler): the dense sweeps and the lazy
// replay below must produce the same bits from the same inputs whatever code surrounds the inlined body.
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a) {
#pragma clang fp contract(off)
  g = __builtin_fmaf(g, a.grad_scale, a.wd * p);
  m = __builtin_fmaf(g - m, a.omb1, m);
  const float g2 = (a.omb2 * g) * g;
  v = __builtin_fmaf(v, a.beta2, g2);
  const float denom = __builtin_fmaf(sqrtf(v), a.inv_bc2_sqrt, a.eps);
  p = __builtin_fmaf(-a.step_size, m / denom, p);
}

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
  const long long n4 = a.n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 p = reinterpret_cast<f32x4*>(a.p)[i];
    f32x4 g = reinterpret_cast<f32x4*>(a.g)[i];
    f32x4 m = reinterpret_cast<f32x4*>(a.m)[i];
    f32x4 v = reinterpret_cast<f32x4*>(a.v)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float pk = p[k], mk = m[k], vk = v[k];
      adam_one(pk, g[k], mk, vk, a);
      p[k] = pk; m[k] = mk; v[k] = vk;
    }
    reinterpret_cast<f32x4*>(a.p)[i] = p;
    reinterpret_cast<f32x4*>(a.m)[i] = m;
    reinterpret_cast<f32x4*>(a.v)[i] = v;
    if (a.zero_grad) reinterpret_cast<f32x4*>(a.g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // tail
  const long long t0 = n4 << 2;
  for (long long i = t0 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
    float p = a.p[i], m = a.m[i], v = a.v[i];
    adam_one(p, a.g[i], m, v, a);
    a.p[i] = p; a.m[i] = m; a.v[i] = v;
    if (a.zero_grad) a.g[i] = 0.f;
  }
}

// The same update over a [rows, d] table whose gradient is all zero except in the rows flagged in row_flags (set by the
// embedding scatter-add): the gradient row is read, cleared and its flag reset only where the flag is set, elsewhere g = 0
// is used without touching memory -- identical arithmetic, 6 instead of 8 streams over an embedding table.
// d / 4 lanes per row (d in {16, 32, 64, 128, 256}), rows strided over the grid.
__global__ __launch_bounds__(256) void adam_rows_kernel(AdamArgs a, int d, unsigned char* __restrict__ row_flags) {
  const int lpr = d >> 2;                                   // lanes per row
  const long long rows = a.n / d;
  const int sub = threadIdx.x % lpr;
  const long long r0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / lpr;
  const long long stride = (long long)gridDim.x * blockDim.x / lpr;
  for (long long r = r0; r < rows; r += stride) {
    const long long i = r * lpr + sub;
    const bool hit = row_flags[r] != 0;
    f32x4 p = reinterpret_cast<f32x4*>(a.p)[i];
    f32x4 m = reinterpret_cast<f32x4*>(a.m)[i];
    f32x4 v = reinterpret_cast<f32x4*>(a.v)[i];
    f32x4 g = f32x4{0.f, 0.f, 0.f, 0.f};
    if (hit) g = reinterpret_cast<f32x4*>(a.g)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float pk = p[k], mk = m[k], vk = v[k];
      adam_one(pk, g[k], mk, vk, a);
      p[k] = pk; m[k] = mk; v[k] = vk;
    }
    reinterpret_cast<f32x4*>(a.p)[i] = p;
    reinterpret_cast<f32x4*>(a.m)[i] = m;
    reinterpret_cast<f32x4*>(a.v)[i] = v;
    if (hit) {
      reinterpret_cast<f32x4*>(a.g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (sub == 0) row_flags[r] = 0;
    }
  }
}

int launch_adam_rows(float* p, float* g, float* m, float* v, long long rows, int d, unsigned char* row_flags, float lr,
                     float beta1, float beta2, float eps, float wd, int step, float grad_scale, hipStream_t st) {
  if (rows <= 0) return 0;
  INTEL_CHECK_ARG(step >= 1, "adam: step must be >= 1");
  INTEL_CHECK_ARG(d == 16 || d == 32 || d == 64 || d == 128 || d == 256, "adam_rows: row width %d unsupported", d);
  INTEL_CHECK_ARG(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                    reinterpret_cast<uintptr_t>(v)) & 15) == 0, "adam: tensors must be 16-byte aligned");
  AdamArgs a;
  a.p = p; a.g = g; a.m = m; a.v = v; a.n = rows * d;
  adam_scalars(lr, beta1, beta2, step, &a.step_size, &a.inv_bc2_sqrt, &a.omb1, &a.omb2);
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = wd; a.grad_scale = grad_scale; a.zero_grad = 1;
  long long blocks = ((a.n >> 2) + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  LAUNCH_W(0.0, 24.0 * (double)a.n + (double)rows, adam_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, d, row_flags);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// two parameter groups (torch's "decay" / "no decay" groups differ in the weight decay only) in ONE launch: the update of a group is a
// pure function of its own elements, so the grid is simply cut in two
__global__ __launch_bounds__(256) void adam_pair_kernel(AdamArgs a, AdamArgs b, int blocks_a) {
  const bool first = (int)blockIdx.x < blocks_a;
  const AdamArgs& c = first ? a : b;
  const long long blk = first ? blockIdx.x : blockIdx.x - blocks_a;
  const long long nblk = first ? blocks_a : (long long)gridDim.x - blocks_a;
  const long long stride = nblk * blockDim.x;
  const long long n4 = c.n >> 2;
  for (long long i = blk * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 p = reinterpret_cast<f32x4*>(c.p)[i];
    f32x4 g = reinterpret_cast<f32x4*>(c.g)[i];
    f32x4 m = reinterpret_cast<f32x4*>(c.m)[i];
    f32x4 v = reinterpret_cast<f32x4*>(c.v)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float pk = p[k], mk = m[k], vk = v[k];
      adam_one(pk, g[k], mk, vk, c);
      p[k] = pk; m[k] = mk; v[k] = vk;
    }
    reinterpret_cast<f32x4*>(c.p)[i] = p;
    reinterpret_cast<f32x4*>(c.m)[i] = m;
    reinterpret_cast<f32x4*>(c.v)[i] = v;
    if (c.zero_grad) reinterpret_cast<f32x4*>(c.g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (long long i = (n4 << 2) + blk * blockDim.x + threadIdx.x; i < c.n; i += stride) {
    float p = c.p[i], m = c.m[i], v = c.v[i];
    adam_one(p, c.g[i], m, v, c);
    c.p[i] = p; c.m[i] = m; c.v[i] = v;
    if (c.zero_grad) c.g[i] = 0.f;
  }
}

static void adam_fill(AdamArgs& a, float* p, float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps, float wd,
                      int step, float grad_scale, int zero_grad) {
  a.p = p; a.g = g; a.m = m; a.v = v; a.n = n;
  adam_scalars(lr, beta1, beta2, step, &a.step_size, &a.inv_bc2_sqrt, &a.omb1, &a.omb2);
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = wd; a.grad_scale = grad_scale; a.zero_grad = zero_grad;
}

int launch_adam_pair(float* const* p, float* const* g, float* const* m, float* const* v, const long long* n, const float* wd, float lr, float beta1,
                     float beta2, float eps, int step, float grad_scale, int zero_grad, hipStream_t st) {
  INTEL_CHECK_ARG(step >= 1, "adam: step must be >= 1");
  if (n[0] <= 0 && n[1] <= 0) return 0;
  for (int k = 0; k < 2; ++k)
    INTEL_CHECK_ARG(n[k] <= 0 || ((reinterpret_cast<uintptr_t>(p[k]) | reinterpret_cast<uintptr_t>(g[k]) | reinterpret_cast<uintptr_t>(m[k]) |
                                   reinterpret_cast<uintptr_t>(v[k])) & 15) == 0, "adam: tensors must be 16-byte aligned");
  AdamArgs a, b;
  adam_fill(a, p[0], g[0], m[0], v[0], n[0] > 0 ? n[0] : 0, lr, beta1, beta2, eps, wd[0], step, grad_scale, zero_grad);
  adam_fill(b, p[1], g[1], m[1], v[1], n[1] > 0 ? n[1] : 0, lr, beta1, beta2, eps, wd[1], step, grad_scale, zero_grad);
  auto blocks_for = [](long long nn) { long long bl = ((nn >> 2) + 255) / 256; return (int)(bl < 1 ? 1 : (bl > 2048 ? 2048 : bl)); };
  const int ba = blocks_for(a.n), bb = blocks_for(b.n);
  LAUNCH_W(0.0, (zero_grad ? 32.0 : 28.0) * (double)(a.n + b.n), adam_pair_kernel, dim3((unsigned)(ba + bb)), dim3(256), 0, st, a, b, ba);
  INTEL_CHECK_LAUNCH();
  return 0;
}

int launch_adam(float* p, float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                float wd, int step, float grad_scale, int zero_grad, hipStream_t st) {
  if (n <= 0) return 0;
  INTEL_CHECK_ARG(step >= 1, "adam: step must be >= 1");
  INTEL_CHECK_ARG(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                    reinterpret_cast<uintptr_t>(v)) & 15) == 0, "adam: tensors must be 16-byte aligned");
  AdamArgs a;
  a.p = p; a.g = g; a.m = m; a.v = v; a.n = n;
  adam_scalars(lr, beta1, beta2, step, &a.step_size, &a.inv_bc2_sqrt, &a.omb1, &a.omb2);
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = wd; a.grad_scale = grad_scale; a.zero_grad = zero_grad;
  long long blocks = ((n >> 2) + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);     // grid-stride: 256 CUs x 8
  LAUNCH_W(0.0, (zero_grad ? 32.0 : 28.0) * (double)n, adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------
// Lazy form of the table's dense Adam.  A row whose gradient is zero still moves every step (moments decay, coupled L2), but its
// update reads nothing except its own (p, m, v) and the step's two scalars (lr / bc1, 1 / sqrt(bc2)): the row can be brought up
// to date LATER, by replaying the missed steps one after the other with exactly the arithmetic of the dense sweep (adam_one,
// g = 0).  last[r] = the step row r has been updated through; sched[s - base - 1] = the scalars of step s.  A step then
// touches only the rows that carry a gradient (flag sweep: replay, then the step with its gradient) and, ahead of a forward
// pass, the rows that pass is about to gather (id sweep); everything a reader can observe is bit-identical to the dense sweep.
// ------------------------------------------------------------------------------------------
struct LazyArgs {
  float* p; float* g; float* m; float* v; int* last; float* sched; unsigned char* flags;
  long long rows; int d, base, step;
  float beta1, beta2, eps, wd, grad_scale, zero, step_size, inv_bc2_sqrt, omb1, omb2;
};

__device__ __forceinline__ AdamArgs lazy_consts(const LazyArgs& a) {
  AdamArgs c;
  c.p = c.g = c.m = c.v = nullptr; c.n = 0; c.zero_grad = 0;
  c.beta1 = a.beta1; c.beta2 = a.beta2; c.eps = a.eps; c.wd = a.wd; c.grad_scale = a.grad_scale;
  c.step_size = a.step_size; c.inv_bc2_sqrt = a.inv_bc2_sqrt; c.omb1 = a.omb1; c.omb2 = a.omb2;
  return c;
}

// steps from+1 .. upto with g = 0 (a.zero: a run-time 0 -- the same instruction sequence as a gradient that happens to be 0)
__device__ __forceinline__ void lazy_replay(f32x4& p, f32x4& m, f32x4& v, int from, int upto, const LazyArgs& a, AdamArgs& c) {
  for (int s = from + 1; s <= upto; ++s) {
    // the schedule entries are read with device-scope (cache-bypassing) loads: an entry is written once, by one thread of its step's
    // kernel, and read from then on by kernels of other streams -- with plain loads single compute units were measured to return the
    // zero the buffer was created with (a forward pass then gathered rows one step stale: tests/test_fullsize_gpu.py, the stress table)
    float2 sc;
    sc.x = __hip_atomic_load(a.sched + 2 * (s - a.base - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sc.y = __hip_atomic_load(a.sched + 2 * (s - a.base - 1) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    c.step_size = sc.x; c.inv_bc2_sqrt = sc.y;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float pk = p[k], mk = m[k], vk = v[k];
      adam_one(pk, a.zero, mk, vk, c);
      p[k] = pk; m[k] = mk; v[k] = vk;
    }
  }
}

// the step `a.step` for the flagged rows (replay of what they missed first); rows without a flag are left for later
__global__ __launch_bounds__(256) void adam_lazy_rows_kernel(LazyArgs a) {
  const int lpr = a.d >> 2;
  const int sub = threadIdx.x % lpr;
  const long long r0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / lpr;
  const long long stride = (long long)gridDim.x * blockDim.x / lpr;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.sched[2 * (a.step - a.base - 1)] = a.step_size;
    a.sched[2 * (a.step - a.base - 1) + 1] = a.inv_bc2_sqrt;
  }
  AdamArgs c = lazy_consts(a);
  for (long long rb = r0; rb < a.rows; rb += 4 * stride) {
    unsigned char f[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long long r = rb + q * stride;
      f[q] = r < a.rows ? a.flags[r] : (unsigned char)0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (!f[q]) continue;
      const long long r = rb + q * stride;
      const long long i = r * lpr + sub;
      const int from = a.last[r];
      f32x4 p = reinterpret_cast<f32x4*>(a.p)[i];
      f32x4 m = reinterpret_cast<f32x4*>(a.m)[i];
      f32x4 v = reinterpret_cast<f32x4*>(a.v)[i];
      const f32x4 g = reinterpret_cast<f32x4*>(a.g)[i];
      lazy_replay(p, m, v, from, a.step - 1, a, c);
      c.step_size = a.step_size; c.inv_bc2_sqrt = a.inv_bc2_sqrt;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float pk = p[k], mk = m[k], vk = v[k];
        adam_one(pk, g[k], mk, vk, c);
        p[k] = pk; m[k] = mk; v[k] = vk;
      }
      reinterpret_cast<f32x4*>(a.p)[i] = p;
      reinterpret_cast<f32x4*>(a.m)[i] = m;
      reinterpret_cast<f32x4*>(a.v)[i] = v;
      reinterpret_cast<f32x4*>(a.g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (sub == 0) { a.last[r] = a.step; a.flags[r] = 0; }
    }
  }
}

// rows ids_a[...] and ids_b[...] brought up to step `upto`; an id may repeat: the occurrence that raises last[id] owns the row
__global__ __launch_bounds__(256) void adam_lazy_ids_kernel(LazyArgs a, const int* __restrict__ ids_a, long long n_a,
                                                            const int* __restrict__ ids_b, long long n_b, int upto) {
  const int lpr = a.d >> 2;
  const int sub = threadIdx.x % lpr;
  const long long j0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / lpr;
  const long long stride = (long long)gridDim.x * blockDim.x / lpr;
  const long long n = n_a + n_b;
  AdamArgs c = lazy_consts(a);
  // every lane group runs the same number of rounds (the claim is handed to the group's lanes by a shuffle)
  const long long rounds = (n + stride - 1) / stride;
  for (long long t = 0; t < rounds; ++t) {
    const long long j = j0 + t * stride;
    int id = -1;
    if (j < n) id = j < n_a ? ids_a[j] : ids_b[j - n_a];
    const bool valid = id >= 0 && (long long)id < a.rows;
    // a row that is up to date already (the pad id, popular items: thousands of occurrences) is left alone without an atomic.
    // ONE lane of the group reads last[id] and makes the claim; the group follows its decision (other groups may raise the entry
    // concurrently: lanes that looked for themselves could disagree and leave a row partially replayed).
    // Not to be run concurrently with lazy gathers of the same rows: the claim is visible before p / m / v are rewritten.
    int from = upto;
    if (valid && sub == 0 && a.last[id] < upto) from = atomicMax(&a.last[id], upto);
    from = __shfl(from, 0, lpr);
    if (!valid || from >= upto) continue;
    const long long i = (long long)id * lpr + sub;
    f32x4 p = reinterpret_cast<f32x4*>(a.p)[i];
    f32x4 m = reinterpret_cast<f32x4*>(a.m)[i];
    f32x4 v = reinterpret_cast<f32x4*>(a.v)[i];
    lazy_replay(p, m, v, from, upto, a, c);
    reinterpret_cast<f32x4*>(a.p)[i] = p;
    reinterpret_cast<f32x4*>(a.m)[i] = m;
    reinterpret_cast<f32x4*>(a.v)[i] = v;
  }
}

// dst[m, col0 : col0 + d] = the row idx[m] of the table AS OF step `upto` (+ the position row, as gather_rows_kernel): a row that
// is behind is replayed in registers from its stored (p, m, v); nothing is written back (the step's flag sweep does that), so
// repeated ids and concurrent gathers need no ordering
__global__ __launch_bounds__(256) void gather_rows_lazy_kernel(LazyArgs a, const int* __restrict__ idx, int M, float* __restrict__ dst,
                                                               int ldd, int col0, const float* __restrict__ pos,
                                                               const int* __restrict__ row_t, int upto) {
  const int d4 = a.d >> 2;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)M * d4) return;
  const int mrow = (int)(i / d4), c = (int)(i - (long long)mrow * d4);
  const int row = idx[mrow];
  f32x4 p = f32x4{0.f, 0.f, 0.f, 0.f};
  if (row >= 0) {
    // all four loads are issued together (most rows of a training batch are behind: waiting for last[row] before asking for the
    // moments would put a second memory latency in front of the replay)
    const long long e = (long long)row * d4 + c;
    p = reinterpret_cast<const f32x4*>(a.p)[e];
    f32x4 m = reinterpret_cast<const f32x4*>(a.m)[e];
    f32x4 v = reinterpret_cast<const f32x4*>(a.v)[e];
    const int from = a.last[row];
    AdamArgs cst = lazy_consts(a);
    lazy_replay(p, m, v, from, upto, a, cst);
  }
  if (pos) p += *reinterpret_cast<const f32x4*>(pos + (size_t)row_t[mrow] * ldd + col0 + c * 4);
  *reinterpret_cast<f32x4*>(dst + (size_t)mrow * ldd + col0 + c * 4) = p;
}

// every row brought up to step `upto` (state_dict, evaluation through another path, change of hyper-parameters, sched full)
__global__ __launch_bounds__(256) void adam_lazy_flush_kernel(LazyArgs a, int upto) {
  const int lpr = a.d >> 2;
  const int sub = threadIdx.x % lpr;
  const long long r0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / lpr;
  const long long stride = (long long)gridDim.x * blockDim.x / lpr;
  AdamArgs c = lazy_consts(a);
  for (long long rb = r0; rb < a.rows; rb += 4 * stride) {
    int l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long long r = rb + q * stride;
      l[q] = r < a.rows ? a.last[r] : upto;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (l[q] >= upto) continue;
      const long long r = rb + q * stride;
      const long long i = r * lpr + sub;
      f32x4 p = reinterpret_cast<f32x4*>(a.p)[i];
      f32x4 m = reinterpret_cast<f32x4*>(a.m)[i];
      f32x4 v = reinterpret_cast<f32x4*>(a.v)[i];
      lazy_replay(p, m, v, l[q], upto, a, c);
      reinterpret_cast<f32x4*>(a.p)[i] = p;
      reinterpret_cast<f32x4*>(a.m)[i] = m;
      reinterpret_cast<f32x4*>(a.v)[i] = v;
      if (sub == 0) a.last[r] = upto;
    }
  }
}

static int lazy_fill(LazyArgs& a, const IntelLazyTable& t) {
  INTEL_CHECK_ARG(t.p && t.m && t.v && t.last && t.sched, "adam_lazy: null tensor");
  INTEL_CHECK_ARG(t.d == 16 || t.d == 32 || t.d == 64 || t.d == 128 || t.d == 256, "adam_lazy: row width %d unsupported", t.d);
  INTEL_CHECK_ARG(((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.m) | reinterpret_cast<uintptr_t>(t.v)) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(t.sched) & 7) == 0, "adam_lazy: tensors must be 16-byte aligned");
  a.p = t.p; a.g = nullptr; a.m = t.m; a.v = t.v; a.last = t.last; a.sched = t.sched; a.flags = nullptr;
  a.rows = t.rows; a.d = t.d; a.base = t.base; a.step = 0;
  a.beta1 = t.beta1; a.beta2 = t.beta2; a.eps = t.eps; a.wd = t.weight_decay; a.grad_scale = 1.f; a.zero = 0.f;
  a.step_size = 0.f; a.inv_bc2_sqrt = 0.f;
  {
    float u0, u1;
    adam_scalars(1.f, t.beta1, t.beta2, 1, &u0, &u1, &a.omb1, &a.omb2);
  }
  return 0;
}

static unsigned lazy_grid(long long groups, int lpr) {
  long long blocks = (groups * lpr + 255) / 256;
  return (unsigned)(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks));
}

int launch_adam_lazy_step(const IntelLazyTable& t, float* g, unsigned char* row_flags, float lr, int step, hipStream_t st) {
  if (t.rows <= 0) return 0;
  LazyArgs a;
  if (int rc = lazy_fill(a, t)) return rc;
  INTEL_CHECK_ARG(g && row_flags && (reinterpret_cast<uintptr_t>(g) & 15) == 0, "adam_lazy: gradient / row flags missing or misaligned");
  INTEL_CHECK_ARG(step > t.base && step - t.base <= t.cap, "adam_lazy: step %d outside the schedule window (%d, %d]", step, t.base, t.base + t.cap);
  a.g = g; a.flags = row_flags; a.step = step;
  adam_scalars(lr, t.beta1, t.beta2, step, &a.step_size, &a.inv_bc2_sqrt, &a.omb1, &a.omb2);
  LAUNCH_W(0.0, (double)t.rows, adam_lazy_rows_kernel, dim3(lazy_grid(t.rows, t.d >> 2)), dim3(256), 0, st, a);
  INTEL_CHECK_LAUNCH();
  return 0;
}

int launch_adam_lazy_ids(const IntelLazyTable& t, const int* ids_a, long long n_a, const int* ids_b, long long n_b, int upto, hipStream_t st) {
  if (t.rows <= 0 || n_a + n_b <= 0 || upto <= t.base) return 0;
  LazyArgs a;
  if (int rc = lazy_fill(a, t)) return rc;
  INTEL_CHECK_ARG(upto - t.base <= t.cap, "adam_lazy: step %d outside the schedule window", upto);
  LAUNCH_W(0.0, 4.0 * (double)(n_a + n_b), adam_lazy_ids_kernel, dim3(lazy_grid(n_a + n_b, t.d >> 2)), dim3(256), 0, st, a, ids_a, n_a,
           ids_b, n_b, upto);
  INTEL_CHECK_LAUNCH();
  return 0;
}

int launch_gather_rows_lazy(const IntelLazyTable& t, int upto, const int* idx, int M, float* dst, int ldd, int col0, hipStream_t st,
                            const float* pos, const int* row_t) {
  if (!row_t) pos = nullptr;
  if (M <= 0) return 0;
  LazyArgs a;
  if (int rc = lazy_fill(a, t)) return rc;
  INTEL_CHECK_ARG(upto - t.base <= t.cap, "gather_rows_lazy: step %d outside the schedule window", upto);
  INTEL_CHECK_ARG((ldd % 4 == 0) && (col0 % 4 == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) &&
                  ((reinterpret_cast<uintptr_t>(pos) & 15) == 0), "gather_rows_lazy: destination / position rows must be 16-byte aligned");
  const long long n = (long long)M * (t.d / 4);
  LAUNCH_W(0.0, 16.0 * (double)M * t.d + 8.0 * M, gather_rows_lazy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, idx, M, dst,
           ldd, col0, pos, row_t, upto);
  INTEL_CHECK_LAUNCH();
  return 0;
}

int launch_adam_lazy_flush(const IntelLazyTable& t, int upto, hipStream_t st) {
  if (t.rows <= 0 || upto <= t.base) return 0;
  LazyArgs a;
  if (int rc = lazy_fill(a, t)) return rc;
  INTEL_CHECK_ARG(upto - t.base <= t.cap, "adam_lazy: step %d outside the schedule window", upto);
  LAUNCH_W(0.0, 4.0 * (double)t.rows, adam_lazy_flush_kernel, dim3(lazy_grid(t.rows, t.d >> 2)), dim3(256), 0, st, a, upto);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------
// NDCG@k, "All" key of BaseRunner.evaluate_method (helpers/BaseRunner.py:117-126):
//   width = max(L, k); predictions padded with 0, labels with 0 (negatives / pads / unlabelled);
//   gains are LINEAR in the label (3/2/1/0); ties in the prediction resolve to the lower label
//   first (what the reference's label-descending pre-sort + reversed argsort yields).
// One wave per session; top-k by k rounds of wave arg-max.
// ------------------------------------------------------------------------------------------
#define ND_MAXPL 8
__global__ __launch_bounds__(256) void ndcg_kernel(const float* __restrict__ ens, const int* __restrict__ ranking,
                                                   const int* __restrict__ slen, int B, int L, int k, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int len = min(slen[b], L);
  const int width = max(L, k);
  float pv[ND_MAXPL];
  int lb[ND_MAXPL];
  bool used[ND_MAXPL];
  int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // label histogram (labels 0..7; Tmall uses 0..3)
#pragma unroll
  for (int i = 0; i < ND_MAXPL; ++i) {
    const int l = lane + 64 * i;
    const bool in = l < len;
    pv[i] = in ? ens[(size_t)b * L + l] : 0.f;
    int r = in ? ranking[(size_t)b * L + l] : 0;
    r = r < 0 ? 0 : r;
    lb[i] = r;
    used[i] = !(l < width);
    if (l < width) {
#pragma unroll
      for (int q = 0; q < 8; ++q) cnt[q] += (min(r, 7) == q);
    }
  }
  double dcg = 0.0;
  for (int p = 0; p < k; ++p) {
    // best = max prediction; ties -> smaller label, then larger index (any order: equal gain)
    float bv = -INFINITY;
    int bl = 0x7fffffff, bi = -1;
#pragma unroll
    for (int i = 0; i < ND_MAXPL; ++i) {
      if (!used[i] && (pv[i] > bv || (pv[i] == bv && lb[i] < bl))) { bv = pv[i]; bl = lb[i]; bi = lane + 64 * i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o);
      const int ol = __shfl_xor(bl, o), oi = __shfl_xor(bi, o);
      const bool take = (oi >= 0) && (bi < 0 || ov > bv || (ov == bv && (ol < bl || (ol == bl && oi > bi))));
      if (take) { bv = ov; bl = ol; bi = oi; }
    }
    if (bi < 0) break;
#pragma unroll
    for (int i = 0; i < ND_MAXPL; ++i)
      if (lane + 64 * i == bi) used[i] = true;
    dcg += (double)bl / log2((double)p + 2.0);
  }
  // ideal: labels sorted descending
#pragma unroll
  for (int q = 0; q < 8; ++q) cnt[q] = wave_sum_i(cnt[q]);
  double idcg = 0.0;
  {
    int p = 0;
#pragma unroll
    for (int q = 7; q >= 1; --q)
      for (int c = 0; c < cnt[q] && p < k; ++c, ++p) idcg += (double)q / log2((double)p + 2.0);
  }
  if (lane == 0) out[b] = (float)(dcg / idcg);    // 0/0 -> NaN like the reference
}

int launch_ndcg(int B, int L, int k, const float* ens, const int* ranking, const int* slen, float* out, hipStream_t st) {
  if (B <= 0) return 0;
  INTEL_CHECK_ARG(max(L, k) <= 64 * ND_MAXPL, "ndcg: list length %d > %d unsupported", L, 64 * ND_MAXPL);
  INTEL_CHECK_ARG(k >= 1 && k <= 64, "ndcg: k=%d unsupported", k);
  LAUNCH(ndcg_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, ens, ranking, slen, B, L, k, out);
  INTEL_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------
// evaluate_method (helpers/BaseRunner.py:56-131), every key, per session: per-behaviour HR@k / NDCG@k with binary
// relevance (:88-116) and the overall NDCG@k with linear gains (:117-126), for a list of cutoffs.
//
//   * the reference first sorts each list by LABEL, descending (:78-81); "positive for behaviour t" then means
//     "sits in the first all_pos_t slots of that order" (:93-94), all_pos = the pay / fav count, or the sum of all three
//     for click (:89-92).  label_pos[b,l] = the slot of item l in that order.  It depends on the labels only, i.e. on
//     the DATA: the host computes it once per evaluation set with the reference's own numpy call (whose order among
//     equal labels is numpy's) and passes it in; when it is NULL the kernel uses the stable form of the same sort
//     (reversed ascending order: among equal labels the LATER list position comes first).
//   * predictions are padded with 0 and labels with -2 up to `width` = max(longest list of the evaluation set,
//     largest cutoff) (:66-75): a session owns width - len pad slots, which outrank every item with a negative score.
//   * rank order = descending prediction; equal predictions resolve to the LATER slot of the label order first (what a
//     stable ascending argsort read from its end yields, :86 and :117); a pad is the last slot of all.
//   out[b] = { [behaviour pay,fav,click][cutoff][HR, NDCG] , [cutoff] overall NDCG }, doubles; valid[b][t] = all_pos_t > 0
//   (the reference averages a behaviour's keys over those sessions only, :96-98).
// One wave per session, kmax rounds of wave arg-max.
// ------------------------------------------------------------------------------------------
#define EM_MAXK 8
struct EvalArgs {
  const float* ens; const int* ranking; const int* slen; const int* pos_nums; const int* label_pos;
  int B, L, width, nk, kmax;
  int topk[EM_MAXK];
  double* out; unsigned char* valid;
};

__global__ __launch_bounds__(256) void eval_metrics_kernel(EvalArgs a) {
  __shared__ int s_lab[4][64 * ND_MAXPL];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wv;
  if (b >= a.B) return;
  const int L = a.L, len = min(a.slen[b], L);
  float pv[ND_MAXPL];
  int lp[ND_MAXPL], gain[ND_MAXPL];
  bool used[ND_MAXPL];
  int n3 = 0, n2 = 0, n1 = 0, nhi[5] = {0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < ND_MAXPL; ++i) {
    const int l = lane + 64 * i;
    const bool in = l < len;
    pv[i] = in ? a.ens[(size_t)b * L + l] : 0.f;
    const int raw = in ? a.ranking[(size_t)b * L + l] : -2;
    if (l < 64 * ND_MAXPL) s_lab[wv][l] = raw;
    gain[i] = raw < 0 ? 0 : raw;
    used[i] = !in;
    lp[i] = (in && a.label_pos) ? a.label_pos[(size_t)b * L + l] : 0;
    n3 += in && raw == 3; n2 += in && raw == 2; n1 += in && raw == 1;
#pragma unroll
    for (int q = 0; q < 5; ++q) nhi[q] += in && gain[i] == q + 3 + 1;      // labels 4..8 (not used by Tmall / LifeData)
  }
  n3 = wave_sum_i(n3); n2 = wave_sum_i(n2); n1 = wave_sum_i(n1);
#pragma unroll
  for (int q = 0; q < 5; ++q) nhi[q] = wave_sum_i(nhi[q]);
  if (!a.label_pos) {        // stable label-descending order, later list position first among equals
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < ND_MAXPL; ++i) {
      const int l = lane + 64 * i;
      if (l >= len) continue;
      const int mine = s_lab[wv][l];
      int pos = 0;
      for (int j = 0; j < len; ++j) {
        const int o = s_lab[wv][j];
        pos += (o > mine) || (o == mine && j > l);
      }
      lp[i] = pos;
    }
  }
  int allpos[3];
  if (a.pos_nums) {
    allpos[0] = a.pos_nums[(size_t)b * 3 + 0];
    allpos[1] = a.pos_nums[(size_t)b * 3 + 1];
    allpos[2] = allpos[0] + allpos[1] + a.pos_nums[(size_t)b * 3 + 2];
  } else {
    allpos[0] = n3; allpos[1] = n2; allpos[2] = n3 + n2 + n1;
  }
  double dcg_b[3][EM_MAXK], dcg_all[EM_MAXK];
  bool hr[3][EM_MAXK];
#pragma unroll
  for (int ki = 0; ki < EM_MAXK; ++ki) {
    dcg_all[ki] = 0.0;
#pragma unroll
    for (int t = 0; t < 3; ++t) { dcg_b[t][ki] = 0.0; hr[t][ki] = false; }
  }
  int pads_left = a.width - len;
  for (int p = 0; p < a.kmax; ++p) {
    float bv = -INFINITY;
    int bp = -1, bg = 0;
#pragma unroll
    for (int i = 0; i < ND_MAXPL; ++i)
      if (!used[i] && (pv[i] > bv || (pv[i] == bv && lp[i] > bp))) { bv = pv[i]; bp = lp[i]; bg = gain[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o);
      const int op = __shfl_xor(bp, o), og = __shfl_xor(bg, o);
      if (op >= 0 && (bp < 0 || ov > bv || (ov == bv && op > bp))) { bv = ov; bp = op; bg = og; }
    }
    const bool pad_wins = pads_left > 0 && (bp < 0 || bv <= 0.f);     // a pad scores 0 and sits behind every real slot
    if (pad_wins) {
      --pads_left;
      bp = 0x7fffffff; bg = 0;
    } else if (bp < 0) {
      break;                                                            // nothing left (width < cutoff cannot happen)
    } else {
#pragma unroll
      for (int i = 0; i < ND_MAXPL; ++i)
        if (!used[i] && lp[i] == bp) used[i] = true;                    // label_pos is a permutation: unique per item
    }
    const double disc = 1.0 / log2((double)p + 2.0);
#pragma unroll
    for (int ki = 0; ki < EM_MAXK; ++ki) {
      if (ki < a.nk && p < a.topk[ki]) {
        dcg_all[ki] += (double)bg * disc;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const bool hit = bp < allpos[t];
          hr[t][ki] = hr[t][ki] || hit;
          if (hit) dcg_b[t][ki] += disc;
        }
      }
    }
  }
  if (lane == 0) {
    const int nk = a.nk;
    double* o = a.out + (size_t)b * (7 * nk);
    for (int ki = 0; ki < nk; ++ki) {
      const int k = a.topk[ki];
      for (int t = 0; t < 3; ++t) {
        double idcg = 0.0;
        for (int r = 0; r < k && r < allpos[t]; ++r) idcg += 1.0 / log2((double)r + 2.0);
        o[(t * nk + ki) * 2 + 0] = hr[t][ki] ? 1.0 : 0.0;
        o[(t * nk + ki) * 2 + 1] = dcg_b[t][ki] / idcg;
      }
      double idcg = 0.0;
      int r = 0;
      for (int q = 4; q >= 0; --q)
        for (int c = 0; c < nhi[q] && r < k; ++c, ++r) idcg += (double)(q + 4) / log2((double)r + 2.0);
      for (int c = 0; c < n3 && r < k; ++c, ++r) idcg += 3.0 / log2((double)r + 2.0);
      for (int c = 0; c < n2 && r < k; ++c, ++r) idcg += 2.0 / log2((double)r + 2.0);
      for (int c = 0; c < n1 && r < k; ++c, ++r) idcg += 1.0 / log2((double)r + 2.0);
      o[6 * nk + ki] = dcg_all[ki] / idcg;      // 0/0 -> NaN like the reference
    }
    for (int t = 0; t < 3; ++t) a.valid[(size_t)b * 3 + t] = allpos[t] > 0;
  }
}

int launch_eval_metrics(int B, int L, int width, int nk, const int* topk, const float* ens, const int* ranking, const int* slen,
                        const int* pos_nums, const int* label_pos, double* out, unsigned char* valid, hipStream_t st) {
  if (B <= 0) return 0;
  INTEL_CHECK_ARG(nk >= 1 && nk <= EM_MAXK, "eval_metrics: %d cutoffs unsupported (1..%d)", nk, EM_MAXK);
  INTEL_CHECK_ARG(L <= 64 * ND_MAXPL, "eval_metrics: list length %d > %d unsupported", L, 64 * ND_MAXPL);
  EvalArgs a;
  a.ens = ens; a.ranking = ranking; a.slen = slen; a.pos_nums = pos_nums; a.label_pos = label_pos;
  a.B = B; a.L = L; a.nk = nk; a.out = out; a.valid = valid;
  a.kmax = 0;
  for (int i = 0; i < EM_MAXK; ++i) {
    a.topk[i] = i < nk ? topk[i] : 0;
    if (a.topk[i] > a.kmax) a.kmax = a.topk[i];
  }
  INTEL_CHECK_ARG(a.kmax >= 1 && a.kmax <= 64, "eval_metrics: cutoff %d unsupported", a.kmax);
  a.width = width > 0 ? width : max(L, a.kmax);
  INTEL_CHECK_ARG(a.width >= a.kmax && a.width >= L, "eval_metrics: width %d < max(list length %d, cutoff %d)", a.width, L, a.kmax);
  LAUNCH(eval_metrics_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, a);
  INTEL_CHECK_LAUNCH();
  return 0;
}
