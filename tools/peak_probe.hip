// Calibration probe (not part of the library): sustained engine clock under a dependent-FMA chain and the
// achievable v_mfma_f32_16x16x4_f32 rate with 1, 2 and 4 waves per SIMD and 1..8 independent accumulators.
//   hipcc -O3 --offload-arch=gfx950 tools/peak_probe.hip -o tools/peak_probe && tools/peak_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void fma_chain(float* out, int n) {
  float a = threadIdx.x * 1e-9f, b = 1.0000001f, c = 1e-9f;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 64; ++k) a = __builtin_fmaf(a, b, c);
  }
  if (a == 123.456f) out[0] = a;
}

template <int NACC>
__global__ void mfma_chain(float* out, int n) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-6f;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) out[0] = s;
}

// MFMAs fed from LDS (one ds_read_b128 per 4 MFMAs, like the GEMM / attention inner loops) and, optionally,
// a global stream: does the engine clock hold under a mixed load?  ticks = s_memtime delta of wave 0.
template <int STREAM>
__global__ void mfma_lds_mix(float* out, const float* __restrict__ gsrc, size_t gfloats, int n, long long* ticks) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 1e-6f;
  __syncthreads();
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lane = threadIdx.x & 63;
  float b = 1.0f + threadIdx.x * 1e-6f;
  f32x4 gsum = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4* gp = reinterpret_cast<const f32x4*>(gsrc);
  const size_t gv = gfloats / 4;
  size_t gi = ((size_t)blockIdx.x * 256 + threadIdx.x) % gv;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(lds + ((k * 1024 + lane * 4 + i * 16) & 8188));
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b, acc[j], 0, 0, 0);
    }
    if (STREAM) {
      gsum += gp[gi];
      gi += (size_t)gridDim.x * 256;
      if (gi >= gv) gi -= gv;
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = gsum[0] + gsum[1] + gsum[2] + gsum[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <typename F>
static float time_ms(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out;
  hipMalloc(&out, 64);
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  printf("CUs %d\n", cus);
  {
    const int n = 20000;
    float ms = time_ms([&] { hipLaunchKernelGGL(fma_chain, dim3(cus), dim3(256), 0, 0, out, n); });
    // one wave per SIMD: 64*n dependent FMAs, 4 cycles issue each (wave64 on a 16-lane SIMD) -- lower bound of the clock
    printf("fma chain: %.3f ms -> >= %.2f GHz if 4 cycles/op, %.2f GHz if 8 cycles/op\n", ms, 64.0 * n * 4 / (ms * 1e6), 64.0 * n * 8 / (ms * 1e6));
  }
  const int n = 4000;
#define RUN(NACC, WPS)                                                                                          \
  {                                                                                                             \
    float ms = time_ms([&] { hipLaunchKernelGGL(mfma_chain<NACC>, dim3(cus* WPS), dim3(256), 0, 0, out, n); });   \
    double fl = 2048.0 * 8 * NACC * n * 4.0 * cus * WPS;                                                        \
    printf("mfma16x16x4f32 acc=%d waves/SIMD=%d: %.3f ms  %.1f TF/s\n", NACC, WPS, ms, fl / (ms * 1e9));         \
  }
  RUN(1, 1) RUN(2, 1) RUN(4, 1) RUN(8, 1)
  RUN(1, 2) RUN(4, 2) RUN(8, 2)
  RUN(1, 4) RUN(4, 4)
  {
    long long* ticks;
    hipMalloc(&ticks, 64);
    float* big;
    const size_t gfloats = (size_t)256 << 20;   // 1 GiB
    hipMalloc(&big, gfloats * 4);
    hipMemset(big, 0, gfloats * 4);
    const int nn = 4000;
    for (int wps = 1; wps <= 2; ++wps) {
      long long ht = 0;
      float ms = time_ms([&] { hipLaunchKernelGGL(mfma_lds_mix<0>, dim3(cus * wps), dim3(256), 0, 0, out, big, gfloats, nn, ticks); });
      hipMemcpy(&ht, ticks, 8, hipMemcpyDeviceToHost);
      double fl = 2048.0 * 32 * nn * 4.0 * cus * wps;
      printf("mfma+lds  waves/SIMD=%d: %.3f ms %.1f TF/s; wave0 ticks %lld -> %.2f ticks/ns (32 MFMA = %.0f ticks)\n", wps, ms, fl / (ms * 1e9), ht, ht / (ms * 1e6), (double)ht / nn);
      ms = time_ms([&] { hipLaunchKernelGGL(mfma_lds_mix<1>, dim3(cus * wps), dim3(256), 0, 0, out, big, gfloats, nn, ticks); });
      hipMemcpy(&ht, ticks, 8, hipMemcpyDeviceToHost);
      double gb = 16.0 * 256 * nn * cus * wps;
      printf("mfma+lds+stream waves/SIMD=%d: %.3f ms %.1f TF/s %.2f TB/s; ticks/ns %.2f\n", wps, ms, fl / (ms * 1e9), gb / (ms * 1e9), ht / (ms * 1e6));
    }
  }
  return 0;
}
