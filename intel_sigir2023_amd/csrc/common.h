// Shared device/host helpers for the IntEL gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <math.h>
#include <stdlib.h>

#include "prof.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define INTEL_WAVE 64

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int rup(int a, int b) { return cdiv(a, b) * b; }
static inline size_t rup_sz(size_t a, size_t b) { return (a + b - 1) / b * b; }

// Sum / max over the 64 lanes of a wave (butterfly: every lane gets the result).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// D(16x16) += A(16x4) * B(4x16), exact fp32 (v_mfma_f32_16x16x4_f32).
// lane l supplies A[l&15][l>>4] and B[l>>4][l&15]; D[4*(l>>4)+r][l&15] is register r.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Dynamic LDS above the default per-kernel limit must be enabled explicitly (up to 160 KiB / CU).
template <typename F>
static inline void allow_lds(F* kernel, size_t bytes) {
  if (bytes > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// compute units of the current device (256 on MI355X); sizes the grids of the persistent kernels
static inline int num_cus() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
    return v;
  }();
  return n;
}

// Probe / ablation hooks (per-phase shader clocks of the fused kernels, the GEMM ablation bits, the small-M threshold) exist only in
// builds with -DINTEL_DEBUG (`INTEL_DEBUG_BUILD=1 python -m intel_sigir2023_amd.build`; tools/tower_probe.py, enc_probe.py,
// gemm_ablate.py): the product library reads none of these variables.
#ifdef INTEL_DEBUG
#define INTEL_DEBUG_ENV(name, dflt) ([] { const char* e__ = getenv(name); return e__ ? atoi(e__) : (dflt); }())
#else
#define INTEL_DEBUG_ENV(name, dflt) (dflt)
#endif

// ---- error reporting (host) ----------------------------------------------------------------
void intel_set_error(const char* fmt, ...);
#define INTEL_CHECK_ARG(cond, ...)        \
  do {                                    \
    if (!(cond)) {                        \
      intel_set_error(__VA_ARGS__);       \
      return -1;                          \
    }                                     \
  } while (0)
#define INTEL_CHECK_LAUNCH()                                                        \
  do {                                                                              \
    hipError_t e__ = hipGetLastError();                                             \
    if (e__ != hipSuccess) {                                                        \
      intel_set_error("%s:%d: launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return (int)e__;                                                              \
    }                                                                               \
  } while (0)
