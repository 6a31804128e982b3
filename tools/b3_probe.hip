// Accuracy probe for the three-plane bf16 split: one 16x16x32 tile, six plane products on v_mfma_f32_16x16x32_bf16, against an fp64
// reference and the fp32 fmaf chain.   hipcc -O3 --offload-arch=gfx950 tools/b3_probe.hip -o tools/b3_probe && tools/b3_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// split 8 floats into three bf16x8 planes
__device__ __forceinline__ void split8(const float* x, bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 hh = (__bf16)x[i];
    const float r1 = x[i] - (float)hh;
    const __bf16 mm = (__bf16)r1;
    const float r2 = r1 - (float)mm;
    h[i] = hh; m[i] = mm; l[i] = (__bf16)r2;
  }
}

// D[16x16] = W[16 n][32 k] * X[16 m][32 k]^T : D[n][m]; one wave
__global__ void tile_test(const float* W, const float* X, float* D) {
  const int lane = threadIdx.x, p = lane & 15, j = lane >> 4;
  float w[8], x[8];
  for (int e = 0; e < 8; ++e) { w[e] = W[p * 32 + 8 * j + e]; x[e] = X[p * 32 + 8 * j + e]; }
  bf16x8 wh, wm, wl, xh, xm, xl;
  split8(w, wh, wm, wl);
  split8(x, xh, xm, xl);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc, 0, 0, 0);
  // D row = 4j + r (n), col = p (m)
  for (int r = 0; r < 4; ++r) D[(4 * j + r) * 16 + p] = acc[r];
}

int main() {
  float hW[16 * 32], hX[16 * 32], hD[256];
  srand(1);
  for (int i = 0; i < 512; ++i) { hW[i] = (rand() / (float)RAND_MAX - 0.5f) * 3.f; hX[i] = (rand() / (float)RAND_MAX - 0.5f) * 7.f; }
  float *dW, *dX, *dD;
  hipMalloc(&dW, sizeof(hW)); hipMalloc(&dX, sizeof(hX)); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dW, hW, sizeof(hW), hipMemcpyHostToDevice); hipMemcpy(dX, hX, sizeof(hX), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(tile_test, dim3(1), dim3(64), 0, 0, dW, dX, dD);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  double maxrel = 0, maxrel32 = 0;
  for (int n = 0; n < 16; ++n) for (int m = 0; m < 16; ++m) {
    double ref = 0; float f32 = 0.f; double mag = 0;
    for (int k = 0; k < 32; ++k) { ref += (double)hW[n * 32 + k] * hX[m * 32 + k]; f32 = fmaf(hW[n * 32 + k], hX[m * 32 + k], f32); mag += fabs((double)hW[n * 32 + k] * hX[m * 32 + k]); }
    maxrel = fmax(maxrel, fabs(hD[n * 16 + m] - ref) / mag);
    maxrel32 = fmax(maxrel32, fabs(f32 - ref) / mag);
  }
  printf("bf16x6 max err / sum|terms| = %.3e   (fp32 fma chain: %.3e)\n", maxrel, maxrel32);
  return 0;
}
