// Built-in per-kernel timing with HIP events on the launch stream (bench.py's `roofline` block).
// Disabled by default: a launch then costs one predictable branch.  When enabled every kernel launch
// is bracketed by two events recorded on ITS stream; intel_prof_collect() synchronises and aggregates
// by kernel name (launch count, total ms, algorithmic flops and bytes the launcher declared).
#pragma once
#include <hip/hip_runtime.h>

struct ProfScope {
  int idx;
  hipStream_t st;
  ProfScope(const char* name, hipStream_t s, double flops = 0.0, double bytes = 0.0);
  ProfScope(const char* name, hipStream_t s, double flops, double bytes, int m, int n, int k);   // shape-tagged
  ~ProfScope();
};
bool prof_enabled();

// Launch `kernel` with optional profiling.  WORK_FLOPS / WORK_BYTES: algorithmic work of this launch.
#define LAUNCH_W(WORK_FLOPS, WORK_BYTES, kernel, grid, block, smem, st, ...)        \
  do {                                                                              \
    ProfScope prof_scope__(#kernel, st, (double)(WORK_FLOPS), (double)(WORK_BYTES)); \
    hipLaunchKernelGGL(kernel, grid, block, smem, st, __VA_ARGS__);                 \
  } while (0)
// shape-tagged variant: records as "kernel[MxNxK]" so that bench.py can price each GEMM shape
#define LAUNCH_S(M_, N_, K_, WORK_FLOPS, WORK_BYTES, kernel, grid, block, smem, st, ...)              \
  do {                                                                                                \
    ProfScope prof_scope__(#kernel, st, (double)(WORK_FLOPS), (double)(WORK_BYTES), (M_), (N_), (K_)); \
    hipLaunchKernelGGL(kernel, grid, block, smem, st, __VA_ARGS__);                                   \
  } while (0)
#define LAUNCH(kernel, grid, block, smem, st, ...) LAUNCH_W(0.0, 0.0, kernel, grid, block, smem, st, __VA_ARGS__)
