#!/bin/bash
# Builds ablated copies of libintel_hip.so (gemm.hip recompiled with -DWB_ABLATE=<bits>: the bf16-pipe weight-gradient kernel
# without its MFMAs / LDS staging / HBM loads) into tools/ablate/, for tools/wgrad_ablate.py.  Run after the normal build.
set -e
cd "$(dirname "$0")/.."
OBJ=intel_sigir2023_amd/build
mkdir -p tools/ablate
for bits in "$@"; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DWB_ABLATE=$bits -x hip -c intel_sigir2023_amd/csrc/gemm.hip -o tools/ablate/gemm_w$bits.o &
done
wait
for bits in "$@"; do
  objs=$(ls $OBJ/*.o | grep -v gemm.hip.o)
  hipcc -shared -fPIC --offload-arch=gfx950 -o tools/ablate/libintel_hip_w$bits.so $objs tools/ablate/gemm_w$bits.o
done
ls -la tools/ablate/*.so
