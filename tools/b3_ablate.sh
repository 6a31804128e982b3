#!/bin/bash
# Builds ablated copies of libintel_hip.so (gemm.hip recompiled with -DB3_ABLATE=<bits>, the other objects reused from
# intel_sigir2023_amd/build/) into tools/ablate/, for tools/b3_ablate.py.  Run after the normal build.
set -e
cd "$(dirname "$0")/.."
OBJ=intel_sigir2023_amd/build
for bits in "$@"; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DB3_ABLATE=$bits -x hip -c intel_sigir2023_amd/csrc/gemm.hip -o tools/ablate/gemm_$bits.o &
done
wait
for bits in "$@"; do
  objs=$(ls $OBJ/*.o | grep -v gemm.hip.o)
  hipcc -shared -fPIC --offload-arch=gfx950 -o tools/ablate/libintel_hip_$bits.so $objs tools/ablate/gemm_$bits.o
done
ls -la tools/ablate/*.so
