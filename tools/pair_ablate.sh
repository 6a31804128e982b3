#!/bin/bash
# debug build (INTEL_DEBUG_BUILD=1): the one-pass linear backward with parts of its tile loop removed -- where do the cycles of a tile go?
for a in 0 1 2 3 4 8 16 24 28 27 31; do
  echo -n "ABL=$a  "; INTEL_PAIR_ABL=$a python tools/pair_bench.py 204800 2>&1 | grep "d=128 mask=1 kernels" | head -1
done
