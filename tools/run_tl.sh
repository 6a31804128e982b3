# usage: bash tools/run_tl.sh <tag> [ENV=VAL ...]: one rocprofv3 kernel-trace of bench.py with the given environment, summarised by tools/rocprof_timeline.py
export TMPDIR=/tmp
tag=$1; shift
for kv in "$@"; do export "$kv"; done
PMCARGS="--no_cpu_baseline --no_roofline --no_feed --no_bf16_line --no_workloads --spread_blocks 0"
mkdir -p gpurun_out/tl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl/$tag -- python3 bench.py --steps 8 --warmup 3 --eval_steps 0 $PMCARGS > gpurun_out/tl/$tag.log 2>&1 < /dev/null
python3 tools/rocprof_timeline.py gpurun_out/tl/$tag 3 > gpurun_out/tl/$tag.txt 2>&1
rm -rf gpurun_out/tl/$tag
head -3 gpurun_out/tl/$tag.txt
