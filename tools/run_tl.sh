export TMPDIR=/tmp
PMCARGS="--no_cpu_baseline --no_roofline --no_feed --no_bf16_line --no_workloads --spread_blocks 0"
mkdir -p gpurun_out/tl
INTEL_PAIR_BWD=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl/base -- python3 bench.py --steps 8 --warmup 3 --eval_steps 0 $PMCARGS > gpurun_out/tl/base.log 2>&1 < /dev/null
python3 tools/rocprof_timeline.py gpurun_out/tl/base 3 > gpurun_out/tl/base.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl/pair -- python3 bench.py --steps 8 --warmup 3 --eval_steps 0 $PMCARGS > gpurun_out/tl/pair.log 2>&1 < /dev/null
python3 tools/rocprof_timeline.py gpurun_out/tl/pair 3 > gpurun_out/tl/pair.txt 2>&1
find gpurun_out/tl -name "*.csv" -delete; find gpurun_out/tl -name "*.db" -delete
head -5 gpurun_out/tl/base.txt gpurun_out/tl/pair.txt
