export TMPDIR=/tmp
PMCARGS="--no_cpu_baseline --no_roofline --no_feed --no_bf16_line --no_workloads --spread_blocks 0"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kr -- python3 bench.py --steps 4 --warmup 2 --eval_steps 2 $PMCARGS > gpurun_out/kr.log 2>&1 < /dev/null
python3 tools/kernel_resources.py gpurun_out/kr > gpurun_out/kernel_resources.txt; rm -rf gpurun_out/kr; cat gpurun_out/kernel_resources.txt
