#!/bin/bash
# Runs on the GPU box (through gpurun): GPU tests, the default bench line, rocprofv3 kernel stats (single stream and
# concurrent branches), the two PMC passes (each counter in its own run, kernel-trace only) and the other workloads.
# Everything lands under gpurun_out/<tag>/; copy what should be judged into profiles/ afterwards.
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1 < /dev/null
tail -2 $out/tests.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err < /dev/null
INTEL_STREAMS=0 INTEL_OVERLAP_TABLE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1s -- python3 bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_roofline --no_feed > $out/stats1s.log 2>&1 < /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_roofline --no_feed > $out/stats.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python3 bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline --no_feed > $out/fetch.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python3 bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline --no_feed > $out/write.log 2>&1 < /dev/null
timeout 600 python bench.py --workload lifedata --batch 2048 --no_cpu_baseline > $out/bench_lifedata.json 2>/dev/null < /dev/null
timeout 600 python bench.py --workload stress --batch 256 --steps 5 --warmup 2 --no_cpu_baseline > $out/bench_stress.json 2>/dev/null < /dev/null
timeout 600 python bench.py --loss IntListloss --cal_diversity 1 --no_cpu_baseline > $out/bench_pl.json 2>/dev/null < /dev/null
timeout 600 python bench.py --encoder GRU4Rec --no_cpu_baseline > $out/bench_gru.json 2>/dev/null < /dev/null
ls $out
