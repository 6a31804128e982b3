#!/bin/bash
# Runs on the GPU box (through gpurun): the default bench line, rocprofv3 kernel stats (single stream and concurrent
# branches), the PMC passes (each counter in its own run, kernel-trace only; training steps and evaluation steps
# separately) and the other workloads.  Everything lands under gpurun_out/<tag>/; tools/pmc_summary.py and a copy into
# profiles/ follow in the build container.
tag=${1:-r06}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err < /dev/null
PMCARGS="--no_cpu_baseline --no_roofline --no_feed --no_bf16_line --no_workloads --spread_blocks 0"
INTEL_STREAMS=0 INTEL_OVERLAP_TABLE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1s -- python3 bench.py --steps 10 --warmup 3 --eval_steps 0 $PMCARGS > $out/stats1s.log 2>&1 < /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 10 --warmup 3 --eval_steps 0 $PMCARGS > $out/stats.log 2>&1 < /dev/null
INTEL_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_eval -- python3 bench.py --steps 0 --warmup 0 --eval_steps 10 $PMCARGS > $out/stats_eval.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python3 bench.py --steps 3 --warmup 1 --eval_steps 0 $PMCARGS > $out/fetch.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python3 bench.py --steps 3 --warmup 1 --eval_steps 0 $PMCARGS > $out/write.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch_eval -- python3 bench.py --steps 0 --warmup 0 --eval_steps 4 $PMCARGS > $out/fetch_eval.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write_eval -- python3 bench.py --steps 0 --warmup 0 --eval_steps 4 $PMCARGS > $out/write_eval.log 2>&1 < /dev/null
# bf16 mode: its own bench line, kernel stats and traffic passes
timeout 600 python bench.py --dtype bf16 --no_cpu_baseline > $out/bench_bf16.json 2>/dev/null < /dev/null
INTEL_STREAMS=0 INTEL_OVERLAP_TABLE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1s_bf16 -- python3 bench.py --dtype bf16 --steps 10 --warmup 3 --eval_steps 0 $PMCARGS > $out/stats1s_bf16.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch_bf16 -- python3 bench.py --dtype bf16 --steps 3 --warmup 1 --eval_steps 0 $PMCARGS > $out/fetch_bf16.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write_bf16 -- python3 bench.py --dtype bf16 --steps 3 --warmup 1 --eval_steps 0 $PMCARGS > $out/write_bf16.log 2>&1 < /dev/null
# the lazy form of the table's Adam forced onto the headline shape (auto keeps the dense sweep there): traffic passes
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch_lazy -- python3 bench.py --adam lazy --steps 3 --warmup 1 --eval_steps 0 $PMCARGS > $out/fetch_lazy.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write_lazy -- python3 bench.py --adam lazy --steps 3 --warmup 1 --eval_steps 0 $PMCARGS > $out/write_lazy.log 2>&1 < /dev/null
# round 5: the one-kernel tower backward (tower_bwd.hip).  bf16 mode takes it by default, fp32 does not: same-box A/B lines of both modes, the fp32
# step's HBM traffic and kernel stats with the kernel FORCED on (what it saves in bytes and what it costs in time), the deferred table wait
timeout 600 python tools/ab_bench.py "--steps 100 --warmup 10 --no_roofline --dtype bf16" INTEL_FUSE_TOWER_BWD=a,0,a,0 > $out/ab_tower_bwd_bf16.txt 2>&1 < /dev/null
timeout 600 python tools/ab_bench.py "--steps 100 --warmup 10 --no_roofline" INTEL_FUSE_TOWER_BWD=a,1,a,1 > $out/ab_tower_bwd_f32.txt 2>&1 < /dev/null
INTEL_FUSE_TOWER_BWD=1 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch_fusedbwd -- python3 bench.py --steps 3 --warmup 1 --eval_steps 0 $PMCARGS > $out/fetch_fusedbwd.log 2>&1 < /dev/null
INTEL_FUSE_TOWER_BWD=1 timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write_fusedbwd -- python3 bench.py --steps 3 --warmup 1 --eval_steps 0 $PMCARGS > $out/write_fusedbwd.log 2>&1 < /dev/null
INTEL_FUSE_TOWER_BWD=1 INTEL_STREAMS=0 INTEL_OVERLAP_TABLE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1s_fusedbwd -- python3 bench.py --steps 10 --warmup 3 --eval_steps 0 $PMCARGS > $out/stats1s_fusedbwd.log 2>&1 < /dev/null
INTEL_FUSE_TOWER_BWD=1 timeout 300 python tools/step_timeline.py f32 train full > $out/timeline_f32_train_fusedbwd.txt 2>&1 < /dev/null
( for i in 1 2 3; do timeout 200 python bench.py --dtype bf16 --no_cpu_baseline --no_feed --no_workloads --spread_blocks 0 --no_roofline --steps 100 --warmup 10 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 next forward under the table sweep', d['value'], d['ms_per_step'])"; timeout 200 python bench.py --dtype bf16 --no_defer_table --no_cpu_baseline --no_feed --no_workloads --spread_blocks 0 --no_roofline --steps 100 --warmup 10 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 every step waits for the sweep    ', d['value'], d['ms_per_step'])"; done ) > $out/ab_defer_table.txt 2>&1 < /dev/null
# the step as it overlaps (HIP-event timeline, branches on their streams)
timeout 300 python tools/step_timeline.py f32 train full > $out/timeline_f32_train.txt 2>&1 < /dev/null
timeout 300 python tools/step_timeline.py bf16 train full > $out/timeline_bf16_train.txt 2>&1 < /dev/null
timeout 300 python tools/step_timeline.py f32 eval full > $out/timeline_f32_eval.txt 2>&1 < /dev/null
# round 4: the reference's published hyper-parameters (tmall_pub, batch 512) -- kernel stats, the step as it overlaps, the GPU-bound step time
# with every launch queued ahead, and the same-box A/B lines of the two kernel families that replaced the kernel-per-op pipeline there
INTEL_STREAMS=0 INTEL_OVERLAP_TABLE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1s_pub -- python3 bench.py --workload tmall_pub --steps 20 --warmup 5 --eval_steps 0 $PMCARGS > $out/stats1s_pub.log 2>&1 < /dev/null
timeout 300 python tools/step_timeline.py f32 train full tmall_pub 512 > $out/timeline_pub_f32_train.txt 2>&1 < /dev/null
timeout 300 python tools/step_timeline.py f32 eval full tmall_pub 512 > $out/timeline_pub_f32_eval.txt 2>&1 < /dev/null
INTEL_STREAMS=0 INTEL_OVERLAP_TABLE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1s_pub_mse -- python3 bench.py --workload tmall_pub_mse --loss IntMSEloss --steps 20 --warmup 5 --eval_steps 0 $PMCARGS > $out/stats1s_pub_mse.log 2>&1 < /dev/null
timeout 300 python tools/step_timeline.py f32 train full tmall_pub_mse 512 > $out/timeline_pub_mse_f32_train.txt 2>&1 < /dev/null
timeout 300 python tools/gpu_bound_probe.py tmall_pub 512 30 > $out/gpu_bound_pub.txt 2>&1 < /dev/null
timeout 300 python tools/gpu_bound_probe.py tmall 4096 20 > $out/gpu_bound_tmall.txt 2>&1 < /dev/null
timeout 900 python tools/ab_bench.py "--workload tmall_pub --steps 300 --warmup 30" INTEL_TOWER32=1,0 INTEL_HEAD_FUSED=1,0 > $out/ab_pub.txt 2>&1 < /dev/null
timeout 600 python tools/ab_bench.py "--workload tmall_pub_mse --loss IntMSEloss --steps 300 --warmup 30 --no_bf16_line" INTEL_ENC32=1,0,1,0 > $out/ab_mse_enc32.txt 2>&1 < /dev/null
# the general attention kernels on the bf16 pipe (attn_p3.hip) against the exact-fp32 ones: steps and kernels
timeout 600 python tools/ab_bench.py "--workload lifedata --batch 2048 --steps 30 --warmup 5 --no_bf16_line" INTEL_ATTN_P3=1,0,1,0 > $out/ab_attn_p3_lifedata.txt 2>&1 < /dev/null
timeout 600 python tools/ab_bench.py "--workload stress --batch 1024 --steps 12 --warmup 3 --no_bf16_line" INTEL_ATTN_P3=1,0,1,0 > $out/ab_attn_p3_stress.txt 2>&1 < /dev/null
( timeout 300 python tools/attn_bench.py long; INTEL_ATTN_P3=0 timeout 300 python tools/attn_bench.py long ) > $out/attn_bench_long.txt 2>&1 < /dev/null
# round 6: the one-pass linear backward (pair.hip), the table sweep behind the backward's last reduction, the in-kernel gather of the inference tower --
# same-box A/B lines (alternating, 100 steps), the kernel alone against the two it replaces, its SQ counters, and the step as the hardware queues ran it
timeout 600 python tools/ab_bench.py "--steps 100 --warmup 10 --no_roofline" INTEL_PAIR_BWD=a,0,a,0 > $out/ab_pair_bwd.txt 2>&1 < /dev/null
timeout 600 python tools/ab_bench.py "--steps 100 --warmup 10 --no_roofline" INTEL_TABLE_AFTER_FLUSH=1,0,1,0 > $out/ab_table_after_flush.txt 2>&1 < /dev/null
timeout 600 python tools/ab_bench.py "--steps 40 --warmup 5 --eval_steps 60" INTEL_TOWER_GATHER=1,0,1,0 > $out/ab_tower_gather.txt 2>&1 < /dev/null
timeout 300 python tools/pair_bench.py 204800 > $out/pair_bench.txt 2>&1 < /dev/null
timeout 600 tools/pmc_kernel.sh $tag/pair_pmc linear_bwd_pair -- python3 tools/pair_bench.py 204800 > $out/pair_pmc.txt 2>&1 < /dev/null
timeout 600 tools/pmc_kernel.sh $tag/qkv_pmc linear_bwd_qkv -- python3 tools/pair_bench.py 204800 > $out/qkv_pmc.txt 2>&1 < /dev/null
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py --steps 8 --warmup 3 --eval_steps 0 $PMCARGS > $out/trace.log 2>&1 < /dev/null
python3 tools/rocprof_timeline.py $out/trace 3 > $out/rocprof_timeline_f32_train.txt 2>&1
INTEL_PAIR_BWD=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/trace_nopair -- python3 bench.py --steps 8 --warmup 3 --eval_steps 0 $PMCARGS > $out/trace_nopair.log 2>&1 < /dev/null
python3 tools/rocprof_timeline.py $out/trace_nopair 3 > $out/rocprof_timeline_f32_train_nopair.txt 2>&1
rm -rf $out/trace $out/trace_nopair
if [ "$2" != "quick" ]; then
timeout 600 python bench.py --workload tmall_pub --steps 300 --warmup 30 --no_cpu_baseline > $out/bench_tmall_pub_long.json 2>/dev/null < /dev/null
timeout 600 python bench.py --workload tmall_pub_mse --loss IntMSEloss --steps 300 --warmup 30 --no_cpu_baseline --no_bf16_line --no_feed > $out/bench_tmall_pub_mse.json 2>/dev/null < /dev/null
INTEL_TOWER32=0 INTEL_HEAD_FUSED=0 timeout 600 python bench.py --workload tmall_pub_mse --loss IntMSEloss --steps 300 --warmup 30 --no_cpu_baseline --no_bf16_line --no_feed --no_roofline > $out/bench_tmall_pub_mse_kernel_per_op.json 2>/dev/null < /dev/null
timeout 600 python bench.py --workload lifedata --batch 2048 --no_cpu_baseline > $out/bench_lifedata.json 2>/dev/null < /dev/null
timeout 600 python bench.py --workload stress --batch 256 --steps 20 --warmup 3 --no_cpu_baseline > $out/bench_stress.json 2>/dev/null < /dev/null
timeout 600 python bench.py --workload stress --batch 1024 --steps 20 --warmup 3 --no_cpu_baseline > $out/bench_stress_b1024.json 2>/dev/null < /dev/null
timeout 600 python bench.py --workload stress --batch 256 --adam dense --steps 20 --warmup 3 --no_cpu_baseline --no_bf16_line > $out/bench_stress_dense.json 2>/dev/null < /dev/null
timeout 600 python bench.py --workload stress --batch 1024 --adam dense --steps 20 --warmup 3 --no_cpu_baseline --no_bf16_line > $out/bench_stress_b1024_dense.json 2>/dev/null < /dev/null
timeout 600 python bench.py --batch 1024 --adam dense --no_cpu_baseline --no_bf16_line --no_feed > $out/bench_tmall_b1024_dense.json 2>/dev/null < /dev/null
timeout 600 python bench.py --batch 1024 --no_cpu_baseline --no_bf16_line --no_feed > $out/bench_tmall_b1024.json 2>/dev/null < /dev/null
timeout 600 python bench.py --adam lazy --no_cpu_baseline --no_bf16_line --no_feed > $out/bench_tmall_lazy.json 2>/dev/null < /dev/null
timeout 600 python bench.py --loss IntListloss --cal_diversity 1 --no_cpu_baseline > $out/bench_pl_div.json 2>/dev/null < /dev/null
timeout 600 python bench.py --encoder GRU4Rec --no_cpu_baseline > $out/bench_gru4rec.json 2>/dev/null < /dev/null
timeout 600 python bench.py --workload tmall_pub --no_cpu_baseline > $out/bench_tmall_pub.json 2>/dev/null < /dev/null
INTEL_GRU_SEQ=0 timeout 600 python bench.py --encoder GRU4Rec --no_cpu_baseline --no_feed > $out/bench_gru4rec_steps.json 2>/dev/null < /dev/null
INTEL_GRU_SEQ=0 timeout 600 python bench.py --workload tmall_pub --no_cpu_baseline --no_feed > $out/bench_tmall_pub_steps.json 2>/dev/null < /dev/null
timeout 600 python bench.py --zipf 1 --no_cpu_baseline > $out/bench_zipf.json 2>/dev/null < /dev/null
INTEL_FUSE_TOWER=0 timeout 600 python bench.py --no_cpu_baseline --no_bf16_line > $out/bench_unfused.json 2>/dev/null < /dev/null
INTEL_BWD_SCHEDULE=phased timeout 600 python bench.py --no_cpu_baseline --no_bf16_line > $out/bench_phased.json 2>/dev/null < /dev/null
INTEL_ENC_FUSED=0 timeout 600 python bench.py --no_cpu_baseline > $out/bench_enc_unfused.json 2>/dev/null < /dev/null
INTEL_ENC_FUSED_BWD=0 timeout 600 python bench.py --no_cpu_baseline --no_bf16_line > $out/bench_enc_unfused_bwd.json 2>/dev/null < /dev/null
timeout 600 python bench.py --workload lifedata --batch 4096 --no_cpu_baseline > $out/bench_lifedata_b4096.json 2>/dev/null < /dev/null
for b in 8192 16384 32768; do timeout 600 python bench.py --batch $b --steps 10 --no_cpu_baseline --no_roofline --no_feed --nbatches 4 > $out/bench_b$b.json 2>/dev/null < /dev/null; done
fi
# keep the merge small: only the summaries travel back
find $out -name "*.csv" ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete 2>/dev/null
find $out -name "*.db" -delete 2>/dev/null
ls -R $out | head -60
