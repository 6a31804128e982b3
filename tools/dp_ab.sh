# same-box comparison of the backward schedules with the data-parallel exchange forced on (one-rank RCCL group)
cd /root/repo
run() { python bench.py --no_cpu_baseline --no_bf16_line --no_roofline --no_feed 2>gpurun_out/dp_err_$1.log >gpurun_out/dp_out_$1.log; python -c "import json; d=[json.loads(l) for l in open('gpurun_out/dp_out_$1.log') if l.startswith('{')][-1]; print(d['value'], d['ms_per_step'])" 2>/dev/null || { grep -v Warning gpurun_out/dp_err_$1.log | tail -25; tail -3 gpurun_out/dp_out_$1.log; }; }
n=0
for sch in wide phased wide; do for ex in sparse dense sharded; do
n=$((n+1)); echo "== $sch $ex"; INTEL_DP_FORCE=1 INTEL_DP_EXCHANGE=$ex INTEL_BWD_SCHEDULE=$sch run $n
done; done
echo "== plain"; run 0
