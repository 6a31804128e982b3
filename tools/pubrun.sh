# bench lines of the published hyper-parameters in both arithmetic modes (same box): bash tools/pubrun.sh
for w in tmall_pub tmall_pub_mse; do
  for dt in f32 bf16 f32 bf16; do
    python bench.py --workload $w --dtype $dt --steps 200 --warmup 20 --no_workloads --spread_blocks 0 --no_bf16_line --no_cpu_baseline --no_feed --no_roofline --eval_steps 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$w $dt', round(d['value']), d['ms_per_step'], 'eval', round(d['eval_sessions_per_s']))"
  done
done
