for rep in 1 2; do
for m in dense lazy; do
for st in 20 100; do
python bench.py --adam $m --steps $st --warmup 5 --no_cpu_baseline --no_feed --no_workloads --spread_blocks 0 --no_bf16_line --no_roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('adam=$m steps=$st', d['value'], d['ms_per_step'])
"
done; done; done
