for g in 1 0; do
INTEL_TOWER_GATHER=$g python bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_feed --no_workloads --spread_blocks 0 --no_bf16_line --shapes 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('gather=$g eval', d['eval_sessions_per_s'])
for k,v in d['eval_shapes'].items(): print('   ',k,v)
"
done
