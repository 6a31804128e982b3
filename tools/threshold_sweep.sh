#!/bin/bash
# Same-box sweeps that justify the batch thresholds left in the policy layer: every line is one bench.py run of 100 timed steps,
# the two settings of a threshold alternate three times at every batch size, all in ONE gpurun call (box-to-box spread is +-3 %).
#   1. backward chain launches (csrc/model.cpp head_fused_ok: <= 1024 sessions with 64/128-wide towers (the sweep under profiles/ was measured with the limit at 768),, <= 4096 with 32-wide towers + GRU4Rec,
#      any batch with 32-wide towers + BERT4Rec): INTEL_HEAD_FUSED=1 (the policy) against 2 (backward chains at any batch)
#   2. length-ordered GRU workgroups (default: from 1024 sessions): INTEL_GRU_ORDER_MIN_B=1 (always ordered) against 1000000 (never)
# usage (GPU box): bash tools/threshold_sweep.sh > gpurun_out/threshold_sweep.txt
run() {   # label, bench args, env assignment
  local out
  out=$(env $3 python bench.py --no_cpu_baseline --no_feed --no_workloads --spread_blocks 0 --no_bf16_line --no_roofline --eval_steps 0 --steps 100 --warmup 10 $2 2>/dev/null | tail -1)
  python - "$1" "$3" "$out" <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[3])
    print('%-38s %-30s %10.0f sessions/s  %.4f ms/step' % (sys.argv[1], sys.argv[2], d['value'], d['ms_per_step']))
except Exception as e:
    print(sys.argv[1], sys.argv[2], 'ERR', e)
PY
}
if [ "$1" != "gru" ]; then
echo "== backward chain launches: policy (INTEL_HEAD_FUSED=1) vs always (=2)"
for b in 768 1024 2048 3072 4096; do
  for rep in 1 2 3; do
    run "tmall (128/64-wide) B=$b" "--workload tmall --batch $b" INTEL_HEAD_FUSED=1
    run "tmall (128/64-wide) B=$b" "--workload tmall --batch $b" INTEL_HEAD_FUSED=2
  done
done
for b in 2048 4096 8192; do
  for rep in 1 2 3; do
    run "tmall_pub (32-wide, GRU4Rec) B=$b" "--workload tmall_pub --batch $b" INTEL_HEAD_FUSED=1
    run "tmall_pub (32-wide, GRU4Rec) B=$b" "--workload tmall_pub --batch $b" INTEL_HEAD_FUSED=2
  done
done
fi
echo "== length-ordered GRU workgroups (default: from 1024 sessions): always (INTEL_GRU_ORDER_MIN_B=1) vs never (=1000000)"
for b in 512 1024 2048; do
  for rep in 1 2 3; do
    run "tmall_pub B=$b" "--workload tmall_pub --batch $b" INTEL_GRU_ORDER_MIN_B=1000000
    run "tmall_pub B=$b" "--workload tmall_pub --batch $b" INTEL_GRU_ORDER_MIN_B=1
  done
done
