python bench.py --workload tmall_pub --steps 200 --warmup 20 --no_workloads --spread_blocks 0 --no_bf16_line > gpurun_out/gru_pub.json 2>gpurun_out/gru_pub.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/gru_pub.json"))
print(d["value"], d["ms_per_step"])
k=d["kernel_ms_per_step"]
for n,v in sorted(k.items(), key=lambda x:-x[1])[:8]: print("  ",n,v)
PY
