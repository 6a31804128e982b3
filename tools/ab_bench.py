"""Same-box A/B of bench.py under environment switches (box-to-box spread is +-3 %: compare within ONE gpurun call).
usage: python tools/ab_bench.py "<bench args>" VAR=a,b [VAR2=c,d ...]      e.g.  "--zipf 1" INTEL_SCATTER_SORTED=1,0"""
import itertools
import json
import os
import subprocess
import sys

args = sys.argv[1].split()
axes = [(a.split('=')[0], a.split('=')[1].split(',')) for a in sys.argv[2:]]
for combo in itertools.product(*[v for _, v in axes]):
    env = dict(os.environ)
    for (k, _), v in zip(axes, combo):
        env[k] = v
    r = subprocess.run([sys.executable, 'bench.py', '--no_cpu_baseline', '--no_feed', '--no_workloads', '--spread_blocks', '0', '--no_bf16_line'] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        top = list(d.get('kernel_ms_per_step', {}).items())[:6]
        print(' '.join('%s=%s' % (k, v) for (k, _), v in zip(axes, combo)), '| %.0f sessions/s  %.3f ms/step  eval %.0f  |' % (d['value'], d['ms_per_step'], d['eval_sessions_per_s']), top)
    except Exception as e:
        print(combo, 'ERR', e, r.stdout[-300:])
