"""bench.py as the driver runs it: the default one-GPU line carries the contract's fields (`roofline`, `cpu_baseline`), and the N > 1 launch line
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) works end to end -- here with two ranks sharing the one test GPU over gloo
(INTEL_SINGLE_DEVICE / INTEL_DIST_BACKEND: RCCL refuses two ranks on one device; the RCCL collectives themselves run in tests/test_dp_gpu.py)."""
import json
import os
import subprocess
import sys

import pytest

from tests.helpers import ROOT

pytestmark = pytest.mark.gpu


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out[-2000:]            # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_default_line_has_the_contract_fields():
    # (--cpu_budget: the CPU baseline's bounded sample, 24 s by default -- thread sweep, B = 512 and B = 4096 -- is 6 s at one thread count and B = 512 here and the four short lines of the other workloads, the bf16-mode line and the feed measurement are left out -- the
    # line's contract fields are what is checked)
    r = subprocess.run([sys.executable, 'bench.py', '--steps', '4', '--warmup', '2', '--cpu_budget', '6', '--no_workloads', '--no_bf16_line', '--no_feed'], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _line(r.stdout)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in j, k
    assert j['n_gpus'] == 1 and j['steps'] == 4 and j['warmup'] == 2 and j['unit'] == 'sessions/s' and j['value'] > 0 and j['vs_baseline'] is None
    assert j['dtype'] == 'f32' and j['data'] == 'synthetic' and 'workload' in j['config'] and 'model' not in j['config']
    ro = j['roofline']
    assert ro['bound'] in ('hbm', 'mfma') and ro['unit'] in ('GB/s', 'TFLOP/s') and abs(ro['frac'] - ro['achieved'] / ro['peak']) < 1e-3 and 0 < ro['frac'] < 1
    assert ro['traffic'] is None or ro['traffic'] > 0
    cb = j['cpu_baseline']
    assert cb['kind'] == 'port' and cb['value'] > 0 and cb['cores'] >= 1 and cb['sample']
    assert j['value'] > 50 * cb['value']           # a sanity bound, not a target


def test_two_rank_launch_line_runs_end_to_end():
    # plain `python bench.py --gpus 2`: bench.py starts its own two ranks (torch.distributed.run as a child process), relays rank 0's line and checks
    # that the process group really spans two ranks (`rccl_ranks`)
    env = dict(os.environ, INTEL_SINGLE_DEVICE='1', INTEL_DIST_BACKEND='gloo')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, 'bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1', '--no_cpu_baseline', '--no_feed', '--no_bf16_line']
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _line(r.stdout)
    assert j['n_gpus'] == 2 and j['rccl_ranks'] == 2 and j['backend'] == 'gloo' and j['scaling'] == 'weak' and j['config']['global_batch'] == 2 * j['config']['per_gpu_batch'] and j['config']['parallelism'] == 'dp2'
    assert j['value'] > 0 and abs(j['value'] - j['config']['global_batch'] / (j['ms_per_step'] * 1e-3)) <= 1e-3 * j['value']
    # data parallel: the line says which exchange form ran and what it cost / how much of it stayed exposed (HIP events around every collective)
    ex = j['exchange']
    assert ex['form'] in ('dense', 'touched_rows', 'sharded') and ex['steps'] == 5
    assert ex['table_exchange_ms'] > 0 and ex['table_branch_ms'] >= ex['table_exchange_ms'] and ex['dense_buckets_allreduce_ms'] > 0
    assert 0 <= ex['exposed_ms'] <= 10 * j['ms_per_step']
