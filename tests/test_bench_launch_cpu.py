"""`python bench.py --gpus N` must produce an N-rank line on its own (the metric is quoted at 1/2/4/8 MI355X; the driver's scaling run may call
it without torch.distributed.run).  Without a GPU the launch path is exercised up to the first GPU call: --dry_launch stops once the process
group (gloo here, RCCL on the box) has confirmed its rank count."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = dict(os.environ)
    env.update({'INTEL_DIST_BACKEND': 'gloo', 'OMP_NUM_THREADS': '1'})
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          timeout=timeout, cwd=ROOT)


def test_gpus_2_launches_two_ranks_by_itself():
    r = _run(['--gpus', '2', '--dry_launch'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout          # rank 0's line only
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['rccl_ranks'] == 2 and d['backend'] == 'gloo' and d['dry_launch'] is True


def test_world_size_mismatch_is_an_error_not_a_one_gpu_line():
    r = _run(['--gpus', '4', '--dry_launch'], {'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert 'WORLD_SIZE=1' in r.stderr


def test_single_rank_dry_launch_stays_in_process():
    r = _run(['--dry_launch'])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['n_gpus'] == 1 and d['rccl_ranks'] == 1 and d['backend'] == 'none'
