"""Data-parallel engine on real kernels: two ranks (gloo, both on cuda:0 -- RCCL refuses two ranks per GPU and
the test box has one) on the two halves of a global batch must reproduce the single-process step on the whole
batch: same losses (mean of the shard losses) and same parameters after 2 steps."""
import os
import socket
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

from tests.helpers import Fixture, build_model

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, out_dir, exchange='auto'):
    os.environ['INTEL_DP_EXCHANGE'] = exchange
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), INTEL_DIST_BACKEND='gloo', INTEL_SINGLE_DEVICE='1')
    from intel_sigir2023_amd import parallel
    from intel_sigir2023_amd.engine import IntELEngine
    if world > 1:
        parallel.init_distributed()
    dev = torch.device('cuda:0')
    fx = Fixture('default')
    model, args = build_model(fx, dev)
    model.train()
    args.cal_diversity = 1
    eng = IntELEngine(model, 'IntBPRloss', args, lr=1e-3, l2=1e-4)
    parallel.broadcast_(eng.param_buckets())
    batch = fx.batch(dev)
    local = parallel.shard_batch(batch, rank, world)
    lo, hi = parallel.shard_range(batch['batch_size'], rank, world)
    losses = []
    for step in range(2):
        noise = torch.from_numpy(fx['adam/noise%d' % step]).to(dev)[lo:hi].contiguous()
        loss, _, _ = eng.train_step(local, noise=noise)
        losses.append(float(loss))
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.save({'sd': sd, 'losses': losses}, os.path.join(out_dir, 'w%d_r%d.pt' % (world, rank)))
    if world > 1:
        parallel.barrier()
        torch.distributed.destroy_process_group()


@pytest.mark.parametrize('world,exchange', [(2, 'dense'), (2, 'sparse'), (4, 'sparse')])
def test_n_rank_engine_equals_single_process(world, exchange):
    """exchange: dense all-reduce of the item-id table gradient, or the touched-rows all-gather.
    world 4 = one session per rank of the 4-session fixture batch."""
    assert torch.cuda.is_available()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_run, args=(1, _free_port(), d), nprocs=1, join=True)
        mp.spawn(_run, args=(world, _free_port(), d, exchange), nprocs=world, join=True)
        one = torch.load(os.path.join(d, 'w1_r0.pt'))
        ranks = [torch.load(os.path.join(d, 'w%d_r%d.pt' % (world, r))) for r in range(world)]
    for s in range(2):
        assert abs(sum(r['losses'][s] for r in ranks) / world - one['losses'][s]) < 2e-5
    for k, v in one['sd'].items():
        for r in ranks[1:]:
            assert torch.equal(ranks[0]['sd'][k], r['sd'][k]), 'replicas diverged: ' + k
        if 'k_linear.bias' in k:
            continue            # analytically-zero gradient: Adam direction is rounding noise
        err = float((ranks[0]['sd'][k] - v).abs().max())
        assert err < 5e-5, (k, err)
