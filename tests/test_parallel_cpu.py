"""Data-parallel scheme (intel_sigir2023_amd/parallel.py) over gloo, world_size 2, on CPU.

Compute = the oracle (the HIP kernels need a GPU); what is under test is the host-side sharding /
gradient exchange: N ranks on contiguous shards of a global batch, loss gradients scaled by 1/N, one
all-reduce(sum) per flat bucket and the same Adam update on every rank must reproduce the single-process
step on the whole batch."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import intel_oracle as O
from tests.helpers import Fixture


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _train(sd0, batch, cfg, noise, world, rank, steps, lr, l2):
    from intel_sigir2023_amd import parallel
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd0.items()}
    params = [(k, v) for k, v in sd.items() if v.requires_grad]
    opt = torch.optim.Adam(O.adam_groups(params, l2), lr=lr)
    local = parallel.shard_batch(batch, rank, world)
    lo, hi = parallel.shard_range(batch['batch_size'], rank, world)
    for s in range(steps):
        opt.zero_grad()
        out = O.forward(sd, local, cfg)
        loss, _, _ = O.int_bpr_loss(out, local, cfg, noise[s][lo:hi])
        (loss / world).backward()                                   # local mean * 1/world
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for _, p in params]
        flat = torch.cat([g.reshape(-1) for g in grads])
        parallel.allreduce_sum_([flat])
        off = 0
        for (_, p), g in zip(params, grads):
            p.grad = flat[off:off + g.numel()].view_as(p).clone()
            off += g.numel()
        opt.step()
    return {k: v.detach().clone() for k, v in sd.items()}


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from intel_sigir2023_amd import parallel
    torch.set_num_threads(1)
    r, w, _ = parallel.init_distributed(backend='gloo')
    assert (r, w) == (rank, world) and parallel.world_size() == world
    fx = Fixture('default')
    cfg = O.Config(**dict(fx.args, cal_diversity=1))
    batch = fx.batch()
    g = torch.Generator().manual_seed(7)
    B, L = batch['i_id_s'].shape
    noise = [torch.rand(B, L, L, generator=g) for _ in range(2)]
    sd = _train(fx.state_dict(), batch, cfg, noise, world, rank, 2, 1e-3, 1e-4)
    t = parallel.allreduce_max_float(float(rank), torch.device('cpu'))
    assert t == world - 1
    parallel.barrier()
    torch.save(sd, os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.destroy_process_group()


def test_two_rank_step_equals_single_process_step():
    fx = Fixture('default')
    cfg = O.Config(**dict(fx.args, cal_diversity=1))
    batch = fx.batch()
    g = torch.Generator().manual_seed(7)
    B, L = batch['i_id_s'].shape
    noise = [torch.rand(B, L, L, generator=g) for _ in range(2)]
    ref = _train(fx.state_dict(), batch, cfg, noise, 1, 0, 2, 1e-3, 1e-4)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        r0 = torch.load(os.path.join(d, 'rank0.pt'))
        r1 = torch.load(os.path.join(d, 'rank1.pt'))
    for k in ref:
        if not ref[k].is_floating_point():
            continue
        assert torch.equal(r0[k], r1[k]), 'replicas diverged: ' + k
        if 'k_linear.bias' in k:
            continue          # analytically zero gradient: Adam's direction is rounding noise (see test_oracle_golden)
        err = float((r0[k] - ref[k]).abs().max())
        assert err < 5e-5, (k, err)


def test_shard_helpers():
    from intel_sigir2023_amd import parallel
    assert parallel.shard_range(8, 1, 2) == (4, 8)
    with pytest.raises(ValueError):
        parallel.shard_range(7, 0, 2)
    b = {'batch_size': 4, 'x': torch.arange(8).view(4, 2), 'phase': 'train', 'w': torch.arange(3)}
    s = parallel.shard_batch(b, 1, 2)
    assert s['batch_size'] == 2 and torch.equal(s['x'], torch.tensor([[4, 5], [6, 7]])) and torch.equal(s['w'], b['w'])
    assert parallel.world_size() == 1 and parallel.rank() == 0
    t = torch.ones(3)
    parallel.allreduce_sum_([t])
    assert torch.equal(t, torch.ones(3))
