"""Data parallelism: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference is single-device; sessions are independent units and every loss is a
mean over the local batch (loss/BPRloss.py:33-34, Listloss.py:14-15, BaseIntloss.py:43,55), so the path
shards by splitting the global batch contiguously across ranks: each rank scales its loss gradients
by 1/world, gradients are summed with ONE all-reduce per flat bucket, and every rank applies the same
dense Adam update to its replica.  No other collective is on the data path.

Backend-agnostic on purpose (the CPU tests run it over gloo with the oracle as compute).
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            # INTEL_DIST_BACKEND=gloo lets several ranks share one GPU (RCCL refuses that): used by the
            # single-GPU test of the data-parallel path
            backend = os.environ.get('INTEL_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if os.environ.get('INTEL_SINGLE_DEVICE'):
            local_rank = 0
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def shard_range(n, r, w):
    """Contiguous shard [lo, hi) of n units for rank r of w (equal shards: n % w == 0 required for the
    1-rank == N-rank equivalence, because each rank's loss is a mean over its shard)."""
    if n % w != 0:
        raise ValueError('global batch %d is not divisible by world size %d' % (n, w))
    per = n // w
    return r * per, (r + 1) * per


def shard_batch(batch, r, w):
    """Rows [lo,hi) of every per-session tensor.  Padded lengths are kept (pad rows are real rows,
    SURVEY.md §0.5: every rank must see the GLOBAL max length)."""
    B = batch['batch_size']
    lo, hi = shard_range(B, r, w)
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == B:
            out[k] = v[lo:hi].contiguous()
        else:
            out[k] = v
    out['batch_size'] = hi - lo
    return out


def allreduce_sum_(tensors):
    """In-place sum over ranks of each flat gradient bucket (no-op for a single process)."""
    if world_size() == 1:
        return
    for t in tensors:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def allreduce_max_(t):
    """In-place element-wise maximum over ranks (the OR of 0/1 byte flags)."""
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)


def allreduce_sum_async(t):
    """Start an in-place sum over ranks and return the work handle (None for a single process).  The collective
    is ordered after everything already enqueued on the current stream; ``handle.wait()`` orders the current
    stream after the collective."""
    if world_size() == 1:
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)


def allgather(t):
    """[world, *t.shape] tensor holding every rank's ``t`` (same shape on all ranks)."""
    w = world_size()
    if w == 1:
        return t.unsqueeze(0)
    out = torch.empty((w,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    if dist.get_backend() == 'nccl':
        dist.all_gather_into_tensor(out, t.contiguous())      # RCCL: one collective into the stacked buffer
    else:
        dist.all_gather(list(out.unbind(0)), t.contiguous())
    return out


def broadcast_(tensors, src=0):
    if world_size() == 1:
        return
    for t in tensors:
        dist.broadcast(t, src=src)


def barrier():
    if world_size() > 1:
        dist.barrier()


def allreduce_max_float(x, device):
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
