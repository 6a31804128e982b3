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
    # INTEL_DP_FORCE=1: build the process group and take every collective branch even with ONE rank (each collective is
    # then an identity that still runs through RCCL): how the one-GPU test box executes the nccl code paths
    if (world > 1 or os.environ.get('INTEL_DP_FORCE') == '1') and not dist.is_initialized():
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


def active():
    """Are the data-parallel exchange steps to be executed?  More than one rank -- or a forced single-rank group."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get('INTEL_DP_FORCE') == '1')


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
    out.pop('his_rows', None)             # totals of the WHOLE batch: the shard runs its histories padded
    out.pop('hisitem_rows', None)
    out.pop('_intel', None)
    return out


def allreduce_sum_(tensors):
    """In-place sum over ranks of each flat gradient bucket (no-op for a single process)."""
    if not active():
        return
    for t in tensors:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def allreduce_max_(t):
    """In-place element-wise maximum over ranks (the OR of 0/1 byte flags)."""
    if active():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)


def allreduce_sum_async(t):
    """Start an in-place sum over ranks and return the work handle (None for a single process).  The collective
    is ordered after everything already enqueued on the current stream; ``handle.wait()`` orders the current
    stream after the collective."""
    if not active():
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)


def reduce_scatter_sum(full, out):
    """out <- this rank's chunk of the sum over ranks of ``full`` (world equal chunks of out.numel() elements, rank order).
    RCCL: one reduce-scatter (the first half of a ring all-reduce).  gloo (tests) has no reduce-scatter: all-reduce + slice."""
    w, r = world_size(), rank()
    n = out.numel()
    assert full.numel() == w * n
    if not active():
        out.copy_(full.reshape(-1)[:n])
    elif dist.get_backend() == 'nccl':
        dist.reduce_scatter_tensor(out, full)
    else:
        tmp = full.clone()          # the fallback must not clobber the caller's buffer (the RCCL branch does not either)
        dist.all_reduce(tmp, op=dist.ReduceOp.SUM)
        out.copy_(tmp.reshape(-1)[r * n:(r + 1) * n])


def allgather(t):
    """[world, *t.shape] tensor holding every rank's ``t`` (same shape on all ranks)."""
    w = world_size()
    if not active():
        return t.unsqueeze(0)
    out = torch.empty((w,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    if dist.get_backend() == 'nccl':
        dist.all_gather_into_tensor(out, t.contiguous())      # RCCL: one collective into the stacked buffer
    else:
        dist.all_gather(list(out.unbind(0)), t.contiguous())
    return out


def allgather_inplace(full, r, n):
    """full[k * n:(k + 1) * n] <- rank k's full[k * n:(k + 1) * n] for every k (this rank's own chunk is already in place): the second half of a
    ring all-reduce, written straight into the destination.  RCCL: one in-place all_gather_into_tensor; gloo (tests): all_gather into the chunks."""
    if not active():
        return
    w = world_size()
    assert full.numel() == w * n
    mine = full[r * n:(r + 1) * n]
    if dist.get_backend() == 'nccl':
        dist.all_gather_into_tensor(full, mine)
    else:
        dist.all_gather([full[k * n:(k + 1) * n] for k in range(w)], mine.clone())


def allgather_object(obj):
    """List of every rank's picklable ``obj`` in rank order (evaluation records; off the hot path)."""
    if world_size() == 1:
        return [obj]
    out = [None] * world_size()
    dist.all_gather_object(out, obj)
    return out


def global_max_(values, device):
    """Element-wise maximum over ranks of a short list of ints (the padded batch shape: pad rows are keys, SURVEY.md 0.5,
    so every rank must pad to the GLOBAL maximum of the step)."""
    if world_size() == 1:
        return [int(v) for v in values]
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [int(v) for v in t.cpu().tolist()]


def pad_batch_to(batch, L, H, Hi):
    """Pad a collated batch dict (BaseModel.py:121-142 layout) with the reference's padding values (ids / scores /
    labels 0: pad_sequence, BaseModel.py:133) up to list length L and history lengths H / Hi."""
    def pad(t, dim, n):
        if t.shape[dim] >= n:
            return t
        shape = list(t.shape)
        shape[dim] = n - t.shape[dim]
        return torch.cat([t, torch.zeros(shape, dtype=t.dtype, device=t.device)], dim=dim)
    out = dict(batch)
    for k in ('i_id_s', 'i_class_c', 'scores', 'ranking'):
        if k in out and torch.is_tensor(out[k]):
            out[k] = pad(out[k], 1, L)
    for k in ('his_context_mh', 'his_intents'):
        if k in out:
            out[k] = pad(out[k], 1, H)
    for k in ('his_item_id', 'his_item_int', 'his_item_idx'):
        if k in out:
            out[k] = pad(out[k], 1, Hi)
    return out


def broadcast_(tensors, src=0):
    if not active():
        return
    for t in tensors:
        dist.broadcast(t, src=src)


def barrier():
    if active():
        dist.barrier()


def allreduce_max_float(x, device):
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def backend_name():
    """'nccl' (= RCCL on ROCm), 'gloo' (CPU tests), or 'none' without a process group."""
    return dist.get_backend() if dist.is_initialized() else 'none'


def count_ranks():
    """How many ranks the collectives really span: an all-reduce of ones through the group (1 without a group).  bench.py refuses to print a line
    whose n_gpus the group does not confirm."""
    if not dist.is_initialized():
        return 1
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.ones(1, dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
