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


def _run(rank, world, port, out_dir, exchange='auto', backend='gloo', seeded=False, overlap='1', tag='', schedule='wide', lazy='0'):
    """One rank of a 2-step training run on its shard of the 'default' fixture batch.  backend='nccl' with world 1 builds
    a REAL one-rank RCCL group (INTEL_DP_FORCE=1) so that every collective branch of the engine runs through RCCL.
    seeded: the BPR tie-breaking noise is drawn inside the loss kernel (common seed, counter keyed by the global session
    index) instead of passed as a tensor."""
    os.environ['INTEL_DP_EXCHANGE'] = exchange
    os.environ['INTEL_OVERLAP_TABLE'] = overlap
    os.environ['INTEL_ADAM_LAZY'] = lazy               # 1: the lazy form of the item-id table's Adam (engine.py)
    os.environ['INTEL_BWD_SCHEDULE'] = schedule       # wide: the one-call backward, the exchange under its tail; phased: two calls
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), INTEL_DIST_BACKEND=backend, INTEL_SINGLE_DEVICE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    if backend == 'nccl':
        os.environ['INTEL_DP_FORCE'] = '1'
    from intel_sigir2023_amd import parallel
    from intel_sigir2023_amd.engine import IntELEngine
    if world > 1 or backend == 'nccl':
        parallel.init_distributed()          # before any other GPU call of this process
        if backend == 'nccl':
            assert torch.distributed.get_backend() == 'nccl' and parallel.active()
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
    selects = []
    for step in range(2):
        if seeded:
            loss, _, _ = eng.train_step(local, noise_seed=1234567 + step)
            selects.append(eng._bufs['select'].cpu().clone())
        else:
            noise = torch.from_numpy(fx['adam/noise%d' % step]).to(dev)[lo:hi].contiguous()
            loss, _, _ = eng.train_step(local, noise=noise)
        losses.append(float(loss))
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.save({'sd': sd, 'losses': losses, 'selects': selects}, os.path.join(out_dir, 'w%d_r%d%s.pt' % (world, rank, tag)))
    if torch.distributed.is_initialized():
        parallel.barrier()
        torch.distributed.destroy_process_group()


def _run_tmall(rank, world, port, out_dir, exchange='auto', tag=''):
    """One rank of a 2-step run at the Tmall SHAPE (BASELINE.json configs[2]: list 50, K = 3, every embedding 64-d, BERT4Rec
    encoders on packed histories -- the fused encoder kernels, the one-kernel 64-wide tower) on a 20 000-item table: the global
    batch of 64 synthetic sessions, contiguous shards, BPR tie-breaks drawn in the kernel keyed by the global session index."""
    os.environ['INTEL_DP_EXCHANGE'] = exchange
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), INTEL_DIST_BACKEND='gloo', INTEL_SINGLE_DEVICE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    from intel_sigir2023_amd import parallel, synth
    from intel_sigir2023_amd.engine import IntELEngine
    from intel_sigir2023_amd.model import IntEL
    if world > 1:
        parallel.init_distributed()
    dev = torch.device('cuda:0')
    over = dict(items=20000, users=2000)
    torch.manual_seed(11)
    args = synth.make_args('tmall', dev, cal_diversity=1)
    corpus, _ = synth.make_corpus('tmall', **over)
    model = IntEL(args, corpus).to(dev)
    model.train()
    eng = IntELEngine(model, 'IntBPRloss', args, lr=1e-3, l2=1e-4)
    parallel.broadcast_(eng.param_buckets())
    losses = []
    for step in range(2):
        batch = synth.make_batch('tmall', 64, dev, seed=300 + step, ragged=True, corpus_over=over)
        local = parallel.shard_batch(batch, rank, world)
        if 'history_len' in local:          # host totals of the shard's packed histories (the producer of a shard knows them)
            local['his_rows'], local['hisitem_rows'] = int(local['history_len'].sum()), int(local['history_item_len'].sum())
        loss, _, _ = eng.train_step(local, noise_seed=4242 + step)
        losses.append(float(loss))
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.save({'sd': sd, 'losses': losses}, os.path.join(out_dir, 'tm_w%d_r%d%s.pt' % (world, rank, tag)))
    if torch.distributed.is_initialized():
        parallel.barrier()
        torch.distributed.destroy_process_group()


_BASE = {}


def _baseline(fn, *args):
    """The single-process run a data-parallel case is compared with, computed ONCE per configuration for the whole module (every case used to
    spawn its own copy: ~3 s of process start each, a quarter of this file's wall time)."""
    key = (fn.__name__,) + tuple(args)
    if key not in _BASE:
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(fn, args=(1, _free_port(), d) + tuple(args), nprocs=1, join=True)
            files = [x for x in os.listdir(d) if x.endswith('.pt')]
            assert len(files) == 1, files
            _BASE[key] = torch.load(os.path.join(d, files[0]))
    return _BASE[key]


@pytest.mark.parametrize('world,exchange', [(2, 'dense'), (2, 'sparse'), (4, 'sparse'), (2, 'sharded'), (4, 'sharded')])
def test_n_rank_engine_equals_single_process_at_tmall_shape(world, exchange):
    """N ranks on the contiguous shards of a 64-session Tmall-shape batch == one process on the whole batch: losses (mean of the
    shard means), replicas bit-identical, parameters after two Adam steps."""
    assert torch.cuda.is_available()
    with tempfile.TemporaryDirectory() as d:
        one = _baseline(_run_tmall)
        mp.spawn(_run_tmall, args=(world, _free_port(), d, exchange), nprocs=world, join=True)
        ranks = [torch.load(os.path.join(d, 'tm_w%d_r%d.pt' % (world, r))) for r in range(world)]
    for s in range(2):
        assert abs(sum(r['losses'][s] for r in ranks) / world - one['losses'][s]) < 2e-5
    for k, v in one['sd'].items():
        for r in ranks[1:]:
            assert torch.equal(ranks[0]['sd'][k], r['sd'][k]), 'replicas diverged: ' + k
        if 'k_linear.bias' in k:
            continue            # analytically-zero gradient: Adam direction is rounding noise
        err = float((ranks[0]['sd'][k] - v).abs().max())
        assert err < 5e-5, (k, err)


@pytest.mark.parametrize('world,exchange,schedule', [(2, 'dense', 'wide'), (2, 'sparse', 'wide'), (2, 'sparse', 'phased'), (2, 'sharded', 'wide'),
                                                     (4, 'sharded', 'wide'), (2, 'sharded', 'phased')])      # (4 ranks x sparse / dense: the Tmall-shape test above)
def test_n_rank_engine_equals_single_process(world, exchange, schedule):
    """exchange: dense all-reduce of the item-id table gradient, or the touched-rows all-gather.
    schedule: the one-call backward with the exchange under its tail (default), or the two-call order.
    world 4 = one session per rank of the 4-session fixture batch."""
    assert torch.cuda.is_available()
    with tempfile.TemporaryDirectory() as d:
        one = _baseline(_run)
        mp.spawn(_run, args=(world, _free_port(), d, exchange, 'gloo', False, '1', '', schedule), nprocs=world, join=True)
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


@pytest.mark.parametrize('exchange,backend', [('dense', 'gloo'), ('sparse', 'gloo'), ('dense', 'nccl'), ('sparse', 'nccl')])
def test_lazy_table_adam_data_parallel(exchange, backend):
    """The lazy table Adam under data parallelism: every rank brings its OWN batch's rows up to date ahead of the forward pass and
    applies the step to the UNION of the ranks' touched rows (all-reduced row marks / all-gathered indices); after the
    state_dict flush the replicas are bit-identical and equal the dense single-process run.  nccl: a one-rank RCCL group."""
    assert torch.cuda.is_available()
    world = 2 if backend == 'gloo' else 1
    with tempfile.TemporaryDirectory() as d:
        one = _baseline(_run)
        mp.spawn(_run, args=(world, _free_port(), d, exchange, backend, False, '1', '_lazy', 'wide', '1'), nprocs=world, join=True)
        ranks = [torch.load(os.path.join(d, 'w%d_r%d_lazy.pt' % (world, r))) for r in range(world)]
    for s in range(2):
        assert abs(sum(r['losses'][s] for r in ranks) / world - one['losses'][s]) < 2e-5
    for k, v in one['sd'].items():
        for r in ranks[1:]:
            assert torch.equal(ranks[0]['sd'][k], r['sd'][k]), 'replicas diverged: ' + k
        if 'k_linear.bias' in k:
            continue
        assert float((ranks[0]['sd'][k] - v).abs().max()) < 5e-5, k


def test_seeded_bpr_noise_is_keyed_by_global_session():
    """The in-kernel BPR tie-breaks (intel_bpr_loss_seeded): 2 ranks on the halves of the batch draw exactly what one
    process draws for the whole batch (common seed, counter keyed by rank * B_loc + b: SURVEY.md 8-e), so the sampled
    negatives, the losses and the parameters after 2 steps agree -- and the two shards do NOT repeat each other's draws."""
    assert torch.cuda.is_available()
    with tempfile.TemporaryDirectory() as d:
        one = _baseline(_run, 'auto', 'gloo', True)
        mp.spawn(_run, args=(2, _free_port(), d, 'dense', 'gloo', True), nprocs=2, join=True)
        ranks = [torch.load(os.path.join(d, 'w2_r%d.pt' % r)) for r in range(2)]
    for s in range(2):
        whole = one['selects'][s]
        got = torch.cat([r['selects'][s] for r in ranks])
        assert torch.equal(whole, got), 'sampled negatives differ between 1 rank and 2 ranks at step %d' % s
        assert abs(sum(r['losses'][s] for r in ranks) / 2 - one['losses'][s]) < 2e-5
    for k, v in one['sd'].items():
        if 'k_linear.bias' in k:
            continue
        assert float((ranks[0]['sd'][k] - v).abs().max()) < 5e-5, k


@pytest.mark.parametrize('exchange,overlap,schedule', [('dense', '1', 'wide'), ('sparse', '1', 'wide'), ('sparse', '1', 'phased'), ('dense', '0', 'wide'),
                                                       ('sharded', '1', 'wide'), ('sharded', '1', 'phased')])
def test_rccl_world1_runs_every_collective_branch(exchange, overlap, schedule):
    """RCCL itself: a one-rank `nccl` process group on the test GPU with the engine forced onto its data-parallel branches
    (INTEL_DP_FORCE=1), in both backward schedules -- the (phased: asynchronous) all-reduce on the side stream, the uint8 MAX all-reduce of the row
    marks, all_gather_into_tensor of the touched rows, the bucket all-reduces, broadcast and barrier all execute through
    RCCL (identities at world 1), and the result must equal the plain single-process run (two steps; the embedding
    rows are accumulated with float atomics, whose order varies from run to run, so step 2 is compared to 1e-6, not bit
    for bit)."""
    assert torch.cuda.is_available()
    with tempfile.TemporaryDirectory() as d:
        one = _baseline(_run, 'auto', 'gloo', False, overlap, '', schedule)      # the same backward schedule: the comparison isolates the collectives (the two schedules round differently: chain launches only in the one-call form)
        mp.spawn(_run, args=(1, _free_port(), d, exchange, 'nccl', False, overlap, '_nccl', schedule), nprocs=1, join=True)
        got = torch.load(os.path.join(d, 'w1_r0_nccl.pt'))
    assert one['losses'][0] == got['losses'][0]
    assert abs(one['losses'][1] - got['losses'][1]) < 1e-6
    for k, v in one['sd'].items():
        if 'k_linear.bias' in k:
            continue            # analytically-zero gradient: Adam direction is rounding noise
        assert float((v - got['sd'][k]).abs().max()) < 2e-6, k


@pytest.mark.parametrize('rows,share', [(1, 1.0), (4095, 0.5), (4096, 0.0), (4097, 0.02), (1000003, 0.29), (10000000, 0.001)])
def test_rows_compact_and_mark_kernels(rows, share):
    """intel_rows_compact (the touched-rows list of the data-parallel table exchange, from the backward's row marks) against torch.nonzero:
    ascending rows, -1 padding up to the static capacity, marks beyond the capacity dropped; intel_rows_mark sets the marks of an index
    list (with -1 holes) and nothing else."""
    from intel_sigir2023_amd import _lib as L
    dev = torch.device('cuda:0')
    lib = L.lib()
    g = torch.Generator(device=dev).manual_seed(rows)
    flags = (torch.rand(rows, device=dev, generator=g) < share).to(torch.uint8) * 7      # any non-zero byte is a mark
    want = torch.nonzero(flags).flatten().to(torch.int32)
    st = L.stream_ptr(dev)
    scratch = torch.empty(int(lib.intel_rows_compact_scratch_ints(rows)), dtype=torch.int32, device=dev)
    for cap in (int(want.numel()) + 5, max(1, int(want.numel()) // 2)):
        idx = torch.full((cap,), 12345, dtype=torch.int32, device=dev)
        L.check(lib.intel_rows_compact(L.ptr(flags), rows, L.ptr(idx), cap, L.ptr(scratch), st), 'intel_rows_compact')
        torch.cuda.synchronize()
        n = min(cap, int(want.numel()))
        assert torch.equal(idx[:n], want[:n])
        assert bool((idx[n:] == -1).all())
    marks = torch.zeros(rows, dtype=torch.uint8, device=dev)
    lst = torch.cat([want[::3], torch.full((4,), -1, dtype=torch.int32, device=dev)])
    L.check(lib.intel_rows_mark(L.ptr(marks), L.ptr(lst), lst.numel(), st), 'intel_rows_mark')
    torch.cuda.synchronize()
    ref = torch.zeros(rows, dtype=torch.uint8, device=dev)
    ref[want[::3].long()] = 1
    assert torch.equal(marks, ref)
