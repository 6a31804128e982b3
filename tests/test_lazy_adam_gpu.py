"""The lazy form of the item-id table's dense Adam (include/intel_hip.h: IntelLazyTable; engine.py: lazy_table=True) against the
dense sweep it replaces (intel_adam_step_rows == torch.optim.Adam over the whole table, helpers/BaseRunner.py:182-188):
kernels bit for bit, the engine at every point where the table is observed (forward outputs, losses, state_dict)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch.device('cuda:0')


@pytest.mark.parametrize('d,rows,wd', [(64, 5003, 1e-4), (16, 777, 0.0), (128, 2048, 1e-2)])
def test_lazy_kernels_equal_dense_sweep_bit_for_bit(d, rows, wd):
    """14 steps with random touched rows and a changing learning rate: after every step the rows a reader asks for (catch-up by
    id, with repeated and out-of-range ids) and at the end all rows (flush) hold exactly the bits of the dense sweep -- parameter
    and both moments; the lazy step leaves gradient and marks clean; the schedule window is moved once on the way."""
    from intel_sigir2023_amd import _lib as L
    dev = _dev()
    lib = L.lib()
    g = torch.Generator(device=dev).manual_seed(d + rows)
    p0 = torch.randn(rows, d, device=dev, generator=g) * 0.1
    dense = dict(p=p0.clone(), m=torch.zeros_like(p0), v=torch.zeros_like(p0), g=torch.zeros_like(p0),
                 f=torch.zeros(rows, dtype=torch.uint8, device=dev))
    lazy = dict(p=p0.clone(), m=torch.zeros_like(p0), v=torch.zeros_like(p0), g=torch.zeros_like(p0),
                f=torch.zeros(rows, dtype=torch.uint8, device=dev))
    cap = 9
    last = torch.zeros(rows, dtype=torch.int32, device=dev)
    sched = torch.zeros(cap, 2, device=dev)
    b1, b2, eps = 0.9, 0.999, 1e-8
    t = L.IntelLazyTable(p=lazy['p'].data_ptr(), m=lazy['m'].data_ptr(), v=lazy['v'].data_ptr(), last=last.data_ptr(),
                         sched=sched.data_ptr(), rows=rows, d=d, base=0, cap=cap, beta1=b1, beta2=b2, eps=eps, weight_decay=wd)
    st = L.stream_ptr(dev)
    observed = 0
    for step in range(1, 15):
        lr = 1e-3 if step < 8 else 5e-4
        n = int(torch.randint(1, max(2, rows // 4), (1,), generator=torch.Generator().manual_seed(step)).item())
        idx = torch.randperm(rows, device=dev, generator=g)[:n]
        grad = torch.randn(n, d, device=dev, generator=g) * 1e-2
        for s in (dense, lazy):
            s['g'][idx] = grad
            s['f'][idx] = 1
        L.check(lib.intel_adam_step_rows(L.ptr(dense['p']), L.ptr(dense['g']), L.ptr(dense['m']), L.ptr(dense['v']), rows, d,
                                         L.ptr(dense['f']), lr, b1, b2, eps, wd, step, 1.0, st), 'dense')
        if step - t.base > t.cap:                      # what engine._lazy_step does when the window is full
            L.check(lib.intel_adam_lazy_flush(C.byref(t), step - 1, st), 'flush')
            t.base = step - 1
        L.check(lib.intel_adam_lazy_step(C.byref(t), L.ptr(lazy['g']), L.ptr(lazy['f']), lr, step, st), 'lazy')
        assert float(lazy['g'].abs().max()) == 0.0 and int(lazy['f'].max()) == 0
        # a reader: some rows, with repeats, -1 and an id past the table
        want = torch.randint(0, rows, (max(1, rows // 7),), device=dev, generator=g).to(torch.int32)
        ids_a = torch.cat([want, want[:5], torch.tensor([-1, rows + 3], dtype=torch.int32, device=dev)])
        ids_b = want[:3].clone()
        L.check(lib.intel_adam_lazy_catchup(C.byref(t), L.ptr(ids_a), ids_a.numel(), L.ptr(ids_b), ids_b.numel(), step, st), 'catchup')
        w = want.long()
        for k in ('p', 'm', 'v'):
            assert torch.equal(lazy[k][w], dense[k][w]), (k, step, float((lazy[k][w] - dense[k][w]).abs().max()), int((lazy[k][w] != dense[k][w]).sum()))
        assert bool((last[w] == step).all())
        observed += int((last < step).sum())
    assert observed > 0                                # rows really were left behind on the way
    assert not torch.equal(lazy['p'], dense['p'])
    L.check(lib.intel_adam_lazy_flush(C.byref(t), 14, st), 'flush')
    for k in ('p', 'm', 'v'):
        assert torch.equal(lazy[k], dense[k]), k
    assert bool((last == 14).all())


def _engines(monkeypatch, workload='tiny', cap=None, **engine_kw):
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.engine import IntELEngine
    from intel_sigir2023_amd.model import IntEL
    dev = _dev()
    out = []
    for lazy in (False, True):
        torch.manual_seed(5)
        args = synth.make_args(workload, dev, cal_diversity=0)
        corpus, _ = synth.make_corpus(workload)
        model = IntEL(args, corpus).to(dev)
        if cap is not None:
            monkeypatch.setattr(IntELEngine, 'LAZY_CAP', cap)
        eng = IntELEngine(model, 'IntBPRloss', args, lr=1e-3, l2=1e-4, lazy_table=lazy, **engine_kw)
        assert (eng._lazy is not None) == lazy
        out.append((model, eng))
    return out


@pytest.mark.parametrize('cap', [None, 4])
def test_lazy_engine_equals_dense_engine(cap, monkeypatch):
    """Nine fused steps over three alternating batches (rows come back after being left behind), a learning-rate change, a
    weight-decay change and an evaluation in between: every loss, the evaluation outputs and the final state_dict of the lazy engine equal the dense
    engine's (up to the order of the embedding scatter's float atomics); before the flush the lazy table really differs.
    cap = 4: the schedule window is moved twice on the way."""
    from intel_sigir2023_amd import synth
    dev = _dev()
    (m0, e0), (m1, e1) = _engines(monkeypatch, cap=cap)
    batches = [synth.make_batch('tiny', 16, dev, seed=40 + i, ragged=True) for i in range(3)]
    held_out = synth.make_batch('tiny', 16, dev, seed=77, ragged=True)
    for step in range(9):
        if step == 5:
            e0.set_lr(5e-4), e1.set_lr(5e-4)
        if step == 7:
            e0.l2 = e1.l2 = 5e-4          # another hyper-parameter: the lazy engine settles the table first (one set per replay)
        b = batches[step % 3]
        l0 = e0.train_step(b, noise_seed=100 + step)
        l1 = e1.train_step(b, noise_seed=100 + step)
        for a, c in zip(l0, l1):
            assert abs(float(a) - float(c)) < 1e-6, step
        if step == 6:
            (o0, n0), (o1, n1) = e0.eval_step(held_out), e1.eval_step(held_out)
            for k in o0:
                assert float((o0[k] - o1[k]).abs().max()) < 1e-6, k
            assert float((n0 - n1).abs().max()) < 1e-6
    torch.cuda.synchronize()
    w0, w1 = m0.iid_embeddings.weight.detach(), m1.iid_embeddings.weight.detach()
    behind = (e1._lazy_last < e1.step_count)
    assert 0 < int(behind.sum()) and float((w0 - w1).abs().max()) > 0.0
    sd0, sd1 = m0.state_dict(), m1.state_dict()                      # the hook settles the table
    assert bool((e1._lazy_last == e1.step_count).all())
    for k in sd0:
        if 'k_linear.bias' in k:
            continue            # analytically-zero gradient: Adam direction is rounding noise
        assert float((sd0[k] - sd1[k]).abs().max()) < 2e-6, k
    for k in ('iid',):
        assert float((e0.m[k] - e1.m[k]).abs().max()) < 2e-6 and float((e0.v[k] - e1.v[k]).abs().max()) < 2e-6
    # the model's own forward (no engine call) brings its rows up to date as well
    l0 = e0.train_step(batches[0], noise_seed=7)
    l1 = e1.train_step(batches[0], noise_seed=7)
    m0.eval(), m1.eval()
    with torch.no_grad():
        a, c = m0(held_out), m1(held_out)
    for k in a:
        assert float((a[k] - c[k]).abs().max()) < 1e-6, k


def test_lazy_engine_load_state_dict_keeps_the_moments(monkeypatch):
    """load_state_dict() in the middle of training: the hook settles the pending rows BEFORE the parameters are replaced, so
    the moments of both engines agree afterwards and training continues identically."""
    from intel_sigir2023_amd import synth
    dev = _dev()
    (m0, e0), (m1, e1) = _engines(monkeypatch)
    batches = [synth.make_batch('tiny', 16, dev, seed=50 + i, ragged=True) for i in range(2)]
    start = {k: v.detach().clone() for k, v in m0.state_dict().items()}
    for step in range(4):
        e0.train_step(batches[step % 2], noise_seed=step), e1.train_step(batches[step % 2], noise_seed=step)
    m0.load_state_dict(start), m1.load_state_dict(start)
    for step in range(3):
        l0 = e0.train_step(batches[step % 2], noise_seed=10 + step)
        l1 = e1.train_step(batches[step % 2], noise_seed=10 + step)
        assert abs(float(l0[0]) - float(l1[0])) < 1e-6
    sd0, sd1 = m0.state_dict(), m1.state_dict()
    for k in sd0:
        if 'k_linear.bias' in k:
            continue
        assert float((sd0[k] - sd1[k]).abs().max()) < 2e-6, k
