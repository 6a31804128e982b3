"""Long-horizon parity against the REFERENCE.

(1) tests/golden/trajectory_*.npz (tests/golden/make_trajectory_golden.py, build container only): the reference's own training
    loop -- helpers/BaseRunner.py:279-290 with torch.optim.Adam over BaseModel.customize_parameters (coupled L2, BaseRunner.py:182-188)
    and a StepLR halving after step 20 (BaseRunner.py:200-203) -- run for 40 steps over three alternating batches.  The engine
    (IntELEngine.train_step: forward, Int* loss, backward, fused Adam) replays it with the dense AND the lazy item-id table Adam:
    every per-step loss and the final parameters.
(2) the Adam kernels themselves against torch.optim.Adam at step numbers the 2-step fixture F7 never reaches: bias corrections at
    t = 1, 2, 10, 1 000, 65 537, the flagged-rows form, the two-group launch, and the lazy replay across a schedule-window move past
    the 65 536-step window.

Tolerances.  The first 10 losses must agree to 1e-5 (north_star's bar for ONE step).  After that two fp32 implementations of the same
trajectory drift apart: every step's parameters carry the previous steps' rounding differences (summation order of the matrix
products, the embedding scatter's float atomics), and Adam's update lr * m / (sqrt(v) + eps) is not a contraction.  The bound grows
linearly, 1e-5 * (1 + (step - 9) / 4) -- 8.75e-5 at step 40; measured on MI355X: see LOSS_DRIFT_MEASURED below."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, grad_projection, make_args, make_corpus

pytestmark = pytest.mark.gpu

# worst |loss - reference| / bound observed on MI355X per fixture (dense, lazy), for the record: all well inside the bound
LOSS_DRIFT_MEASURED = {}


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch.device('cuda:0')


def _load(name):
    z = np.load(os.path.join(GOLDEN, 'trajectory_%s.npz' % name))
    meta = json.loads(str(z['cfg']))
    zsd = z if any(k.startswith('sd/') for k in z.files) else np.load(os.path.join(GOLDEN, 'trajectory_w64_IntListloss.npz'))
    sd = {k[3:]: torch.from_numpy(zsd[k]) for k in zsd.files if k.startswith('sd/')}
    batches = []
    for i in range(3):
        b = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('in%d/' % i)}
        b['batch_size'] = int(b['u_id_c'].shape[0])
        b['phase'] = 'train'
        batches.append(b)
    return z, meta, sd, batches


@pytest.mark.parametrize('lazy', [False, True], ids=['dense_table_adam', 'lazy_table_adam'])
@pytest.mark.parametrize('name', ['w64_IntListloss', 'w64_IntBPRloss', 'pub_IntBPRloss'])
def test_forty_reference_steps(name, lazy):
    from intel_sigir2023_amd import model as M
    from intel_sigir2023_amd.engine import IntELEngine
    dev = _dev()
    z, meta, sd, batches = _load(name)
    args = make_args(meta['args'], dev)
    model = M.IntEL(args, make_corpus(meta['shape']))
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train()
    eng = IntELEngine(model, meta['loss'], args, lr=meta['lr'], l2=meta['l2'], lazy_table=lazy)
    dbatches = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()} for b in batches]
    B, L = batches[0]['i_id_s'].shape
    worst = 0.0
    for step in range(meta['steps']):
        if step == meta['lr_drop_at']:
            eng.set_lr(meta['lr'] * 0.5)
        torch.manual_seed(meta['noise_seed'] + step)
        noise = torch.rand(B, L, L)          # the reference's draw (loss/BPRloss.py: torch.rand on the CPU generator), re-drawn with its seed
        if step < 3:
            assert np.array_equal(noise.numpy(), z['noise%d' % step]), 'the CPU generator does not reproduce the fixture\'s noise draw'
        loss, ens, itl = eng.train_step(dbatches[step % 3], noise=noise.to(dev))
        ref = z['losses'][step]
        bound = 1e-5 if step < 10 else 1e-5 * (1.0 + (step - 9) / 4.0)
        for got, want, what in ((loss, ref[0], 'loss'), (ens, ref[1], 'ensemble loss'), (itl, ref[2], 'intent loss')):
            err = abs(float(got) - float(want))
            worst = max(worst, err / bound)
            assert err <= bound, (name, step, what, float(got), float(want), err, bound)
    torch.cuda.synchronize()
    LOSS_DRIFT_MEASURED[(name, lazy)] = worst
    # final parameters (state_dict settles the lazy table).  40 Adam steps move a parameter by up to ~40 lr = 0.03; the two trajectories may
    # differ by a small fraction of that: 2e-3 of the largest movement of the tensor (measured: <= 3e-4), projections of the big matrices likewise
    final = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    rows = {k[len('final_rows/'):]: torch.from_numpy(z[k]) for k in z.files if k.startswith('final_rows/')}
    checked, bad = 0, []
    for k in z.files:
        if k.startswith('final/'):
            pn = k[len('final/'):]
            got, init = final[pn], sd[pn]
            if pn in rows:
                got, init = got[rows[pn]], init[rows[pn]]
            want = torch.from_numpy(z[k])
            move = float((want - init).abs().max())
            err = float((got - want).abs().max())
            if pn.endswith('k_linear.bias'):
                # The key bias of an attention block has NO gradient (a constant added to every key shifts each query's scores by a constant:
                # softmax ignores it); what autograd returns is rounding noise of ~1e-9, and Adam divides noise by (sqrt(noise^2) + eps):
                # a random walk of a few percent of lr per step that no two implementations share.  Both walks stay within 1 % of 40 lr.
                if not (err <= 0.01 * meta['steps'] * meta['lr'] and move <= 0.01 * meta['steps'] * meta['lr']):
                    bad.append((pn, 'noise-driven parameter moved too far', err, move))
            elif err > 2e-3 * move + 1e-7:
                bad.append((pn, err, move))
            checked += 1
        elif k.startswith('finalproj/'):
            pn = k[len('finalproj/'):]
            upd = (final[pn] - sd[pn]).numpy()
            got, want = grad_projection(upd), z[k]
            if abs(got[1] - want[1]) > 2e-3 * want[1] + 1e-9 or abs(got[0] - want[0]) > 2e-3 * want[1] + 1e-9:
                bad.append((pn, 'projection / norm of the update', tuple(got), tuple(want)))
            checked += 1
    print('trajectory %s lazy=%s: worst loss error / bound %.3f, %d tensors checked' % (name, lazy, worst, checked))
    assert not bad, (name, bad)
    assert checked > 40
    # untouched table rows moved by weight decay alone -- and they DID move (dense semantics)
    w0, w1 = sd['iid_embeddings.weight'], final['iid_embeddings.weight']
    assert float((w1 - w0).abs().min()) > 0.0 or float(w0.abs().min()) == 0.0


def _torch_adam_step(p, g, m, v, t, lr, b1, b2, eps, wd):
    """One torch.optim.Adam step on the CPU from the given state (step counter t - 1 -> t)."""
    pp = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([pp], lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd)
    opt.state[pp] = {'step': torch.tensor(float(t - 1)), 'exp_avg': m.clone(), 'exp_avg_sq': v.clone()}
    pp.grad = g.clone()
    opt.step()
    st = opt.state[pp]
    assert float(st['step']) == float(t)
    return pp.detach(), st['exp_avg'], st['exp_avg_sq']


@pytest.mark.parametrize('t', [1, 2, 10, 1000, 65537])
@pytest.mark.parametrize('wd', [0.0, 1e-4])
def test_adam_kernels_match_torch_adam_at_step(t, wd):
    """intel_adam_step, intel_adam_step_pair and intel_adam_step_rows: one step at step number t from a plausible state (t = 1: zero moments)
    against torch.optim.Adam with its state's step counter set to t - 1.  A wrong bias correction is off by >= 1e-3 of the update at every t
    tested; the bound is 2e-5 of a full-size update (lr) + 4 ulp of the parameter, moments to 1e-6 of their largest element."""
    from intel_sigir2023_amd import _lib as L
    dev = _dev()
    lib = L.lib()
    gen = torch.Generator().manual_seed(1000 + t)
    rows, d = 257, 64
    n = rows * d
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    PTOL = 2e-5 * lr + 4e-8
    p = torch.randn(n, generator=gen) * 0.1
    g = torch.randn(n, generator=gen) * 1e-2
    g[::7] = 0.0                                    # exact zeros: rows / elements without gradient
    m = torch.zeros(n) if t == 1 else torch.randn(n, generator=gen) * 1e-3
    v = torch.zeros(n) if t == 1 else m * m * (0.5 + 3.5 * torch.rand(n, generator=gen)) + 1e-12      # |m| / sqrt(v) <= 1.5, as a real Adam state has it
    rp, rm, rv = _torch_adam_step(p, g, m, v, t, lr, b1, b2, eps, wd)
    st = L.stream_ptr(dev)

    def check(dp, dm, dv, what):
        torch.cuda.synchronize()
        assert float((dp.cpu() - rp).abs().max()) <= PTOL, (what, t, float((dp.cpu() - rp).abs().max()))
        assert float((dm.cpu() - rm).abs().max()) <= 1e-6 * max(1e-30, float(rm.abs().max())) + 1e-12, what
        assert float((dv.cpu() - rv).abs().max()) <= 1e-6 * max(1e-30, float(rv.abs().max())) + 1e-15, what

    # dense kernel
    dp, dg, dm, dv = (x.clone().to(dev) for x in (p, g, m, v))
    L.check(lib.intel_adam_step(L.ptr(dp), L.ptr(dg), L.ptr(dm), L.ptr(dv), n, lr, b1, b2, eps, wd, t, 1.0, 1, st), 'intel_adam_step')
    check(dp, dm, dv, 'intel_adam_step')
    assert float(dg.abs().max()) == 0.0                                              # zero_grad = 1 cleared it
    # two groups in one launch: this tensor with its decay + a second one without
    dp2, dg2, dm2, dv2 = (x.clone().to(dev) for x in (p, g, m, v))
    q = [x[:999].clone().to(dev) for x in (p, g, m, v)]
    arr = lambda a, b: (C.c_void_p * 2)(a.data_ptr(), b.data_ptr())
    L.check(lib.intel_adam_step_pair(arr(dp2, q[0]), arr(dg2, q[1]), arr(dm2, q[2]), arr(dv2, q[3]), (C.c_longlong * 2)(n, 999),
                                     (C.c_float * 2)(wd, 0.0), lr, b1, b2, eps, t, 1.0, 1, st), 'intel_adam_step_pair')
    check(dp2, dm2, dv2, 'intel_adam_step_pair group 0')
    qp, qm, qv = _torch_adam_step(p[:999], g[:999], m[:999], v[:999], t, lr, b1, b2, eps, 0.0)
    torch.cuda.synchronize()
    assert float((q[0].cpu() - qp).abs().max()) <= PTOL
    # flagged-rows form: the gradient is zero outside the flagged rows (the kernel never reads it there)
    g2 = g.clone().reshape(rows, d)
    flags = torch.zeros(rows, dtype=torch.uint8)
    flags[::3] = 1
    g2[flags == 0] = 0.0
    rp2, rm2, rv2 = _torch_adam_step(p, g2.reshape(-1), m, v, t, lr, b1, b2, eps, wd)
    dp3, dm3, dv3 = (x.clone().reshape(rows, d).to(dev) for x in (p, m, v))
    dg3 = g2.clone().to(dev)
    dg3[flags.to(dev) == 0] = 123.0                  # garbage where no flag is set must not be read ... and is not cleared
    df = flags.to(dev)
    L.check(lib.intel_adam_step_rows(L.ptr(dp3), L.ptr(dg3), L.ptr(dm3), L.ptr(dv3), rows, d, L.ptr(df), lr, b1, b2, eps, wd, t, 1.0, st), 'intel_adam_step_rows')
    torch.cuda.synchronize()
    assert float((dp3.cpu().reshape(-1) - rp2).abs().max()) <= PTOL
    assert float((dm3.cpu().reshape(-1) - rm2).abs().max()) <= 1e-6 * float(rm2.abs().max()) + 1e-12
    assert int(df.max()) == 0


def test_lazy_replay_across_the_65536_step_window_matches_torch_adam():
    """The lazy table Adam at step numbers around the engine's schedule window (IntELEngine.LAZY_CAP = 65 536): a table whose state is at step
    65 530 takes 12 more steps with a few touched rows each -- window [65 530, 65 538) is full after 8 of them and is moved (flush + new base,
    what engine._lazy_step does) -- and is then settled.  Reference: torch.optim.Adam on the CPU from the same state (step counter 65 530),
    dense gradient with zeros in the untouched rows."""
    from intel_sigir2023_amd import _lib as L
    dev = _dev()
    lib = L.lib()
    gen = torch.Generator().manual_seed(7)
    rows, d, t0 = 301, 64, 65530
    lr, b1, b2, eps, wd = 1e-3, 0.9, 0.999, 1e-8, 1e-4
    p = torch.randn(rows, d, generator=gen) * 0.1
    m = torch.randn(rows, d, generator=gen) * 1e-3
    v = m * m * (0.5 + 3.5 * torch.rand(rows, d, generator=gen)) + 1e-12
    pp = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([pp], lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd)
    opt.state[pp] = {'step': torch.tensor(float(t0)), 'exp_avg': m.clone(), 'exp_avg_sq': v.clone()}
    dp, dm, dv = p.clone().to(dev), m.clone().to(dev), v.clone().to(dev)
    dg = torch.zeros(rows, d, device=dev)
    df = torch.zeros(rows, dtype=torch.uint8, device=dev)
    cap = 8
    last = torch.full((rows,), t0, dtype=torch.int32, device=dev)
    sched = torch.zeros(cap, 2, device=dev)
    t = L.IntelLazyTable(p=dp.data_ptr(), m=dm.data_ptr(), v=dv.data_ptr(), last=last.data_ptr(), sched=sched.data_ptr(), rows=rows, d=d,
                         base=t0, cap=cap, beta1=b1, beta2=b2, eps=eps, weight_decay=wd)
    st = L.stream_ptr(dev)
    for step in range(t0 + 1, t0 + 13):
        cur_lr = lr if step < t0 + 7 else lr * 0.5
        idx = torch.randperm(rows, generator=gen)[:17]
        grad = torch.randn(17, d, generator=gen) * 1e-2
        full = torch.zeros(rows, d)
        full[idx] = grad
        for gr in opt.param_groups:
            gr['lr'] = cur_lr
        pp.grad = full
        opt.step()
        dg[idx.to(dev)] = grad.to(dev)
        df[idx.to(dev)] = 1
        if step - t.base > t.cap:
            L.check(lib.intel_adam_lazy_flush(C.byref(t), step - 1, st), 'flush')
            t.base = step - 1
        L.check(lib.intel_adam_lazy_step(C.byref(t), L.ptr(dg), L.ptr(df), cur_lr, step, st), 'lazy step')
    L.check(lib.intel_adam_lazy_flush(C.byref(t), t0 + 12, st), 'flush')
    torch.cuda.synchronize()
    assert float(opt.state[pp]['step']) == float(t0 + 12)
    assert float((dp.cpu() - pp.detach()).abs().max()) <= 12 * 2e-5 * lr + 4e-8
    assert float((dm.cpu() - opt.state[pp]['exp_avg']).abs().max()) <= 1e-6 * float(opt.state[pp]['exp_avg'].abs().max())
    assert float((dv.cpu() - opt.state[pp]['exp_avg_sq']).abs().max()) <= 1e-6 * float(opt.state[pp]['exp_avg_sq'].abs().max())
