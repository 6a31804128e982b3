"""The bf16 mode (`--dtype bf16`, BASELINE.json config 2; SURVEY.md 7 hard-part 6): the reference is fp32-only, so there is
no reference output to compare with -- the mode is gated against this repo's own fp32 build: forward outputs within bf16
rounding of the fp32 ones, and after the same 300 planted-signal training steps the held-out NDCG@3 of the two builds
agrees and the loss trajectories stay together."""
import numpy as np
import pytest
import torch

from tests.helpers import Fixture, make_args, make_corpus

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch.device('cuda:0')


@pytest.mark.parametrize('name', ['tmall64', 'stress', 'lifedata'])
def test_bf16_gradients_match_the_emulating_oracle(name):
    """The backward of the mode against oracle.forward_bf16's autograd: its products are Functions whose backward rounds the
    gradient operands to bf16 exactly where the HIP backward does (dx = bf(dy) bf(w), dw = bf(dy)^T bf(x), the attention's
    dP / dV / dQ / dK products), fp32 everywhere else.  Every parameter gradient within 2e-3 of the tensor's largest element
    for nine tensors in ten, 5e-3 for the worst (fp32 parity is 2e-4; one bf16 rounding is 4e-3 relative) -- the fp32 oracle's
    gradients are 1e-2 .. 1e-1 away."""
    from oracle import intel_oracle as O
    from intel_sigir2023_amd import loss as LS
    from intel_sigir2023_amd.model import IntEL
    fx = Fixture(name)
    dev = _dev()
    a = dict(fx.args)
    a['dtype'] = 'bf16'
    args = make_args(a, dev)
    model = IntEL(args, make_corpus(fx.shape))
    model.load_state_dict(fx.state_dict(), strict=True)
    model = model.to(dev).train()
    batch = fx.batch(dev)
    out = model(batch)
    loss, _, _ = LS.IntListloss(args)(out, batch)
    loss.backward()
    cfg = O.Config(**fx.args)
    cfg.cal_diversity = args.cal_diversity
    grads = {}
    for emu in (True, False):
        sd = {k: v.clone().requires_grad_(True) for k, v in fx.state_dict().items()}
        ref = O.forward_bf16(sd, fx.batch(), cfg) if emu else O.forward(sd, fx.batch(), cfg)
        rl = O.int_list_loss(ref, fx.batch(), cfg)[0]
        rl.backward()
        grads[emu] = {k: v.grad for k, v in sd.items() if v.grad is not None}
        if emu:
            assert abs(float(loss) - float(rl)) < 2e-5 * max(1.0, abs(float(rl)))
    worst, wk, far, errs = 0.0, None, 0.0, []
    for k, p in model.named_parameters():
        g = grads[True].get(k)
        if g is None or p.grad is None:
            continue
        gm = float(g.abs().max())
        if gm < 1e-12:
            continue
        err = float((p.grad.detach().cpu() - g).abs().max()) / gm
        far = max(far, float((grads[False][k] - g).abs().max()) / gm)
        if 'k_linear.bias' in k:
            continue            # analytically zero (a key bias shifts every score of a row alike): rounding noise in both
        errs.append(err)
        if err > worst:
            worst, wk = err, k
    errs.sort()
    print('bf16 gradients vs emulating oracle: worst %.2e (%s), median %.2e, 90th percentile %.2e; fp32 oracle is %.2e away'
          % (worst, wk, errs[len(errs) // 2], errs[int(0.9 * len(errs))], far))
    # An operand that sits on a bf16 rounding boundary can round the other way in the two implementations (their fp32 values differ
    # in the last bits): one such flip moves a product term by 4e-3 of its size, and the fixtures' B-row gradients are sums over 2-4
    # sessions.  Hence 5e-3 for the single worst element of any tensor, 2e-3 for nine tensors in ten (measured: worst 3.4e-3).
    assert worst <= 5e-3, (wk, worst)
    assert errs[int(0.9 * len(errs))] <= 2e-3, errs[int(0.9 * len(errs))]
    assert far > 5e-3                                          # the emulation is not the fp32 arithmetic


@pytest.mark.parametrize('name', ['gru_bpr'])
def test_bf16_forward_and_gradients_track_fp32(name):
    from intel_sigir2023_amd import loss as LS
    from intel_sigir2023_amd.model import IntEL
    fx = Fixture(name)
    dev = _dev()
    outs, grads = {}, {}
    for dt in ('f32', 'bf16'):
        a = dict(fx.args)
        a['dtype'] = dt
        args = make_args(a, dev)
        model = IntEL(args, make_corpus(fx.shape))
        model.load_state_dict(fx.state_dict(), strict=True)
        model = model.to(dev)
        model.train()
        batch = fx.batch(dev)
        out = model(batch)
        loss, _, _ = LS.IntListloss(args)(out, batch)
        loss.backward()
        outs[dt] = {k: v.detach().cpu() for k, v in out.items()}
        grads[dt] = {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None}
    for k in ('weights', 'ens_score', 'intents'):
        ref = outs['f32'][k]
        err = float((outs['bf16'][k] - ref).abs().max()) / max(1.0, float(ref.abs().max()))
        assert 0 < err < 3e-2, (k, err)                       # differs (one bf16 product), by bf16 rounding only
    worst, wk = 0.0, None
    gmax = max(float(g.norm()) for g in grads['f32'].values())
    for k, g in grads['f32'].items():
        if 'k_linear.bias' in k or float(g.norm()) < 1e-4 * gmax:
            continue            # analytically-zero (attention key bias) or negligible gradients are rounding noise in both builds
        rel = float((grads['bf16'][k] - g).norm()) / float(g.norm())
        if rel > worst:
            worst, wk = rel, k
    assert worst < 0.15, (wk, worst)                          # every gradient tensor within 15 % in norm


@pytest.mark.parametrize('name', ['tmall64', 'stress', 'lifedata'])
def test_bf16_gradients_against_the_reference_fixtures(name):
    """A check that does NOT go through the emulating oracle (which restates the build's own rounding points: a bug shared by build and emulation
    is invisible to it): the bf16 build's loss and parameter gradients against the fp32 REFERENCE's, as the fixtures hold them
    (tests/golden/make_golden.py ran the unmodified reference).  One bf16 rounding is 2^-9 of an operand; a gradient tensor is a sum of products
    of rounded operands over the batch's rows (2 - 4 sessions in these fixtures: a relu that flips behind a rounding moves a bias gradient by several per
    cent), so the bar is per TENSOR: direction (cosine >= 0.99; measured worst 0.9973) and size (norm within 8 %; measured 3e-3) of every tensor that is not
    negligible -- for the two large fixtures a random projection and the norm within 12 % of the tensor's norm (measured 5.9e-2 / 2.9e-2) --, and the loss
    within 2e-3: a wrong tile, a dropped term or a stale operand is O(1) off in at least one tensor."""
    from intel_sigir2023_amd import loss as LS
    from intel_sigir2023_amd.model import IntEL
    from tests.helpers import grad_projection
    fx = Fixture(name)
    dev = _dev()
    a = dict(fx.args)
    a['dtype'] = 'bf16'
    args = make_args(a, dev)
    args.cal_diversity = 1
    model = IntEL(args, make_corpus(fx.shape))
    model.load_state_dict(fx.state_dict(), strict=True)
    model = model.to(dev).train()
    batch = fx.batch(dev)
    tag = 'bpr' if fx.detail == 'bpr' else 'pl'      # (the tmall64 fixture keeps the IntBPRloss gradients, with the reference's tie-breaking draw)
    if tag == 'bpr':
        batch['bpr_noise'] = torch.from_numpy(fx['bpr/noise']).to(dev)
    out = model(batch)
    loss, _, _ = (LS.IntBPRloss if tag == 'bpr' else LS.IntListloss)(args)(out, batch)
    loss.backward()
    ref_loss = float(fx['int%s/loss' % tag])
    assert abs(float(loss) - ref_loss) < 2e-3 * max(1.0, abs(ref_loss)), (float(loss), ref_loss)
    named = dict(model.named_parameters())
    report = []
    if fx.detail == 'proj':      # the large fixtures keep two random projections per gradient tensor instead of the tensor
        gmax = max(float(ref[1]) for ref in fx.group('gradproj_pl').values())
        for k, ref in fx.group('gradproj_pl').items():
            if 'k_linear.bias' in k or float(ref[1]) < 1e-4 * gmax:
                continue        # analytically zero (attention key bias) or negligible: rounding noise in any arithmetic
            g = named[k].grad
            got = grad_projection(np.zeros(tuple(named[k].shape), np.float32) if g is None else g.cpu().numpy())
            report.append((k, abs(got[0] - ref[0]) / max(1e-6, ref[1]), abs(got[1] - ref[1]) / max(1e-6, ref[1])))
        worst = max(report, key=lambda t: max(t[1], t[2]))
        print('bf16 gradients vs the reference (projections): worst %s %.3e / %.3e of the tensor norm' % worst)
        assert max(worst[1], worst[2]) < 0.12, worst
        return
    rows = fx.group('grad_%s_rows' % tag)
    refs = fx.group('grad_' + tag)
    gmax = max(float(np.linalg.norm(r)) for r in refs.values())
    for k, ref in refs.items():
        if 'k_linear.bias' in k or float(np.linalg.norm(ref)) < 1e-4 * gmax:
            continue            # analytically zero (attention key bias) or negligible: rounding noise in any arithmetic
        g = named[k].grad
        g = torch.zeros_like(named[k]) if g is None else g
        g = g.cpu()
        if k in rows:
            g = g[torch.from_numpy(rows[k])]
        g = g.numpy().astype(np.float64).ravel()
        r = ref.astype(np.float64).ravel()
        cos = float(g @ r / max(1e-30, np.linalg.norm(g) * np.linalg.norm(r)))
        size = float(abs(np.linalg.norm(g) - np.linalg.norm(r)) / np.linalg.norm(r))
        report.append((k, cos, size))
    wc = min(report, key=lambda t: t[1])
    wsz = max(report, key=lambda t: t[2])
    print('bf16 gradients vs the reference: worst direction %s cos %.5f, worst size %s %.3e (of %d tensors)' % (wc[0], wc[1], wsz[0], wsz[2], len(report)))
    assert wc[1] >= 0.99, wc
    assert wsz[2] <= 8e-2, wsz


@pytest.mark.parametrize('name', ['tmall64', 'stress', 'lifedata', 'default'])
def test_bf16_forward_matches_the_emulating_oracle(name):
    """oracle.forward_bf16 restates WHAT the mode computes: the reference's forward with both operands of a product rounded
    to bf16 exactly where the HIP build runs it on the bf16 pipe (64- / 128-wide linears, whole-sequence attention products)
    and fp32 everywhere else.  The HIP outputs equal it to summation order (measured 5e-8 .. 3e-6 of the output scale, where the
    fp32 forward is 1e-4 .. 1e-3 away): the mode is the reference's arithmetic with those rounding points, nothing else."""
    from oracle import intel_oracle as O
    from intel_sigir2023_amd.model import IntEL
    fx = Fixture(name)
    dev = _dev()
    a = dict(fx.args)
    a['dtype'] = 'bf16'
    args = make_args(a, dev)
    model = IntEL(args, make_corpus(fx.shape))
    model.load_state_dict(fx.state_dict(), strict=True)
    model = model.to(dev).eval()
    with torch.no_grad():
        got = {k: v.cpu() for k, v in model(fx.batch(dev)).items()}
    cfg = O.Config(**fx.args)
    sd, batch = fx.state_dict(), fx.batch()
    with torch.no_grad():
        emu = O.forward_bf16(sd, batch, cfg)
        f32 = O.forward(sd, batch, cfg)
    for k in ('weights', 'ens_score', 'intents'):
        scale = max(1.0, float(f32[k].abs().max()))
        e_emu = float((got[k] - emu[k]).abs().max()) / scale
        e_f32 = float((got[k] - f32[k]).abs().max()) / scale
        d_emu = float((emu[k] - f32[k]).abs().max()) / scale
        if name != 'default':
            assert d_emu > 1e-5, (k, d_emu)                  # the mode is engaged: the emulation differs from fp32
        else:
            assert d_emu < 1e-6, (k, d_emu)                  # 16/32-wide model: nothing runs on the bf16 pipe, in either (the pooling's
                                                             # algebraic form differs: fp32 rounding only)
        assert e_emu <= 2e-5, (k, e_emu, e_f32)              # measured: 5e-8 .. 3e-6 (summation order) against 1e-4 .. 1e-3 to fp32


def test_bf16_training_reaches_the_fp32_ndcg():
    """The gate of the mode: 300 fused training steps on a planted ranking signal (labels = top items of base ranker 0) with
    identical data, initialisation and BPR tie-breaks; held-out NDCG@3 of the bf16 build within 1e-3 of the fp32 build's and
    the loss trajectories within 2 % of each other on average."""
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.engine import IntELEngine
    from intel_sigir2023_amd.model import IntEL
    dev = _dev()
    # the headline widths (64-d embeddings: every tower / encoder linear is a K = 64 / 128 product, where the mode applies) on a
    # small corpus
    synth.WORKLOADS['tiny64'] = dict(flags=dict(synth.WORKLOADS['tmall']['flags']), corpus=dict(synth.WORKLOADS['tiny']['corpus']),
                                     batch=dict(synth.WORKLOADS['tiny']['batch']))

    def planted(seed):
        b = synth.make_batch('tiny64', 256, dev, seed=seed, ragged=True)
        s0 = b['scores'][:, :, 0].float()
        valid = torch.arange(s0.shape[1], device=dev)[None, :] < b['session_len'][:, None]
        order = torch.where(valid, s0, torch.full_like(s0, -1.0)).argsort(dim=1, descending=True)
        labels = torch.tensor([3, 2, 1, 1, 1], device=dev, dtype=torch.int32)
        r = torch.zeros_like(b['ranking'])
        r.scatter_(1, order[:, :5], labels[None, :].expand(r.shape[0], 5))
        b['ranking'] = (r * valid).int()
        return b
    held = [planted(9000 + i) for i in range(4)]
    res = {}
    for dt in ('f32', 'bf16'):
        torch.manual_seed(3)
        args = synth.make_args('tiny64', dev, dtype=dt)
        corpus, _ = synth.make_corpus('tiny64')
        model = IntEL(args, corpus).to(dev)
        eng = IntELEngine(model, 'IntBPRloss', args, lr=2e-3, l2=0.0)
        model.train()
        losses = [float(eng.train_step(planted(step), noise_seed=777 + step)[0]) for step in range(300)]
        model.eval()
        nd = float(torch.cat([eng.eval_step(h)[1].float() for h in held]).mean())
        res[dt] = (np.array(losses), nd)
    (l32, n32), (l16, n16) = res['f32'], res['bf16']
    print('NDCG@3 fp32 %.5f bf16 %.5f; mean |dloss| / loss = %.2e' % (n32, n16, float(np.mean(np.abs(l16 - l32) / l32))))
    assert n32 > 0.85 and float(np.abs(l16 - l32).max()) > 0      # the mode is really on
    assert abs(n16 - n32) <= 1e-3, (n32, n16)
    assert float(np.mean(np.abs(l16 - l32) / l32)) < 2e-2


def test_sweep_case_274_lies_inside_the_oracles_own_cloud():
    """Case 274 of the bf16-mode randomised sweep (`tools/fuzz_parity.py 500 61616 --dtype bf16`: lists of 65, two heads on the 64-wide score tower, two tied
    layers, five sessions) misses the emulating oracle by 5.5 x the sweep's bar on s_attn_head.v_linear.weight (29 % of the tensor's norm).  Bug or
    rounding-flip amplification?  Decided by the oracle itself (tools/bf16_cloud.py): re-run with every parameter multiplied by (1 + 1e-6 N(0, 1)) -- the size
    of the build's own last-bit deviations -- the oracle's gradients of that tensor differ from EACH OTHER by up to 69 % of its norm (operands on a bf16
    rounding boundary fall the other way, two layers later a few relu units of a few of the 325 rows switch), and the build sits 1.3 % from the nearest
    member: it is one more sample of the cloud.  A structural error (wrong tile, stale LDS, dropped term) would be O(1) away from EVERY member.
    Asserted: the oracle's own cloud is wider than the sweep's bar on that tensor (the amplification is real, not a loose bar), and for every gradient
    tensor the build's distance to the nearest of 12 members (24 in tools/bf16_cloud.py's own run, the numbers above) is no larger than the cloud's own spread (or the
    sweep's 5e-2 bar)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('bf16_cloud', os.path.join(os.path.dirname(__file__), '..', 'tools', 'bf16_cloud.py'))
    cloud = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cloud)
    h = cloud.build_case(274, 61616, _dev())
    assert h['desc'].startswith('IntListloss B=5 L=65'), h['desc']
    worst, wk, rows = cloud.cloud_check(h['model'], h['batch'], h['ref_batch'], h['cfg'], h['loss_name'], h['noise'], members=12, eps=1e-6)
    by = {k: (d0, dn, sp, ratio) for k, d0, dn, sp, ratio in rows}
    d0, dn, sp, _ = by['s_attn_head.v_linear.weight']
    print('case 274: s_attn_head.v_linear.weight %.3f of its norm from the unperturbed oracle, %.4f from the nearest of 12 members, cloud spread %.3f; worst ratio %.2f at %s'
          % (d0, dn, sp, worst, wk))
    assert worst <= 1.0, (wk, worst)        # every gradient tensor of the build: no farther from its nearest member than the members are from each other
    assert sp > 5e-2                        # the oracle disagrees with itself by more than the sweep's bar on this tensor (CPU-deterministic)
    # (measured: 0.287 of the norm from the unperturbed oracle, 0.013 from the nearest member.  Which branch the build lands on depends on its own last bits --
    # float atomics in the embedding scatter -- so the distance to the nearest of 24 samples is reported, not asserted beyond the cloud criterion above)


def test_sweep_case_348_is_one_relu_unit_from_the_oracle():
    """Case 348 of the second bf16-mode sweep (`tools/fuzz_parity.py 500 62626 --dtype bf16`: three sessions, lists of 33, 128-wide BERT4Rec history
    encoder) misses the emulating oracle by 1.46 x the sweep's bar on the history encoder's LAST block (linear1.weight / .bias 7.4 % of the norm, every
    other history-encoder tensor a uniform 1.5 %), and the parameter-noise cloud of case 274 does NOT contain it (1e-6 and 1e-5 noise, 24 / 48 members:
    the members' flips are random; none reaches this unit).  Decided constructively instead (tools/bf16_cloud.py: flip_probe): the whole difference
    of linear1.bias sits in ONE of its 128 elements (hidden unit 2); in the oracle that unit's pre-activation at session 0's last history row -- one of
    the three rows the pruned last block computes -- is 5.5e-4 against a row maximum of 1.6, the build's upstream bf16 roundings put it on the other
    side of zero; pushing that one entry across the kink in the ORACLE (its own test hook, nothing in the build changes) takes the build's worst
    gradient tensor from 7.4e-2 of the norm to 9.6e-3 (a score-tower tensor whose own cloud spread is 2.2e-2; the history encoder's tensors: 3.5e-3, their
    cloud-spread level).  A structural error would not collapse by flipping one activation.
    Asserted: before > the sweep's bar => after <= 2e-2 (0.4 of the bar) and the flipped pre-activation is below 1e-3 of its row's largest."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('bf16_cloud', os.path.join(os.path.dirname(__file__), '..', 'tools', 'bf16_cloud.py'))
    cloud = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cloud)
    h = cloud.build_case(348, 62626, _dev())
    assert h['desc'].startswith('IntListloss B=3 L=33'), h['desc']
    before, after, which = cloud.flip_probe(h['model'], h['ref_batch'], h['cfg'], h['loss_name'], h['noise'], tries=2, verbose=False)
    print('case 348: worst gradient tensor %.3e of its norm from the oracle, %.3e from the oracle with %s pushed across the relu kink' % (before, after, which))
    if before <= 5e-2:
        return      # the build's own last bits (float atomics in the embedding scatter) put the unit on the oracle's side this run: nothing to explain
    assert which is not None and which[0] == 'encoder.transformer_block.1.linear1' and which[4] < 1e-3, which
    assert after <= 2e-2, (before, after, which)
