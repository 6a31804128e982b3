"""Dropout in the HIP path: with the reference's keep masks the training-mode forward, IntMSEloss and every parameter
gradient match the reference (tests/golden/intel_dropout.npz); with the built-in generator the keep rate is 1 - p and
evaluation never drops."""
import numpy as np
import pytest
import torch

from tests.helpers import Fixture, build_model

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device('cuda:0')


def test_training_step_with_pinned_dropout_matches_reference():
    from intel_sigir2023_amd import loss as LS
    dev = _dev()
    fx = Fixture('dropout')
    model, args = build_model(fx, dev)
    model.train()
    keep = np.concatenate([fx['keep_i'].ravel(), fx['keep_s'].ravel()]).astype(np.float32)
    model._dropout_keep = torch.from_numpy(keep).to(dev)
    batch = fx.batch(dev)
    out = model(batch)
    for k in ('weights', 'ens_score', 'intents'):
        ref = fx['out/' + k]
        err = float(np.abs(out[k].detach().cpu().numpy() - ref).max())
        assert err <= 3e-5 * max(1.0, float(np.abs(ref).max())), (k, err)
    loss, el, il = LS.IntMSEloss(args)(out, batch)
    assert abs(float(loss) - float(fx['loss'])) < 1e-5 and abs(float(el) - float(fx['loss_ens'])) < 1e-5
    loss.backward()
    named = dict(model.named_parameters())
    for name, ref in fx.group('grad').items():
        g = named[name].grad
        g = np.zeros(ref.shape, np.float32) if g is None else g.cpu().numpy()
        tol = 1e-6 + 2e-4 * float(np.abs(ref).max())
        assert float(np.abs(g - ref).max()) <= tol, name


def test_builtin_generator_keep_rate_and_eval_mode():
    dev = _dev()
    fx = Fixture('dropout')
    model, args = build_model(fx, dev)
    batch = fx.batch(dev)
    model.eval()
    with torch.no_grad():
        e1 = model(batch)['ens_score'].clone()
        e2 = model(batch)['ens_score'].clone()
    assert torch.equal(e1, e2)                     # evaluation: no dropout, deterministic
    model.train()
    torch.manual_seed(1)
    t1 = model(batch)['ens_score'].detach().clone()
    t2 = model(batch)['ens_score'].detach().clone()
    assert float((t1 - t2).abs().max()) > 1e-4     # a fresh mask per forward
    assert float((t1 - e1).abs().max()) > 1e-4
    torch.manual_seed(1)
    t3 = model(batch)['ens_score'].detach().clone()
    assert torch.equal(t1, t3)                     # reproducible under torch.manual_seed


def test_engine_trains_int_mse_with_dropout():
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.engine import IntELEngine
    from intel_sigir2023_amd.model import IntEL
    dev = _dev()
    args = synth.make_args('tiny', dev, dropout=0.5)
    corpus, _ = synth.make_corpus('tiny')
    model = IntEL(args, corpus).to(dev)
    model.train()
    eng = IntELEngine(model, 'IntMSEloss', args)
    batch = synth.make_batch('tiny', 32, dev, seed=5, ragged=True)
    losses = [float(eng.train_step(batch)[0]) for _ in range(30)]
    assert all(np.isfinite(losses)) and np.mean(losses[-5:]) < np.mean(losses[:5])


@pytest.mark.parametrize('layers,heads', [(1, 1), (2, 2)])
def test_wide_towers_with_pinned_dropout_match_the_oracle_autograd(layers, heads):
    """Dropout at the benchmarked widths (item tower 128, score tower 64): the kernel-per-op tower pipeline with the masked add + LayerNorm, and in the
    backward the one-pass linear backward (pair.hip) fed with dZ * mask -- outputs, IntMSEloss and every parameter gradient against the oracle's autograd
    with the SAME keep flags (oracle.forward(dropout_keep=...): the reference's nn.Dropout draw pinned, tests/golden/intel_dropout.npz pins the oracle)."""
    from intel_sigir2023_amd import loss as LS
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.model import IntEL
    from oracle import intel_oracle as O
    from tests.helpers import KernelTrace
    dev = _dev()
    name = 'drop_wide_%d_%d' % (layers, heads)
    synth.WORKLOADS[name] = dict(flags=dict(synth.WORKLOADS['tmall']['flags'], num_layers=layers, num_heads=heads),
                                 corpus=dict(items=3000, users=300, classes=40, ctx=50, I=30), batch=dict(L=50, H=20))
    torch.manual_seed(7 + layers)
    args = synth.make_args(name, dev, dropout=0.5, cal_diversity=0)
    corpus, c = synth.make_corpus(name)
    model = IntEL(args, corpus).to(dev).train()
    B, L = 9, 50
    batch = synth.make_batch(name, B, dev, seed=3, ragged=True)
    g = torch.Generator().manual_seed(11)
    d_i, d_s = args.i_emb_size + args.im_emb_size, args.s_emb_size
    keep_i = [(torch.rand(B, L, d_i, generator=g) > 0.5).float() for _ in range(layers)]
    keep_s = [(torch.rand(B, L, d_s, generator=g) > 0.5).float() for _ in range(layers)]
    model._dropout_keep = torch.cat([k.reshape(-1) for k in keep_i + keep_s]).to(dev)
    with KernelTrace() as kt:
        out = model(batch)
        loss, _, _ = LS.IntMSEloss(args)(out, batch)
        loss.backward()
    kt.check(['linear_bwd_pair_kernel', 'dropout_mask_kernel'], ['tw32_fwd_kernel', 'tower_bwd_fused_kernel'])
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    ref_batch = synth.to_reference_layout(batch, c['I'])
    cfg = O.Config(**{k: v for k, v in vars(args).items() if k not in ('device', 'dtype')})
    ref = O.forward(sd, ref_batch, cfg, dropout_keep=(keep_i, keep_s))
    rl = O.int_mse_loss(ref, ref_batch, cfg)
    rl[0].backward()
    for k in ('weights', 'ens_score', 'intents'):
        err = float((out[k].detach().cpu() - ref[k].detach()).abs().max())
        assert err <= 3e-5 * max(1.0, float(ref[k].detach().abs().max())), (k, err)
    assert abs(float(loss) - float(rl[0])) < 1e-5
    for k, p in model.named_parameters():
        r = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        gg = p.grad.cpu() if p.grad is not None else torch.zeros_like(r)
        tol = 1e-6 + 2e-4 * float(r.abs().max())
        assert float((gg - r).abs().max()) <= tol, (k, float((gg - r).abs().max()), tol)
