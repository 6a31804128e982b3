"""Packed histories (IntelBatch.his_off / hisitem_off): the BERT4Rec encoders run on the valid history rows only.  The
padded positions never reach a valid row in the reference (masked keys, row-wise blocks, `seq * valid`, GeneralSeq.py:95-105),
so the packed run must reproduce the padded run: outputs bit for bit (every row's products are computed identically),
gradients up to the summation order of the weight gradients (their row tiles differ)."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import Fixture, build_model

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch.device('cuda:0')


@pytest.mark.parametrize('name', ['default', 'tmall64', 'noxatt', 'stress'])      # stress: histories of 200 -> the general attention kernels on packed rows
def test_packed_histories_equal_padded_histories(name):
    from intel_sigir2023_amd import loss as LS
    fx = Fixture(name)
    dev = _dev()
    res = {}
    for packed in (False, True):
        model, args = build_model(fx, dev)
        model.train()
        args.cal_diversity = 1
        batch = fx.batch(dev)
        if packed:
            batch['his_rows'] = int(batch['history_len'].sum())
            batch['hisitem_rows'] = int(batch['history_item_len'].sum())
            assert batch['his_rows'] < batch['his_context_mh'].numel()          # the fixture does hold padding
        batch['bpr_noise'] = torch.from_numpy(fx['bpr/noise']).to(dev)
        out = model(batch)
        loss, _, _ = LS.IntBPRloss(args)(out, batch)
        loss.backward()
        import ctypes as C
        assert bool(model._ctx) and packed == ('his_off' in model.prepare_batch(batch)[1])
        res[packed] = ({k: v.detach().cpu() for k, v in out.items()}, float(loss),
                       {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None})
    # the fused encoder kernels (csrc/enc.hip: packed rows, width 128, history <= 32) tile the rows differently from the
    # kernel-per-op pipeline the padded run takes: equal to summation order there, bit for bit everywhere else
    fused = name == 'tmall64' and os.environ.get('INTEL_ENC_FUSED', '1') != '0'
    for k in ('weights', 'ens_score', 'intents'):
        if fused:
            assert float((res[True][0][k] - res[False][0][k]).abs().max()) <= 3e-6 * max(1.0, float(res[False][0][k].abs().max())), k
        else:
            assert torch.equal(res[True][0][k], res[False][0][k]), k
    assert res[True][1] == res[False][1] or (fused and abs(res[True][1] - res[False][1]) < 1e-6)
    for k, g in res[False][2].items():
        tol = 1e-7 + (2e-5 if fused else 2e-6) * float(g.abs().max())
        assert float((res[True][2][k] - g).abs().max()) <= tol, k


def test_packed_engine_step_matches_oracle_with_short_histories():
    """The synthetic generator supplies the host totals, so IntELEngine runs packed: one training step against the oracle on
    a batch whose histories are mostly short (lengths 1..5 of 20: 85 % of the padded rows are skipped)."""
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.engine import IntELEngine
    from intel_sigir2023_amd.model import IntEL
    from oracle import intel_oracle as O
    dev = _dev()
    torch.manual_seed(2)
    over = dict(items=20000, users=2000)
    args = synth.make_args('tmall', dev, cal_diversity=1)
    corpus, c = synth.make_corpus('tmall', **over)
    model = IntEL(args, corpus).to(dev)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    batch = synth.make_batch('tmall', 40, dev, seed=7, ragged=True, corpus_over=over)
    for key in ('history_len', 'history_item_len'):
        batch[key] = (batch[key] % 5 + 1).int()
    batch['history_len'][0] = 20
    hv = torch.arange(20, device=dev)[None, :] < batch['history_len'][:, None]
    hiv = torch.arange(20, device=dev)[None, :] < batch['history_item_len'][:, None]
    batch['his_intents'] = batch['his_intents'] * hv[:, :, None]
    batch['his_context_mh'] = batch['his_context_mh'] * hv
    batch['his_item_id'] = batch['his_item_id'] * hiv
    batch['his_item_idx'] = torch.where(hiv, batch['his_item_idx'].clamp_min(0), torch.full_like(batch['his_item_idx'], -1))
    batch['his_rows'], batch['hisitem_rows'] = int(batch['history_len'].sum()), int(batch['history_item_len'].sum())
    ref_batch = synth.to_reference_layout(batch, c['I'])
    cfg = O.Config(**{k: v for k, v in vars(args).items() if k != 'device'})
    eng = IntELEngine(model, 'IntBPRloss', args, lr=1e-3, l2=1e-4)
    out, _ = eng.eval_step(batch, k=3)
    with torch.no_grad():
        ref = O.forward(sd, ref_batch, cfg)
    for k in ('weights', 'ens_score', 'intents'):
        err = float((out[k].cpu() - ref[k]).abs().max())
        assert err <= 3e-5 * max(1.0, float(ref[k].abs().max())), (k, err)
    L = batch['i_id_s'].shape[1]
    noise = torch.rand(40, L, L, device=dev)
    loss, _, _ = eng.train_step(batch, noise=noise)
    ref_loss, _, _ = O.int_bpr_loss(ref, ref_batch, cfg, noise.cpu())
    assert abs(float(loss) - float(ref_loss)) < 1e-5


def test_packed_long_histories_with_empty_and_boundary_lengths():
    """Histories of up to 200 events (the general attention kernels on packed rows) with the lengths that matter pinned: empty histories
    (no rows at all for the session), one event, exactly 64 / 65 events (one / two 64-row blocks) and the full 200: the packed run equals the
    padded run bit for bit, and everything is finite."""
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.model import IntEL
    dev = _dev()
    torch.manual_seed(0)
    over = dict(items=50000)
    args = synth.make_args('stress', dev, cal_diversity=0)
    corpus, c = synth.make_corpus('stress', **over)
    model = IntEL(args, corpus).to(dev).eval()
    b = synth.make_batch('stress', 24, dev, seed=3, ragged=True, corpus_over=over)
    H = b['his_context_mh'].shape[1]
    for key in ('history_len', 'history_item_len'):
        for i, v in ((1, 0), (5, 0), (7, 1), (9, H), (11, 65), (12, 64), (23, 0)):
            b[key][i] = v
    hv = torch.arange(H, device=dev)[None, :] < b['history_len'][:, None]
    hiv = torch.arange(H, device=dev)[None, :] < b['history_item_len'][:, None]
    b['his_intents'] = b['his_intents'] * hv[:, :, None]
    b['his_context_mh'] = b['his_context_mh'] * hv
    b['his_item_id'] = b['his_item_id'] * hiv
    b['his_item_idx'] = torch.where(hiv, b['his_item_idx'].clamp_min(0), torch.full_like(b['his_item_idx'], -1))
    outs = {}
    for packed in (False, True):
        bb = {k: v for k, v in b.items() if k not in ('his_rows', 'hisitem_rows')}
        if packed:
            bb['his_rows'], bb['hisitem_rows'] = int(b['history_len'].sum()), int(b['history_item_len'].sum())
        with torch.no_grad():
            o = model(bb)
        assert packed == ('his_off' in model.prepare_batch(bb)[1])
        outs[packed] = {k: v.clone() for k, v in o.items()}
    for k in ('weights', 'ens_score', 'intents'):
        assert bool(torch.isfinite(outs[True][k]).all()), k
        assert torch.equal(outs[True][k], outs[False][k]), (k, float((outs[True][k] - outs[False][k]).abs().max()))
    # ... and through the backward: every parameter gradient, up to the summation order of the weight gradients (their row tiles differ)
    from intel_sigir2023_amd import loss as LS
    L = b['i_id_s'].shape[1]
    noise = torch.rand(24, L, L, device=dev)
    grads = {}
    model.train()
    for packed in (False, True):
        bb = {k: v for k, v in b.items() if k not in ('his_rows', 'hisitem_rows')}
        if packed:
            bb['his_rows'], bb['hisitem_rows'] = int(b['history_len'].sum()), int(b['history_item_len'].sum())
        bb['bpr_noise'] = noise
        model.zero_grad()
        loss, _, _ = LS.IntBPRloss(args)(model(bb), bb)
        loss.backward()
        grads[packed] = (float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    assert grads[True][0] == grads[False][0]
    for k, g in grads[False][1].items():
        assert bool(torch.isfinite(grads[True][1][k]).all()), k
        assert float((grads[True][1][k] - g).abs().max()) <= 1e-7 + 2e-6 * float(g.abs().max()), k
