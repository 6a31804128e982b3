"""The fused BERT4Rec encoder kernels (csrc/enc.hip, csrc/enc_bwd.hip): models/GeneralSeq.py:89-106 on packed history rows as
tiles of whole sessions.  Through the model (C ABI) against the oracle: outputs, the IntBPRloss value and every parameter gradient
(the oracle's autograd), on batches with enough sessions for many row tiles -- histories of 1 .. history_max rows, sessions of exactly
16 / 17 rows (one / two query tiles), the longest history at a tile boundary, history_max 7 / 20 / 32 (tile windows of 58 / 45 / 33
rows).  The switch INTEL_ENC_FUSED is read once per process: the kernel-per-op encoder is re-checked in a child process."""
import os
import subprocess
import sys

import pytest
import torch

from tests.helpers import ROOT, relu_flip_forgiven_error

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch.device('cuda:0')


def _case(history_max, B, seed, train, dtype='f32', L=12):
    from intel_sigir2023_amd import loss as LS
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.model import IntEL
    from oracle import intel_oracle as O
    dev = _dev()
    name = 'enc%d_%d' % (history_max, L)
    w = dict(synth.WORKLOADS['tmall'])
    w['flags'] = dict(w['flags'], history_max=history_max)
    w['corpus'] = dict(items=5000, users=500, classes=60, ctx=100, I=30)
    w['batch'] = dict(L=L, H=history_max)
    synth.WORKLOADS[name] = w
    torch.manual_seed(seed)
    args = synth.make_args(name, dev, cal_diversity=0, dtype=dtype)
    bf = dtype == 'bf16'
    oracle_forward = O.forward_bf16 if bf else O.forward      # bf16 mode: the emulating oracle, forward and autograd (tests/test_bf16_gpu.py)
    corpus, c = synth.make_corpus(name)
    model = IntEL(args, corpus).to(dev)
    sd = {k: v.detach().cpu().clone().requires_grad_(train) for k, v in model.state_dict().items()}
    batch = synth.make_batch(name, B, dev, seed=seed, ragged=True)
    # pin the interesting lengths
    H = history_max
    for key in ('history_len', 'history_item_len'):
        ln = batch[key]
        for i, v in ((0, H), (1, 1), (2, min(H, 16)), (3, min(H, 17)), (B - 1, H)):
            if 0 <= i < B:
                ln[i] = v
    hv = torch.arange(H, device=dev)[None, :] < batch['history_len'][:, None]
    hiv = torch.arange(H, device=dev)[None, :] < batch['history_item_len'][:, None]
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    batch['his_intents'] = torch.softmax(torch.rand(B, H, 30, generator=g, device=dev), dim=-1) * hv[:, :, None]
    batch['his_context_mh'] = (torch.randint(0, 100, (B, H), generator=g, device=dev) * hv).int()
    batch['his_item_id'] = (torch.randint(1, 5000, (B, H), generator=g, device=dev) * hiv).int()
    batch['his_item_idx'] = torch.where(hiv, torch.randint(0, 30, (B, H), generator=g, device=dev), torch.full((B, H), -1, device=dev)).int()
    batch['his_rows'], batch['hisitem_rows'] = int(batch['history_len'].sum()), int(batch['history_item_len'].sum())
    ref_batch = synth.to_reference_layout(batch, c['I'])
    cfg = O.Config(**{k: v for k, v in vars(args).items() if k not in ('device', 'dtype')})
    L = batch['i_id_s'].shape[1]
    noise = torch.rand(B, L, L, device=dev)
    batch['bpr_noise'] = noise
    if train:
        model.train()
        out = model(batch)
        assert 'his_off' in model.prepare_batch(batch)[1]          # the encoders do run packed
        loss, _, _ = LS.IntBPRloss(args)(out, batch)
        loss.backward()
        taps = {}
        ref = oracle_forward(sd, ref_batch, cfg) if bf else O.forward(sd, ref_batch, cfg, taps=taps)
        rl = O.int_bpr_loss(ref, ref_batch, cfg, noise.cpu())
        rl[0].backward()
        assert abs(float(loss) - float(rl[0])) < (2e-5 if bf else 1e-5)
    else:
        model.eval()
        with torch.no_grad():
            out = model(batch)
            ref = oracle_forward(sd, ref_batch, cfg)
    flipped = torch.zeros(B, dtype=torch.bool)
    for k in ('weights', 'ens_score', 'intents'):
        e = (out[k].detach().cpu() - ref[k].detach()).abs().float().reshape(B, -1).max(dim=1)[0]      # per session
        scale = max(1.0, float(ref[k].abs().max()))
        if bf:
            # bf16 mode against its emulation: equal to summation order (<= 3e-5 like fp32) EXCEPT in a session where a pre-rounding value sits
            # on a bf16 boundary and the two summation orders round it to different neighbours: one such flip moves that SESSION's outputs
            # by up to ~2^-9 of an activation (sessions are independent: nothing else moves).  Counted below per session, not per element.
            assert float(e.max()) <= 2e-3 * scale, (k, float(e.max()))
            flipped |= e > 3e-5 * scale
        else:
            assert float(e.max()) <= 3e-5 * scale, (k, float(e.max()))
    if bf:
        # how many sessions may carry a visible flip.  A session rounds R ~ 2.5e4 activations to bf16 (two encoders x ~10 packed rows x 128
        # columns x 6 rounded tensors + the towers' 12 x 192 x 5); two fp32 summation orders of a 128-term product differ by ~2^-22 of the
        # value against the bf16 half-spacing 2^-9: p ~ 2^-13 per element, i.e. ~3 boundary flips per session -- but a flipped element is
        # one of 128 terms of the next product (2^-9 / 128 of its output, far below 3e-5); only a flip in the last few values of a session's
        # chain (the 2 x 128 encoder outputs and the pooled / fused vectors, ~5e2 values) is visible: 5e2 x 2^-13 ~ 0.06 per session.  The
        # bound is that expectation plus two sessions; measured on the GPU box (round 4): 1 of 130, 2 of 70, 5 of 257 sessions.
        n_flip = int(flipped.sum())
        print('bf16 fused-encoder case H=%d B=%d: %d sessions with a visible rounding flip' % (history_max, B, n_flip))
        assert n_flip <= 2 + int(0.06 * B), (n_flip, B)
    if train:
        worst = 0.0
        for k, p in model.named_parameters():
            gr = sd[k].grad
            if gr is None:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
                continue
            tol = (5e-3 if bf else 2e-4) * max(float(gr.abs().max()), 1e-6) + 1e-7      # bf16 mode: the bar of test_bf16_gradients_match_the_emulating_oracle
            err = float((p.grad.detach().cpu() - gr).abs().max())
            if not bf and err > tol:
                # one flipped relu moves one row of a first-linear gradient: forgiven only where the ORACLE's pre-activation of that hidden
                # unit touches zero, at most two units (these batches hold up to 257 sessions), each within 50x (tests/helpers.py)
                err = relu_flip_forgiven_error(k, p.grad.detach().cpu(), gr, tol, taps, max_units=2)
            worst = max(worst, err / tol)
            assert err <= tol, (k, err, tol)
        return worst
    return 0.0


@pytest.mark.parametrize('history_max,B,seed', [(20, 257, 0), (32, 130, 1), (7, 300, 2), (20, 16, 3), (20, 2, 4), (17, 64, 5)])
def test_fused_encoder_inference_matches_oracle(history_max, B, seed):
    _case(history_max, B, seed, train=False)


@pytest.mark.parametrize('history_max,B,seed', [(20, 257, 10), (32, 130, 11), (7, 200, 12), (20, 5, 13)])
def test_fused_encoder_training_step_matches_oracle_autograd(history_max, B, seed):
    _case(history_max, B, seed, train=True)


@pytest.mark.parametrize('history_max,B,seed,train', [(20, 130, 20, True), (32, 70, 21, True), (20, 257, 22, False)])
def test_fused_encoder_bf16_mode_matches_the_emulating_oracle(history_max, B, seed, train):
    """--dtype bf16: the NP = 1 variants of the forward AND backward kernels (hi planes only, single bf16 MFMAs in the attention) against
    oracle.forward_bf16 and its autograd, which restate the mode's rounding points."""
    _case(history_max, B, seed, train=train, dtype='bf16')


def test_kernel_per_op_encoder_when_forced_off():
    env = dict(os.environ, INTEL_ENC_FUSED='0')
    r = subprocess.run([sys.executable, '-m', 'pytest', 'tests/test_enc_gpu.py', 'tests/test_pack_gpu.py', '-m', 'gpu', '-x', '-q', '-p', 'no:cacheprovider',
                        '-k', 'not forced_off and not key_tile_count'], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]


@pytest.mark.parametrize('dtype', ['bf16', 'f32'])
@pytest.mark.parametrize('L,seed', [(5, 30), (16, 31), (17, 32), (33, 33), (48, 34), (64, 35)])
def test_training_step_at_every_key_tile_count(L, seed, dtype):
    """Lists of 1 .. 4 key tiles of 16 (full and ragged last tile) through the towers' kernels of either mode -- in bf16 mode the one-kernel tower
    backward (tower_bwd.hip), whose attention works on 32-key blocks: an odd tile count leaves half a block that must read as zeros (round 5: it
    read whatever the LDS slot held, NaN patterns included, at lists of <= 16 and 33 .. 48).  bf16 mode on 96 sessions: its 5e-3 bar is about one
    rounding flip in a SUM over sessions and list positions (33 sessions of 5 items: 5.5e-3 on an encoder weight, with either tower backward)."""
    _case(20, 96 if dtype == 'bf16' else 33, seed, train=True, dtype=dtype, L=L)
