"""The one-kernel 32-wide tower (csrc/tower32.hip: models/IntEL/IntEL.py:182-197 at the reference's default / published widths) and the
session-head chain launches (csrc/chain.hip: IntEL.py:147-153, 201-215) through the model (C ABI) against the oracle's autograd:
outputs, the Int* loss and EVERY parameter gradient, at the shapes that matter for these kernels -- lists of 2 / 16 / 17 / 33 / 50 /
64 / 65 / 90 / 96 candidates (1 .. 6 row tiles, tile boundaries), 1 and 2 heads, 1 .. 3 tied layers, both encoders, batches that are
not a multiple of the chains' 16 sessions per workgroup; inference also at 128 candidates (training: kernel-per-op pipeline there)."""
import importlib.util
import os
import random

os.environ.setdefault('INTEL_GRU_ORDER_MIN_B', '1')

import pytest
import torch

pytestmark = pytest.mark.gpu

_spec = importlib.util.spec_from_file_location('fuzz_parity', os.path.join(os.path.dirname(__file__), '..', 'tools', 'fuzz_parity.py'))
fuzz = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(fuzz)


def _check(tr, present, absent, what):
    from tests.helpers import KernelTrace
    kt = KernelTrace.__new__(KernelTrace)
    kt.names = tr['names']
    kt.check(present, absent, what)


W32 = dict(i_emb_size=16, im_emb_size=16, s_emb_size=32, u_emb_size=32, intent_emb_size=32, context_emb_size=64, cross_attn_qsize=32)


@pytest.mark.parametrize('L,heads,layers,B,encoder,loss', [
    (2, 1, 1, 3, 'BERT4Rec', 'IntBPRloss'), (16, 2, 2, 5, 'GRU4Rec', 'IntBPRloss'), (17, 1, 2, 17, 'GRU4Rec', 'IntListloss'),
    (33, 2, 1, 16, 'BERT4Rec', 'IntMSEloss'), (50, 2, 2, 33, 'GRU4Rec', 'IntBPRloss'), (64, 1, 3, 4, 'BERT4Rec', 'IntListloss'),
    (65, 2, 2, 7, 'GRU4Rec', 'IntBPRloss'), (90, 1, 1, 6, 'BERT4Rec', 'IntBPRloss'), (96, 2, 1, 3, 'GRU4Rec', 'IntMSEloss'),
    (128, 2, 2, 3, 'GRU4Rec', 'IntBPRloss')])
def test_reference_widths_match_oracle_autograd(L, heads, layers, B, encoder, loss):
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    force = dict(W32, L=L, B=B, I=30, num_heads=heads, num_layers=layers, encoder=encoder, history_max=20, model_num=3, loss=loss,
                 cross_attention=1, cal_diversity=1)
    tr = {}
    worst, bad, desc = fuzz.one_case(random.Random(L * 100 + heads * 10 + layers), 7000 + L, torch.device('cuda:0'), big=False, force=force, trace=tr)
    assert worst <= 1.0, (worst, bad, desc)
    # ... and it was the one-kernel tower + the chain launches that produced these numbers (training lists of 97 .. 128 candidates: kernel-per-op pipeline)
    if L <= 96:
        _check(tr, ['tw32_fwd_kernel', 'tw32_bwd_kernel', 'chain_kernel'], ['tower_fwd_fused_kernel'], desc)
        assert tr['count']['tw32_fwd_kernel'] == 2 and tr['count']['tw32_bwd_kernel'] == 2, tr['count']      # both towers
    else:
        _check(tr, ['chain_kernel'], ['tw32_fwd_kernel', 'tw32_bwd_kernel'], desc)
    _check(tr, ['gru_seq_fwd_kernel', 'gru_seq_bwd_kernel'] if encoder == 'GRU4Rec' else [], ['enc32_fwd_kernel', 'enc32_bwd_kernel'], desc)


def test_gate_variant_keeps_the_one_kernel_towers():
    """--cross_attention 0 (IntEL.py:206-209): the chains do not apply, the 32-wide towers still do."""
    force = dict(W32, L=50, B=9, I=30, num_heads=2, num_layers=2, encoder='GRU4Rec', history_max=20, model_num=3, loss='IntBPRloss',
                 cross_attention=0, cal_diversity=0)
    tr = {}
    worst, bad, desc = fuzz.one_case(random.Random(5), 7999, torch.device('cuda:0'), big=False, force=force, trace=tr)
    assert worst <= 1.0, (worst, bad, desc)
    _check(tr, ['tw32_fwd_kernel', 'tw32_bwd_kernel', 'gate_fwd_kernel', 'gate_bwd_kernel'], ['chain_kernel'], desc)


E32 = dict(i_emb_size=16, im_emb_size=16, s_emb_size=32, u_emb_size=32, intent_emb_size=16, context_emb_size=16, cross_attn_qsize=32)


@pytest.mark.parametrize('H,heads,layers,B,loss', [
    (1, 1, 1, 3, 'IntBPRloss'), (5, 2, 2, 5, 'IntMSEloss'), (6, 1, 2, 17, 'IntListloss'), (16, 2, 1, 16, 'IntBPRloss'), (16, 2, 2, 9, 'IntMSEloss'),
    (17, 1, 2, 7, 'IntBPRloss'), (20, 2, 2, 33, 'IntMSEloss'), (32, 2, 2, 4, 'IntListloss'), (32, 1, 1, 6, 'IntBPRloss'), (33, 2, 2, 5, 'IntBPRloss')])
def test_one_kernel_bert4rec_encoder_matches_oracle_autograd(H, heads, layers, B, loss):
    """The whole 32-wide BERT4Rec encoder in one kernel per direction (tower32.hip: enc32_*; models/sequential/BERT4Rec.py's blocks at the
    reference's default widths: context 16 + intent 16, item id 16 + intent 16): histories of 1 .. 32 events (one and two 16-row tiles, tile
    boundaries), 1 and 2 heads, 1 and 2 blocks, ragged lengths; 33 events fall back to the kernel-per-op encoder."""
    force = dict(E32, L=20, B=B, I=30, num_heads=heads, num_layers=layers, encoder='BERT4Rec', history_max=H, model_num=3, loss=loss,
                 cross_attention=1, cal_diversity=1)
    tr = {}
    worst, bad, desc = fuzz.one_case(random.Random(H * 100 + heads * 10 + layers), 7300 + H, torch.device('cuda:0'), big=False, force=force, trace=tr)
    assert worst <= 1.0, (worst, bad, desc)
    if H <= 32:
        _check(tr, ['enc32_fwd_kernel', 'enc32_bwd_kernel', 'tw32_fwd_kernel', 'tw32_bwd_kernel'], ['enc_block_fwd_kernel', 'attn_lastq_fwd_kernel', 'attn_seq_fwd_kernel'], desc)
    else:
        _check(tr, ['tw32_fwd_kernel', 'tw32_bwd_kernel'], ['enc32_fwd_kernel', 'enc32_bwd_kernel'], desc)


@pytest.mark.parametrize('B', [1024, 1025])
def test_one_kernel_encoder_batch_limit(B):
    """Training takes the one-kernel encoder up to 1 024 sessions per step (one gradient slab per session and block: ENC32_MAXB, tower32.hip) and
    the kernel-per-op encoder above; both sides of the limit against the oracle."""
    force = dict(E32, L=4, B=B, I=10, num_heads=2, num_layers=2, encoder='BERT4Rec', history_max=20, model_num=2, loss='IntBPRloss',
                 cross_attention=1, cal_diversity=0)
    tr = {}
    worst, bad, desc = fuzz.one_case(random.Random(B), 7400 + (B & 1), torch.device('cuda:0'), big=False, force=force, trace=tr)
    assert worst <= 1.0, (worst, bad, desc)
    if B <= 1024:
        _check(tr, ['enc32_fwd_kernel', 'enc32_bwd_kernel'], [], desc)
    else:
        _check(tr, [], ['enc32_fwd_kernel', 'enc32_bwd_kernel'], desc)


def test_one_kernel_encoder_inference_beyond_its_grid():
    """Inference has no batch limit on the one-kernel encoder (no gradient slabs): its 2 048 workgroups walk the sessions with a stride.  A batch of 2 100
    sessions must give, session by session, exactly what its two halves give (sessions are independent; published IntEL-MSE widths, small tables)."""
    from intel_sigir2023_amd import parallel, synth
    from intel_sigir2023_amd.model import IntEL
    dev = torch.device('cuda:0')
    over = dict(items=5000, users=500)
    torch.manual_seed(3)
    args = synth.make_args('tmall_pub_mse', dev)
    corpus, _ = synth.make_corpus('tmall_pub_mse', **over)
    model = IntEL(args, corpus).to(dev)
    model.eval()
    batch = synth.make_batch('tmall_pub_mse', 2100, dev, seed=5, ragged=True, corpus_over=over)
    with torch.no_grad():
        whole = {k: v.clone() for k, v in model(batch).items() if torch.is_tensor(v)}
        parts = []
        for r in range(2):
            parts.append({k: v.clone() for k, v in model(parallel.shard_batch(batch, r, 2)).items() if torch.is_tensor(v)})
    for k, v in whole.items():
        if v.dim() >= 1 and v.shape[0] == 2100:
            assert torch.equal(v, torch.cat([p[k] for p in parts])), k
