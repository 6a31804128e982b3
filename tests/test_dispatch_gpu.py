"""Which kernel family serves which shape under the DEFAULT policy, asserted from the library's own launch records
(tests/helpers.py: KernelTrace over csrc/prof.cpp).  Dispatch is decided by shape / batch predicates in csrc/model.cpp
(tower_fused_supported / tower_fused_wanted, tower_bwd_fused_*, enc_fused_supported, attn_p3_supported, head_fused_ok, ...):
every case below checks parity against the oracle's autograd AND that the kernels the case exists for really produced the
numbers -- flipping a predicate turns these red instead of silently re-testing the kernel-per-op pipeline."""
import importlib.util
import os
import random

os.environ.setdefault('INTEL_GRU_ORDER_MIN_B', '1')

import pytest
import torch

from tests.helpers import KernelTrace

pytestmark = pytest.mark.gpu

_spec = importlib.util.spec_from_file_location('fuzz_parity', os.path.join(os.path.dirname(__file__), '..', 'tools', 'fuzz_parity.py'))
fuzz = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(fuzz)

# the benchmarked widths: item tower 64 + 64 = 128, score tower 64, encoders 64 + 64 = 128
W64 = dict(i_emb_size=64, im_emb_size=64, s_emb_size=64, u_emb_size=64, intent_emb_size=64, context_emb_size=64, cross_attn_qsize=64)


def _run(force, idx):
    tr = {}
    worst, bad, desc = fuzz.one_case(random.Random(idx), idx, torch.device('cuda:0'), big=False, force=force, trace=tr)
    assert worst <= 1.0, (worst, bad, desc)
    kt = KernelTrace.__new__(KernelTrace)
    kt.names, kt.count = tr['names'], tr['count']
    return kt, desc


@pytest.mark.parametrize('L,B', [(50, 17), (64, 5), (20, 33)])
def test_headline_widths_short_lists(L, B):
    """Tmall-shape widths, lists <= 64, histories <= 32, fp32: the 64-wide score tower's forward as one-kernel layers (tower.hip), the 128-wide item
    tower and both backward passes on the kernel-per-op pipeline (policy: tower_fused_wanted, tower_bwd_fused_wanted -- the one-kernel backward is
    the bf16 mode's) with each feed-forward linear's and each fused q/k/v projection's data + weight gradient in one pass (pair.hip: no K > 128 row GEMM left), fused BERT4Rec blocks, forward chain launches, the pooling with the folded LayerNorm tail, weight gradients / row GEMMs on
    the bf16 pipe."""
    kt, desc = _run(dict(W64, L=L, B=B, I=30, num_heads=1, num_layers=1, encoder='BERT4Rec', history_max=20, model_num=3, loss='IntBPRloss',
                         cross_attention=1, cal_diversity=0), 9100 + L)
    kt.check(['tower_fwd_fused_kernel', 'attn_seq_fwd_kernel', 'attn_seq_bwd_fused_kernel', 'enc_block_fwd_kernel', 'enc_last_fwd_kernel',
              'enc_block_bwd_kernel', 'enc_last_bwd_kernel', 'chain_kernel', 'xatt_pool_fwd_reg_kernel', 'xatt_pool_ln_bwd_reg_kernel', 'wgrad_b3_kernel',
              'linear_bwd_pair_kernel', 'linear_bwd_qkv_kernel'],
             ['attn_fwd_kernel', 'tw32_fwd_kernel', 'enc32_fwd_kernel', 'gru_seq_fwd_kernel', 'tower_bwd_fused_kernel', 'gemm_rows_b3k_kernel'], desc)
    # (q/k/v one-pass backward: one per tower + per encoder one for the full block and one for the pruned last block's k/v)
    assert kt.count['tower_fwd_fused_kernel'] == 1 and kt.count['attn_seq_bwd_fused_kernel'] == 2 and kt.count['linear_bwd_pair_kernel'] == 4 and \
        kt.count['linear_bwd_qkv_kernel'] == 6, kt.count


@pytest.mark.parametrize('B,bwd_chains', [(1000, True), (1100, False)])
def test_backward_chain_launches_stop_at_1024_sessions(B, bwd_chains):
    """64 / 128-wide towers: the session head's BACKWARD runs as chain launches up to 1024 sessions per step (head_fused_ok; 768 before the
    weight-gradient chains moved behind the score tower in round 5), as one launch per link above; the forward chains run at any batch.  Parity on both sides."""
    kt, desc = _run(dict(W64, L=7, B=B, I=30, num_heads=1, num_layers=1, encoder='BERT4Rec', history_max=5, model_num=3, loss='IntListloss',
                         cross_attention=1, cal_diversity=0), 9600 + B)
    kt.check(['chain_kernel', 'tower_fwd_fused_kernel'], ['tw32_fwd_kernel'], desc)
    assert (kt.count['chain_kernel'] > 2) == bwd_chains, (kt.count['chain_kernel'], desc)      # two forward chain launches in either case


def test_headline_widths_two_layers_two_heads():
    kt, desc = _run(dict(W64, L=50, B=9, I=30, num_heads=2, num_layers=2, encoder='BERT4Rec', history_max=20, model_num=3, loss='IntListloss',
                         cross_attention=1, cal_diversity=1), 9200)
    # fp32: both backward passes on the kernel-per-op pipeline by policy (two whole-sequence attention backward launches per tower: two tied layers)
    kt.check(['tower_fwd_fused_kernel', 'attn_seq_bwd_fused_kernel'], ['tw32_bwd_kernel', 'tower_bwd_fused_kernel'], desc)


def test_long_lists_take_the_plane_attention_kernels():
    """LifeData shape (lists of 100): general attention on the bf16 pipe at fp32 accuracy (attn_p3.hip), no one-kernel tower layer."""
    kt, desc = _run(dict(W64, L=100, B=5, I=10, num_heads=1, num_layers=1, encoder='BERT4Rec', history_max=20, model_num=5, loss='IntBPRloss',
                         cross_attention=1, cal_diversity=0), 9300)
    kt.check(['attn_fwd_p3_kernel', 'attn_bwd_dkv_p3_kernel', 'attn_bwd_dq_ds_p3_kernel', 'enc_block_fwd_kernel'],
             ['tower_fwd_fused_kernel', 'tower_bwd_fused_kernel', 'attn_seq_fwd_kernel', 'attn_fwd_kernel'], desc)


def test_long_histories_leave_the_fused_encoder():
    kt, desc = _run(dict(W64, L=20, B=5, I=10, num_heads=1, num_layers=1, encoder='BERT4Rec', history_max=70, model_num=3, loss='IntBPRloss',
                         cross_attention=1, cal_diversity=0), 9400)
    kt.check(['attn_fwd_p3_kernel', 'attn_lastq_fwd_kernel'], ['enc_block_fwd_kernel', 'enc_block_bwd_kernel'], desc)


def test_gru4rec_recurrence_kernels():
    kt, desc = _run(dict(W64, L=50, B=19, I=30, num_heads=1, num_layers=1, encoder='GRU4Rec', history_max=20, model_num=3, loss='IntBPRloss',
                         cross_attention=1, cal_diversity=0), 9500)
    kt.check(['gru_seq_fwd_kernel', 'gru_seq_bwd_kernel'], ['gru_gate_fwd_kernel', 'enc_block_fwd_kernel', 'tower_bwd_fused_kernel'], desc)


def test_gru_output_projection_with_one_chain_capable_width():
    """context_emb_size 8 makes encoder 0 24 wide (not a multiple of 16) and encoder 1 32 wide.  The head's chains take over
    the GRU output projections only as a pair: with one width unsuitable BOTH encoders must compute their own projection (csrc/model.cpp:
    forward_impl) -- outputs, loss and every gradient against the oracle."""
    kt, desc = _run(dict(i_emb_size=16, im_emb_size=16, s_emb_size=32, u_emb_size=32, intent_emb_size=16, context_emb_size=8, cross_attn_qsize=32,
                         L=20, B=7, I=30, num_heads=1, num_layers=1, encoder='GRU4Rec', history_max=20, model_num=3, loss='IntBPRloss',
                         cross_attention=1, cal_diversity=1), 9600)
    kt.check(['gru_seq_fwd_kernel', 'chain_kernel'], [], desc)


def test_bf16_mode_kernels():
    """--dtype bf16: the one-kernel tower layers in both directions at one plane (backward: dZ -> dX at both widths, the 64-wide tower's q/k/v weight
    gradients inside too), the transposing-read weight gradient for what is left."""
    from intel_sigir2023_amd import loss as LS
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.model import IntEL
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    over = dict(items=5000, users=500)
    args = synth.make_args('tmall', dev, dtype='bf16')
    corpus, _ = synth.make_corpus('tmall', **over)
    model = IntEL(args, corpus).to(dev).train()
    batch = synth.make_batch('tmall', 64, dev, seed=2, ragged=True, corpus_over=over)
    with KernelTrace() as kt:
        out = model(batch)
        loss, _, _ = LS.IntListloss(args)(out, batch)
        loss.backward()
    assert bool(torch.isfinite(loss))
    kt.check(['tower_fwd_fused_kernel', 'tower_bwd_fused_kernel', 'wgrad_tr_kernel', 'enc_block_fwd_kernel', 'enc_block_bwd_kernel', 'chain_kernel'],
             ['attn_seq_bwd_fused_kernel', 'attn_seq_fwd_kernel'], 'bf16 mode, Tmall shape')      # (round 5: the session head's chains in bf16 mode too, their 64 / 128-deep links rounding like the bf16 pipe)
    assert kt.count['tower_fwd_fused_kernel'] == 2 and kt.count['tower_bwd_fused_kernel'] == 2, kt.count


@pytest.mark.parametrize('L,B,heads', [(50, 19, 1), (64, 3, 2), (7, 40, 1)])
def test_inference_builds_the_tower_inputs_inside_the_one_kernel_layer(L, B, heads):
    """Evaluation at the benchmarked widths: the item tower's candidate rows go from the embedding tables (id | class embeddings, IntEL.py:170-173)
    straight into the first layer's LDS tile (tower.hip: TowerInput) -- no gather_rows launch for the tower, no [B*L, d] input tensor -- and the outputs
    still equal the oracle's forward."""
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.model import IntEL
    from oracle import intel_oracle as O
    dev = torch.device('cuda:0')
    name = 'gather%d_%d' % (L, heads)
    synth.WORKLOADS[name] = dict(flags=dict(synth.WORKLOADS['tmall']['flags'], num_heads=heads), corpus=dict(items=3000, users=300, classes=40, ctx=50, I=30),
                                 batch=dict(L=L, H=20))
    torch.manual_seed(L + heads)
    args = synth.make_args(name, dev)
    corpus, c = synth.make_corpus(name)
    model = IntEL(args, corpus).to(dev).eval()
    batch = synth.make_batch(name, B, dev, seed=L, ragged=True)
    with KernelTrace() as kt, torch.no_grad():
        out = model(batch)
    kt.check(['tower_fwd_fused_kernel'], ['tw32_fwd_kernel', 'attn_seq_fwd_kernel'])
    assert kt.count['tower_fwd_fused_kernel'] == 2, kt.count
    # what is left of the gathers belongs to the encoders: the context / item-id rows of the two histories
    assert kt.count.get('gather_rows_kernel', 0) <= 2, kt.count
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = O.Config(**{k: v for k, v in vars(args).items() if k not in ('device', 'dtype')})
    with torch.no_grad():
        ref = O.forward(sd, synth.to_reference_layout(batch, c['I']), cfg)
    for k in ('weights', 'ens_score', 'intents'):
        err = float((out[k].cpu() - ref[k]).abs().max())
        assert err <= 3e-5 * max(1.0, float(ref[k].abs().max())), (k, err)
