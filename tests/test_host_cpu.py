"""CPU tests of the host-side logic: runner metrics vs the reference fixture, CLI/flag contract,
state_dict compatibility, parameter-slot map, synthetic workloads, data-parallel scheme over gloo."""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from oracle import intel_oracle as O
from tests.helpers import CONFIG_NAMES, Fixture, GOLDEN, build_model


def test_runner_evaluate_method_matches_reference_fixture():
    from intel_sigir2023_amd.runner import BaseRunner
    z = np.load(GOLDEN + '/metrics.npz')
    n = int(z['n'])
    preds = [z['pred/%d' % i] for i in range(n)]
    ranks = [z['rank/%d' % i] for i in range(n)]
    pos = {k: z['pos/' + k] for k in ('c_paynum_i', 'c_favnum_i', 'c_clicknum_i')}
    res = BaseRunner.evaluate_method(preds, ranks, pos, [int(k) for k in z['topk']], ['NDCG', 'HR'], z['session_len'])
    keys = json.loads(str(z['keys']))
    assert sorted(res.keys()) == keys
    for k in keys:
        assert abs(float(res[k]) - float(z['metric/' + k])) < 1e-12, k


@pytest.mark.parametrize('name', CONFIG_NAMES)
def test_state_dict_is_interchangeable_with_the_reference(name):
    fx = Fixture(name)
    model, _ = build_model(fx, torch.device('cpu'))          # strict load of the reference state_dict
    sd = model.state_dict()
    assert sorted(sd.keys()) == sorted(fx.group('sd').keys())
    slots = model.slot_items()
    assert len({s for s, _, _ in slots}) == len(slots)
    named = dict(model.named_parameters())
    assert {n for _, n, _ in slots} == set(named.keys()), 'every parameter must map to one ABI slot'
    groups = model.customize_parameters()
    assert all('bias' in n for n, p in named.items() if any(p is q for q in groups[1]['params']))
    assert groups[1]['weight_decay'] == 0


def test_cli_flags_match_the_reference_defaults():
    from intel_sigir2023_amd.main import build_parser
    init, parser = build_parser(['--model_name', 'IntEL', '--loss_name', 'IntListloss'])
    a, _ = parser.parse_known_args([])
    # IntEL.py:17-34, GeneralSeq.py:16, BaseModel.py:26, BaseRunner.py:22-54, BaseIntloss.py:13-20, Baseloss.py:9-12
    expect = dict(encoder='BERT4Rec', context_emb_size=16, i_emb_size=16, u_emb_size=32, s_emb_size=32, im_emb_size=16,
                  intent_emb_size=16, cross_attn_qsize=32, num_heads=1, dropout=0, num_layers=1, cross_attention=1,
                  history_max=20, model_num=2, epoch=200, early_stop=10, lr=1e-3, l2=0, batch_size=256, eval_batch_size=100,
                  optimizer='Adam', topk='1,3,5', metrics='NDCG,HR', main_metric='NDCG@1', intent_weight=0.1,
                  ensemble_weight=1, kl_temp=2, kl_weight=0.5, cal_diversity=0, diversity_alpha=0.01)
    for k, v in expect.items():
        assert getattr(a, k) == v, k
    with pytest.raises(SystemExit):
        build_parser(['--model_name', 'NoSuchModel'])


def test_invalid_encoder_raises_like_the_reference():
    from intel_sigir2023_amd.model import IntEL
    from tests.helpers import make_args, make_corpus
    fx = Fixture('default')
    a = make_args(dict(fx.args, encoder='LSTM'), torch.device('cpu'))
    with pytest.raises(ValueError, match='Invalid sequence encoder'):
        IntEL(a, make_corpus(fx.shape))


def test_synthetic_batch_layout_and_oracle_consumes_it():
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.model import IntEL
    cpu = torch.device('cpu')
    b = synth.make_batch('tiny', 6, cpu, seed=1, ragged=True)
    L = synth.WORKLOADS['tiny']['batch']['L']
    assert b['i_id_s'].shape == (6, L) and b['i_id_s'].dtype == torch.int32
    assert b['scores'].dtype == torch.float64 and b['scores'].shape == (6, L, 3)
    valid = torch.arange(L)[None, :] < b['session_len'][:, None]
    assert float(b['scores'][~valid].abs().max() if (~valid).any() else 0) == 0.0          # pads are 0 (pad_sequence)
    sc = b['scores'][0][: int(b['session_len'][0])]
    assert float(sc.min()) == 0.0 and abs(float(sc.max()) - 1.0) < 1e-5                    # per-list min-max normalisation
    assert ((b['ranking'] > 0).sum(1) == 5).all()
    ref = synth.to_reference_layout(b, 30)
    assert ref['his_item_int'].shape == (6, 20, 30) and ref['i_id_s'].dtype == torch.int64
    assert ((ref['his_item_int'].sum(-1) == 1) == (b['his_item_idx'] >= 0)).all()
    args = synth.make_args('tiny', cpu)
    corpus, c = synth.make_corpus('tiny')
    torch.manual_seed(0)
    sd = IntEL(args, corpus).state_dict()
    out = O.forward(sd, ref, O.Config(**{k: v for k, v in vars(args).items() if k != 'device'}))
    assert out['ens_score'].shape == (6, L) and bool(torch.isfinite(out['ens_score']).all())
    # same seed -> same batch
    b2 = synth.make_batch('tiny', 6, cpu, seed=1, ragged=True)
    assert all(torch.equal(b[k], b2[k]) for k in b if torch.is_tensor(b[k]))


def test_reference_default_init_is_reproduced():
    """Same torch.manual_seed -> same initial weights as the reference (skipped where /root/reference is absent)."""
    ref_src = '/root/reference/IntEL/src'
    if not os.path.isdir(ref_src):
        pytest.skip('reference tree not present on this box')
    import subprocess
    import sys
    code = r'''
import sys, types, json
sys.dont_write_bytecode = True
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import numpy as np, torch
for n, v in (('object', object), ('float', float), ('int', int), ('bool', bool)):
    if not hasattr(np, n): setattr(np, n, v)
from tests.helpers import Fixture, make_args, make_corpus
from models.IntEL.IntEL import IntEL as Ref
from intel_sigir2023_amd.model import IntEL as Mine
for name in ('default', 'gru_bpr', 'noxatt'):
    fx = Fixture(name)
    a = make_args(fx.args, torch.device('cpu'))
    torch.manual_seed(5); r = Ref(a, make_corpus(fx.shape)).state_dict()
    torch.manual_seed(5); m = Mine(a, make_corpus(fx.shape)).state_dict()
    assert list(r.keys()) == list(m.keys()), name
    assert all(torch.equal(r[k], m[k]) for k in r), name
print('same-init-ok')
''' % (ref_src, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert 'same-init-ok' in out.stdout, out.stderr[-2000:]

