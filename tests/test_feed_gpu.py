"""Columnar feed, GPU side: intel_feed_collate == the per-sample pipeline, bit for bit, when it is
handed the permutations the reference's np.random.choice draws; the device-drawn shuffle is a valid permutation of
every list and different per seed; the assembled batch drives a training step."""
import numpy as np
import pytest
import torch

from intel_sigir2023_amd import data, feed
from oracle import feed_oracle
from tests.test_feed_cpu import _Model, compare, corpus, reference_batch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device('cuda:0')


def _host(batch):
    return {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in batch.items() if not k.startswith('_')}


@pytest.mark.parametrize('phase,max_his', [('train', 20), ('dev', 20), ('test', 3), ('train', 0)])
def test_collate_kernel_matches_reference_pipeline_bit_exact(phase, max_his):
    c = corpus()
    I = len(c.zero_int)
    ds = data.Dataset(_Model(max_his, I, 3), c, phase)
    st = feed.ColumnarStore(c, phase, 3, I, max_his).to(_dev())
    n = len(ds)
    for lo in range(0, n, 9):
        idx = list(range(lo, min(n, lo + 9)))
        ref, perms = reference_batch(ds, idx, seed=7 + lo)
        got = _host(st.collate(idx, shuffle='host', perm=perms))
        compare(ref, got, I)
        ora = feed_oracle.collate(st.host, max_his, idx, perms)
        for k, v in ora.items():
            assert np.array_equal(v, got[k]), k


def test_device_shuffle_is_a_permutation_and_depends_on_the_seed():
    c = corpus()
    I = len(c.zero_int)
    st = feed.ColumnarStore(c, 'train', 3, I, 20).to(_dev())
    idx = np.arange(st.n_sessions)
    base = feed_oracle.collate(st.host, 20, idx, None)
    a = _host(st.collate(idx, shuffle='device', seed=1))
    b = _host(st.collate(idx, shuffle='device', seed=2))
    same = _host(st.collate(idx, shuffle='none'))
    for k, v in base.items():
        assert np.array_equal(v, same[k]), k
    moved = 0
    for s in range(len(idx)):
        n = int(base['session_len'][s])
        for got in (a, b):
            # the shuffled list is the stored list under ONE permutation applied to ids, classes, labels and scores alike
            order = np.argsort(got['i_id_s'][s, :n], kind='stable')
            ref_order = np.argsort(base['i_id_s'][s, :n], kind='stable')
            for k in ('i_id_s', 'i_class_c', 'ranking'):
                assert np.array_equal(got[k][s, :n][order], base[k][s, :n][ref_order]), k
            assert np.array_equal(got['scores'][s, :n][order], base['scores'][s, :n][ref_order])
            assert not got['i_id_s'][s, n:].any()
        moved += int(not np.array_equal(a['i_id_s'][s, :n], b['i_id_s'][s, :n]))
    assert moved > len(idx) // 2
    for k in ('his_intents', 'his_item_id', 'his_item_idx', 'history_len', 'intents'):
        assert np.array_equal(a[k], base[k]), k


def test_fed_batch_trains():
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.engine import IntELEngine
    from intel_sigir2023_amd.model import IntEL
    c = corpus()
    I = len(c.zero_int)
    args = synth.make_args('tiny', _dev())
    args.model_num = 3
    model = IntEL(args, c).to(_dev())
    st = feed.ColumnarStore(c, 'train', model.model_num, I, model.max_his).to(_dev())
    eng = IntELEngine(model, 'IntBPRloss', args)
    batch = st.collate(np.arange(16), shuffle='device', seed=3)
    losses = [float(eng.train_step(batch)[0]) for _ in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_cli_trains_on_a_csv_corpus_through_the_device_feed(tmp_path):
    import os
    import shutil
    from intel_sigir2023_amd import main as cli
    shutil.copytree(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'minidata'), str(tmp_path / 'minidata'))
    root = str(tmp_path) + os.sep              # a copy: the run writes the corpus cache (main.py:64-72) next to the CSVs
    res = cli.main(['--model_name', 'IntEL', '--loss_name', 'IntBPRloss', '--workload', 'tiny', '--dataset', 'minidata', '--datapath', root,
                    '--intent_note', '_multi', '--max_session_len', '100', '--model_num', '3', '--epoch', '2', '--batch_size', '16',
                    '--eval_batch_size', '16'])
    assert res and all(np.isfinite(v) for v in res.values())
    assert os.path.exists(os.path.join(root, 'minidata', 'SeqReader_100_multi.pkl'))
    # second run: the corpus comes from the cache and gives the same numbers (same seeds)
    res2 = cli.main(['--model_name', 'IntEL', '--loss_name', 'IntBPRloss', '--workload', 'tiny', '--dataset', 'minidata', '--datapath', root,
                     '--intent_note', '_multi', '--max_session_len', '100', '--model_num', '3', '--epoch', '2', '--batch_size', '16',
                     '--eval_batch_size', '16'])
    assert res2.keys() == res.keys() and all(abs(res2[k] - res[k]) < 1e-4 for k in res)
