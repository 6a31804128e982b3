"""Input pipeline (intel_sigir2023_amd/data.py) vs the REFERENCE's SeqReader + IntEL.Dataset + collate_batch,
recorded on the synthetic mini dataset tests/golden/minidata/ by tests/golden/make_data_golden.py (fixture F8).
Integer / index fields must be bit-exact; the float fields are computed by the same numpy expressions and must
be exact too."""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN


@pytest.fixture(scope='module')
def pipeline():
    from intel_sigir2023_amd.data import SeqReader
    from intel_sigir2023_amd.model import IntEL
    z = np.load(os.path.join(GOLDEN, 'data_feed.npz'))
    cfg = json.loads(str(z['cfg']))
    cfg['datapath'] = GOLDEN
    args = argparse.Namespace(**cfg)
    args.device = torch.device('cpu')
    corpus = SeqReader(args)
    torch.manual_seed(0)
    model = IntEL(args, corpus)
    return z, args, corpus, model


def test_corpus_attributes_match_reference(pipeline):
    z, args, corpus, model = pipeline
    assert list(z['corpus/contextfnum']) == corpus.contextfnum
    assert list(z['corpus/itemfnum']) == corpus.itemfnum
    assert list(z['corpus/userfnum']) == corpus.userfnum
    assert list(z['corpus/max_ids']) == [corpus.max_uid, corpus.max_iid]
    assert int(z['corpus/intent_num']) == len(corpus.zero_int) == model.intent_num
    for phase in ('train', 'dev', 'test'):
        for key in ('position', 'item_position', 'c_id_c'):
            np.testing.assert_array_equal(z['corpus/%s/%s' % (phase, key)], np.asarray(corpus.interactions[phase][key]))
    # train lists are cut to --max_session_len, dev/test are not (BaseReader.py:75-78)
    assert max(len(x) for x in corpus.interactions['train']['i_id_s']) <= args.max_session_len
    assert max(len(x) for x in corpus.interactions['test']['i_id_s']) > args.max_session_len


def test_feed_dicts_and_collate_match_reference(pipeline):
    from intel_sigir2023_amd.data import Dataset
    z, args, corpus, model = pipeline
    np.random.seed(1234)                       # same generator state as the reference run
    n_checked = 0
    for phase in ('train', 'dev', 'test'):
        ds = Dataset(model, corpus, phase)
        ds.prepare()
        n = int(z['%s/n' % phase])
        assert len(ds) == n
        fds = [ds[i] for i in range(n)]
        for i, fd in enumerate(fds):
            ref_keys = sorted(k.split('/', 2)[2] for k in z.files if k.startswith('%s/%d/' % (phase, i)))
            assert sorted(fd.keys()) == ref_keys, (phase, i)
            for k in ref_keys:
                ref = z['%s/%d/%s' % (phase, i, k)]
                got = np.asarray(fd[k])
                assert got.shape == ref.shape, (phase, i, k)
                np.testing.assert_array_equal(got, ref, err_msg='%s/%d/%s' % (phase, i, k))
                n_checked += 1
        batch = ds.collate_batch(fds[:5])
        for k in [f.split('/', 2)[2] for f in z.files if f.startswith('%s/batch/' % phase)]:
            ref = z['%s/batch/%s' % (phase, k)]
            got = batch[k].numpy()
            assert got.dtype == ref.dtype and got.shape == ref.shape, (phase, k, got.dtype, ref.dtype)
            np.testing.assert_array_equal(got, ref, err_msg='%s/batch/%s' % (phase, k))
        assert batch['batch_size'] == 5 and batch['phase'] == phase
    assert n_checked > 500
    # edge cases present in the fixture: a session without history and a history longer than history_max
    pos = np.concatenate([z['corpus/%s/position' % p] for p in ('train', 'dev', 'test')])
    assert (pos == 0).any() and (pos > args.history_max).any()


def test_standin_intents_recipe_is_deterministic_and_normalised():
    from intel_sigir2023_amd.data import standin_intents
    root = os.path.join(GOLDEN, 'minidata')
    a = standin_intents(root)
    b = standin_intents(root)
    assert a == b and len(a) > 20
    for v in a.values():
        assert len(v) == 3 * 7 and abs(sum(v) - 1.0) < 1e-9


TOY = '/root/reference/IntEL/data/Tmall_toy'


@pytest.mark.skipif(not os.path.isdir(TOY), reason='toy Tmall sample only exists in the build container')
def test_toy_tmall_sample_matches_reference_pipeline_and_runs_batch2(tmp_path):
    """BASELINE.json configs[0] plumbing: the bundled toy Tmall sample (+ the deterministic stand-in for its
    missing intents_multi.json) through OUR reader/Dataset vs the REFERENCE's, then batch=2 through the oracle."""
    import shutil
    import subprocess
    import sys
    from intel_sigir2023_amd.data import Dataset, SeqReader, standin_intents
    from intel_sigir2023_amd.model import IntEL
    from oracle import intel_oracle as O
    root = tmp_path / 'Tmall_toy'
    root.mkdir()
    for f in ('train.csv', 'dev.csv', 'test.csv', 'item_metadata.json', 'user_metadata.json'):
        os.symlink(os.path.join(TOY, f), root / f)
    json.dump(standin_intents(str(root)), open(root / 'intents_multi.json', 'w'))
    cfg = dict(datapath=str(tmp_path), dataset='Tmall_toy', sep='\t', intent_note='_multi', max_session_len=100, model_num=3,
               history_max=20, model_path='', buffer=1, encoder='BERT4Rec', context_emb_size=16, i_emb_size=16, u_emb_size=32,
               s_emb_size=32, im_emb_size=16, intent_emb_size=16, cross_attn_qsize=32, num_heads=1, dropout=0, num_layers=1,
               cross_attention=1, intent_weight=0.1, ensemble_weight=1, kl_temp=2, kl_weight=0.5, cal_diversity=1,
               diversity_alpha=0.01)
    args = argparse.Namespace(**cfg)
    args.device = torch.device('cpu')
    corpus = SeqReader(args)
    assert corpus.max_iid == 266340 and len(corpus.zero_int) == 1071 and corpus.contextfnum == [931]
    torch.manual_seed(0)
    model = IntEL(args, corpus)
    np.random.seed(11)
    ds = Dataset(model, corpus, 'train')
    mine = [ds[i] for i in range(6)]
    # the reference on the same files / same numpy seed (separate process: it needs its own import shims)
    code = r'''
import sys, json, argparse
sys.dont_write_bytecode = True
sys.path.insert(0, %r); sys.path.insert(0, %r)
from make_golden import install_shims
install_shims()
import numpy as np, torch
from helpers.SeqReader import SeqReader
from models.IntEL.IntEL import IntEL
cfg = json.loads(%r)
args = argparse.Namespace(**cfg); args.device = torch.device('cpu')
corpus = SeqReader(args)
torch.manual_seed(0)
model = IntEL(args, corpus)
np.random.seed(11)
ds = IntEL.Dataset(model, corpus, 'train')
out = {}
for i in range(6):
    fd = ds[i]
    for k, v in fd.items():
        out['%%d/%%s' %% (i, k)] = np.asarray(v)
np.savez(%r, **out)
print('ref-ok')
''' % (os.path.join(os.path.dirname(GOLDEN), 'golden'), os.path.dirname(os.path.dirname(GOLDEN)), json.dumps(cfg), str(tmp_path / 'ref.npz'))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert 'ref-ok' in r.stdout, r.stderr[-3000:]
    ref = np.load(tmp_path / 'ref.npz')
    for i, fd in enumerate(mine):
        for k, v in fd.items():
            np.testing.assert_array_equal(np.asarray(v), ref['%d/%s' % (i, k)], err_msg='%d/%s' % (i, k))
    # batch = 2 through the oracle (the CPU-runnable configuration of the reference)
    batch = ds.collate_batch(mine[:2])
    ocfg = O.Config(**{k: v for k, v in cfg.items() if k not in ('datapath', 'dataset', 'sep', 'intent_note', 'max_session_len')})
    with torch.no_grad():
        out = O.forward(model.state_dict(), batch, ocfg)
        loss, _, _ = O.int_list_loss(out, batch, ocfg)
    assert out['ens_score'].shape[0] == 2 and bool(torch.isfinite(loss))
