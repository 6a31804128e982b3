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
