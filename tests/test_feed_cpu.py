"""Columnar feed, CPU side: the flattened corpus + the numpy restatement of the assembly kernel
reproduce what the per-sample pipeline (data.Dataset, pinned to the reference by tests/golden/data_feed.npz) builds."""
import argparse
import os

import numpy as np
import pytest
import torch

from intel_sigir2023_amd import data, feed
from oracle import feed_oracle

HERE = os.path.dirname(os.path.abspath(__file__))


class _Model(object):
    buffer = 0

    def __init__(self, max_his, intent_num, model_num):
        self.max_his, self.intent_num, self.model_num = max_his, intent_num, model_num


def corpus():
    ns = argparse.Namespace(datapath=os.path.join(HERE, 'golden') + os.sep, dataset='minidata', sep='\t', intent_note='_multi',
                            max_session_len=100)
    return data.SeqReader(ns)


def reference_batch(ds, idx, seed):
    """data.Dataset path; returns (collated batch, the permutations it drew)."""
    np.random.seed(seed)
    perms = [np.random.choice(np.arange(len(ds.data['i_id_s'][i])), len(ds.data['i_id_s'][i]), replace=False).astype(int) for i in idx]
    np.random.seed(seed)
    return ds.collate_batch([ds._get_feed_dict(i) for i in idx]), perms


def compare(ref, got, I):
    for k in ('i_id_s', 'i_class_c', 'ranking', 'session_len', 'u_id_c', 'context_mh', 'his_context_mh', 'history_len', 'his_item_id',
              'history_item_len'):
        r, g = np.asarray(ref[k]), np.asarray(got[k])
        assert r.shape == g.shape, (k, r.shape, g.shape)
        assert np.array_equal(r.astype(np.int64), g.astype(np.int64)), k
    for k in ('scores', 'intents', 'his_intents'):
        r = np.asarray(ref[k]).astype(np.float32)               # the model's .float()
        assert np.array_equal(r, np.asarray(got[k])), k         # bit-exact
    onehot = np.zeros(np.asarray(got['his_item_idx']).shape + (I,), dtype=np.float64)
    gi = np.asarray(got['his_item_idx'])
    for b in range(gi.shape[0]):
        for t in range(gi.shape[1]):
            if gi[b, t] >= 0:
                onehot[b, t, gi[b, t]] = 1
    assert np.array_equal(np.asarray(ref['his_item_int']), onehot)


@pytest.mark.parametrize('phase,max_his', [('train', 20), ('dev', 20), ('test', 3), ('train', 0)])
def test_columnar_store_and_oracle_match_the_per_sample_pipeline(phase, max_his):
    c = corpus()
    I = len(c.zero_int)
    ds = data.Dataset(_Model(max_his, I, 3), c, phase)
    st = feed.ColumnarStore(c, phase, 3, I, max_his)
    n = len(ds)
    assert st.n_sessions == n
    for lo in range(0, n, 7):                                   # ragged batches, every session covered
        idx = list(range(lo, min(n, lo + 7)))
        ref, perms = reference_batch(ds, idx, seed=100 + lo)
        got = feed_oracle.collate(st.host, max_his, idx, perms)
        assert st.batch_shape(np.asarray(idx)) == (ref['i_id_s'].shape[1], ref['his_context_mh'].shape[1], ref['his_item_id'].shape[1])
        compare(ref, got, I)


def test_store_requires_a_gpu_and_has_no_cpu_path():
    c = corpus()
    st = feed.ColumnarStore(c, 'dev', 3, len(c.zero_int), 20)
    with pytest.raises(Exception):
        st.collate([0, 1])
    with pytest.raises(Exception):
        st.to(torch.device('cpu'))


def test_corpus_pickle_cache_round_trip(tmp_path):
    """main.load_corpus (the reference's main.py:64-72): the first call parses the CSV / JSON files and writes
    <reader>_<max_session_len><note>.pkl next to them, the second call reads the cache; --regenerate parses again.
    The cached corpus feeds the device store exactly like the fresh one."""
    import shutil
    from intel_sigir2023_amd import main as cli
    shutil.copytree(os.path.join(HERE, 'golden', 'minidata'), str(tmp_path / 'minidata'))
    ns = argparse.Namespace(datapath=str(tmp_path) + os.sep, dataset='minidata', sep='\t', intent_note='_multi', max_session_len=100,
                            regenerate=0)
    fresh = cli.load_corpus(ns, data.SeqReader)
    path = tmp_path / 'minidata' / 'SeqReader_100_multi.pkl'
    assert path.exists()
    calls = []

    def counting_reader(a):
        calls.append(1)
        return data.SeqReader(a)
    cached = cli.load_corpus(ns, counting_reader)
    assert not calls, 'the cache was ignored'
    assert cached.pos_types == ['c_paynum_i', 'c_favnum_i', 'c_clicknum_i']
    for phase in ('train', 'dev', 'test'):
        a = feed.ColumnarStore(fresh, phase, 3, len(fresh.zero_int), 20)
        b = feed.ColumnarStore(cached, phase, 3, len(cached.zero_int), 20)
        assert a.host.keys() == b.host.keys()
        for k in a.host:
            np.testing.assert_array_equal(a.host[k], b.host[k])
    ns.regenerate = 1
    cli.load_corpus(ns, counting_reader)
    assert calls == [1]
