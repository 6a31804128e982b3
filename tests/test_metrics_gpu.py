"""evaluate_method (helpers/BaseRunner.py:56-131) on the device: intel_eval_metrics against the reference's own 25-key
result (fixture F6, tests/golden/metrics.npz: ragged lengths, negative scores that pads outrank, fav / pay counts that
split tie groups) and against the numpy restatement on random evaluation sets."""
import json

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN

pytestmark = pytest.mark.gpu


def _device_eval(preds, ranks, slen, pos, topk, metrics, exact_order=True, chunk=16):
    from intel_sigir2023_amd.runner import BaseRunner
    dev = torch.device('cuda:0')
    n = len(slen)
    width = int(max(int(np.max(slen)), max(topk)))
    lp_all = BaseRunner.label_positions(ranks, slen, width) if exact_order else None
    sums = torch.zeros(7 * len(topk), dtype=torch.float64, device=dev)
    counts = torch.zeros(3, dtype=torch.float64, device=dev)
    for lo in range(0, n, chunk):                       # batches of different padded length, like an evaluation loop
        idx = list(range(lo, min(n, lo + chunk)))
        Lb = max(int(slen[i]) for i in idx)
        e = np.zeros((len(idx), Lb), np.float32)
        r = np.zeros((len(idx), Lb), np.int32)
        lp = np.zeros((len(idx), Lb), np.int32)
        for j, i in enumerate(idx):
            m = min(int(slen[i]), len(preds[i]))
            e[j, :m] = np.asarray(preds[i])[:m]
            r[j, :m] = np.asarray(ranks[i])[:m]
            if lp_all is not None:
                lp[j, :m] = lp_all[i, :m]
        pn = None
        if pos is not None:
            pn = torch.from_numpy(np.stack([pos['c_paynum_i'][idx], pos['c_favnum_i'][idx], pos['c_clicknum_i'][idx]], 1).astype(np.int32)).to(dev)
        vals, valid = BaseRunner.evaluate_method_device(torch.from_numpy(e).to(dev), torch.from_numpy(r).to(dev),
                                                        torch.from_numpy(np.asarray(slen)[idx].astype(np.int32)).to(dev), topk, metrics,
                                                        width=width, pos_nums=pn, label_pos=torch.from_numpy(lp).to(dev) if lp_all is not None else None)
        nk = len(topk)
        w = torch.ones(len(idx), 7 * nk, dtype=torch.float64, device=dev)
        for t in range(3):
            w[:, t * nk * 2:(t + 1) * nk * 2] = valid[:, t:t + 1].double()
        sums += torch.where(w > 0, vals, torch.zeros_like(vals)).sum(0)
        counts += valid.double().sum(0)
    return BaseRunner.reduce_device_metrics(sums.cpu().numpy(), counts.cpu().numpy(), n, topk, metrics)


def test_device_metrics_match_the_reference_fixture():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    z = np.load(GOLDEN + '/metrics.npz')
    n = int(z['n'])
    preds = [z['pred/%d' % i] for i in range(n)]
    ranks = [z['rank/%d' % i] for i in range(n)]
    pos = {k: z['pos/' + k] for k in ('c_paynum_i', 'c_favnum_i', 'c_clicknum_i')}
    topk = [int(k) for k in z['topk']]
    res = _device_eval(preds, ranks, z['session_len'], pos, topk, ['NDCG', 'HR'])
    keys = json.loads(str(z['keys']))
    assert sorted(res.keys()) == keys
    for k in keys:
        assert abs(res[k] - float(z['metric/' + k])) < 1e-12, (k, res[k], float(z['metric/' + k]))


@pytest.mark.parametrize('seed,maxlen,neg', [(0, 12, False), (1, 16, True), (2, 60, True), (3, 100, False), (4, 200, True)])
def test_device_metrics_match_numpy_on_random_sets(seed, maxlen, neg):
    """Against runner.evaluate_method (the bit-for-bit numpy restatement) on random sets: short lists (fewer items than the
    cutoff: pads enter the top-k), all-negative scores (pads outrank everything), lists of the full width (no pad slots),
    several pay / fav items (the fav boundary splits a tie group of the label pre-sort)."""
    from intel_sigir2023_amd.runner import BaseRunner
    assert torch.cuda.is_available()
    rs = np.random.RandomState(seed)
    n = 57
    slen = rs.randint(1, maxlen + 1, n)
    slen[0] = maxlen
    preds, ranks = [], []
    for i in range(n):
        p = rs.randn(slen[i]).astype(np.float32)
        if neg and i % 3 == 0:
            p = -np.abs(p) - 0.1
        r = np.zeros(slen[i], np.int64)
        k = min(slen[i], rs.randint(0, 7))
        r[rs.permutation(slen[i])[:k]] = rs.randint(1, 4, k)
        if slen[i] > 3 and i % 5 == 0:
            r[-2:] = -1                              # unlabelled tail
        preds.append(p)
        ranks.append(r)
    pos = {'c_paynum_i': np.array([(r == 3).sum() for r in ranks]), 'c_favnum_i': np.array([(r == 2).sum() for r in ranks]),
           'c_clicknum_i': np.array([(r == 1).sum() for r in ranks])}
    topk = [3, 1, 5, 10]
    ref = BaseRunner.evaluate_method(preds, ranks, {k: v.copy() for k, v in pos.items()}, topk, ['NDCG', 'HR'], slen)
    for with_pos in (True, False):
        res = _device_eval(preds, ranks, slen, pos if with_pos else None, topk, ['NDCG', 'HR'])
        assert sorted(res) == sorted(ref)
        for k in ref:
            a, b = res[k], float(ref[k])
            assert (np.isnan(a) and np.isnan(b)) or abs(a - b) < 1e-12, (k, a, b, with_pos)
    # label_pos = NULL: the kernel's built-in order is the STABLE form of the label pre-sort (later list position first among
    # equal labels).  numpy's own order among equal labels is implementation-specific (x86-simd-sort on AVX-512 hosts is not
    # stable even for short rows), which is why the product passes the slots computed by the reference's call; the
    # built-in order must equal the documented rule handed in explicitly.
    width = int(max(int(slen.max()), max(topk)))
    stable = np.zeros((n, width), np.int32)
    for i in range(n):
        r = np.asarray(ranks[i])
        for l in range(slen[i]):
            stable[i, l] = int((r > r[l]).sum() + (r[l + 1:] == r[l]).sum())
    import intel_sigir2023_amd.runner as R
    orig = R.BaseRunner.label_positions
    try:
        R.BaseRunner.label_positions = staticmethod(lambda rl, sl, w: stable[:, :w])
        explicit = _device_eval(preds, ranks, slen, pos, topk, ['NDCG', 'HR'])
    finally:
        R.BaseRunner.label_positions = staticmethod(orig)
    builtin = _device_eval(preds, ranks, slen, pos, topk, ['NDCG', 'HR'], exact_order=False)
    for k in ref:
        assert (np.isnan(builtin[k]) and np.isnan(explicit[k])) or builtin[k] == explicit[k], k


def test_runner_evaluate_device_equals_numpy_flow():
    """BaseRunner.evaluate with the device metrics (default) == the reference flow (predictions to the host, numpy
    evaluate_method) on real model outputs over batches of different padded length."""
    import argparse
    from intel_sigir2023_amd import loss as LS
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.model import IntEL
    from intel_sigir2023_amd.runner import BaseRunner
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    args = synth.make_args('tiny', dev)
    p = argparse.ArgumentParser()
    BaseRunner.parse_runner_args(p)
    ra = p.parse_args(['--topk', '3,1,5,10', '--main_metric', 'NDCG@3'])
    for k, v in vars(args).items():
        setattr(ra, k, v)
    corpus, _ = synth.make_corpus('tiny')
    model = IntEL(args, corpus).to(dev)
    batches = [synth.make_batch('tiny', 33, dev, seed=50 + i, ragged=True) for i in range(3)]
    batches[1] = {k: (v[:, :v.shape[1] - 3].contiguous() if k in ('i_id_s', 'i_class_c', 'scores', 'ranking') else v) for k, v in batches[1].items()}
    batches[1]['session_len'] = batches[1]['session_len'].clamp(max=batches[1]['i_id_s'].shape[1])
    crit = LS.IntListloss(args)
    res = {}
    for flag in (1, 0):
        runner = BaseRunner(ra)
        runner.device_metrics = bool(flag)
        res[flag] = runner.evaluate(model, batches, runner.topk, runner.metrics, crit)
    assert abs(res[1][0] - res[0][0]) < 1e-6
    assert sorted(res[1][1]) == sorted(res[0][1]) and len(res[1][1]) >= 25
    for k, v in res[0][1].items():
        assert abs(res[1][1][k] - float(v)) < 1e-9, (k, res[1][1][k], v)
