"""GPU parity of the fast path (IntELEngine: flat buckets + fused Adam + on-device NDCG) and of the loss /
NDCG kernels on larger random shapes, against the CPU oracle on identical inputs."""
import numpy as np
import pytest
import torch

from oracle import intel_oracle as O
from tests.helpers import Fixture, build_model
from tests.test_oracle_golden import check_adam_result

pytestmark = pytest.mark.gpu


def per_session_ndcg(ens, ranking, slen, k):
    """NDCG@k of every session with the padding width evaluate_method uses for the WHOLE set
    (max(max(session_len), k), helpers/BaseRunner.py:66): each session is evaluated together with a
    full-length dummy list so that the width is fixed, then the dummy's share of the mean is removed."""
    Lm = ens.shape[1]
    dummy_e = np.linspace(1.0, 2.0, Lm, dtype=np.float32)[None, :]
    dummy_r = np.zeros((1, Lm), dtype=np.int64)
    dummy_r[0, 0] = 1
    d = O.ndcg_at_k(dummy_e, dummy_r, np.array([Lm]), k)
    out = []
    for i in range(ens.shape[0]):
        m = O.ndcg_at_k(np.concatenate([ens[i:i + 1], dummy_e]), np.concatenate([ranking[i:i + 1], dummy_r]),
                        np.array([int(slen[i]), Lm]), k)
        out.append(2 * m - d)
    return np.array(out)


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch.device('cuda:0')


@pytest.mark.parametrize('name', ['default', 'gru_bpr', 'noxatt'])
def test_two_phase_backward_equals_whole_backward(name):
    """intel_backward_phase(1) + (2) with the item-id table's Adam sweep on the side stream (the default, data-parallel
    overlap path) == ONE intel_backward followed by the three Adam sweeps (overlap_table_update = False: the path
    bench.py's roofline steps and INTEL_OVERLAP_TABLE=0 take), bit for bit apart from the atomically accumulated
    embedding rows."""
    from intel_sigir2023_amd.engine import IntELEngine
    fx = Fixture(name)
    dev = _dev()
    res = []
    for phased in (False, True):
        model, args = build_model(fx, dev)
        model.train()
        args.cal_diversity = 1
        eng = IntELEngine(model, 'IntBPRloss', args, lr=1e-3, l2=1e-4)
        eng.overlap_table_update = phased
        loss, _, _ = eng.train_step(fx.batch(dev), noise=torch.from_numpy(fx['adam/noise0']).to(dev))
        res.append((float(loss), {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}))
    assert res[0][0] == res[1][0]
    for k, v in res[0][1].items():
        if 'embeddings.weight' in k and ('iid' in k or 'uid' in k or 'context' in k or 'item_' in k):
            assert float((v - res[1][1][k]).abs().max()) < 1e-6, k       # atomics: summation order may differ
        else:
            assert torch.equal(v, res[1][1][k]), k


@pytest.mark.parametrize('name', ['default', 'gru_bpr', 'noxatt'])
def test_engine_two_steps_match_reference_adam(name):
    """IntELEngine.train_step x2 == the reference's 2 torch.optim.Adam steps (fixture F7)."""
    from intel_sigir2023_amd.engine import IntELEngine
    fx = Fixture(name)
    dev = _dev()
    model, args = build_model(fx, dev)
    model.train()
    args.cal_diversity = 1
    lr, l2 = [float(x) for x in fx['adam/lr_l2']]
    eng = IntELEngine(model, 'IntBPRloss', args, lr=lr, l2=l2)
    batch = fx.batch(dev)
    for step in range(2):
        loss, ens, itl = eng.train_step(batch, noise=torch.from_numpy(fx['adam/noise%d' % step]).to(dev))
        assert abs(float(loss) - float(fx['adam/losses'][step])) < 1e-5
    rows = fx.group('adam_rows')
    named = dict(model.named_parameters())
    for pname, ref in fx.group('adam').items():
        if pname in ('losses', 'lr_l2', 'noise0', 'noise1'):
            continue
        got = named[pname].detach().cpu()
        if pname in rows:
            got = got[torch.from_numpy(rows[pname])]
        check_adam_result(fx, pname, got.numpy(), ref, lr)
    # gradients were consumed and cleared by the fused Adam sweep
    assert all(float(g.abs().max()) == 0.0 for g in eng.buckets())


@pytest.mark.parametrize('workload,B', [('tiny', 64), ('tmall', 48), ('tmall', 37), ('lifedata', 16), ('lifedata', 5), ('toyshape', 6)])
def test_engine_step_matches_oracle_on_synthetic_workloads(workload, B):
    """The benchmark workloads' shapes (lists, histories, widths) on a 20 000-item / 2 000-user table (the full-size tables are
    covered by tests/test_fullsize_gpu.py): loss of one engine step vs the oracle, NDCG@3 on device vs
    evaluate_method, and a size-independent property: the dense Adam sweep moves EVERY table row (weight decay)
    while only touched rows carry gradient."""
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.engine import IntELEngine
    from intel_sigir2023_amd.model import IntEL
    dev = _dev()
    over = dict(items=20000, users=2000) if workload not in ('tiny', 'toyshape') else None
    torch.manual_seed(1)
    args = synth.make_args(workload, dev, cal_diversity=1)
    corpus, c = synth.make_corpus(workload, **(over or {}))
    model = IntEL(args, corpus).to(dev)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    batch = synth.make_batch(workload, B, dev, seed=5, ragged=True, corpus_over=over)
    ref_batch = synth.to_reference_layout(batch, c['I'])
    cfg = O.Config(**{k: v for k, v in vars(args).items() if k != 'device'})
    eng = IntELEngine(model, 'IntBPRloss', args, lr=1e-3, l2=1e-4)
    # eval first (same parameters as the oracle's)
    out, ndcg = eng.eval_step(batch, k=3)
    with torch.no_grad():
        ref = O.forward(sd, ref_batch, cfg)
    for k in ('weights', 'ens_score', 'intents'):
        err = float((out[k].cpu() - ref[k]).abs().max())
        assert err <= 3e-5 * max(1.0, float(ref[k].abs().max())), (k, err)
    ref_nd = per_session_ndcg(ref['ens_score'].numpy(), ref_batch['ranking'].numpy(), ref_batch['session_len'].numpy(), 3)
    np.testing.assert_allclose(ndcg.cpu().numpy(), ref_nd, atol=1e-4)
    whole = O.ndcg_at_k(ref['ens_score'].numpy(), ref_batch['ranking'].numpy(), ref_batch['session_len'].numpy(), 3)
    assert abs(float(ndcg.mean()) - whole) <= 1e-4
    L = batch['i_id_s'].shape[1]
    noise = torch.rand(B, L, L, device=dev)
    p_before = model.iid_embeddings.weight.detach().clone()
    loss, _, _ = eng.train_step(batch, noise=noise)
    ref_loss, _, _ = O.int_bpr_loss(ref, ref_batch, cfg, noise.cpu())
    assert abs(float(loss) - float(ref_loss)) < 1e-5
    moved = (model.iid_embeddings.weight.detach() != p_before).any(dim=1)
    assert float(moved.float().mean()) > 0.999          # dense semantics: decay reaches untouched rows too


@pytest.mark.parametrize('B,L,K', [(33, 50, 3), (7, 200, 8), (19, 100, 5)])
def test_loss_kernels_match_oracle_random(B, L, K):
    from intel_sigir2023_amd import loss as LS
    import argparse
    dev = _dev()
    g = torch.Generator().manual_seed(B * L + K)
    ens = torch.randn(B, L, generator=g)
    weights = torch.randn(B, L, K, generator=g)
    scores = torch.rand(B, L, K, generator=g, dtype=torch.float64)
    slen = torch.randint(3, L + 1, (B,), generator=g)
    slen[0] = L
    ranking = torch.zeros(B, L, dtype=torch.int64)
    for b in range(B):
        n = int(slen[b])
        r = torch.zeros(n, dtype=torch.int64)
        r[: min(n, 6)] = torch.tensor([3, 2, 2, 1, 1, 1])[: min(n, 6)]
        ranking[b, :n] = r[torch.randperm(n, generator=g)]
    noise = torch.rand(B, L, L, generator=g)
    intents = torch.softmax(torch.randn(B, 30, generator=g, dtype=torch.float64), -1)
    pred = torch.softmax(torch.randn(B, 30, generator=g), -1)
    a = argparse.Namespace(cal_diversity=1, diversity_alpha=0.01, intent_weight=0.1, ensemble_weight=1.0, kl_temp=2.0, kl_weight=0.5)
    cb = {'ranking': ranking, 'session_len': slen, 'scores': scores, 'intents': intents, 'bpr_noise': noise}
    db = {k: v.to(dev) for k, v in cb.items()}
    for cls, fn in ((LS.IntBPRloss, 'bpr'), (LS.IntListloss, 'pl')):
        e_d, w_d, p_d = ens.to(dev).requires_grad_(True), weights.to(dev).requires_grad_(True), pred.to(dev).requires_grad_(True)
        loss, el, il = cls(a)({'ens_score': e_d, 'weights': w_d, 'intents': p_d}, db)
        loss.backward()
        e_c, w_c, p_c = ens.clone().requires_grad_(True), weights.clone().requires_grad_(True), pred.clone().requires_grad_(True)
        cfg = O.Config(**vars(a))
        out = {'ens_score': e_c, 'weights': w_c, 'intents': p_c}
        ref, rel, ril = (O.int_bpr_loss(out, cb, cfg, noise) if fn == 'bpr' else O.int_list_loss(out, cb, cfg))
        ref.backward()
        assert abs(float(loss) - float(ref)) < 1e-5 and abs(float(el) - float(rel)) < 1e-5 and abs(float(il) - float(ril)) < 1e-5
        for got, want, nm in ((e_d.grad, e_c.grad, 'd_ens'), (w_d.grad, w_c.grad, 'd_weights'), (p_d.grad, p_c.grad, 'd_intents')):
            err = float((got.cpu() - want).abs().max())
            assert err <= 1e-6 + 2e-4 * float(want.abs().max()), (fn, nm, err)


def test_ndcg_kernel_matches_evaluate_method_on_ragged_lists():
    from intel_sigir2023_amd import _lib as L
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    B, Lm = 257, 90
    slen = torch.randint(1, Lm + 1, (B,), generator=g)
    slen[0], slen[1] = 1, 2                      # shorter than k
    ens = torch.randn(B, Lm, generator=g)         # negative scores: padded slots (0) outrank them
    ranking = torch.randint(-1, 4, (B, Lm), generator=g)
    ranking[:, 0] = 3                             # at least one positive inside every list
    for k in (1, 3, 10):
        out = torch.empty(B, dtype=torch.float32, device=dev)
        e, r, s = ens.to(dev).contiguous(), ranking.to(torch.int32).to(dev).contiguous(), slen.to(torch.int32).to(dev)
        L.check(L.lib().intel_ndcg(B, Lm, k, L.ptr(e), L.ptr(r), L.ptr(s), L.ptr(out), L.stream_ptr(dev)), 'intel_ndcg')
        ref = per_session_ndcg(ens.numpy(), ranking.numpy(), slen.numpy(), k)
        np.testing.assert_allclose(out.cpu().numpy(), ref, atol=1e-6)


def test_reference_layout_batches_from_the_input_pipeline_run_end_to_end():
    """config[0]-style plumbing on the GPU: CSV/JSON mini dataset -> SeqReader -> Dataset -> collate_batch
    (the reference's own batch layout: int64 ids, float64 scores, dense one-hot history) -> IntEL.forward +
    IntListloss on the HIP path, batch=2, against the oracle on the same batch."""
    import argparse
    import json
    import os
    from intel_sigir2023_amd import loss as LS
    from intel_sigir2023_amd.data import Dataset, SeqReader
    from intel_sigir2023_amd.model import IntEL
    from tests.helpers import GOLDEN
    dev = _dev()
    z = np.load(os.path.join(GOLDEN, 'data_feed.npz'))
    cfg = json.loads(str(z['cfg']))
    cfg['datapath'] = GOLDEN
    args = argparse.Namespace(**cfg)
    args.device = dev
    corpus = SeqReader(args)
    torch.manual_seed(0)
    model = IntEL(args, corpus).to(dev)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    np.random.seed(7)
    ds = Dataset(model, corpus, 'train')
    args.cal_diversity = 1
    crit = LS.IntListloss(args)
    ocfg = O.Config(**{k: v for k, v in vars(args).items() if k not in ('device', 'datapath', 'dataset', 'sep', 'intent_note', 'max_session_len')})
    for start in (0, 2, 4):
        batch = ds.collate_batch([ds[i] for i in range(start, start + 2)])      # batch = 2
        gbatch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
        model.train()
        out = model(gbatch)
        loss, el, il = crit(out, gbatch)
        loss.backward()
        with torch.no_grad():
            ref = O.forward(sd, batch, ocfg)
            rl, _, _ = O.int_list_loss(ref, batch, ocfg)
        for k in ('weights', 'ens_score', 'intents'):
            assert float((out[k].detach().cpu() - ref[k]).abs().max()) <= 3e-5 * max(1.0, float(ref[k].abs().max())), k
        assert abs(float(loss) - float(rl)) < 1e-5
        model.zero_grad()


def test_full_size_tmall_properties():
    """BASELINE.json configs[1] at FULL size (1M-item table, 4096 sessions, list=50): the oracle cannot run this
    in seconds, so the HIP path is checked through size-independent properties of the domain:
      1. sessions are independent: permuting the batch permutes every output row identically (bit-exact);
      2. the losses are means over sessions: loss(batch) == mean of the losses of its two halves;
      3. gradients are linear in the batch: grad(batch) == mean of the half-batch gradients;
      4. a sub-batch small enough for the oracle agrees with it (forward 3e-5, loss 1e-5);
      5. NDCG@3 in [0,1], equal to the mean of the per-half NDCG."""
    from intel_sigir2023_amd import loss as LS
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.engine import IntELEngine
    from intel_sigir2023_amd.model import IntEL
    dev = _dev()
    torch.manual_seed(0)
    args = synth.make_args('tmall', dev, cal_diversity=1)
    corpus, c = synth.make_corpus('tmall')
    model = IntEL(args, corpus).to(dev)
    B = 4096
    batch = synth.make_batch('tmall', B, dev, seed=21, ragged=True)
    L = batch['i_id_s'].shape[1]

    def take(b, idx):
        return {k: (v[idx].contiguous() if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == B else v) for k, v in b.items()}
    model.eval()
    with torch.no_grad():
        out = model(batch)
        perm = torch.randperm(B, device=dev)
        pb = take(batch, perm)
        pb['batch_size'] = B
        outp = model(pb)
    for k in ('weights', 'ens_score', 'intents'):
        assert torch.equal(outp[k], out[k][perm]), 'permutation equivariance broken for ' + k
    # 2 + 3: halves
    crit = LS.IntBPRloss(args)
    noise = torch.rand(B, L, L, device=dev)
    named = dict(model.named_parameters())
    check = ['i_W1.weight', 'i_attn_head.q_linear.weight', 's_W2.bias', 'weight_embeddings.weight', 'pred_layer.weight',
             'encoder.transformer_block.0.linear1.weight', 'item_encoder.transformer_block.1.masked_attn_head.v_linear.weight',
             'intent_embeddings.weight', 'context_embeddings.weight']

    def run(idx):
        sub = take(batch, idx)
        sub['batch_size'] = int(idx.numel())
        sub['bpr_noise'] = noise[idx].contiguous()
        model.train()
        model.zero_grad()
        o = model(sub)
        loss, el, il = crit(o, sub)
        loss.backward()
        return float(loss), float(el), {n: named[n].grad.detach().clone() for n in check}
    full = run(torch.arange(B, device=dev))
    h1 = run(torch.arange(0, B // 2, device=dev))
    h2 = run(torch.arange(B // 2, B, device=dev))
    assert abs(full[0] - 0.5 * (h1[0] + h2[0])) < 5e-6 and abs(full[1] - 0.5 * (h1[1] + h2[1])) < 5e-6
    for n in check:
        want = 0.5 * (h1[2][n] + h2[2][n])
        err = float((full[2][n] - want).abs().max())
        assert err <= 1e-7 + 2e-4 * float(want.abs().max()), (n, err)
    model.zero_grad()
    # 4: oracle on a 6-session sub-batch of the same full-size model
    idx = torch.tensor([0, 1, 2, 777, 2048, 4095], device=dev)
    sub = take(batch, idx)
    sub['batch_size'] = 6
    ref_batch = synth.to_reference_layout(sub, c['I'])
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = O.Config(**{k: v for k, v in vars(args).items() if k != 'device'})
    with torch.no_grad():
        ref = O.forward(sd, ref_batch, cfg)
    for k in ('weights', 'ens_score', 'intents'):
        err = float((out[k][idx].cpu() - ref[k]).abs().max())
        assert err <= 3e-5 * max(1.0, float(ref[k].abs().max())), (k, err)
    rl, _, _ = O.int_bpr_loss(ref, ref_batch, cfg, noise[idx].cpu())
    sub['bpr_noise'] = noise[idx].contiguous()
    with torch.no_grad():
        gl, _, _ = crit({k: v[idx].contiguous() for k, v in out.items()}, sub)
    assert abs(float(gl) - float(rl)) < 1e-5
    # 5: NDCG
    eng = IntELEngine(model, 'IntBPRloss', args)
    _, nd = eng.eval_step(batch, k=3)
    nd = nd.float()
    assert bool(((nd >= 0) & (nd <= 1 + 1e-6)).all())


def test_training_learns_a_planted_ranking_signal():
    """End to end: when the labels are a function of the base scores (top items of ranker 0), a few hundred fused steps
    (forward, IntBPR loss, hand-written backward, dense Adam) must lift the held-out NDCG@3 from chance towards 1."""
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.engine import IntELEngine
    from intel_sigir2023_amd.model import IntEL
    dev = _dev()
    torch.manual_seed(3)
    args = synth.make_args('tiny', dev)
    corpus, _ = synth.make_corpus('tiny')
    model = IntEL(args, corpus).to(dev)
    eng = IntELEngine(model, 'IntBPRloss', args, lr=2e-3, l2=0.0)

    def planted(seed):
        b = synth.make_batch('tiny', 256, dev, seed=seed, ragged=True)
        s0 = b['scores'][:, :, 0].float()
        valid = torch.arange(s0.shape[1], device=dev)[None, :] < b['session_len'][:, None]
        order = torch.where(valid, s0, torch.full_like(s0, -1.0)).argsort(dim=1, descending=True)
        labels = torch.tensor([3, 2, 1, 1, 1], device=dev, dtype=torch.int32)
        r = torch.zeros_like(b['ranking'])
        r.scatter_(1, order[:, :5], labels[None, :].expand(r.shape[0], 5))
        b['ranking'] = (r * valid).int()
        return b

    held_out = planted(9999)
    model.eval()
    before = float(eng.eval_step(held_out)[1].float().mean())
    model.train()
    for step in range(300):
        eng.train_step(planted(step))
    model.eval()
    after = float(eng.eval_step(held_out)[1].float().mean())
    assert before < 0.5, before
    assert after > 0.85 and after > before + 0.3, (before, after)


@pytest.mark.parametrize('workload,B', [('tiny', 64), ('tmall', 48)])
def test_flagged_row_adam_equals_dense_adam(workload, B, monkeypatch):
    """The item-id table's Adam sweep that reads the gradient only in the rows the backward marked (intel_adam_step_rows)
    == the dense sweep (intel_adam_step): parameters and both moments after three steps, untouched rows bit for bit,
    touched rows up to the summation order of the embedding atomics; the gradient table and the marks are clean after
    every step."""
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.engine import IntELEngine
    from intel_sigir2023_amd.model import IntEL
    dev = _dev()
    over = dict(items=20000, users=2000) if workload != 'tiny' else None
    res = []
    for rows_mode in ('1', '0'):
        monkeypatch.setenv('INTEL_ADAM_ROWS', rows_mode)
        torch.manual_seed(3)
        args = synth.make_args(workload, dev, cal_diversity=0)
        corpus, c = synth.make_corpus(workload, **(over or {}))
        model = IntEL(args, corpus).to(dev)
        eng = IntELEngine(model, 'IntBPRloss', args, lr=1e-3, l2=1e-4)
        assert (eng._iid_flags is not None) == (rows_mode == '1')
        touched = torch.zeros(model.iid_embeddings.weight.shape[0], dtype=torch.bool, device=dev)
        for step in range(3):
            batch = synth.make_batch(workload, B, dev, seed=20 + step, ragged=True, corpus_over=over)
            L = batch['i_id_s'].shape[1]
            noise = torch.rand(B, L, L, device=dev, generator=torch.Generator(device=dev).manual_seed(step))
            eng.train_step(batch, noise=noise)
            touched[batch['i_id_s'].reshape(-1).long().clamp_min(0)] = True
            touched[batch['his_item_id'].reshape(-1).long().clamp_min(0)] = True
            torch.cuda.synchronize()
            assert float(eng.gflat['iid'].abs().max()) == 0.0
            if eng._iid_flags is not None:
                assert int(eng._iid_flags.max()) == 0
        res.append((model.iid_embeddings.weight.detach().clone(), eng.m['iid'].clone(), eng.v['iid'].clone(), touched))
    (p1, m1, v1, t1), (p0, m0, v0, t0) = res
    assert torch.equal(t1, t0) and 0 < int(t1.sum()) < t1.numel()
    d = p1.shape[1]
    for a, b in ((p1, p0), (m1[:p1.numel()].view(-1, d), m0[:p0.numel()].view(-1, d)), (v1[:p1.numel()].view(-1, d), v0[:p0.numel()].view(-1, d))):
        assert torch.equal(a[~t1], b[~t1])
        assert float((a[t1] - b[t1]).abs().max()) < 1e-6


@pytest.mark.parametrize('packed', [True, False])
def test_sorted_embedding_scatter_equals_unsorted(packed, monkeypatch):
    """The id-sorted embedding-gradient scatter (runs of equal ids summed in registers, boundary runs merged through LDS) == the
    plain atomic scatter on a batch with heavily repeated item ids (Zipf popularity), for the tower rows and for the packed and
    padded item-history rows: the parameters after two fused steps agree up to the order of the float additions."""
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.engine import IntELEngine
    from intel_sigir2023_amd.model import IntEL
    dev = _dev()
    over = dict(items=5000, users=500)
    res = []
    for mode in ('0', '1'):
        monkeypatch.setenv('INTEL_SCATTER_SORTED', mode)
        torch.manual_seed(4)
        args = synth.make_args('tmall', dev)
        corpus, _ = synth.make_corpus('tmall', **over)
        model = IntEL(args, corpus).to(dev)
        eng = IntELEngine(model, 'IntBPRloss', args, lr=1e-3, l2=1e-4)
        for step in range(2):
            batch = synth.make_batch('tmall', 96, dev, seed=30 + step, ragged=True, zipf=True, corpus_over=over)
            batch['his_item_id'] = batch['i_id_s'][:, :batch['his_item_id'].shape[1]].clone() * (batch['his_item_id'] > 0)   # repeated ids in the histories too
            if not packed:
                batch.pop('his_rows'), batch.pop('hisitem_rows')
            eng.train_step(batch, noise_seed=99 + step)
        torch.cuda.synchronize()
        ids = batch['i_id_s'].reshape(-1)
        assert torch.unique(ids).numel() < 0.5 * ids.numel()                 # the batch does repeat its ids
        res.append({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    for k, v in res[0].items():
        if 'k_linear.bias' in k:
            continue
        assert float((v - res[1][k]).abs().max()) < 2e-6, k


def test_out_of_range_ids_raise_when_checked(monkeypatch):
    """INTEL_CHECK_IDS=1: an id outside the table raises before any kernel runs, as nn.Embedding does (the gather / scatter
    kernels themselves do not look)."""
    from intel_sigir2023_amd import _lib, synth
    from intel_sigir2023_amd.model import IntEL
    dev = _dev()
    args = synth.make_args('tiny', dev)
    corpus, c = synth.make_corpus('tiny')
    model = IntEL(args, corpus).to(dev)
    batch = synth.make_batch('tiny', 4, dev, seed=1)
    monkeypatch.setenv('INTEL_CHECK_IDS', '1')
    with torch.no_grad():
        model(batch)                                   # in range: fine
    bad = dict(batch)
    bad['i_id_s'] = batch['i_id_s'].clone()
    bad['i_id_s'][0, 0] = c['items']                   # one past the last row
    with pytest.raises(_lib.IntelHipError):
        with torch.no_grad():
            model(bad)


def test_packed_weight_reuse_never_serves_stale_parameters():
    """Evaluation over a frozen model reuses the packed weight images of the previous forward (intel_set_params_unchanged);
    every way the parameters can change -- the engine's fused Adam (raw pointers), an in-place torch op, load_state_dict --
    must repack: each evaluation equals the one a fresh context computes from the same parameters."""
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.engine import IntELEngine
    from intel_sigir2023_amd.model import IntEL
    dev = _dev()
    torch.manual_seed(3)
    args = synth.make_args('tiny', dev)
    corpus, _ = synth.make_corpus('tiny')
    model = IntEL(args, corpus).to(dev)
    eng = IntELEngine(model, 'IntBPRloss', args, lr=1e-2, l2=1e-4)
    batch = synth.make_batch('tiny', 32, dev, seed=2, ragged=True)
    other = synth.make_batch('tiny', 32, dev, seed=9, ragged=True)

    def fresh():
        torch.manual_seed(3)
        m2 = IntEL(args, corpus).to(dev)
        m2.load_state_dict(model.state_dict())
        m2.eval()
        with torch.no_grad():
            return m2(batch)['ens_score'].clone()

    model.eval()
    a0 = eng.eval_step(batch)[0]['ens_score'].clone()
    a1 = eng.eval_step(batch)[0]['ens_score'].clone()              # reuses the images
    assert model._packed_key is not None and torch.equal(a0, a1) and torch.equal(a0, fresh())
    eng.eval_step(other)                                            # same shape, other data: still reusable
    assert torch.equal(eng.eval_step(batch)[0]['ens_score'], a0)
    model.train()
    eng.train_step(batch)                                           # fused Adam behind torch's version counters
    model.eval()
    b0 = eng.eval_step(batch)[0]['ens_score'].clone()
    assert not torch.equal(b0, a0) and torch.equal(b0, fresh())
    with torch.no_grad():
        model.pred_layer.weight.mul_(1.5)                           # in-place torch op
    c0 = eng.eval_step(batch)[0]['ens_score'].clone()
    assert not torch.equal(c0, b0) and torch.equal(c0, fresh())
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    sd['weight_embeddings.weight'] = sd['weight_embeddings.weight'] * 0.5
    model.load_state_dict(sd)
    d0 = eng.eval_step(batch)[0]['ens_score'].clone()
    assert not torch.equal(d0, c0) and torch.equal(d0, fresh())


def test_timeline_records_and_borrowed_stream():
    """intel_prof_timeline returns one record per launch with start <= end and the streams of the concurrent branches;
    intel_side_stream hands out the context's streams (the engine's table sweep borrows one instead of creating a fifth)."""
    import json
    from intel_sigir2023_amd import _lib, synth
    from intel_sigir2023_amd.engine import IntELEngine
    from intel_sigir2023_amd.model import IntEL
    dev = _dev()
    torch.manual_seed(0)
    args = synth.make_args('tiny', dev)
    corpus, _ = synth.make_corpus('tiny')
    model = IntEL(args, corpus).to(dev)
    eng = IntELEngine(model, 'IntBPRloss', args)
    batch = synth.make_batch('tiny', 32, dev, seed=4, ragged=True)
    eng.train_step(batch)
    torch.cuda.synchronize()
    lib = _lib.lib()
    ptrs = [lib.intel_side_stream(model._context(), i) for i in range(3)]
    assert all(ptrs) and len(set(ptrs)) == 3 and lib.intel_side_stream(model._context(), 3) is None
    assert eng._table_stream().cuda_stream == ptrs[1]
    lib.intel_prof_enable(1)
    eng.train_step(batch)
    tl = json.loads(lib.intel_prof_timeline().decode())
    lib.intel_prof_enable(0)
    assert len(tl) > 40 and all(r['t1'] >= r['t0'] for r in tl)      # (60 launches at the reference's default widths since the one-kernel towers / encoders / chain launches)
    assert len({r['stream'] for r in tl}) >= 4                 # the caller's stream + three branches
    names = {r['name'].split('[')[0].strip('()').split('<')[0] for r in tl}
    assert {'bpr_loss_kernel', 'adam_rows_kernel', 'slab_reduce_batch_kernel'} <= names
    assert json.loads(lib.intel_prof_timeline().decode()) == []      # cleared


@pytest.mark.parametrize('workload,lazy', [('tiny', False), ('tmall', False), ('tmall', True)])
def test_next_forward_under_the_table_sweep_changes_nothing(workload, lazy):
    """engine.defer_table_wait (bench.py, runner.fit): a step returns with the item-id table's Adam sweep still running on its side stream; the next
    forward's item-id gathers wait for it inside intel_forward (intel_set_table_wait_event), the other branches start under it.  Eight steps over
    three alternating batches, an evaluation and a state_dict in between: losses, evaluation outputs and final parameters equal the engine that
    waits after every step (up to the order of the embedding scatter's float atomics)."""
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.engine import IntELEngine
    from intel_sigir2023_amd.model import IntEL
    dev = _dev()
    over = dict(items=30000, users=2000) if workload != 'tiny' else None
    res = []
    for defer in (False, True):
        torch.manual_seed(11)
        args = synth.make_args(workload, dev)
        corpus, _ = synth.make_corpus(workload, **(over or {}))
        model = IntEL(args, corpus).to(dev)
        eng = IntELEngine(model, 'IntBPRloss', args, lr=1e-3, l2=1e-4, lazy_table=lazy)
        eng.defer_table_wait = defer
        batches = [synth.make_batch(workload, 24, dev, seed=60 + i, ragged=True, corpus_over=over) for i in range(3)]
        held = synth.make_batch(workload, 24, dev, seed=99, ragged=True, corpus_over=over)
        losses, mid = [], None
        for step in range(8):
            l = eng.train_step(batches[step % 3], noise_seed=500 + step)
            losses.append([float(x) for x in l])
            if step == 4:
                out, nd = eng.eval_step(held)
                mid = ({k: v.clone() for k, v in out.items() if torch.is_tensor(v)}, nd.clone(),
                       {k: v.detach().clone() for k, v in model.state_dict().items()})
        assert (eng._table_ev is not None) == defer          # the last step's sweep is still pending in the deferred engine ...
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}      # ... and state_dict() orders itself behind it
        assert eng._table_ev is None
        res.append((losses, mid, sd))
    (l0, m0, s0), (l1, m1, s1) = res
    for a, b in zip(l0, l1):
        for x, y in zip(a, b):
            assert abs(x - y) < 1e-6, (l0, l1)
    for k in m0[0]:
        assert float((m0[0][k] - m1[0][k]).abs().max()) < 1e-6, k
    assert float((m0[1] - m1[1]).abs().max()) < 1e-6
    for k in s0:
        # (the attention key biases have no gradient: what they receive is rounding noise, and Adam divides noise by noise -- tests/test_trajectory_gpu.py)
        tol = 1e-4 if k.endswith('k_linear.bias') else 1e-6
        assert float((s0[k] - s1[k]).abs().max()) < tol, k
        assert float((m0[2][k].cpu() - m1[2][k].cpu()).abs().max()) < tol, k
