"""BASELINE.json configurations at their FULL sizes, where the CPU oracle cannot run a whole batch in seconds: the HIP path is
checked through size-independent properties of the domain (sessions are independent, losses are means over sessions, gradients are
linear in the batch) plus the oracle on a sub-batch the oracle finishes in seconds.  tests/test_engine_gpu.py::
test_full_size_tmall_properties holds configs[1] in fp32; this file adds
  * configs[1] in the bf16 mode it names (1 M items, 4096 sessions): the sub-batch against oracle.forward_bf16,
  * configs[3] LifeData-shape (K = 5, 10 intents, list 100) at 2048 sessions,
  * configs[4] stress (10 M-item table, histories and lists of 200, K = 8) at 256 sessions per step with the LAZY form of the table's
    dense Adam, which is what `bench.py --workload stress` runs: three training steps against the dense sweep, bit for bit."""
import pytest
import torch

from oracle import intel_oracle as O

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch.device('cuda:0')


def _take(b, idx, B):
    out = {k: (v[idx].contiguous() if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == B else v) for k, v in b.items()}
    out['batch_size'] = int(idx.numel())
    for k in ('his_rows', 'hisitem_rows'):
        out.pop(k, None)
    if 'history_len' in out:                      # the host totals of the packed histories, recomputed for the subset
        out['his_rows'] = int(out['history_len'].sum())
        out['hisitem_rows'] = int(out['history_item_len'].sum())
    return out


@pytest.mark.parametrize('workload,dtype,B,sub', [('tmall', 'bf16', 4096, [0, 1, 2, 777, 2048, 4095]), ('lifedata', 'f32', 2048, [0, 1, 1000, 2047]),
                                                  # the reference's PUBLISHED hyper-parameters (script/IntEL.sh:15: 16/16/32/32-wide, GRU4Rec, 2 heads x 2 tied
                                                  # layers) at its batch of 512 and at 1024: 25 600 / 51 200 candidate rows = both sides of the 32 768-row
                                                  # thresholds of the short-list attention backward (csrc/attn.hip) and the batched small weight gradients (csrc/gemm.hip)
                                                  ('tmall_pub', 'f32', 512, [0, 1, 2, 255, 256, 511]), ('tmall_pub', 'f32', 1024, [0, 1, 2, 511, 512, 1023])])
def test_full_size_properties(workload, dtype, B, sub):
    from intel_sigir2023_amd import loss as LS
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.engine import IntELEngine
    from intel_sigir2023_amd.model import IntEL
    dev = _dev()
    torch.manual_seed(0)
    args = synth.make_args(workload, dev, cal_diversity=1, dtype=dtype)
    corpus, c = synth.make_corpus(workload)
    model = IntEL(args, corpus).to(dev)
    batch = synth.make_batch(workload, B, dev, seed=21, ragged=True)
    L = batch['i_id_s'].shape[1]
    # 1. sessions are independent: permuting the batch permutes every output row identically, bit for bit
    model.eval()
    with torch.no_grad():
        out = model(batch)
        perm = torch.randperm(B, device=dev)
        outp = model(_take(batch, perm, B))
    for k in ('weights', 'ens_score', 'intents'):
        assert torch.equal(outp[k], out[k][perm]), 'permutation equivariance broken for ' + k
    # 2 + 3. the loss is a mean over sessions and the gradients are linear in the batch: full == mean of the halves
    crit = LS.IntBPRloss(args)
    noise = torch.rand(B, L, L, device=dev)
    named = dict(model.named_parameters())
    check = ['i_W1.weight', 'i_attn_head.q_linear.weight', 's_W2.bias', 'weight_embeddings.weight', 'pred_layer.weight',
             'intent_embeddings.weight', 'context_embeddings.weight', 'i_layer_norm.weight', 's_attn_head.v_linear.weight', 'score_embeddings.weight',
             'intent_item_attention.key_layer.weight', 'item_embeddings.weight']
    if args.encoder == 'GRU4Rec':
        check += ['encoder.rnn.weight_ih_l0', 'item_encoder.rnn.weight_hh_l0', 'item_encoder.out.weight', 'encoder.rnn.bias_hh_l0']
    else:
        check += ['encoder.transformer_block.0.linear1.weight', 'item_encoder.transformer_block.1.masked_attn_head.v_linear.weight',
                  'item_encoder.transformer_block.0.layer_norm1.weight']

    def run(idx):
        sb = _take(batch, idx, B)
        sb['bpr_noise'] = noise[idx].contiguous()
        model.train()
        model.zero_grad()
        o = model(sb)
        loss, el, il = crit(o, sb)
        loss.backward()
        return float(loss), float(el), {n: named[n].grad.detach().clone() for n in check}
    full = run(torch.arange(B, device=dev))
    h1 = run(torch.arange(0, B // 2, device=dev))
    h2 = run(torch.arange(B // 2, B, device=dev))
    assert abs(full[0] - 0.5 * (h1[0] + h2[0])) < 5e-6 and abs(full[1] - 0.5 * (h1[1] + h2[1])) < 5e-6
    gtol = 2e-4 if dtype == 'f32' else 2e-3        # bf16 mode: the gradient operands are rounded per product, the halves' sums re-associate them
    for n in check:
        want = 0.5 * (h1[2][n] + h2[2][n])
        err = float((full[2][n] - want).abs().max())
        assert err <= 1e-7 + gtol * float(want.abs().max()), (n, err)
    model.zero_grad()
    # 4. the oracle on a sub-batch of the same full-size model (bf16 mode: the emulating oracle)
    idx = torch.tensor(sub, device=dev)
    sb = _take(batch, idx, B)
    ref_batch = synth.to_reference_layout(sb, c['I'])
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = O.Config(**{k: v for k, v in vars(args).items() if k not in ('device', 'dtype')})
    with torch.no_grad():
        ref = (O.forward_bf16 if dtype == 'bf16' else O.forward)(sd, ref_batch, cfg)
    for k in ('weights', 'ens_score', 'intents'):
        err = float((out[k][idx].cpu() - ref[k]).abs().max())
        assert err <= 3e-5 * max(1.0, float(ref[k].abs().max())), (k, err)
    rl, _, _ = O.int_bpr_loss(ref, ref_batch, cfg, noise[idx].cpu())
    sb['bpr_noise'] = noise[idx].contiguous()
    with torch.no_grad():
        gl, _, _ = crit({k: v[idx].contiguous() for k, v in out.items()}, sb)
    assert abs(float(gl) - float(rl)) < 1e-5
    # 5. NDCG@3 on the device in [0, 1]
    eng = IntELEngine(model, 'IntBPRloss', args)
    _, nd = eng.eval_step(batch, k=3)
    nd = nd.float()
    assert bool(((nd >= 0) & (nd <= 1 + 1e-6)).all())


@pytest.mark.parametrize('B', [256, 1024])      # 1024 = the per-GPU batch SURVEY 8-d gives configs[4]
def test_stress_table_lazy_adam_equals_the_dense_sweep_at_full_size(B):
    """configs[4] on one GPU: the 10 M-item table (2.56 GB; parameter + gradient + two moments = 10 GB per engine), lists and
    histories of 200, K = 8, 256 / 1024 sessions per step.  Three fused training steps with the lazy form of the table's Adam (rows without
    a gradient replayed when they are next read) against three with the dense sweep, same initialisation, batches and BPR
    tie-breaks: equal losses, and after the flush the table and both moments bit for bit in the 97 % of the rows no batch touched
    (the touched rows to the order of the scatter's float atomics).  The first step's loss is checked
    against the oracle on a two-session sub-batch."""
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.engine import IntELEngine
    from intel_sigir2023_amd.model import IntEL
    dev = _dev()
    args = synth.make_args('stress', dev, cal_diversity=1)
    corpus, c = synth.make_corpus('stress')
    assert c['items'] == 10000000
    batches = [synth.make_batch('stress', B, dev, seed=40 + i) for i in range(3)]
    res = {}
    for lazy in (False, True):
        torch.manual_seed(5)
        model = IntEL(args, corpus).to(dev)
        eng = IntELEngine(model, 'IntBPRloss', args, lr=1e-3, l2=1e-4, lazy_table=lazy)
        assert (eng._lazy is not None) == lazy
        eng._w0 = model.iid_embeddings.weight.detach().clone()
        if not lazy:                 # the oracle on two sessions of the first batch, before any update
            idx = torch.tensor([0, B - 1], device=dev)
            sb = _take(batches[0], idx, B)
            ref_batch = synth.to_reference_layout(sb, c['I'])
            sd = {k: v.detach().cpu() for k, v in model.state_dict().items() if 'iid_embeddings' not in k}
            # the oracle gets a compact item table: the rows this sub-batch gathers, ids remapped (row 0 stays the padding row)
            ids = torch.cat([torch.zeros(1, dtype=torch.long), ref_batch['i_id_s'].reshape(-1), ref_batch['his_item_id'].reshape(-1)]).unique()
            sd['iid_embeddings.weight'] = model.iid_embeddings.weight.detach()[ids.to(dev)].cpu()
            for key in ('i_id_s', 'his_item_id'):
                ref_batch[key] = torch.searchsorted(ids, ref_batch[key].contiguous())
            cfg = O.Config(**{k: v for k, v in vars(args).items() if k != 'device'})
            model.eval()
            with torch.no_grad():
                got = model(sb)
                ref = O.forward(sd, ref_batch, cfg)
            for k in ('weights', 'ens_score', 'intents'):
                err = float((got[k].cpu() - ref[k]).abs().max())
                assert err <= 3e-5 * max(1.0, float(ref[k].abs().max())), (k, err)
            model.train()
        losses = [float(eng.train_step(bt, noise_seed=100 + i)[0]) for i, bt in enumerate(batches)]
        eng.flush()
        torch.cuda.synchronize()
        res[lazy] = (losses, model.iid_embeddings.weight.detach().clone(), eng)
    # the embedding scatter sums its rows with float atomics: two runs of the SAME engine differ in the last bits of the rows a batch
    # touches.  Rows no batch touched are decayed by the sweep / the replay alone: those must agree bit for bit.
    for a, b in zip(res[True][0], res[False][0]):
        assert abs(a - b) < 1e-6, (res[True][0], res[False][0])
    wl, wd = res[True][1], res[False][1]
    ed, el = res[False][2], res[True][2]
    touched = torch.zeros(wl.shape[0], dtype=torch.bool, device=dev)
    for bt in batches:
        touched[bt['i_id_s'].reshape(-1).long()] = True
        touched[bt['his_item_id'].reshape(-1).long()] = True
    assert 0.005 < float(touched.float().mean()) < 0.25
    quiet = ~touched
    assert torch.equal(wl[quiet], wd[quiet]) and float((wl[quiet] - el._w0[quiet]).abs().max()) > 0      # moved (weight decay), identically
    assert torch.equal(el.m['iid'].view_as(wl)[quiet], ed.m['iid'].view_as(wl)[quiet])
    assert torch.equal(el.v['iid'].view_as(wl)[quiet], ed.v['iid'].view_as(wl)[quiet])
    assert float((wl - wd).abs().max()) < 2e-6
    assert float((el.m['iid'] - ed.m['iid']).abs().max()) < 1e-6 and float((el.v['iid'] - ed.v['iid']).abs().max()) < 1e-8
    sd_d, sd_l = ed.model.state_dict(), el.model.state_dict()
    for k in sd_d:
        if 'k_linear.bias' in k:
            continue                # analytically-zero gradients: Adam normalises rounding noise
        assert float((sd_d[k] - sd_l[k]).abs().max()) < 2e-6, k
