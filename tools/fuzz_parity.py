"""Randomised parity sweep: forward outputs, the Int* losses and EVERY parameter gradient of the HIP path against the CPU
oracle's autograd on shapes the fixtures do not hold (odd batch / list sizes, 1-2 heads, 1-2 tied layers, both encoders,
with and without cross attention, 16-128-wide embeddings).  Test infrastructure like tests/: imports oracle/.
usage (GPU box): python tools/fuzz_parity.py [n_cases] [seed] [--only i,j] [--dtype bf16]
--dtype bf16: the same draws in bf16 mode (four cases in five at the benchmarked widths, where the mode has products to round) against the EMULATING oracle
(oracle.forward_bf16 and its autograd) at bars of 2e-3 of scale on the outputs and 1e-2 of a gradient tensor's largest element: batches of 2 .. 17 sessions
put one rounding flip (4e-3 of one term) at several 1e-3 of a SUM of a few terms, so this sweep looks for structural errors (NaNs, wrong tiles, stale
LDS: O(1) off), not for the last digit -- that is tests/test_bf16_gpu.py's job on real batch sizes."""
import random
import sys

sys.path.insert(0, '.')
import numpy as np
import torch

from intel_sigir2023_amd import loss as LS
from intel_sigir2023_amd import synth
from intel_sigir2023_amd.model import IntEL
from oracle import intel_oracle as O


from tests.helpers import relu_flip_forgiven_error

import os
import ast
FORCE = ast.literal_eval(os.environ['FUZZ_FORCE']) if os.environ.get('FUZZ_FORCE') else None      # debugging: a dict that pins flags of the drawn case, e.g. "{'num_layers': 1}"
VERBOSE = os.environ.get('FUZZ_VERBOSE') == '1'      # every parameter above 0.3 of its tolerance, with the magnitudes


def one_case(rng, idx, dev, big=None, force=None, dry=False, trace=None, dtype='f32', keep=None):
    """big: None = decided per case (about one case in 25), True / False = forced.  A 'big' case is the reference's own tower widths
    (16/16/32/32, script/IntEL.sh:15,21) at lists of 50 / 90 with more than 32 768 candidate rows per batch: the far side of the
    row-count thresholds of the short-list attention backward and the batched small weight gradients."""
    e = lambda: rng.choice([16, 32, 64])
    flags = dict(model_num=rng.choice([2, 3, 5]), context_emb_size=e(), i_emb_size=e(), u_emb_size=e(), s_emb_size=rng.choice([32, 64, 128]),
                 im_emb_size=e(), intent_emb_size=e(), cross_attn_qsize=rng.choice([16, 64]), num_heads=rng.choice([1, 2]),
                 num_layers=rng.choice([1, 1, 2]), encoder=rng.choice(['BERT4Rec', 'BERT4Rec', 'GRU4Rec']), history_max=rng.choice([5, 20, 20, 70]))      # 70: packed histories through the general attention kernels
    if dtype == 'bf16':
        flags['encoder'] = 'BERT4Rec'      # (the emulating oracle states the mode's rounding points for BERT4Rec encoders)
    if rng.random() < (0.8 if dtype == 'bf16' else 0.5):      # the benchmarked widths (fused tower tails, register-resident pooling)
        flags.update(i_emb_size=64, im_emb_size=64, s_emb_size=64)
    L = rng.choice([2, 7, 20, 33, 50, 52, 53, 64, 65, 100])
    B = rng.choice([2, 3, 5, 17])
    I = rng.choice([4, 10, 30])
    if big is None:
        big = random.Random(7919 * idx + 13).random() < 0.04      # its own generator: the other cases keep their draws
    if big:
        flags.update(i_emb_size=16, im_emb_size=16, s_emb_size=32, history_max=min(flags['history_max'], 20))
        L = rng.choice([50, 90])
        B = 32768 // L + rng.choice([1, 40, 300])
    if force:      # tests pin shapes: keys of `flags` plus 'L', 'B', 'I', 'loss', 'cross_attention', 'cal_diversity'
        for k, v in force.items():
            if k in flags:
                flags[k] = v
        L, B, I = force.get('L', L), force.get('B', B), force.get('I', I)
    H = flags['history_max']
    name = 'fuzz%d' % idx
    synth.WORKLOADS[name] = dict(flags=flags, corpus=dict(items=3000, users=300, classes=40, ctx=50, I=I), batch=dict(L=L, H=H))
    over = dict(cross_attention=rng.choice([1, 1, 0]), cal_diversity=rng.choice([0, 1]))
    loss_name = rng.choice(['IntBPRloss', 'IntListloss', 'IntMSEloss'])
    if force:
        over.update({k: force[k] for k in ('cross_attention', 'cal_diversity') if k in force})
        loss_name = force.get('loss', loss_name)
    if dry:      # only advance the generator (re-running single cases of a sweep: `--only`)
        return None
    torch.manual_seed(100 + idx)
    bf = dtype == 'bf16'
    args = synth.make_args(name, dev, dtype=dtype, **over)
    corpus, c = synth.make_corpus(name)
    model = IntEL(args, corpus).to(dev)
    model.train()
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    batch = synth.make_batch(name, B, dev, seed=idx, ragged=True)
    ref_batch = synth.to_reference_layout(batch, c['I'])
    cfg = O.Config(**{k: v for k, v in vars(args).items() if k not in ('device', 'dtype')})
    noise = torch.rand(B, L, L, device=dev)
    batch['bpr_noise'] = noise
    crit = getattr(LS, loss_name)(args)
    if trace is not None:      # tests: which kernels did this case run (tests/helpers.py: KernelTrace)
        from tests.helpers import KernelTrace
        with KernelTrace() as kt:
            out = model(batch)
            loss, ens, itl = crit(out, batch)
            loss.backward()
        trace['names'] = kt.names
        trace['count'] = kt.count
    else:
        out = model(batch)
        loss, ens, itl = crit(out, batch)
        loss.backward()
    taps = {}
    ref = O.forward_bf16(sd, ref_batch, cfg) if bf else O.forward(sd, ref_batch, cfg, taps=taps)
    if loss_name == 'IntBPRloss':
        rl = O.int_bpr_loss(ref, ref_batch, cfg, noise.cpu())
    elif loss_name == 'IntListloss':
        rl = O.int_list_loss(ref, ref_batch, cfg)
    else:
        rl = O.int_mse_loss(ref, ref_batch, cfg)
    rl[0].backward()
    desc = '%s B=%d L=%d I=%d %s' % (loss_name, B, L, I, {k: v for k, v in list(flags.items()) + list(over.items())})
    named = dict(model.named_parameters())

    def compare(ref, rl, sd, taps, forgive):
        worst, bad = 0.0, None
        for k in ('weights', 'ens_score', 'intents'):
            err = float((out[k].detach().cpu() - ref[k].detach()).abs().max()) / max(1.0, float(ref[k].detach().abs().max()))
            if VERBOSE:
                print('    output %-12s err %.3e of scale' % (k, err))
            worst = max(worst, err / (2e-3 if bf else 3e-5))
        worst = max(worst, abs(float(loss.detach()) - float(rl[0].detach())) / ((2e-4 if bf else 1e-5) * max(1.0, abs(float(rl[0].detach())) if bf else 1.0)))
        for k, p in named.items():
            g = p.grad.cpu() if p.grad is not None else torch.zeros(p.shape)
            r = sd[k].grad if sd[k].grad is not None else torch.zeros(p.shape)
            tol = 1e-6 + 2e-4 * float(r.abs().max())            # fp32: the fixture tests' tolerance
            err = float((g - r).abs().max())
            if not (err == err):
                return float('inf'), k      # NaN
            if bf:
                if 'k_linear.bias' in k:
                    continue      # analytically zero: rounding noise in both implementations (tests/test_bf16_gpu.py)
                # bf16 mode: the error of the TENSOR (Frobenius norm) against 5e-2 of the reference's norm.  Elementwise bars do not work at these batch
                # sizes: a bf16 rounding that falls the other way (4e-3 of one activation) can flip a relu downstream, and one flipped (row, unit) moves that
                # unit's weight-gradient row by 1 / sqrt(rows) of its size -- 6 % at 5 x 50 rows, 0.2 % at the benchmark's 204 800.  Such errors are sparse
                # (a few rows of a few tensors); a wrong tile, a stale LDS slot or a NaN is dense and O(1).
                tol = 1e-5 + 5e-2 * float(r.norm())
                err = float((g - r).norm())
            if err > tol and forgive:      # the relu-flip exemption (tests/helpers.py: one row of a first-linear gradient, only when the oracle's pre-activation touches zero)
                err = relu_flip_forgiven_error(k, g, r, tol, taps)
            if VERBOSE and err / tol > 0.3:
                print('    %-50s err %.3e  tol %.3e  max|ref| %.3e  max|hip| %.3e' % (k, err, tol, float(r.abs().max()), float(g.abs().max())))
            if err / tol > worst:
                worst, bad = err / tol, k
        return worst, bad

    def oracle_losses(ref):
        if loss_name == 'IntBPRloss':
            return O.int_bpr_loss(ref, ref_batch, cfg, noise.cpu())
        if loss_name == 'IntListloss':
            return O.int_list_loss(ref, ref_batch, cfg)
        return O.int_mse_loss(ref, ref_batch, cfg)

    worst, bad = compare(ref, rl, sd, taps, not bf)
    if keep is not None:      # tools/bf16_cloud.py: the case's objects for a closer look
        keep.update(model=model, batch=batch, ref_batch=ref_batch, cfg=cfg, loss_name=loss_name, noise=noise.cpu(), desc=desc, worst=worst, bad=bad)
    if worst > 1.0 and not bf:
        # A hidden unit whose pre-activation is ZERO at rounding-noise level in the oracle sits on the kink of the relu: both one-sided derivatives
        # are legitimate, the two implementations may take different ones entry by entry, and at a handful of rows per batch the difference reaches
        # every upstream parameter.  Decide it properly: for every such ENTRY (|pre| < 1e-5 max(1, max|pre|) of a feed-forward block's first linear)
        # the ORACLE is re-run with the entry pushed to either side (a constant added before the relu: oracle._nudged; outputs move by less than their
        # tolerance) and the kernels' gradients must match ONE assignment at the PLAIN tolerances -- no exemption at all.  <= 8 entries (256 runs).
        entries = []
        for key, pres in taps.items():
            if key == '__nudge__' or (key + '.bias') not in named:
                continue
            # only the hidden units whose OWN gradient row is off (a unit on the kink that both sides resolved alike needs no decision)
            g, r = named[key + '.bias'].grad.cpu(), sd[key + '.bias'].grad
            gw, rw = named[key + '.weight'].grad.cpu(), sd[key + '.weight'].grad
            off = ((g - r).abs() > 1e-6 + 2e-4 * float(r.abs().max())) | ((gw - rw).abs().max(dim=1)[0] > 1e-6 + 2e-4 * float(rw.abs().max()))
            big = max(1.0, max(float(t.abs().max()) for t in pres))
            for li, t in enumerate(pres):
                idx = torch.nonzero((t.abs() < 1e-5 * big) & off.reshape(*([1] * (t.dim() - 1)), -1))
                for e in idx.tolist():
                    entries.append((key, li, tuple(e), 4e-5 * big, t.shape))
        if len(entries) > 8:
            desc += '  [%d entries on the relu kink: not decided]' % len(entries)
        elif entries:
            import itertools
            best, bestbad = 1e30, None
            for signs in itertools.product((1.0, -1.0), repeat=len(entries)):
                nudge = {}
                for (key, li, e, dlt, shape), sg in zip(entries, signs):
                    t = nudge.setdefault(key, {}).setdefault(li, torch.zeros(shape))
                    t[e] = sg * dlt
                sd2 = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
                ref2 = O.forward(sd2, ref_batch, cfg, taps={'__nudge__': nudge})
                rl2 = oracle_losses(ref2)
                rl2[0].backward()
                w2, b2 = compare(ref2, rl2, sd2, None, False)
                if w2 < best:
                    best, bestbad = w2, b2
                if w2 <= 1.0:
                    desc += '  [relu kink: %d entries of %s decided, %.2f of tolerance]' % (len(entries), sorted(set(k for k, *_ in entries)), w2)
                    return w2, None, desc
            desc += '  [relu kink: %d entries of %s, no assignment matched (best %.2f at %s)]' % (len(entries), sorted(set(k for k, *_ in entries)), best, bestbad)
    return worst, bad, desc


def explain_bf16(h):
    """A bf16-mode case over the bar: inside the oracle's own cloud, or one relu unit from the oracle?  Returns the sentence, or None."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bf16_cloud', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bf16_cloud.py'))
    cloud = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cloud)
    worst, wk, _ = cloud.cloud_check(h['model'], h['batch'], h['ref_batch'], h['cfg'], h['loss_name'], h['noise'], members=12, eps=1e-6)
    if worst <= 1.0:
        return 'explained: inside the oracle\'s own parameter-noise cloud (12 members, 1e-6; worst ratio %.2f at %s)' % (worst, wk)
    before, after, which = cloud.flip_probe(h['model'], h['ref_batch'], h['cfg'], h['loss_name'], h['noise'], tries=4, verbose=False)
    if which is not None and after <= 2e-2 and which[4] < 2e-3:
        return 'explained: ONE relu unit from the oracle -- %s layer %d row %d unit %d, pre-activation %.1e of its row\'s largest; flipped in the oracle: worst tensor %.3e -> %.3e of |g|' % (which[0], which[1], which[2], which[3], which[4], before, after)
    return None


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = random.Random(seed)
    dev = torch.device('cuda:0')
    fails = unexplained = 0
    only = None
    dtype = sys.argv[sys.argv.index('--dtype') + 1] if '--dtype' in sys.argv else 'f32'
    if '--only' in sys.argv:      # fuzz_parity.py N seed --only 56,136: the same cases as the full sweep draws, only these are run
        only = set(int(x) for x in sys.argv[sys.argv.index('--only') + 1].split(','))
    for i in range(n):
        if only is not None and i not in only:
            one_case(rng, i, dev, dry=True, dtype=dtype)
            continue
        hold = {} if dtype == 'bf16' else None
        try:
            worst, bad, desc = one_case(rng, i, dev, dtype=dtype, force=FORCE, keep=hold)
        except Exception as ex:      # an unsupported shape must fail loudly, not silently
            print('case %d ERROR %s: %s' % (i, type(ex).__name__, str(ex)[:300]))
            fails += 1
            continue
        ok = worst <= 1.0
        fails += 0 if ok else 1
        print('case %d %s worst=%.2f of tolerance%s  %s' % (i, 'ok  ' if ok else 'FAIL', worst, (' at ' + bad) if bad else '', desc))
        if not ok and hold:
            # bf16 mode: a gradient over the structural bar is either a bug or ONE bf16 rounding amplified through relu kinks (tools/bf16_cloud.py) -- decided
            # right here by the oracle itself: its own parameter-noise cloud, then the constructive one-unit flip
            try:
                why = explain_bf16(hold)
            except Exception as ex:
                why = None
                print('    explanation failed: %s: %s' % (type(ex).__name__, str(ex)[:200]))
            print('    case %d %s' % (i, why or 'UNEXPLAINED by the oracle\'s cloud or a single relu flip'))
            unexplained += 0 if why else 1
    print('fuzz: %d cases, %d failures' % (n, fails) + ((', %d of them unexplained' % unexplained) if dtype == 'bf16' else ''))
    sys.exit(1 if fails else 0)


if __name__ == '__main__':
    main()
