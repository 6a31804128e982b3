"""Is a bf16-mode gradient that misses the emulating oracle by more than the sweep's bar a BUG or rounding-flip amplification?

At a handful of sessions the bf16 mode is chaotic at the granularity of ONE bf16 rounding: an operand on a rounding boundary falls the other way when
its fp32 value differs in the last bits (4e-3 of the element), two layers later pre-activations differ by 1e-3 of their scale, a few relu units of a
few rows switch, and a weight-gradient tensor summed over 300 rows moves by several per cent.  The oracle shows this against ITSELF: re-run with every
parameter multiplied by (1 + eps * N(0, 1)), eps = the size of the build's own last-bit deviations, its gradients form a CLOUD.  A correct build lies
inside that cloud (its distance to the nearest member is no larger than the members' distances to each other); a structural error (wrong tile, stale
operand, dropped term) is O(1) away from every member.
When the cloud does not contain the build (its members' flips are random; a specific unit a few 1e-4 of its row's scale from zero is not reached by 1e-6
noise), flip_probe names the flip constructively instead.
usage (GPU box): python tools/bf16_cloud.py <case index of `fuzz_parity.py N seed --dtype bf16`> [seed = 61616] [members = 24] [eps = 1e-6]"""
import itertools
import os
import random
import sys

sys.path.insert(0, '.')
import torch


def cloud_check(model, batch, ref_batch, cfg, loss_name, noise, members=24, eps=1e-6, verbose=False):
    """(worst ratio, tensor, report rows): per gradient tensor d_hip = min over members of |g_hip - g_member|, spread = max pairwise |g_s - g_t|;
    ratio = d_hip / max(spread, 5e-2 |g_0|) -- <= 1: inside the cloud or inside the sweep's structural bar."""
    from oracle import intel_oracle as O

    def oracle_grads(sd):
        sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        ref = O.forward_bf16(sd, ref_batch, cfg)
        rl = O.int_bpr_loss(ref, ref_batch, cfg, noise) if loss_name == 'IntBPRloss' else (O.int_list_loss(ref, ref_batch, cfg) if loss_name == 'IntListloss' else O.int_mse_loss(ref, ref_batch, cfg))
        rl[0].backward()
        return {k: (v.grad if v.grad is not None else torch.zeros_like(v)).detach() for k, v in sd.items()}
    sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    cloud = [oracle_grads(sd0)]
    for t in range(members):
        gen = torch.Generator().manual_seed(1000 + t)
        cloud.append(oracle_grads({k: (v * (1 + eps * torch.randn(v.shape, generator=gen))) if v.is_floating_point() else v for k, v in sd0.items()}))
    named = dict(model.named_parameters())
    worst, wk, rows = 0.0, None, []
    for k, p in named.items():
        g0 = cloud[0][k]
        n0 = float(g0.norm())
        if 'k_linear.bias' in k or n0 < 1e-12:
            continue      # analytically zero (attention key bias): rounding noise in any arithmetic
        g = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().cpu()
        d_hip = min(float((g - c[k]).norm()) for c in cloud)
        d0 = float((g - g0).norm())
        spread = max(float((a[k] - b[k]).norm()) for a, b in itertools.combinations(cloud, 2))
        ratio = d_hip / max(spread, 5e-2 * n0, 1e-5)
        rows.append((k, d0 / n0, d_hip / n0, spread / n0, ratio))
        if ratio > worst:
            worst, wk = ratio, k
    if verbose:
        for k, a, b, c, r in sorted(rows, key=lambda t: -t[1])[:int(os.environ.get("CLOUD_ROWS", "14"))]:
            print('    %-52s to the unperturbed oracle %.3e  to the nearest member %.3e  cloud spread %.3e  (of |g|)  ratio %.2f' % (k, a, b, c, r))
    return worst, wk, rows


def flip_probe(model, ref_batch, cfg, loss_name, noise, tries=6, verbose=True):
    """Names the rounding flip behind an outlier that the parameter-noise cloud does not reach.  One relu unit of one row switching side shows up as ONE
    element of the first feed-forward linear's bias gradient carrying the whole difference; the oracle's pre-ReLU activations are tapped (its own test
    hook), the feed-forward block whose bias-gradient difference is the most concentrated gives the unit, the rows where that unit's pre-activation is
    nearest zero are pushed across the kink one at a time, and the build is re-measured against each flipped oracle over ALL gradient tensors.  A build
    that is one flip from the oracle lands at cloud-spread distance from one of them.
    Returns (worst relative distance before, after, (block, layer, row index, unit, pre-activation / its row's largest))."""
    from oracle import intel_oracle as O

    def grads(nudge):
        sd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
        taps = {'__nudge__': nudge} if nudge else {}
        O._EMU['on'] = True
        try:
            ref = O.forward(sd, ref_batch, cfg, taps=taps)
        finally:
            O._EMU['on'] = False
        rl = O.int_bpr_loss(ref, ref_batch, cfg, noise) if loss_name == 'IntBPRloss' else (O.int_list_loss(ref, ref_batch, cfg) if loss_name == 'IntListloss' else O.int_mse_loss(ref, ref_batch, cfg))
        rl[0].backward()
        return {k: (v.grad if v.grad is not None else torch.zeros_like(v)).detach() for k, v in sd.items() if v.is_floating_point()}, taps
    named = {k: (p.grad if p.grad is not None else torch.zeros_like(p)).detach().cpu() for k, p in model.named_parameters()}

    def worst(g):
        return max((float((named[k] - g[k]).norm()) / float(g[k].norm()), k) for k in named if 'k_linear.bias' not in k and float(g[k].norm()) > 1e-12)
    g0, taps = grads(None)
    before = worst(g0)
    pick = None
    for key in taps:
        if key != '__nudge__' and key + '.bias' in named:
            d = named[key + '.bias'] - g0[key + '.bias']
            score = float(d.abs().max()) / max(float(g0[key + '.bias'].norm()), 1e-30)      # one element's share of the tensor
            if pick is None or score > pick[0]:
                pick = (score, key, int(d.abs().argmax()))
    _, key, u = pick
    cands = []
    for l, pre in enumerate(taps[key]):
        rows = pre.reshape(-1, pre.shape[-1])
        rel = rows[:, u].abs() / rows.abs().amax(dim=1).clamp_min(1e-30)
        cands += [(float(rel[r]), l, r) for r in torch.argsort(rel)[:tries].tolist()]
    best = (before[0], None)
    for rel, l, r in sorted(cands)[:tries]:
        pre = taps[key][l]
        n = torch.zeros_like(pre).reshape(-1, pre.shape[-1])
        n[r, u] = -2.0 * float(pre.reshape(-1, pre.shape[-1])[r, u])
        g, _ = grads({key: {l: n.view_as(pre)}})
        w = worst(g)
        if verbose:
            print('    %s layer %d row %d unit %d (pre-activation %.1e of its row\'s largest) pushed across the kink: worst tensor %.3e -> %.3e of |g| (%s)' % (key, l, r, u, rel, before[0], w[0], w[1]))
        if w[0] < best[0]:
            best = (w[0], (key, l, r, u, rel))
    return before[0], best[0], best[1]


def build_case(idx, seed, dev):
    """The configuration, model, batch and loss of case `idx` of the bf16 sweep (the draws of tools/fuzz_parity.py: one_case)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('fuzz_parity', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fuzz_parity.py'))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    rng = random.Random(seed)
    for i in range(idx):
        fuzz.one_case(rng, i, dev, dry=True, dtype='bf16')
    hold = {}
    fuzz.one_case(rng, idx, dev, dtype='bf16', keep=hold)
    return hold


if __name__ == '__main__':
    idx = int(sys.argv[1])
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 61616
    members = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    eps = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-6
    h = build_case(idx, seed, torch.device('cuda:0'))
    print('case %d: %s; against the emulating oracle alone: %.2f of the sweep bar at %s' % (idx, h['desc'], h['worst'], h['bad']))
    worst, wk, _ = cloud_check(h['model'], h['batch'], h['ref_batch'], h['cfg'], h['loss_name'], h['noise'], members, eps, verbose=True)
    if worst > 1.0 and os.environ.get('CLOUD_FLIP', '1') != '0':
        b, a, which = flip_probe(h['model'], h['ref_batch'], h['cfg'], h['loss_name'], h['noise'])
        print('flip probe: worst tensor %.3e of |g| from the oracle, %.3e from the oracle with %s pushed across the relu kink' % (b, a, which))
    print('cloud check (%d members, eps %.0e): worst ratio %.2f at %s -> %s' % (members, eps, worst, wk, 'inside the cloud' if worst <= 1.0 else 'OUTSIDE'))
