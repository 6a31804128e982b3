"""Is a bf16-mode gradient that misses the emulating oracle by more than the sweep's bar a BUG or rounding-flip amplification?

At a handful of sessions the bf16 mode is chaotic at the granularity of ONE bf16 rounding: an operand on a rounding boundary falls the other way when
its fp32 value differs in the last bits (4e-3 of the element), two layers later pre-activations differ by 1e-3 of their scale, a few relu units of a
few rows switch, and a weight-gradient tensor summed over 300 rows moves by several per cent.  The oracle shows this against ITSELF: re-run with every
parameter multiplied by (1 + eps * N(0, 1)), eps = the size of the build's own last-bit deviations, its gradients form a CLOUD.  A correct build lies
inside that cloud (its distance to the nearest member is no larger than the members' distances to each other); a structural error (wrong tile, stale
operand, dropped term) is O(1) away from every member.
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
        for k, a, b, c, r in sorted(rows, key=lambda t: -t[1])[:14]:
            print('    %-52s to the unperturbed oracle %.3e  to the nearest member %.3e  cloud spread %.3e  (of |g|)  ratio %.2f' % (k, a, b, c, r))
    return worst, wk, rows


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
    print('cloud check (%d members, eps %.0e): worst ratio %.2f at %s -> %s' % (members, eps, worst, wk, 'inside the cloud' if worst <= 1.0 else 'OUTSIDE'))
