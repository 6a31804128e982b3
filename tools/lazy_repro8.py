"""Step 2 of the lazy engine vs the dense engine, piece by piece (debugging aid)."""
import sys
sys.path.insert(0, '.')
import torch
from intel_sigir2023_amd import synth
from intel_sigir2023_amd.engine import IntELEngine
from intel_sigir2023_amd.model import IntEL

dev = torch.device('cuda:0')
wl, items, B = 'stress', 200000, 256
args = synth.make_args(wl, dev, cal_diversity=1)
corpus, c = synth.make_corpus(wl, items=items)
b1 = synth.make_batch(wl, B, dev, seed=40, corpus_over=dict(items=items))
b2 = synth.make_batch(wl, B, dev, seed=41, corpus_over=dict(items=items))


def run(lazy):
    torch.manual_seed(5)
    model = IntEL(args, corpus).to(dev)
    eng = IntELEngine(model, 'IntBPRloss', args, lr=1e-3, l2=1e-4, lazy_table=lazy)
    torch.cuda.synchronize()
    eng.train_step(b1, noise_seed=100)
    torch.cuda.synchronize()
    model.eval()
    o, _ = eng.eval_step(b2, k=3)                      # lazy gathers at upto = 1
    o = {k: v.clone() for k, v in o.items()}
    model.train()
    torch.cuda.synchronize()
    l2 = float(eng.train_step(b2, noise_seed=101)[0].detach())
    torch.cuda.synchronize()
    fl = int(eng._iid_flags.sum())
    gl = int((eng.gflat['iid'].view(items, -1).abs().amax(1) > 0).sum())
    eng.flush()
    torch.cuda.synchronize()
    return o, l2, model.iid_embeddings.weight.detach().clone(), fl, gl


ref = run(False)
for trial in range(5):
    got = run(True)
    d = {k: float((got[0][k] - ref[0][k]).abs().max()) for k in ref[0]}
    ses = (got[0]['ens_score'] - ref[0]['ens_score']).abs().amax(1)
    dw = (got[2] - ref[2]).abs().amax(1)
    print('trial', trial, 'eval-forward diff', d, 'sessions off', int((ses > 1e-6).sum()), 'loss2 diff %.2e' % abs(got[1] - ref[1]), 'flags/g left', got[3], got[4],
          'table rows off', int((dw > 1e-7).sum()), float(dw.max()))
