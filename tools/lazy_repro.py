"""Run-to-run reproducibility probe of the lazy table Adam at large table sizes (debugging aid)."""
import sys
sys.path.insert(0, '.')
import torch
from intel_sigir2023_amd import synth
from intel_sigir2023_amd.engine import IntELEngine
from intel_sigir2023_amd.model import IntEL

dev = torch.device('cuda:0')
wl = sys.argv[1] if len(sys.argv) > 1 else 'stress'
items = int(sys.argv[2]) if len(sys.argv) > 2 else 10000000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
args = synth.make_args(wl, dev, cal_diversity=1)
corpus, c = synth.make_corpus(wl, items=items)
batches = [synth.make_batch(wl, B, dev, seed=40 + i, corpus_over=dict(items=items)) for i in range(3)]
for lazy in (False, True, False, True, True):
    torch.manual_seed(5)
    model = IntEL(args, corpus).to(dev)
    eng = IntELEngine(model, 'IntBPRloss', args, lr=1e-3, l2=1e-4, lazy_table=lazy)
    losses = [float(eng.train_step(batches[i % 3], noise_seed=100 + i)[0]) for i in range(5)]
    eng.flush()
    torch.cuda.synchronize()
    print('lazy' if lazy else 'dense', ['%.12f' % x for x in losses], float(model.iid_embeddings.weight.double().sum()))
    del eng, model
