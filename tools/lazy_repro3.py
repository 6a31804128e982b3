"""Run-to-run reproducibility of the first training steps (debugging aid): python tools/lazy_repro3.py <lazy 0|1> [workload] [items] [B]"""
import sys
sys.path.insert(0, '.')
import torch
from intel_sigir2023_amd import synth
from intel_sigir2023_amd.engine import IntELEngine
from intel_sigir2023_amd.model import IntEL

dev = torch.device('cuda:0')
lazy = sys.argv[1] == '1'
wl = sys.argv[2] if len(sys.argv) > 2 else 'stress'
items = int(sys.argv[3]) if len(sys.argv) > 3 else 200000
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
args = synth.make_args(wl, dev, cal_diversity=1)
corpus, c = synth.make_corpus(wl, items=items)
batches = [synth.make_batch(wl, B, dev, seed=40 + i, corpus_over=dict(items=items)) for i in range(3)]
for trial in range(6):
    torch.manual_seed(5)
    model = IntEL(args, corpus).to(dev)
    eng = IntELEngine(model, 'IntBPRloss', args, lr=1e-3, l2=1e-4, lazy_table=lazy)
    losses = [float(eng.train_step(batches[i % 3], noise_seed=100 + i)[0].detach()) for i in range(3)]
    eng.flush()
    torch.cuda.synchronize()
    print('trial', trial, ['%.10f' % x for x in losses], '%.9f' % float(model.iid_embeddings.weight.double().sum()))
    del eng, model
    junk = torch.full((1 << 28,), float(trial + 1), device=dev)      # dirty the freed blocks differently every trial
    del junk
