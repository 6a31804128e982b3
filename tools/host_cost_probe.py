"""How long does the host take to enqueue one training step?  (B small -> GPU work negligible -> wall time = host time.)"""
import sys, time
sys.path.insert(0, '.')
import torch
from intel_sigir2023_amd import synth
from intel_sigir2023_amd.engine import IntELEngine
from intel_sigir2023_amd.model import IntEL
dev = torch.device('cuda:0')
wl = sys.argv[1] if len(sys.argv) > 1 else 'tmall'              # usage: host_cost_probe.py [workload] [batch ...]
batches = [int(x) for x in sys.argv[2:]] or [16, 256, 1024]
args = synth.make_args(wl, dev)
corpus, _ = synth.make_corpus(wl, items=20000, users=2000)
for B in batches:
    torch.manual_seed(0)
    m = IntEL(args, corpus).to(dev)
    e = IntELEngine(m, 'IntBPRloss', args)
    b = synth.make_batch(wl, B, dev, seed=1, corpus_over=dict(items=20000, users=2000))
    for _ in range(5):
        e.train_step(b)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 30
    for _ in range(n):
        e.train_step(b)
    t1 = time.time()
    torch.cuda.synchronize()
    t2 = time.time()
    print('B=%d: enqueue %.3f ms/step, incl. drain %.3f ms/step' % (B, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
if len(sys.argv) > 1:
    sys.exit(0)
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    e.train_step(b)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
