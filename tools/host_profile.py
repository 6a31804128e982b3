"""cProfile of the host side of IntELEngine.train_step (where does the enqueue time of a short step go?).
usage (GPU box): python tools/host_profile.py [workload=tmall_pub] [batch=512]"""
import cProfile
import pstats
import sys
import time

sys.path.insert(0, '.')
import torch

from intel_sigir2023_amd import synth
from intel_sigir2023_amd.engine import IntELEngine
from intel_sigir2023_amd.model import IntEL

wl = sys.argv[1] if len(sys.argv) > 1 else 'tmall_pub'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device('cuda:0')
args = synth.make_args(wl, dev)
corpus, _ = synth.make_corpus(wl)
torch.manual_seed(0)
m = IntEL(args, corpus).to(dev)
lr, l2 = synth.WORKLOADS[wl].get('optim', (1e-3, 1e-6))
e = IntELEngine(m, 'IntBPRloss', args, lr=lr, l2=l2)
bs = [synth.make_batch(wl, B, dev, seed=i) for i in range(4)]
for b in bs:
    b['_intel'] = m.prepare_batch(b)
    b['_intel'][1]['ranking_i32'] = b['ranking']
for i in range(8):
    e.train_step(bs[i % 4])
torch.cuda.synchronize()
t0 = time.time()
for i in range(50):
    e.train_step(bs[i % 4])
t1 = time.time()
torch.cuda.synchronize()
t2 = time.time()
print('enqueue %.3f ms/step, drained %.3f ms/step' % ((t1 - t0) / 50 * 1e3, (t2 - t0) / 50 * 1e3))
pr = cProfile.Profile()
pr.enable()
for i in range(50):
    e.train_step(bs[i % 4])
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
