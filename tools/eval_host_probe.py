"""Is the evaluation step host-bound?  Enqueue time vs drained time of eval_step, and the host profile.
usage: eval_host_probe.py [f32|bf16] [workload=tmall] [batch=4096]"""
import sys, time
sys.path.insert(0, '.')
import torch
from intel_sigir2023_amd import synth
from intel_sigir2023_amd.engine import IntELEngine
from intel_sigir2023_amd.model import IntEL
dev = torch.device('cuda:0')
dtype = sys.argv[1] if len(sys.argv) > 1 else 'f32'
wl = sys.argv[2] if len(sys.argv) > 2 else 'tmall'
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
args = synth.make_args(wl, dev, dtype=dtype)
corpus, _ = synth.make_corpus(wl)
torch.manual_seed(0)
m = IntEL(args, corpus).to(dev)
e = IntELEngine(m, 'IntBPRloss', args)
bs = [synth.make_batch(wl, B, dev, seed=i) for i in range(4)]
for b in bs:
    b['_intel'] = m.prepare_batch(b)
    b['_intel'][1]['ranking_i32'] = b['ranking']
m.eval()
for i in range(5):
    e.eval_step(bs[i % 4])
torch.cuda.synchronize()
n = 40
t0 = time.time()
for i in range(n):
    e.eval_step(bs[i % 4])
t1 = time.time()
torch.cuda.synchronize()
t2 = time.time()
print('eval %s B=%d' % (wl, B)); print('eval %s: enqueue %.3f ms/step, incl. drain %.3f ms/step' % (dtype, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    e.eval_step(bs[i % 4])
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
