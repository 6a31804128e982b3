"""GPU-bound step time: the host enqueues N training steps while the GPU is held busy by a spin kernel, so the steps then run
back to back with every launch already queued -- the step time the GPU alone would allow (against the eager loop's, which at small
batches is bounded by the host's enqueue rate).
usage (GPU box): python tools/gpu_bound_probe.py [workload=tmall_pub] [batch=512] [steps=20]"""
import sys
import time

sys.path.insert(0, '.')
import torch

from intel_sigir2023_amd import synth
from intel_sigir2023_amd.engine import IntELEngine
from intel_sigir2023_amd.model import IntEL

wl = sys.argv[1] if len(sys.argv) > 1 else 'tmall_pub'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
N = int(sys.argv[3]) if len(sys.argv) > 3 else 20
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
# eager loop
t0 = time.time()
for i in range(N):
    e.train_step(bs[i % 4])
t1 = time.time()
torch.cuda.synchronize()
t2 = time.time()
print('eager: enqueue %.3f ms/step, drained %.3f ms/step' % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
# GPU held busy while the host enqueues
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(int(2.0e9 * 0.001 * (N * 1.2 + 5)))      # ~ (1.2 N + 5) ms at 2 GHz
ev0.record()
t0 = time.time()
for i in range(N):
    e.train_step(bs[i % 4])
t1 = time.time()
ev1.record()
torch.cuda.synchronize()
print('queued ahead: host enqueue %.3f ms/step; GPU %.3f ms/step' % ((t1 - t0) / N * 1e3, ev0.elapsed_time(ev1) / N))
