"""Experiment: capture one fused training step (all launches of the C ABI on torch's stream + the library's forked side
streams) in a HIP graph through torch.cuda.CUDAGraph and replay it.  (A replay repeats the captured step's scalars -- Adam's
bias corrections, the noise seed -- so this times the launch path only.)
usage: python tools/graph_probe.py [workload=tmall] [batch=4096]"""
import sys, time
sys.path.insert(0, '.')
import torch
from intel_sigir2023_amd import synth
from intel_sigir2023_amd.engine import IntELEngine
from intel_sigir2023_amd.model import IntEL
dev = torch.device('cuda:0')
wl = sys.argv[1] if len(sys.argv) > 1 else 'tmall'
BS = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
args = synth.make_args(wl, dev)
corpus, _ = synth.make_corpus(wl)
torch.manual_seed(0)
m = IntEL(args, corpus).to(dev)
e = IntELEngine(m, 'IntBPRloss', args)
b = synth.make_batch(wl, BS, dev, seed=1)
b['_intel'] = m.prepare_batch(b)
b['_intel'][1]['ranking_i32'] = b['ranking']
B, Lm = b['i_id_s'].shape
noise = torch.rand(B, Lm, Lm, device=dev)


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


print('eager: %.3f ms/step' % timeit(lambda: e.train_step(b, noise=noise)))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        e.train_step(b, noise=noise)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        out = e.train_step(b, noise=noise)
    torch.cuda.synchronize()
    print('captured')
    print('graph replay: %.3f ms/step' % timeit(lambda: g.replay()))
    print('loss after replays', float(out[0]))
except Exception as ex:
    print('capture failed:', repr(ex)[:400])
