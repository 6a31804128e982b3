"""Experiment: capture one fused training step (all launches of the C ABI on torch's stream + the library's forked side
streams) in a HIP graph through torch.cuda.CUDAGraph and replay it."""
import sys, time
sys.path.insert(0, '.')
import torch
from intel_sigir2023_amd import synth
from intel_sigir2023_amd.engine import IntELEngine
from intel_sigir2023_amd.model import IntEL
dev = torch.device('cuda:0')
args = synth.make_args('tmall', dev)
corpus, _ = synth.make_corpus('tmall')
torch.manual_seed(0)
m = IntEL(args, corpus).to(dev)
e = IntELEngine(m, 'IntBPRloss', args)
b = synth.make_batch('tmall', 4096, dev, seed=1)
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
