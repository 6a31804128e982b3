"""Experiment: does running two half-batch pipelines on two streams beat one full-batch pipeline?  (Kernel boundaries --
drain, launch, ramp-up -- cost ~10 us each; an independent second pipeline can fill them.)"""
import sys, time
sys.path.insert(0, '.')
import torch
from intel_sigir2023_amd import synth
from intel_sigir2023_amd.engine import IntELEngine
from intel_sigir2023_amd.model import IntEL

dev = torch.device('cuda:0')
args = synth.make_args('tmall', dev)
corpus, _ = synth.make_corpus('tmall')


def mk(B, seed):
    torch.manual_seed(0)
    m = IntEL(args, corpus).to(dev)
    e = IntELEngine(m, 'IntBPRloss', args)
    b = synth.make_batch('tmall', B, dev, seed=seed)
    return e, b


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


e1, b1 = mk(4096, 1)
print('one pipeline  B=4096: %.3f ms/step' % timeit(lambda: e1.train_step(b1)))
eh, bh = mk(2048, 3)
print('one pipeline  B=2048: %.3f ms/step' % timeit(lambda: eh.train_step(bh)))
del e1
ea, ba = mk(2048, 1)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def two():
    with torch.cuda.stream(sa):
        ea.train_step(ba)
    with torch.cuda.stream(sb):
        eh.train_step(bh)


print('two pipelines B=2048 each, two streams: %.3f ms per pair (4096 sessions)' % timeit(two))
