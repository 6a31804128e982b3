"""Per-phase shader-clock breakdown of the one-kernel tower layer (csrc/tower.hip) at the headline shape.
usage (GPU box; library built with INTEL_DEBUG_BUILD=1 python -m intel_sigir2023_amd.build): INTEL_TOWER_DBG=1 INTEL_FUSE_TOWER=1 python tools/tower_probe.py [workload] [batch] [f32|bf16]"""
import sys
import time

sys.path.insert(0, '.')
import torch

from intel_sigir2023_amd import synth
from intel_sigir2023_amd.engine import IntELEngine
from intel_sigir2023_amd.model import IntEL

wl = sys.argv[1] if len(sys.argv) > 1 else 'tmall'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dtype = sys.argv[3] if len(sys.argv) > 3 else 'f32'
dev = torch.device('cuda:0')
args = synth.make_args(wl, dev, dtype=dtype)
corpus, _ = synth.make_corpus(wl)
model = IntEL(args, corpus).to(dev)
eng = IntELEngine(model, 'IntBPRloss', args)
batch = synth.make_batch(wl, B, dev, seed=1)
for mode in ('eval', 'train'):
    for i in range(3):
        print('---', mode, i, file=sys.stderr)
        if mode == 'eval':
            eng.eval_step(batch)
        else:
            eng.train_step(batch)
        torch.cuda.synchronize()
