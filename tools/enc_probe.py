"""Per-phase shader-clock breakdown of the fused encoder block kernel (INTEL_ENC_DBG=1; synchronises every launch).
usage (GPU box; library built with INTEL_DEBUG_BUILD=1 python -m intel_sigir2023_amd.build): INTEL_ENC_DBG=1 python tools/enc_probe.py [bf16]"""
import sys

sys.path.insert(0, '.')
import torch

from intel_sigir2023_amd import synth
from intel_sigir2023_amd.engine import IntELEngine
from intel_sigir2023_amd.model import IntEL

dev = torch.device('cuda:0')
over = {'dtype': 'bf16'} if len(sys.argv) > 1 and sys.argv[1] == 'bf16' else {}
args = synth.make_args('tmall', dev, **over)
corpus, c = synth.make_corpus('tmall', items=100000)
torch.manual_seed(0)
model = IntEL(args, corpus).to(dev)
eng = IntELEngine(model, 'IntBPRloss', args, lr=1e-3, l2=1e-4)
batch = synth.make_batch('tmall', 4096, dev, seed=1, corpus_over=dict(items=100000))
for i in range(3):
    eng.train_step(batch)
torch.cuda.synchronize()
model.eval()
for i in range(2):
    eng.eval_step(batch, k=3)
torch.cuda.synchronize()
