"""Three fused training steps at the full headline size with fixed seeds; run once with the defaults (bf16-pipe
products) and once with INTEL_GEMM_B3=0 INTEL_WGRAD_B3=0 (fp32-MFMA products) and compare the printed numbers."""
import sys
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
e = IntELEngine(m, 'IntBPRloss', args, lr=1e-3, l2=1e-4)
b = synth.make_batch('tmall', 4096, dev, seed=1)
B, Lm = b['i_id_s'].shape
g = torch.Generator(device=dev); g.manual_seed(5)
noise = torch.rand(B, Lm, Lm, device=dev, generator=g)
for s in range(3):
    loss, el, il = e.train_step(b, noise=noise)
    print('step %d loss %.9f ens %.9f int %.9f' % (s, float(loss), float(el), float(il)))
m.eval()
out, nd = e.eval_step(b)
print('ndcg3 %.9f  w_sum %.9f  table_sum %.6f' % (float(nd.float().nan_to_num(0).mean()), float(m.i_W1.weight.double().sum()), float(m.iid_embeddings.weight.double().abs().sum())))
