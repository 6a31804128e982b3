"""Per-step wall times (synchronised) of the first steps over 8 resident batches: which steps pay a first-touch cost?
usage (GPU box): python tools/steptimes.py [workload] [batch] [zipf]"""
import sys, time
sys.path.insert(0, '.')
import torch
from intel_sigir2023_amd import synth
from intel_sigir2023_amd.engine import IntELEngine
from intel_sigir2023_amd.model import IntEL
wl = sys.argv[1] if len(sys.argv) > 1 else 'lifedata'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
zipf = len(sys.argv) > 3 and sys.argv[3] == '1'
dev = torch.device('cuda:0')
args = synth.make_args(wl, dev)
corpus, _ = synth.make_corpus(wl)
torch.manual_seed(0)
m = IntEL(args, corpus).to(dev)
e = IntELEngine(m, 'IntBPRloss', args, lazy_table='auto')
bs = [synth.make_batch(wl, B, dev, seed=i, zipf=zipf) if zipf else synth.make_batch(wl, B, dev, seed=i) for i in range(8)]
for b in bs:
    b['_intel'] = m.prepare_batch(b)
    b['_intel'][1]['ranking_i32'] = b['ranking']
ts = []
for i in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e.train_step(bs[i % 8])
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print(wl, B, 'lazy' if e._lazy is not None else 'dense', ' '.join('%.2f' % t for t in ts))
# the bench's bracket: 5 warm-up steps, 20 timed steps without per-step synchronisation, the lazy table's flush inside the bracket
import os
DEFER = os.environ.get('DEFER', '1') == '1'
for nb in (4, 8):
    m2 = IntEL(args, corpus).to(dev)
    e2 = IntELEngine(m2, 'IntBPRloss', args, lr=1e-3, l2=1e-4, lazy_table='auto')
    e2.defer_table_wait = DEFER
    for b in bs:
        b['_intel'] = m2.prepare_batch(b)
        b['_intel'][1]['ranking_i32'] = b['ranking']
    for i in range(5):
        e2.train_step(bs[i % nb])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20):
        e2.train_step(bs[i % nb])
    torch.cuda.synchronize(); t1 = time.perf_counter()
    e2.flush()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print('%d batches: 20 steps %.3f ms/step, flush %.2f ms' % (nb, 1e3 * (t1 - t0) / 20, 1e3 * (t2 - t1)))
