"""The training (or evaluation) step as it really overlaps: every launch bracketed by two HIP events on its own stream
(intel_prof_enable + intel_prof_timeline), branches left on their streams.  Far less intrusive than a tracing profiler, whose
per-launch host cost makes the step host-bound.
usage (GPU box): python tools/step_timeline.py [f32|bf16] [train|eval] [full] [workload=tmall] [batch=4096]"""
import json
import re
import sys
from collections import Counter

sys.path.insert(0, '.')
import torch

from intel_sigir2023_amd import _lib, synth
from intel_sigir2023_amd.engine import IntELEngine
from intel_sigir2023_amd.model import IntEL

dtype = sys.argv[1] if len(sys.argv) > 1 else 'f32'
mode = sys.argv[2] if len(sys.argv) > 2 else 'train'
full = len(sys.argv) > 3 and sys.argv[3] == 'full'
wl = sys.argv[4] if len(sys.argv) > 4 else 'tmall'
BS = int(sys.argv[5]) if len(sys.argv) > 5 else 4096
dev = torch.device('cuda:0')
args = synth.make_args(wl, dev, dtype=dtype)
corpus, _ = synth.make_corpus(wl)
torch.manual_seed(0)
model = IntEL(args, corpus).to(dev)
eng = IntELEngine(model, 'IntBPRloss', args)
bs = [synth.make_batch(wl, BS, dev, seed=i) for i in range(4)]
for b in bs:
    b['_intel'] = model.prepare_batch(b)
    b['_intel'][1]['ranking_i32'] = b['ranking']
if mode == 'eval':
    model.eval()
step = (lambda i: eng.eval_step(bs[i % 4])) if mode == 'eval' else (lambda i: eng.train_step(bs[i % 4]))
for i in range(6):
    step(i)
torch.cuda.synchronize()
lib = _lib.lib()
lib.intel_prof_enable(1)
N = 4
for i in range(N):
    step(i)
tl = json.loads(lib.intel_prof_timeline().decode())
lib.intel_prof_enable(0)


def short(n):
    n = re.sub(r'^\(', '', n)
    return n.split('<')[0].split('[')[0].split(')')[0]


marker = 'ndcg_kernel' if mode == 'eval' else 'bpr_loss_kernel'
marks = [i for i, r in enumerate(tl) if short(r['name']) == marker]
lo, hi = marks[1], marks[2]
seg = tl[lo:hi]
t0 = seg[0]['t0']
wall = tl[hi]['t0'] - t0
ev = []
end = tl[hi]['t0']
for r in seg:        # (launch order is enqueue order: a launch of this step may run past the next step's first kernel -- clipped)
    a0 = min(r['t0'], end)
    ev.append((a0, 1, short(r['name'])))
    ev.append((max(a0, min(r['t1'], end)), -1, short(r['name'])))
ev.sort(key=lambda x: (x[0], -x[1]))
live, last, busy = [], t0, 0.0
conc, alone, gaps = Counter(), Counter(), Counter()
prev = seg[0]['name']
for t, k, n in ev:
    if live:
        busy += t - last
        conc[min(len(live), 4)] += t - last
        if len(live) == 1:
            alone[live[0]] += t - last
    elif t > last:
        gaps[prev] += t - last
    if k == 1:
        live.append(n)
    else:
        live.remove(n)
        prev = n
    last = t
print('%s %s step: wall %.3f ms, busy (union) %.3f ms, summed %.3f ms, idle %.3f ms' %
      (dtype, mode, wall, busy, sum(max(0.0, min(r['t1'], end) - min(r['t0'], end)) for r in seg), wall - busy))
print('ms with k launches in flight:', {k: round(v, 3) for k, v in sorted(conc.items())})
print('alone:', [(k, round(v, 3)) for k, v in alone.most_common(14)])
print('idle after:', [(k, round(v, 3)) for k, v in gaps.most_common(8)])
if full:
    for r in seg:
        print('%8.1f %7.1f  s%d %s' % ((r['t0'] - t0) * 1e3, (r['t1'] - r['t0']) * 1e3, r['stream'], r['name'][:70]))
