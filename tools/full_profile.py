"""Every kernel of a training step, one at a time on one stream (intel_prof): launches, time, time per launch and the bytes the
launcher declares per second -- the whole list, not the top 16 of the bench line.  usage: python tools/full_profile.py [--shapes] [workload batch]"""
import json, sys
sys.path.insert(0, '.')
import torch
from intel_sigir2023_amd import _lib, synth
from intel_sigir2023_amd.engine import IntELEngine
from intel_sigir2023_amd.model import IntEL
dev = torch.device('cuda:0')
pos = [a for a in sys.argv[1:] if not a.startswith('--')]
wl = pos[0] if pos else 'tmall'
BS = int(pos[1]) if len(pos) > 1 else 4096
args = synth.make_args(wl, dev)
corpus, _ = synth.make_corpus(wl)
torch.manual_seed(0)
m = IntEL(args, corpus).to(dev)
e = IntELEngine(m, 'IntBPRloss', args, lr=1e-3, l2=1e-4)
bs = [synth.make_batch(wl, BS, dev, seed=i) for i in range(4)]
for b in bs:
    b['_intel'] = m.prepare_batch(b); b['_intel'][1]['ranking_i32'] = b['ranking']
for i in range(5): e.train_step(bs[i % 4])
torch.cuda.synchronize()
lib = _lib.lib()
lib.intel_set_concurrency(m._context(), 0)
e.overlap_table_update = False
lib.intel_prof_enable(1)
N = 4
for i in range(N): e.train_step(bs[i % 4])
p = json.loads(lib.intel_prof_collect().decode())
lib.intel_prof_enable(0)
import re, collections
agg = collections.defaultdict(lambda: {'ms': 0.0, 'launches': 0, 'bytes': 0.0})
for k, v in p.items():
    name = re.sub(r'\[.*$', '', k)
    for f in ('ms', 'launches'):
        agg[name][f] += v[f]
    agg[name]['bytes'] += v.get('bytes', 0)
p = agg if '--shapes' not in sys.argv else p
rows = sorted(p.items(), key=lambda kv: -kv[1]['ms'])
tot = sum(v['ms'] for _, v in rows) / N
print('total kernel ms/step %.3f, launches/step %.1f' % (tot, sum(v['launches'] for _, v in rows) / N))
for k, v in rows:
    ms = v['ms'] / N
    gb = v.get('bytes', 0) / N / 1e9
    print('%-70s %5.1f launches  %7.4f ms  %6.1f us/launch  %7.1f GB/s' % (k[:70], v['launches'] / N, ms, 1e3 * v['ms'] / v['launches'], gb / (ms * 1e-3) if ms > 0 else 0))
