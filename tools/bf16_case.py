"""One case of the bf16-mode randomised sweep (tools/fuzz_parity.py N seed --dtype bf16) taken apart: every gradient tensor's error (Frobenius norm,
relative) against the EMULATING oracle and against the plain fp32 oracle, and -- for a first-linear gradient that is off -- which hidden units carry
the difference and how close to zero the oracle's pre-activation of those units comes.
usage (GPU box): python tools/bf16_case.py <case index> [seed = 61616]"""
import random
import sys

sys.path.insert(0, '.')
import torch

import importlib.util
import os
spec = importlib.util.spec_from_file_location('fuzz_parity', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fuzz_parity.py'))
fuzz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fuzz)

idx = int(sys.argv[1])
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 61616
rng = random.Random(seed)
dev = torch.device('cuda:0')
for i in range(idx):
    fuzz.one_case(rng, i, dev, dry=True, dtype='bf16')
fuzz.VERBOSE = True
worst, bad, desc = fuzz.one_case(rng, idx, dev, dtype='bf16')
print('case %d worst %.2f at %s  %s' % (idx, worst, bad, desc))
