"""Times the row GEMM of one ablated library (tools/b3_ablate.sh) on the tower shapes.  usage: b3_ablate.py <bits>"""
import json, os, sys
sys.path.insert(0, '.')
import torch
from intel_sigir2023_amd import _lib
bits = sys.argv[1] if len(sys.argv) > 1 else '0'
if bits != '0':
    _lib.LIB_PATH = os.path.join('tools', 'ablate', 'libintel_hip_%s.so' % bits)
from intel_sigir2023_amd import ops
dev = torch.device('cuda:0')
lib = _lib.lib()
for M, K, N in ((204800, 128, 128), (204800, 128, 384), (204800, 64, 64), (4096, 128, 128)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / 11; b = torch.randn(N, device=dev)
    for _ in range(3): y = ops.linear(x, w, b)
    torch.cuda.synchronize()
    lib.intel_prof_enable(1)
    for _ in range(20): y = ops.linear(x, w, b)
    p = json.loads(lib.intel_prof_collect().decode())
    lib.intel_prof_enable(0)
    for k, v in p.items():
        if 'gemm' in k:
            us = 1e3 * v['ms'] / v['launches']
            print('ablate=%s %dx%dx%d %-40s %.1f us  %.2f TB/s' % (bits, M, N, K, k[:40], us, 4.0 * M * (K + N) / us / 1e6))
