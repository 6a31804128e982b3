import sys, os, time, json
sys.path.insert(0, '.')
import torch
from intel_sigir2023_amd import ops, _lib
dev = torch.device('cuda:0')
M, K, N = 204800, 128, 128
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / 11; b = torch.randn(N, device=dev)
for _ in range(3): y = ops.linear(x, w, b)
torch.cuda.synchronize()
lib = _lib.lib(); lib.intel_prof_enable(1)
for _ in range(10): y = ops.linear(x, w, b)
p = json.loads(lib.intel_prof_collect().decode())
for k, v in p.items():
    if 'gemm' in k: print(os.environ.get('INTEL_DEBUG_GEMM', '0'), k, '%.1f us' % (1e3 * v['ms'] / v['launches']), '%.1f TF/s' % (v['flops'] / v['ms'] / 1e9))
