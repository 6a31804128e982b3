"""Times the attention operators through the C ABI (per-kernel HIP events of the built-in profiler).
usage: attn_bench.py [long] [path of another libintel_hip.so]      (long: the LifeData / stress list shapes, general kernels)"""
import sys, os, json
sys.path.insert(0, '.')
import torch
from intel_sigir2023_amd import ops, _lib
dev = torch.device('cuda:0')
shapes = [(4096, 50, 128, 1), (4096, 50, 64, 1), (4096, 20, 128, 2)]
args = sys.argv[1:]
if args and args[0] == 'long':
    shapes = [(4096, 100, 128, 1), (4096, 100, 64, 1), (1024, 200, 128, 1), (1024, 200, 64, 1), (1024, 200, 128, 2)]
    args = args[1:]
if args:
    _lib.LIB_PATH = args[0]
lib = _lib.lib()
for (B, T, d, heads) in shapes:
    qkv = torch.randn(B * T, 3 * d, device=dev)
    dout = torch.randn(B * T, d, device=dev)
    for _ in range(2):
        out, lse = ops.attention(qkv, B, T, d, heads, None)
        ops.attention_bwd(qkv, out, dout, lse, B, T, d, heads, None)
    torch.cuda.synchronize()
    lib.intel_prof_enable(1)
    for _ in range(10):
        out, lse = ops.attention(qkv, B, T, d, heads, None)
        ops.attention_bwd(qkv, out, dout, lse, B, T, d, heads, None)
    p = json.loads(lib.intel_prof_collect().decode())
    lib.intel_prof_enable(0)
    for k, v in p.items():
        if 'attn' in k:
            print((B, T, d, heads), k.split('[')[0], '%.1f us' % (1e3 * v['ms'] / v['launches']),
                  '%.1f TF/s' % (v['flops'] / v['ms'] / 1e9))
