"""Times the bf16-pipe weight-gradient kernel of one ablated library (tools/wgrad_ablate.sh) on the tower shapes.
usage: wgrad_ablate.py <bits> [slabs]"""
import ctypes as C, json, os, sys
sys.path.insert(0, '.')
import torch
from intel_sigir2023_amd import _lib
bits = sys.argv[1] if len(sys.argv) > 1 else '0'
if bits != '0':
    _lib.LIB_PATH = os.path.join('tools', 'ablate', 'libintel_hip_w%s.so' % bits)
dev = torch.device('cuda:0')
lib = _lib.lib()
st = _lib.stream_ptr(dev)
SHAPES = ((204800, 128, 128), (204800, 384, 128), (204800, 64, 64), (42752, 128, 128))
if os.environ.get('WGRAD_SHAPES'):      # e.g. WGRAD_SHAPES=42566x128x128,42560x384x128
    SHAPES = tuple(tuple(int(v) for v in t.split('x')) for t in os.environ['WGRAD_SHAPES'].split(','))
for M, N, K in SHAPES:
    dy = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev)
    dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
    nb = lib.intel_op_workspace_bytes(M, N, K)
    ws = torch.empty(int(nb) + 256, dtype=torch.uint8, device=dev)
    def run():
        _lib.check(lib.intel_op_linear_wgrad(_lib.ptr(dy), _lib.ptr(x), M, N, K, _lib.ptr(dw), _lib.ptr(db), _lib.ptr(ws), ws.numel(), st), 'wgrad')
    for _ in range(3): run()
    torch.cuda.synchronize()
    lib.intel_prof_enable(1)
    for _ in range(20): run()
    p = json.loads(lib.intel_prof_collect().decode())
    lib.intel_prof_enable(0)
    for k, v in p.items():
        if 'wgrad' in k or 'slab' in k:
            us = 1e3 * v['ms'] / v['launches']
            print('ablate=%s %dx%dx%d %-44s %.1f us  %.2f TB/s' % (bits, M, N, K, k[:44], us, 4.0 * M * (K + N) / us / 1e6))
