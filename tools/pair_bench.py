"""Standalone timing of the one-pass linear backward (csrc/pair.hip) against the two kernels it replaces, at the towers' row counts.
usage: python tools/pair_bench.py [M] ; prints microseconds per call (median of 20, HIP events on the launch stream)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from intel_sigir2023_amd import ops, _lib as L

dev = torch.device('cuda:0')
M = int(sys.argv[1]) if len(sys.argv) > 1 else 204800


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for d in (128, 64):
    g = torch.Generator().manual_seed(1)
    x = torch.relu(torch.randn(M, d, generator=g)).to(dev)
    dy = torch.randn(M, d, generator=g).to(dev)
    w = (torch.randn(d, d, generator=g) / d ** 0.5).to(dev)
    # the op wrappers allocate + pack per call: time the raw entry points with preallocated buffers
    lib = L.lib()
    dx = torch.empty(M, d, device=dev); dw = torch.empty(d, d, device=dev); db = torch.empty(d, device=dev)
    nb = lib.intel_op_linear_bwd_workspace_bytes(M, d)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    nb2 = lib.intel_op_workspace_bytes(M, d, d)
    ws2 = torch.empty(nb2, dtype=torch.uint8, device=dev)
    st = L.stream_ptr(dev)
    for mask in (0, 1):
        t_pair = timeit(lambda: L.check(lib.intel_op_linear_bwd(L.ptr(dy), L.ptr(x), M, d, L.ptr(w), mask, L.ptr(dx), L.ptr(dw), L.ptr(db), L.ptr(ws), nb, st)))
        print('d=%d M=%d mask=%d  pair (pack + kernel + reduce): %.1f us' % (d, M, mask, t_pair))
    t_d = timeit(lambda: L.check(lib.intel_op_linear_dgrad(L.ptr(dy), M, d, L.ptr(w), d, L.ptr(dx), L.ptr(ws2), nb2, st)))
    t_w = timeit(lambda: L.check(lib.intel_op_linear_wgrad(L.ptr(dy), L.ptr(x), M, d, d, L.ptr(dw), L.ptr(db), L.ptr(ws2), nb2, st)))
    print('d=%d M=%d  dgrad (pack + kernel): %.1f us   wgrad (kernel + reduce): %.1f us   sum %.1f us' % (d, M, t_d, t_w, t_d + t_w))
    # kernel-only times from the library's profiler (HIP events around each launch), one variant at a time
    def kernel_us(fn, n=10):
        lib.intel_prof_collect()
        lib.intel_prof_enable(1)
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        import json
        r = json.loads(lib.intel_prof_collect().decode())
        lib.intel_prof_enable(0)
        return {k.split('[')[0].strip('()').split('<')[0]: round(1e3 * v['ms'] / v['launches'], 1) for k, v in r.items()}
    for mask in (1, 0, 1, 0):
        print('d=%d mask=%d kernels (us):' % (d, mask), kernel_us(lambda: L.check(lib.intel_op_linear_bwd(L.ptr(dy), L.ptr(x), M, d, L.ptr(w), mask, L.ptr(dx), L.ptr(dw), L.ptr(db), L.ptr(ws), nb, st))))
    print('d=%d separate kernels (us):' % d, kernel_us(lambda: (L.check(lib.intel_op_linear_dgrad(L.ptr(dy), M, d, L.ptr(w), d, L.ptr(dx), L.ptr(ws2), nb2, st)),
                                                              L.check(lib.intel_op_linear_wgrad(L.ptr(dy), L.ptr(x), M, d, d, L.ptr(dw), L.ptr(db), L.ptr(ws2), nb2, st)))))

# ---- the fused q/k/v projection's backward (linear_bwd_qkv_kernel) against wgrad_b3 (three column blocks) + gemm_rows_b3k
for d, nb in ((128, 3), (64, 3)):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(M, d, generator=g).to(dev)
    dy = torch.randn(M, nb * d, generator=g).to(dev)
    res = torch.randn(M, d, generator=g).to(dev)
    w = (torch.randn(nb * d, d, generator=g) / d ** 0.5).to(dev)
    lib = L.lib()
    dx = torch.empty(M, d, device=dev); dw = torch.empty(nb * d, d, device=dev)
    nb_ws = lib.intel_op_linear_bwd_qkv_workspace_bytes(M, d, nb)
    ws = torch.empty(nb_ws, dtype=torch.uint8, device=dev)
    nb2 = lib.intel_op_workspace_bytes(M, nb * d, nb * d)
    ws2 = torch.empty(nb2, dtype=torch.uint8, device=dev)
    st = L.stream_ptr(dev)

    def kernel_us(fn, n=10):
        import json
        lib.intel_prof_collect()
        lib.intel_prof_enable(1)
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        r = json.loads(lib.intel_prof_collect().decode())
        lib.intel_prof_enable(0)
        return {k.split('[')[0].strip('()').split('<')[0]: round(1e3 * v['ms'] / v['launches'], 1) for k, v in r.items()}
    one = lambda: L.check(lib.intel_op_linear_bwd_qkv(L.ptr(dy), L.ptr(x), L.ptr(res), M, d, nb, L.ptr(w), L.ptr(dx), L.ptr(dw), None, L.ptr(ws), nb_ws, st))
    for _ in range(3):
        one()
    print('qkv d=%d nb=%d one pass (us):' % (d, nb), kernel_us(one))
    print('qkv d=%d nb=%d one pass (us):' % (d, nb), kernel_us(one))
    sep = lambda: (L.check(lib.intel_op_linear_dgrad(L.ptr(dy), M, nb * d, L.ptr(w), d, L.ptr(dx), L.ptr(ws2), nb2, st)),
                   L.check(lib.intel_op_linear_wgrad(L.ptr(dy), L.ptr(x), M, nb * d, d, L.ptr(dw), None, L.ptr(ws2), nb2, st)))
    for _ in range(3):
        sep()
    print('qkv d=%d nb=%d separate kernels (us; the dgrad here is the exact-fp32 K > 128 kernel -- the model uses gemm_rows_b3k with a weight image):' % (d, nb), kernel_us(sep))
