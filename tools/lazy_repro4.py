"""Kernel-level probe at a large table: lazy step + lazy gather against the dense sweep (debugging aid)."""
import ctypes as C
import sys
sys.path.insert(0, '.')
import torch
from intel_sigir2023_amd import _lib as L

dev = torch.device('cuda:0')
lib = L.lib()
rows, d = int(sys.argv[1]) if len(sys.argv) > 1 else 300000, 64
g = torch.Generator(device=dev).manual_seed(1)
p0 = torch.randn(rows, d, device=dev, generator=g) * 0.1
dense = dict(p=p0.clone(), m=torch.zeros_like(p0), v=torch.zeros_like(p0), g=torch.zeros_like(p0), f=torch.zeros(rows, dtype=torch.uint8, device=dev))
lazy = dict(p=p0.clone(), m=torch.zeros_like(p0), v=torch.zeros_like(p0), g=torch.zeros_like(p0), f=torch.zeros(rows, dtype=torch.uint8, device=dev))
cap = 64
last = torch.zeros(rows, dtype=torch.int32, device=dev)
sched = torch.zeros(cap, 2, device=dev)
b1, b2, eps, wd = 0.9, 0.999, 1e-8, 1e-4
t = L.IntelLazyTable(p=lazy['p'].data_ptr(), m=lazy['m'].data_ptr(), v=lazy['v'].data_ptr(), last=last.data_ptr(), sched=sched.data_ptr(),
                     rows=rows, d=d, base=0, cap=cap, beta1=b1, beta2=b2, eps=eps, weight_decay=wd)
st = L.stream_ptr(dev)
print('exports', [n for n in dir(lib) if 'lazy' in n][:10])
bad = 0
for step in range(1, 9):
    n = 60000
    idx = torch.randint(0, rows, (n,), device=dev, generator=g).unique()
    grad = torch.randn(idx.numel(), d, device=dev, generator=g) * 1e-3
    for s in (dense, lazy):
        s['g'][idx] = grad
        s['f'][idx] = 1
    L.check(lib.intel_adam_step_rows(L.ptr(dense['p']), L.ptr(dense['g']), L.ptr(dense['m']), L.ptr(dense['v']), rows, d, L.ptr(dense['f']), 1e-3, b1, b2, eps, wd, step, 1.0, st), 'dense')
    L.check(lib.intel_adam_lazy_step(C.byref(t), L.ptr(lazy['g']), L.ptr(lazy['f']), 1e-3, step, st), 'lazy')
    torch.cuda.synchronize()
    # what a forward pass would gather: compare the dense table rows with a replay in torch of the lazy state? simpler: flush a COPY
    lp, lm, lv, ll = lazy['p'].clone(), lazy['m'].clone(), lazy['v'].clone(), last.clone()
    t2 = L.IntelLazyTable(p=lp.data_ptr(), m=lm.data_ptr(), v=lv.data_ptr(), last=ll.data_ptr(), sched=sched.data_ptr(), rows=rows, d=d, base=0, cap=cap,
                          beta1=b1, beta2=b2, eps=eps, weight_decay=wd)
    L.check(lib.intel_adam_lazy_flush(C.byref(t2), step, st), 'flush')
    torch.cuda.synchronize()
    nb = int((lp != dense['p']).any(1).sum())
    bad += nb
    print('step', step, 'rows differing after flush of a copy:', nb, 'flags left', int(lazy['f'].sum()), 'g left', float(lazy['g'].abs().max()))
print('TOTAL bad', bad)
