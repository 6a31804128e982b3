"""intel_adam_step_pair (ABI v4): both dense parameter groups of BaseModel.customize_parameters (models/BaseModel.py:53-62) in one launch --
bit-identical to two intel_adam_step calls, odd sizes and an empty group included."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('na,nb', [(250013, 771), (64, 0), (4097, 3)])
def test_pair_equals_two_single_group_steps(na, nb):
    from intel_sigir2023_amd import _lib as L
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    dev = torch.device('cuda:0')
    lib = L.lib()
    g = torch.Generator(device=dev)
    g.manual_seed(na + nb)
    mk = lambda n: [torch.randn(max(n, 4), device=dev, generator=g)[:n].contiguous() for _ in range(4)]
    A, Bv = mk(na), mk(nb)
    for t in (A, Bv):
        t[3].abs_()                                  # second moment
    ref = [[x.clone() for x in A], [x.clone() for x in Bv]]
    st = L.stream_ptr(dev)
    for step in (1, 2, 7):
        for grp, wd in ((ref[0], 1e-4), (ref[1], 0.0)):
            if grp[0].numel():
                grp[1].copy_(torch.sin(grp[0] * step))      # a fresh gradient each step
                L.check(lib.intel_adam_step(L.ptr(grp[0]), L.ptr(grp[1]), L.ptr(grp[2]), L.ptr(grp[3]), grp[0].numel(), 1e-3, 0.9, 0.999, 1e-8, wd, step, 1.0, 1, st), 'adam')
        for grp in (A, Bv):
            if grp[0].numel():
                grp[1].copy_(torch.sin(grp[0] * step))
        arr = lambda k: (C.c_void_p * 2)(A[k].data_ptr() if na else None, Bv[k].data_ptr() if nb else None)
        L.check(lib.intel_adam_step_pair(arr(0), arr(1), arr(2), arr(3), (C.c_longlong * 2)(na, nb), (C.c_float * 2)(1e-4, 0.0), 1e-3, 0.9, 0.999, 1e-8, step, 1.0, 1, st),
                'adam_pair')
        torch.cuda.synchronize()
        for got, want in zip(A + Bv, ref[0] + ref[1]):
            assert torch.equal(got, want)
