"""CPU-side checks of the C-ABI boundary: the library builds/loads without a GPU, exports every symbol
include/intel_hip.h declares, the ctypes mirrors match the C structs, argument validation works and the
product refuses to run off-GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

from tests.helpers import ROOT


def _declared_functions():
    txt = open(os.path.join(ROOT, 'include', 'intel_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    names = re.findall(r'\b(intel_[a-z0-9_]+)\s*\(', txt)
    return sorted(set(names))


def test_library_builds_and_exports_every_declared_symbol():
    from intel_sigir2023_amd import build, _lib
    path = build.build_library()
    assert os.path.exists(path)
    lib = _lib.lib()
    declared = _declared_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), 'libintel_hip.so does not export ' + name
    assert sorted(_lib.EXPORTS) == declared, (sorted(set(declared) ^ set(_lib.EXPORTS)))
    assert lib.intel_abi_version() == 5


def test_struct_mirrors_match_header():
    from intel_sigir2023_amd import _lib
    sizes = (C.c_int * 4)()
    _lib.lib().intel_abi_sizes(sizes)
    assert list(sizes) == [C.sizeof(_lib.IntelDesc), C.sizeof(_lib.IntelBatch), C.sizeof(_lib.IntelOut), _lib.P_COUNT]


def _desc(**over):
    from intel_sigir2023_amd import _lib
    d = dict(model_num=3, intent_num=30, item_num=1000, class_num=60, user_num=100, ctx_num=50, d_id=16, d_im=16, d_u=32,
             d_s=32, d_c=16, d_int=16, q_size=32, heads=1, layers=1, cross_attention=1, encoder=0, history_max=20,
             enc_layers=2, enc_heads=2, gru_hidden=128)
    d.update(over)
    return _lib.IntelDesc(**d)


def test_create_validates_descriptor_and_workspace_grows():
    from intel_sigir2023_amd import _lib
    lib = _lib.lib()
    ctx = lib.intel_create(C.byref(_desc()))
    assert ctx
    w1 = lib.intel_workspace_bytes(ctx, 8, 50, 20, 20, 1)
    w2 = lib.intel_workspace_bytes(ctx, 16, 50, 20, 20, 1)
    assert 0 < w1 < w2
    assert lib.intel_workspace_bytes(ctx, 0, 50, 20, 20, 1) == 0
    lib.intel_destroy(ctx)
    assert not lib.intel_create(C.byref(_desc(model_num=0)))
    assert b'model_num' in lib.intel_last_error()
    assert not lib.intel_create(C.byref(_desc(encoder=7)))
    assert b'Invalid sequence encoder' in lib.intel_last_error()        # same message as IntEL.py:111
    assert not lib.intel_create(C.byref(_desc(d_id=10)))


def test_product_has_no_cpu_path():
    from intel_sigir2023_amd import _lib, ops
    from tests.helpers import Fixture, build_model
    with pytest.raises(_lib.IntelHipError):
        ops.linear(torch.zeros(4, 4), torch.zeros(4, 4))
    fx = Fixture('default')
    model, _ = build_model(fx, torch.device('cpu'))
    with pytest.raises(_lib.IntelHipError):
        model(fx.batch())


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'intel_sigir2023_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.cpp', '.hip', '.h')):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in txt and 'from oracle' not in txt and 'oracle/' not in txt.replace('# ', ''), f


def test_dispatcher_ops_are_registered():
    """`torch.ops.intel_mi355x.*` exist with schemas after importing the package's ops module (no GPU needed to register)."""
    from intel_sigir2023_amd import ops
    assert ops.REGISTERED_OPS == ['linear', 'linear_dgrad', 'linear_wgrad', 'attention', 'attention_bwd', 'add_layernorm', 'ndcg', 'intel_forward', 'intel_backward'], \
        getattr(ops, '_REGISTER_ERROR', None)
    for name in ops.REGISTERED_OPS:
        assert getattr(torch.ops.intel_mi355x, name).default._schema.name == 'intel_mi355x::' + name
    with pytest.raises(Exception):          # no CPU implementation: the product has no CPU path
        torch.ops.intel_mi355x.linear(torch.zeros(4, 16), torch.zeros(8, 16), torch.zeros(8), False)
