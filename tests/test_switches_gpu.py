"""Every A/B switch of DESIGN.md section 6 keeps a superseded (or alternative) code path alive.  The switches are read once per
process, so each non-default path is exercised here by re-running a compact parity set in a child process with the switch set:
the reference fixtures' forward outputs and parameter gradients (tests/test_model_gpu.py: six model configurations from the
default 32-wide towers to the stress shape) and the engine steps on the synthetic workloads against the oracle
(tests/test_engine_gpu.py), at the unchanged tolerances."""
import os
import subprocess
import sys

import pytest

from tests.helpers import ROOT

pytestmark = pytest.mark.gpu

MODEL = ['tests/test_model_gpu.py', '-k', 'forward or grads']
ENGINE = ['tests/test_engine_gpu.py', '-k', 'synthetic_workloads or adam']
BF16 = ['tests/test_bf16_gpu.py', '-k', 'emulating_oracle']      # the bf16 mode's forward / gradient parity against its emulating oracle

CASES = [
    ({'INTEL_GEMM_B3': '0'}, MODEL),                                   # fp32-MFMA row GEMMs (LDS-DMA form)
    ({'INTEL_GEMM_B3': '0', 'INTEL_GLDS': '0'}, MODEL),                # ... register-prefetch form
    ({'INTEL_WGRAD_B3': '0'}, MODEL),                                  # fp32-MFMA weight gradients (LDS-DMA form)
    ({'INTEL_WGRAD_B3': '0', 'INTEL_WGRAD_DMA': '0'}, MODEL),          # ... register-prefetch form
    ({'INTEL_ATTN_FUSED_BWD': '0', 'INTEL_ENC_FUSED': '0'}, MODEL),    # whole-sequence attention backward as dK/dV kernel + dQ kernel
    ({'INTEL_ATTN_SEQ': '0', 'INTEL_ENC_FUSED': '0'}, MODEL),          # flash-style general attention for every shape
    ({'INTEL_ATTN_DS': '0'}, MODEL),                                   # general attention backward recomputes S / dP in the dQ pass
    ({'INTEL_ATTN_P3': '0'}, MODEL),                                   # general attention (lists / histories > 64) on exact fp32 MFMAs instead of the three-plane bf16-pipe kernels (attn_p3.hip)
    ({'INTEL_BWD_WIDE': '0'}, MODEL),                                  # one-call backward runs its two branch sets one after the other
    ({'INTEL_FUSE_TAIL': '0'}, MODEL),                                 # towers' last LayerNorm as its own store / kernel
    ({'INTEL_GEMM_SMALL': '0'}, MODEL),                                # odd B-row products on the generic kernel
    ({'INTEL_STREAMS': '0'}, MODEL),                                   # whole step on the caller's stream
    ({'INTEL_POS_GRAD_PACKED': '0'}, MODEL),                           # position-embedding gradient through the LDS-atomic kernel
    ({'INTEL_PACK_HISTORY': '0'}, MODEL),                              # encoders on the padded [B, H] rows
    ({'INTEL_GEMM_XCD': '0'}, MODEL),                                  # row-GEMM grids not rounded to the XCD count
    ({'INTEL_FUSE_TOWER_D64': '0'}, MODEL),                            # fp32 training keeps the 64-wide tower on the kernel-per-op pipeline
    ({'INTEL_ENC_FUSED_BWD': '0'}, MODEL),                             # kernel-per-op encoder backward on the fused forward's stash
    ({'INTEL_WGRAD_SLABS': '64', 'INTEL_WGRAD_CORESIDENT': '0'}, MODEL),
    ({'INTEL_MODEL_OP': '1'}, MODEL),                                  # IntEL.forward through torch.ops.intel_mi355x.intel_forward
    ({'INTEL_STREAMS': '0'}, ENGINE),                                  # ... the engine's table sweep still has to wait for the backward
    ({'INTEL_BWD_WIDE': '0'}, ENGINE),
    ({'INTEL_ADAM_ROWS': '0'}, ENGINE),                                # dense Adam kernel over the item-id table
    ({'INTEL_OVERLAP_TABLE': '0'}, ENGINE),                            # table sweep on the main stream
    ({'INTEL_BWD_SCHEDULE': 'phased'}, ENGINE),                        # two-call backward
    ({'INTEL_SCATTER_SORTED': '1'}, ENGINE),                           # always the sorted embedding scatter
    ({'INTEL_BPR_NOISE': 'tensor'}, ENGINE),                           # BPR tie-breaking noise as a torch.rand tensor
    ({'INTEL_TOWER32': '0'}, MODEL),                                   # 32-wide towers on the kernel-per-op pipeline instead of the one-kernel tower (tower32.hip)
    ({'INTEL_HEAD_FUSED': '0'}, MODEL),                                # session head as one launch per link instead of the chain launches (chain.hip)
    ({'INTEL_HEAD_FUSED': '0'}, ENGINE),
    ({'INTEL_PACK_SIDE': '0'}, MODEL),                                 # weight packing on the caller's stream even where no branch reads a packed image
    ({'INTEL_ENC32': '0'}, MODEL),                                     # 32-wide BERT4Rec encoders on the kernel-per-op pipeline instead of the one-kernel encoder (tower32.hip: enc32_*)
    ({'INTEL_WGRAD_TR': '0'}, BF16),                                   # bf16 mode: 128 x 128 weight gradients through the transposed-staging kernel
]


@pytest.mark.parametrize('env,target', CASES, ids=[','.join('%s=%s' % kv for kv in e.items()) for e, _ in CASES])
def test_parity_set_with_switch(env, target):
    r = subprocess.run([sys.executable, '-m', 'pytest'] + target + ['-m', 'gpu', '-x', '-q', '-p', 'no:cacheprovider'], cwd=ROOT,
                       env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
