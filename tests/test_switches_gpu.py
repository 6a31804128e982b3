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
FUZZ = ['tests/test_model_gpu.py', 'tests/test_fuzz_gpu.py', '-k', 'forward or grads or random_configs']      # + random shapes (1 - 2 heads, 1 - 2 tied layers)
BF16 = ['tests/test_bf16_gpu.py', '-k', 'emulating_oracle']      # the bf16 mode's forward / gradient parity against its emulating oracle

CASES = [
    ({'INTEL_GEMM_B3': '0'}, MODEL, [], ['gemm_rows_b3_kernel', 'gemm_rows_b3k_kernel']),                                   # fp32-MFMA row GEMMs (LDS-DMA form)
    ({'INTEL_WGRAD_B3': '0'}, MODEL, ['wgrad_pipe_kernel'], ['wgrad_b3_kernel', 'wgrad_b3_batch_kernel']),                                  # fp32-MFMA weight gradients (LDS-DMA form)
    ({'INTEL_ATTN_SEQ': '0', 'INTEL_ENC_FUSED': '0'}, MODEL, ['attn_fwd_kernel'], ['attn_seq_fwd_kernel', 'attn_seq_bwd_fused_kernel', 'attn_seq_bwd_kv_kernel', 'enc_block_fwd_kernel']),          # flash-style general attention for every shape
    ({'INTEL_ATTN_P3': '0'}, MODEL, ['attn_fwd_kernel', 'attn_bwd_dkv_kernel'], ['attn_fwd_p3_kernel', 'attn_bwd_dkv_p3_kernel', 'attn_bwd_dq_ds_p3_kernel']),                                   # general attention (lists / histories > 64) on exact fp32 MFMAs instead of the three-plane bf16-pipe kernels (attn_p3.hip)
    ({'INTEL_WGRAD_SLABS': '64'}, MODEL, [], []),                              # 64 partial slabs per weight gradient instead of 128
    ({'INTEL_STREAMS': '0'}, MODEL, [], []),                                   # whole step on the caller's stream
    ({'INTEL_PACK_HISTORY': '0'}, MODEL, [], ['his_pack_kernel', 'enc_block_fwd_kernel']),                              # encoders on the padded [B, H] rows
    ({'INTEL_ENC_FUSED_BWD': '0'}, MODEL, [], ['enc_block_bwd_kernel', 'enc_last_bwd_kernel']),                             # kernel-per-op encoder backward on the fused forward's stash
    ({'INTEL_MODEL_OP': '1'}, MODEL, [], []),                                  # IntEL.forward through torch.ops.intel_mi355x.intel_forward
    ({'INTEL_STREAMS': '0'}, ENGINE, [], []),                                  # ... the engine's table sweep still has to wait for the backward
    ({'INTEL_ADAM_ROWS': '0'}, ENGINE, ['adam_kernel'], []),                                # dense Adam kernel over the item-id table
    ({'INTEL_OVERLAP_TABLE': '0'}, ENGINE, [], []),                            # table sweep on the main stream
    ({'INTEL_BWD_SCHEDULE': 'phased'}, ENGINE, [], []),                        # two-call backward
    ({'INTEL_SCATTER_SORTED': '1'}, ENGINE, ['scatter_add_sorted_kernel'], []),                           # always the sorted embedding scatter
    ({'INTEL_BPR_NOISE': 'tensor'}, ENGINE, [], []),                           # BPR tie-breaking noise as a torch.rand tensor
    ({'INTEL_TOWER32': '0'}, MODEL, [], ['tw32_fwd_kernel', 'tw32_bwd_kernel']),                                   # 32-wide towers on the kernel-per-op pipeline instead of the one-kernel tower (tower32.hip)
    ({'INTEL_HEAD_FUSED': '0'}, MODEL, [], ['chain_kernel']),                                # session head as one launch per link instead of the chain launches (chain.hip)
    ({'INTEL_HEAD_FUSED': '0'}, ENGINE, [], ['chain_kernel']),
    ({'INTEL_PACK_SIDE': '0'}, MODEL, [], []),                                 # weight packing on the caller's stream even where no branch reads a packed image
    ({'INTEL_ENC32': '0'}, MODEL, [], ['enc32_fwd_kernel', 'enc32_bwd_kernel']),                                     # 32-wide BERT4Rec encoders on the kernel-per-op pipeline instead of the one-kernel encoder (tower32.hip: enc32_*)
    ({'INTEL_WGRAD_TR': '0'}, BF16, [], ['wgrad_tr_kernel']),
    ({'INTEL_FUSE_TOWER_BWD': '1'}, FUZZ, ['tower_bwd_fused_kernel'], []),       # the one-kernel tower backward in fp32 too (default: bf16 mode only): fixtures + random shapes
    ({'INTEL_FUSE_TOWER_BWD': '1'}, ENGINE, ['tower_bwd_fused_kernel'], []),     # ... and through the engine's steps (tied layers, Adam)
    ({'INTEL_FUSE_TOWER_BWD': '0'}, BF16, [], ['tower_bwd_fused_kernel']),       # bf16 mode on the kernel-per-op tower backward                                   # bf16 mode: 128 x 128 weight gradients through the transposed-staging kernel
]


@pytest.mark.parametrize('env,target,expect,forbid', CASES, ids=[','.join('%s=%s' % kv for kv in c[0].items()) for c in CASES])
def test_parity_set_with_switch(env, target, expect, forbid):
    """expect / forbid: kernels the child session must / must not launch with the switch set (tests/conftest.py: the dispatch check) -- a switch that no
    longer selects its code path fails here instead of re-testing the default path."""
    env = dict(env, INTEL_EXPECT_KERNELS=','.join(expect), INTEL_FORBID_KERNELS=','.join(forbid))
    r = subprocess.run([sys.executable, '-m', 'pytest'] + target + ['-m', 'gpu', '-x', '-q', '-p', 'no:cacheprovider'], cwd=ROOT,
                       env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
