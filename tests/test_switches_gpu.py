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
FUZZ = ['tests/test_model_gpu.py', 'tests/test_fuzz_gpu.py::test_random_configs_match_oracle_autograd[0]', '-k', 'forward or grads or random_configs']      # + six random shapes (1 - 2 heads, 1 - 2 tied layers)
BF16 = ['tests/test_bf16_gpu.py', '-k', 'emulating_oracle']      # the bf16 mode's forward / gradient parity against its emulating oracle

# Independent switches share a child run (round 6: 25 single-switch runs of ~4 s process start each were a quarter of the GPU suite's wall time).  A group
# combines switches whose code paths do not mask each other; the child must launch every kernel in `expect` and none in `forbid` (tests/conftest.py), so
# a switch that stopped selecting its path still turns the run red.
CASES = [
    # everything on the kernel-per-op pipeline with exact-fp32 MFMA products: fp32-MFMA row GEMMs and weight gradients, general attention (lists /
    # histories > 64) on exact fp32 MFMAs, 32-wide towers / BERT4Rec encoders kernel-per-op instead of the one-kernel forms (tower32.hip), the session head
    # as one launch per link (chain.hip), separate data / weight gradient kernels instead of the one-pass linear / q-k-v backwards (pair.hip), the kernel-per-op
    # encoder backward on the fused forward's stash, packing on the caller's stream, 64 partial slabs per weight gradient
    ({'INTEL_GEMM_B3': '0', 'INTEL_WGRAD_B3': '0', 'INTEL_ATTN_P3': '0', 'INTEL_TOWER32': '0', 'INTEL_ENC32': '0', 'INTEL_HEAD_FUSED': '0', 'INTEL_PAIR_BWD': '0',
      'INTEL_ENC_FUSED_BWD': '0', 'INTEL_PACK_SIDE': '0', 'INTEL_WGRAD_SLABS': '64'}, MODEL, ['wgrad_pipe_kernel', 'attn_fwd_kernel', 'attn_bwd_dkv_kernel'],
     ['gemm_rows_b3_kernel', 'gemm_rows_b3k_kernel', 'wgrad_b3_kernel', 'wgrad_b3_batch_kernel', 'attn_fwd_p3_kernel', 'attn_bwd_dkv_p3_kernel', 'attn_bwd_dq_ds_p3_kernel',
      'tw32_fwd_kernel', 'tw32_bwd_kernel', 'enc32_fwd_kernel', 'enc32_bwd_kernel', 'chain_kernel', 'linear_bwd_pair_kernel', 'linear_bwd_qkv_kernel', 'enc_block_bwd_kernel', 'enc_last_bwd_kernel']),
    # the default (bf16-pipe) kernels, one switch family at a time where a group above would hide them: the b3 GEMMs with the kernel-per-op 32-wide paths
    ({'INTEL_TOWER32': '0', 'INTEL_ENC32': '0', 'INTEL_HEAD_FUSED': '0', 'INTEL_PAIR_BWD': '0'}, MODEL, ['wgrad_b3_kernel', 'gemm_rows_b3_kernel'],
     ['tw32_fwd_kernel', 'tw32_bwd_kernel', 'enc32_fwd_kernel', 'enc32_bwd_kernel', 'chain_kernel', 'linear_bwd_pair_kernel', 'linear_bwd_qkv_kernel']),
    # flash-style general attention for every shape, encoders on the padded [B, H] rows, the whole step on the caller's stream, IntEL.forward through
    # torch.ops.intel_mi355x.intel_forward
    ({'INTEL_ATTN_SEQ': '0', 'INTEL_ENC_FUSED': '0', 'INTEL_PACK_HISTORY': '0', 'INTEL_STREAMS': '0', 'INTEL_MODEL_OP': '1'}, MODEL, ['attn_fwd_kernel'],
     ['attn_seq_fwd_kernel', 'attn_seq_bwd_fused_kernel', 'enc_block_fwd_kernel', 'his_pack_kernel']),
    # engine steps: one stream (the table sweep still has to wait for the backward), dense Adam kernel over the item-id table, always the sorted embedding
    # scatter, BPR tie-breaking noise as a torch.rand tensor, kernel-per-op session head, the table sweep released before the backward's last reduction
    ({'INTEL_STREAMS': '0', 'INTEL_ADAM_ROWS': '0', 'INTEL_SCATTER_SORTED': '1', 'INTEL_BPR_NOISE': 'tensor', 'INTEL_HEAD_FUSED': '0'}, ENGINE,
     ['adam_kernel', 'scatter_add_sorted_kernel'], ['chain_kernel']),
    ({'INTEL_OVERLAP_TABLE': '0'}, ENGINE, [], []),                            # table sweep on the main stream
    ({'INTEL_BWD_SCHEDULE': 'phased', 'INTEL_FUSE_TOWER_BWD': '1'}, ENGINE, ['tower_bwd_fused_kernel'], []),      # two-call backward; the one-kernel tower backward in fp32 through the engine's steps (tied layers, Adam)
    ({'INTEL_TABLE_AFTER_FLUSH': '0'}, ENGINE, [], []),                        # four streams, the table sweep released as soon as the table gradient is complete
    ({'INTEL_FUSE_TOWER_BWD': '1'}, FUZZ, ['tower_bwd_fused_kernel'], []),       # the one-kernel tower backward in fp32 too (default: bf16 mode only): fixtures + random shapes
    # bf16 mode: 128 x 128 weight gradients through the transposed-staging kernel, the kernel-per-op tower backward
    ({'INTEL_WGRAD_TR': '0', 'INTEL_FUSE_TOWER_BWD': '0'}, BF16, [], ['wgrad_tr_kernel', 'tower_bwd_fused_kernel']),
]


@pytest.mark.parametrize('env,target,expect,forbid', CASES, ids=[','.join('%s=%s' % kv for kv in c[0].items()) for c in CASES])
def test_parity_set_with_switch(env, target, expect, forbid):
    """expect / forbid: kernels the child session must / must not launch with the switch set (tests/conftest.py: the dispatch check) -- a switch that no
    longer selects its code path fails here instead of re-testing the default path."""
    env = dict(env, INTEL_EXPECT_KERNELS=','.join(expect), INTEL_FORBID_KERNELS=','.join(forbid))
    r = subprocess.run([sys.executable, '-m', 'pytest'] + target + ['-m', 'gpu', '-x', '-q', '-p', 'no:cacheprovider'], cwd=ROOT,
                       env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
