"""The one-kernel tower layer (csrc/tower.hip) in TRAINING.  By default the fp32 build trains through the kernel-per-op
pipeline (it is the faster step) and uses the one-kernel layer for inference and in bf16 mode; INTEL_FUSE_TOWER=1 forces it
everywhere.  The switch is read once per process, so the fixture / fuzz parity suites are re-run in a child process with
the one-kernel layer forced on: forward outputs, losses and every parameter gradient against the reference fixtures and the
oracle's autograd, at the unchanged tolerances."""
import os
import subprocess
import sys

import pytest

from tests.helpers import ROOT

pytestmark = pytest.mark.gpu


def test_parity_suites_with_the_one_kernel_layer_forced_on():
    env = dict(os.environ, INTEL_FUSE_TOWER='1')
    r = subprocess.run([sys.executable, '-m', 'pytest', 'tests/test_model_gpu.py', 'tests/test_fuzz_gpu.py::test_random_configs_match_oracle_autograd[2]',
                        'tests/test_pack_gpu.py', '-m', 'gpu', '-x', '-q', '-p', 'no:cacheprovider'], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]


def test_kernel_per_op_pipeline_when_forced_off():
    env = dict(os.environ, INTEL_FUSE_TOWER='0')
    r = subprocess.run([sys.executable, '-m', 'pytest', 'tests/test_model_gpu.py', '-m', 'gpu', '-x', '-q', '-p', 'no:cacheprovider', '-k',
                        'forward or grads'], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
