"""Randomised parity sweep (tools/fuzz_parity.py) as a GPU test: outputs, Int* losses and every parameter gradient of the
HIP path against the oracle's autograd on shapes the fixtures do not hold."""
import importlib.util
import os
import random

# the GRU recurrence takes its sessions ordered by history length from batches of 1024 sessions on; the sweep's batches are small:
# order them too, so that the ordered + packed recurrence is compared with the oracle directly (read per prepare_batch call)
os.environ.setdefault('INTEL_GRU_ORDER_MIN_B', '1')

import pytest
import torch

pytestmark = pytest.mark.gpu

_spec = importlib.util.spec_from_file_location('fuzz_parity', os.path.join(os.path.dirname(__file__), '..', 'tools', 'fuzz_parity.py'))
fuzz = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(fuzz)


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_random_configs_match_oracle_autograd(seed):
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    rng = random.Random(seed)
    dev = torch.device('cuda:0')
    for i in range(6):
        worst, bad, desc = fuzz.one_case(rng, 1000 * seed + i, dev)
        assert worst <= 1.0, (worst, bad, desc)
