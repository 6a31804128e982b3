"""The GRU4Rec recurrence (csrc/gru.hip) in its three forms.  The default is the one-kernel recurrence with three-plane bf16
products (INTEL_GRU_SEQ=2), which the fixture / fuzz parity suites exercise in this process; the switch is read once per
process, so the same suites are re-run in child processes with the exact-fp32-MFMA recurrence kernel (1) and with the per-step
form (0: hidden GEMM + gate kernel per step, the round-1 path): forward outputs, losses and every parameter gradient against
the reference fixtures (gru_bpr: models/GeneralSeq.py:58-78 through torch.nn.GRU) and the oracle's autograd, unchanged
tolerances.  A direct comparison of the three forms on one batch with ragged histories (lengths 0 .. T, a batch that is not a
multiple of the 16-session workgroup tile) follows."""
import os
import subprocess
import sys

import pytest

from tests.helpers import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('mode', ['0', '1'])
def test_parity_suites_in_the_other_recurrence_forms(mode):
    env = dict(os.environ, INTEL_GRU_SEQ=mode)
    # the GRU fixture's tests + one seed of the randomised sweep (six configurations, a third of them GRU4Rec): the recurrence form is all that differs
    r = subprocess.run([sys.executable, '-m', 'pytest', 'tests/test_model_gpu.py', 'tests/test_fuzz_gpu.py::test_random_configs_match_oracle_autograd[%s]' % mode,
                        '-m', 'gpu', '-x', '-q', '-p', 'no:cacheprovider', '-k', 'gru or random_configs'], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]


_CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from intel_sigir2023_amd import synth
from intel_sigir2023_amd.model import IntEL
dev = torch.device('cuda:0')
torch.manual_seed(11)
args = synth.make_args('tiny', dev, encoder='GRU4Rec', cal_diversity=0)
corpus, _ = synth.make_corpus('tiny')
model = IntEL(args, corpus).to(dev)
batch = synth.make_batch('tiny', 37, dev, seed=5, ragged=True)
batch['history_len'][:3] = 0                     # sessions without history keep h = 0
batch['history_item_len'][3:6] = 0
model.train()
out = model(batch)
loss = (out['ens_score'] * torch.linspace(0.5, 1.5, out['ens_score'].numel(), device=dev).view_as(out['ens_score'])).sum() + out['intents'].square().sum()
loss.backward()
torch.save({'out': {k: v.detach().cpu() for k, v in out.items()},
            'grads': {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None}}, sys.argv[1])
'''


def test_three_forms_agree_on_ragged_histories(tmp_path):
    import torch
    res = {}
    for mode in ('0', '1', '2'):
        f = str(tmp_path / ('gru%s.pt' % mode))
        r = subprocess.run([sys.executable, '-c', _CHILD % ROOT, f], cwd=ROOT, env=dict(os.environ, INTEL_GRU_SEQ=mode, INTEL_GRU_ORDER_MIN_B='1'),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout[-3000:]
        res[mode] = torch.load(f)
    ref = res['0']
    assert any('rnn' in k for k in ref['grads']), 'GRU parameters carry no gradient'
    for mode in ('1', '2'):
        for k, v in ref['out'].items():
            assert float((res[mode]['out'][k] - v).abs().max()) <= 2e-6 * max(1.0, float(v.abs().max())), (mode, k)
        for k, v in ref['grads'].items():
            err = float((res[mode]['grads'][k] - v).abs().max())
            assert err <= 2e-5 * max(1e-3, float(v.abs().max())), (mode, k, err, float(v.abs().max()))


def test_packed_gru_histories_equal_padded_histories(monkeypatch):
    """GRU4Rec encoders on the valid history rows only (the batch carries 'his_rows' / 'hisitem_rows': input projection, recurrence
    stashes and the weight / input gradient products run on the packed rows) == on the padded [B, T] rows: forward outputs and
    every parameter gradient, ragged histories including empty ones."""
    import torch
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.model import IntEL
    dev = torch.device('cuda:0')
    res = []
    # packed: sessions ordered by length too (the default from 1024 sessions on); padded: batch order
    for packed in (True, False):
        monkeypatch.setenv('INTEL_GRU_ORDER_MIN_B', '1' if packed else '100000')
        torch.manual_seed(12)
        args = synth.make_args('tiny', dev, encoder='GRU4Rec', cal_diversity=0)
        corpus, _ = synth.make_corpus('tiny')
        model = IntEL(args, corpus).to(dev)
        batch = synth.make_batch('tiny', 37, dev, seed=6, ragged=True)
        assert 'his_rows' in batch and 'hisitem_rows' in batch
        if not packed:
            batch.pop('his_rows'), batch.pop('hisitem_rows')
        model.train()
        out = model(batch)
        assert bool(model._ctx) and (batch['history_len'] < batch['his_context_mh'].shape[1]).any()
        assert ('his_order' in model.prepare_batch(batch)[1]) == packed
        loss = (out['ens_score'] * torch.linspace(0.5, 1.5, out['ens_score'].numel(), device=dev).view_as(out['ens_score'])).sum() + out['intents'].square().sum()
        loss.backward()
        res.append(({k: v.detach().cpu() for k, v in out.items()}, {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None}))
    (o1, g1), (o0, g0) = res
    for k, v in o0.items():
        assert float((o1[k] - v).abs().max()) <= 2e-6 * max(1.0, float(v.abs().max())), k
    assert any('rnn' in k for k in g0)
    for k, v in g0.items():
        err = float((g1[k] - v).abs().max())
        assert err <= 2e-5 * max(1e-3, float(v.abs().max())), (k, err, float(v.abs().max()))
