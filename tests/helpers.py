"""Shared helpers for the tests: fixture loading (tests/golden/*.npz)."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
PROJ_SEED = 12345
CONFIG_NAMES = ['default', 'gru_bpr', 'noxatt', 'tmall64', 'lifedata', 'stress']


class Fixture(object):
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, 'intel_%s.npz' % name))
        meta = json.loads(str(self.z['cfg']))
        self.args = meta['args']
        self.shape = meta['shape']
        self.detail = str(self.z['detail'])

    def group(self, prefix):
        pl = len(prefix) + 1
        return {k[pl:]: self.z[k] for k in self.z.files if k.startswith(prefix + '/')}

    def state_dict(self, device='cpu'):
        return {k: torch.from_numpy(v).to(device) for k, v in self.group('sd').items()}

    def batch(self, device='cpu'):
        b = {k: torch.from_numpy(v).to(device) for k, v in self.group('in').items()}
        b['batch_size'] = int(b['u_id_c'].shape[0])
        b['phase'] = 'train'
        return b

    def __getitem__(self, k):
        return self.z[k]


def grad_projection(g):
    r = np.random.default_rng(PROJ_SEED).standard_normal(g.shape)
    g64 = np.asarray(g, dtype=np.float64)
    return np.array([(g64 * r).sum(), np.sqrt((g64 * g64).sum())])


def make_args(fx_args, device):
    import argparse
    ns = argparse.Namespace(**fx_args)
    ns.device = device
    return ns


def make_corpus(shape):
    import types
    return types.SimpleNamespace(itemfnum=[shape['classes']], contextfnum=[shape['ctx']],
                                 zero_int=np.zeros(shape['I']), max_uid=shape['users'] - 1,
                                 max_iid=shape['items'] - 1)


def build_model(fx, device):
    """The product model initialised from a fixture's reference state_dict."""
    from intel_sigir2023_amd import model as M
    args = make_args(fx.args, device)
    model = getattr(M, fx.args.get('model_name', 'IntEL'))(args, make_corpus(fx.shape))
    missing = model.load_state_dict(fx.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return model.to(device), args
