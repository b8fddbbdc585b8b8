import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def has_gpu() -> bool:
    return torch.cuda.is_available()


@pytest.fixture
def emulated_ops(monkeypatch):
    """CPU host-logic tests: swap the HIP op layer for the plain-PyTorch kernel references (tests/kernel_refs.py)
    so the autograd wiring and module logic of lcasr_amd can be checked against the oracle / golden fixtures
    without a GPU.  This is test scaffolding only — the product package has no such switch."""
    import lcasr_amd.functional as Fn
    import kernel_refs
    monkeypatch.setattr(Fn, 'ops', kernel_refs)
    Fn.clear_weight_cache()
    yield kernel_refs
    Fn.clear_weight_cache()


def load_golden(name):
    return np.load(os.path.join(GOLD, name + '.npz'), allow_pickle=False)


def golden_cfg(fx):
    cfg = {}
    for k in fx.files:
        if k.startswith('cfg.'):
            v = fx[k]
            cfg[k[4:]] = v.item() if v.shape == () else v.tolist()
    return cfg


def golden_state_dict(fx):
    return {k[2:]: torch.from_numpy(fx[k].copy()) for k in fx.files if k.startswith('w.')}
