import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + '.npz')))
        return cache[name]
    return load


def weights(d, tag):
    """{'<tag>_w_<name>': arr} -> {name: torch tensor}"""
    import torch
    pre = tag + '_w_'
    return {k[len(pre):]: torch.from_numpy(v) for k, v in d.items() if k.startswith(pre)}
