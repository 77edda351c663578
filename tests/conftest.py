import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: a full-size case whose CPU oracle needs minutes (still part of -m gpu)')


def pytest_collection_modifyitems(config, items):
    """-m gpu tests must never silently pass without a device: skip them on CPU-only hosts only when
    the user did not ask for them explicitly."""
    import torch
    if torch.cuda.is_available():
        return
    mexpr = config.getoption('-m') or ''
    for it in items:
        if 'gpu' in it.keywords and 'gpu' not in mexpr.replace('not gpu', ''):
            it.add_marker(pytest.mark.skip(reason='no GPU in this container'))


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return load
