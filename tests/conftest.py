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
    # the multi-process cases (N ranks sharing the box's one GPU over gloo: the stress configuration of the suite) run LAST, so
    # that under ``-x`` a failure there cannot leave the parity cases unexecuted
    items.sort(key=lambda it: 1 if 'test_gpu_distributed' in it.nodeid else 0)
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


def pytest_sessionstart(session):
    # fill_deterministic's closed-form patterns are functions of (numel, name hash): computed once per session
    from scene_generation_amd import synthetic
    if synthetic.HASH_CACHE is None:
        synthetic.HASH_CACHE = {}
    # the CPU oracle with one thread per physical core inside a 16-CPU cgroup quota (the GPU boxes) spends most of every scheduler
    # period frozen: run the whole session -- oracle and HIP path alike -- with as many threads as the quota allows
    from scene_generation_amd.utils import respect_cpu_quota
    respect_cpu_quota(verbose=False)


import contextlib


@contextlib.contextmanager
def skip_random_init():
    """Construct modules WITHOUT their random initialisation (for objects whose every parameter is overwritten right after
    construction by fill_deterministic / load_state_dict): the normal / uniform draws of a 183 M-parameter generator cost ~8 s
    per Trainer on the host and the full-size step tests build a dozen of them."""
    import torch
    import torch.nn.init as I
    names = ['kaiming_normal_', 'kaiming_uniform_', 'normal_', 'uniform_', 'xavier_uniform_', 'xavier_normal_', 'trunc_normal_',
             'orthogonal_']
    saved = {n: getattr(I, n) for n in names if hasattr(I, n)}
    tn, tu = torch.Tensor.normal_, torch.Tensor.uniform_
    for n in saved:
        setattr(I, n, lambda t, *a, **k: t)
    torch.Tensor.normal_ = lambda self, *a, **k: self
    torch.Tensor.uniform_ = lambda self, *a, **k: self
    try:
        yield
    finally:
        for n, f in saved.items():
            setattr(I, n, f)
        torch.Tensor.normal_, torch.Tensor.uniform_ = tn, tu
