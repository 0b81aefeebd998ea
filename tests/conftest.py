import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'moviigen1.1_amd')
for p in (ROOT, PKG, os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + '.npz')))
    return load


@pytest.fixture(autouse=True)
def _reset_kernel_selection(request):
    """mg_attn_set_variant / mg_gemm_set_variant / the debug hooks are process-global test switches: whatever a GPU
    test selected is undone after it, so no test can change the launches of a later one."""
    yield
    if 'gpu' not in request.keywords:
        return
    from wan.backend import lib
    assert not lib._use_ab, 'a test left an ab_library() scope open'
    h = lib._lib_ab      # the product library has no switches; the A/B one is reset if a test loaded it
    if h is not None:
        h.mg_attn_set_variant(lib.DEFAULT_ATTN_VARIANT)
        h.mg_gemm_set_variant(lib.DEFAULT_GEMM_VARIANT)
        h.mg_attn_w64_debug(0)
        h.mg_attn_w64_flag_counter(None)
