import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # a fresh checkout has no libsnet_hip.so yet: cross-compile it once (hipcc needs no GPU); an existing
    # library is left alone -- rebuilding is `python -m sevennet_amd.build` / __graft_entry__.build()
    from sevennet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH) and not os.environ.get('SNET_HIP_LIB'):
        from sevennet_amd.build import build
        build(verbose=False)


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
