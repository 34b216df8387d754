import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# scratch trees of the hardware repros (tests/tools/hw/README.md builds variant libraries, and once unpacked a whole older tree,
# under tests/tools/hw/_*): never collected -- a second tests/test_abi.py in there once broke the collection of `pytest tests`
collect_ignore_glob = ["tools/hw/_*"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "experiments: needs the experiments build of the library (NPA_EXPERIMENTS=1 python -m "
                                       "neupan_amd.build --force): the kernels DESIGN.md section 7 keeps on record, not in the product")


def experiments_built():
    try:
        from neupan_amd import _lib
        return b"+experiments" in _lib.load().npa_version()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests are skipped automatically when no GPU is visible, so a plain
    `pytest tests/` on a CPU box stays green; the driver selects with -m explicitly."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if any("experiments" in item.keywords for item in items) and not experiments_built():
        skip_x = pytest.mark.skip(reason="the library is the product build (no experiments: NPA_EXPERIMENTS=1 python -m neupan_amd.build --force)")
        for item in items:
            if "experiments" in item.keywords:
                item.add_marker(skip_x)
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
