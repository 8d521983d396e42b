"""pytest configuration: marker registration + import paths.

`-m "not gpu"`: oracle vs the reference's golden answers, host logic, C-ABI symbol checks (CPU only).
`-m gpu`      : parity tests proper -- every call goes through the C-ABI of libcosmo_hip.so on a MI355X.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with `-m gpu` through gpurun)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # a gpu-marked test that is selected on a box without a GPU is an error in the harness, not a skip
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (gpu tests run through gpurun)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
