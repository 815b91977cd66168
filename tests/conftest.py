import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _ensure_library():
    """Build libamp_engine.so if it is missing or older than its sources (hipcc cross-compiles without a GPU)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("_amp_build", os.path.join(ROOT, "humanoid_amp_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build_library(force=False, verbose=False)


def pytest_configure(config):
    _ensure_library()
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU-marked tests are skipped (not failed) when no GPU is visible."""
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
