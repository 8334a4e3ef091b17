"""pytest wiring: makes `pit_hip` / `gq_cuda` (under the hyphenated package dir)
and `oracle` importable, registers the `gpu` marker."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "e2e: quantiser parity THROUGH the conv stack (end-to-end goldens, reproducibility, bench contract)")
    config.addinivalue_line("markers", "convstack: conv-stack kernels / routes (no quantiser parity depends on them)")


def _has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


# Collection order of the -m gpu suite (the driver runs it with -x): hot-path parity first, so that nothing in the conv stack --
# which north_star leaves to PyTorch-ROCm; the hand-written stack is a bonus -- can stop the run before the quantiser's parity
# evidence has been produced (round 4: one conv-stack test failed at #240 of 428 and hid every golden added since round 2).
#   0  C-ABI consumer, kernel-boundary parity vs the oracle          3  end-to-end goldens / reproducibility (marker e2e)
#   1  quantiser modules, goldens, rounding-bound attacks            4  conv-stack kernels vs fp64 (marker convstack)
#   2  randomized stress                                             5  conv-stack routes (test_gpu_convstack_routes.py)
_FILE_TIER = {"test_gpu_cabi.py": 0, "test_gpu_kernel_parity.py": 0, "test_gpu_stress.py": 2, "test_gpu_convstack_routes.py": 5}


def _tier(item) -> int:
    if "gpu" not in item.keywords:
        return -1                      # CPU tests keep their place in front
    name = os.path.basename(str(item.fspath))
    if name in _FILE_TIER:
        return _FILE_TIER[name]
    if "convstack" in item.keywords:
        return 4
    if "e2e" in item.keywords:
        return 3
    return 1


def pytest_collection_modifyitems(config, items):
    items.sort(key=_tier)              # stable: file / definition order inside a tier
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
