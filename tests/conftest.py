import os
import sys

# HIP-graph replay of a captured training step (engine/graph_step.py) needs this BEFORE the HIP runtime starts
# (profiles/r04a_hip_graph_flags.txt); harmless for everything else
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "maskrcnn-benchmark_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests are skipped (not failed) when no GPU is visible."""
    try:
        import torch

        have_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _reset_library_tuning():
    """the libraries' tuning switches (detops_tuning_set) are process-wide: every test starts from the defaults"""
    yield
    emu_mod = sys.modules.get("emu")
    if emu_mod is not None:
        emu_mod.tuning_reset()
    lib_mod = sys.modules.get("maskrcnn_benchmark._lib")
    if lib_mod is not None and hasattr(lib_mod, "tuning_set"):
        import emu as _e  # key list only
        for k in _e.TUNING_KEYS:
            lib_mod.lib.detops_tuning_set(k.encode(), 0)


_C_SURFACE = {}
_C_MODULE = []


@pytest.fixture(autouse=True)
def _operator_surface_is_restored():
    """tests that swap entries of `maskrcnn_benchmark._C` (tests/cpu_shim.py, monkeypatch) must leave the module as they
    found it: a leftover stand-in would silently serve every later test"""
    mod = sys.modules.get("maskrcnn_benchmark._C")
    if mod is not None and not _C_MODULE and hasattr(mod, "on_device"):
        _C_MODULE.append(mod)
        _C_SURFACE.update({k: v for k, v in vars(mod).items() if callable(v) or k == "lib"})
    yield
    if _C_MODULE:
        mod = _C_MODULE[0]
        changed = [k for k, v in _C_SURFACE.items() if vars(mod).get(k) is not v]
        for k in changed:
            setattr(mod, k, _C_SURFACE[k])
        assert not changed, "left patched after the test: %s" % changed
