import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# tests/hipemu (host emulation of the HIP execution model, test infrastructure) as an importable package
import importlib.util as _ilu  # noqa: E402
if "tests_hipemu" not in sys.modules:
    _spec = _ilu.spec_from_file_location("tests_hipemu", os.path.join(ROOT, "tests", "hipemu", "__init__.py"),
                                         submodule_search_locations=[os.path.join(ROOT, "tests", "hipemu")])
    _m = _ilu.module_from_spec(_spec)
    sys.modules["tests_hipemu"] = _m
    _spec.loader.exec_module(_m)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, e.g. `pytest tests/` on the build box."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden(dict):
    """npz fully materialised once (NpzFile re-reads and inflates an array on EVERY access)."""

    @property
    def files(self):
        return list(self.keys())


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return Golden({k: z[k] for k in z.files})


@pytest.fixture(scope="session")
def golden_steps_56():
    return load_golden("steps_56.npz")


@pytest.fixture(scope="session")
def golden_steps_224():
    return load_golden("steps_224.npz")


@pytest.fixture(scope="session")
def golden_steps_56_dual():
    return load_golden("steps_56_dual.npz")


@pytest.fixture(scope="session")
def golden_steps_56_dropout1():
    return load_golden("steps_56_dropout1.npz")


@pytest.fixture(scope="session")
def golden_steps_56_untargeted():
    return load_golden("steps_56_untargeted.npz")


@pytest.fixture(scope="session")
def golden_trace_untargeted():
    return load_golden("trace_56_untargeted.npz")


@pytest.fixture(scope="session")
def golden_geometry():
    return load_golden("geometry.npz")


@pytest.fixture(scope="session", params=["trace_56.npz", "trace_56_fail.npz"])
def golden_trace(request):
    return load_golden(request.param)


@pytest.fixture(scope="session")
def golden_patchcleanser():
    return load_golden("patchcleanser_56.npz")
