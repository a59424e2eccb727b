"""tests/test_skip_satisfied_gpu.py's main case on CPU tensors through the host emulation of the HIP kernels, with a
shallow ResNetV2 of the same block types at 112 x 112 (the smallest input the fused stem pooling takes; the emulation
runs one fiber per GPU thread, so the other cases stay GPU-only)."""
import importlib.util
import os

import pytest
import torch

from tests_hipemu import patch as emu_patch  # noqa: E402  (registered in conftest.py)

HERE = os.path.dirname(os.path.abspath(__file__))

if emu_patch.build_emu.host_compiler() is None:
    pytest.skip("no host clang++ for the HIP emulation build", allow_module_level=True)

_spec = importlib.util.spec_from_file_location("_skip_satisfied_on_emu", os.path.join(HERE, "test_skip_satisfied_gpu.py"))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
_mod.DEV = "cpu"
_mod.H, _mod.S, _mod.B = 112, 4, 1
_mod.LAYERS, _mod.N_CLASSES = (1, 1, 1, 2), 10
_mod.DETERMINISTIC = False            # no run-twice reproducibility probe under emulation
_mod.LAYOUTS = {"split": dict(micro_batch=2, ladder=[1, 2])}
del _mod.pytestmark


@pytest.fixture(autouse=True)
def _emulated(monkeypatch):
    with emu_patch.emulated_ops():
        yield


def test_skipping_satisfied_samples_changes_nothing():
    _mod.test_skipping_satisfied_samples_changes_nothing.__wrapped__("split") \
        if hasattr(_mod.test_skipping_satisfied_samples_changes_nothing, "__wrapped__") \
        else _mod.test_skipping_satisfied_samples_changes_nothing("split")

