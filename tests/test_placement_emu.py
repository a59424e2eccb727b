"""tests/test_placement_gpu.py (the affine-placement EXTENSION) on CPU tensors through the HIP emulation."""
import importlib.util
import os

import pytest
import torch

from tests_hipemu import patch as emu_patch  # noqa: E402  (registered in conftest.py)

HERE = os.path.dirname(os.path.abspath(__file__))

if emu_patch.build_emu.host_compiler() is None:
    pytest.skip("no host clang++ for the HIP emulation build", allow_module_level=True)

_spec = importlib.util.spec_from_file_location("_placement_on_emu", os.path.join(HERE, "test_placement_gpu.py"))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
_mod.DEV = "cpu"
del _mod.pytestmark


@pytest.fixture(autouse=True)
def _emulated(monkeypatch):
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    with emu_patch.emulated_ops():
        yield


for _name in dir(_mod):
    if _name.startswith("test_"):
        globals()[_name] = getattr(_mod, _name)
