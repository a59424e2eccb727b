"""tests/test_end_metric_gpu.py on CPU tensors through the HIP emulation (tests/hipemu): ~10 minutes for the 8
two-stage runs (27 minutes for the 32-problem null test), so opt-in (DORPATCH_EMU_FULL=1); the `-m gpu` run is the gate.
Round 3: the null test run here FAILED exactly like on the GPU (product total 22 635 failures vs a null of 24 362-25 010)
while the old 8-image bands passed — which is what located the aliasing in oracle/ref_shim.py: the
fixtures have since been re-recorded through the corrected shim."""
import importlib.util
import os

import pytest
import torch

from tests_hipemu import patch as emu_patch  # noqa: E402  (registered in conftest.py)

HERE = os.path.dirname(os.path.abspath(__file__))

if os.environ.get("DORPATCH_EMU_FULL", "0") != "1":
    pytest.skip("opt-in: DORPATCH_EMU_FULL=1 (about 10 minutes)", allow_module_level=True)
if emu_patch.build_emu.host_compiler() is None:
    pytest.skip("no host clang++ for the HIP emulation build", allow_module_level=True)

_spec = importlib.util.spec_from_file_location("_end_metric_on_emu", os.path.join(HERE, "test_end_metric_gpu.py"))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
_mod.DEV = "cpu"
del _mod.pytestmark


@pytest.fixture(autouse=True)
def _emulated(monkeypatch):
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    with emu_patch.emulated_ops():
        yield


test_certified_asr_matches_reference = _mod.test_certified_asr_matches_reference
test_end_metric_is_a_plausible_draw_from_the_reference_null = _mod.test_end_metric_is_a_plausible_draw_from_the_reference_null
