"""The whole-loop GPU parity suite (tests/test_attack_gpu.py: HotLoop.step, DorPatch.generate,
collect_failure, PatchCleanser — against fixtures recorded from the unmodified reference and against
the CPU oracle) re-run on CPU tensors through the host emulation of the HIP execution model
(tests/hipemu).  The toy classifiers run in torch-CPU; every dp_* kernel runs as the product's own
translation unit compiled for the host.  Same tolerances as on the device.

Proves here, without a GPU: the host orchestration (sampling, micro-batching, bookkeeping, stage
control, cache files) and the kernels' logic reproduce the reference.  The `-m gpu` suite remains the
parity gate for the gfx950 build.  The emulation library is test infrastructure; the product never
loads it.
"""
import importlib.util
import os

import pytest
import torch

from tests_hipemu import patch as emu_patch  # noqa: E402  (registered in conftest.py)

HERE = os.path.dirname(os.path.abspath(__file__))

if emu_patch.build_emu.host_compiler() is None:
    pytest.skip("no host clang++ for the HIP emulation build", allow_module_level=True)

_spec = importlib.util.spec_from_file_location("_attack_gpu_on_emu", os.path.join(HERE, "test_attack_gpu.py"))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
_mod.DEV = "cpu"
del _mod.pytestmark

# full-size (1.2 GB) property test: a fiber-per-thread emulation of 2048 x 224^2 samples takes too long;
# the same kernels are covered at 56/224/384 by tests/test_kernels_emu.py
SKIP = {"test_config2_size_properties"}
# full DorPatch.generate runs against the recorded reference traces: ~3 min each under emulation; opt in with
# DORPATCH_EMU_FULL=1 (test_generate_short_run_both_stages covers the same control flow in seconds)
if os.environ.get("DORPATCH_EMU_FULL", "0") != "1":
    SKIP.add("test_generate_trajectory_tracks_reference")
    SKIP.add("test_generate_untargeted_run_tracks_reference")


@pytest.fixture(autouse=True)
def _emulated(monkeypatch):
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    with emu_patch.emulated_ops():
        yield


for _name in dir(_mod):
    if _name.startswith("test_") and _name not in SKIP:
        globals()[_name] = getattr(_mod, _name)
