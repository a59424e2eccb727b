"""TEST INFRASTRUCTURE — exec'd by bench.py in every rank process when DORPATCH_BENCH_RANK_HOOK names this file
(tests/test_bench_emu.py: the plain `python bench.py --gpus N --backend gloo` command on a GPU-less box).
Routes dorpatch_amd.ops through the host emulation of the HIP kernels (tests/hipemu) on CPU tensors for the life of
the process.  `bench` is the running bench module (injected by bench.main)."""
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
if "tests_hipemu" not in sys.modules:
    _spec = importlib.util.spec_from_file_location("tests_hipemu", os.path.join(_HERE, "hipemu", "__init__.py"),
                                                   submodule_search_locations=[os.path.join(_HERE, "hipemu")])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules["tests_hipemu"] = _mod
    _spec.loader.exec_module(_mod)
from tests_hipemu import patch as _patch  # noqa: E402

bench.DEVICE_OVERRIDE = "cpu"            # noqa: F821  (injected)
bench._emu_ctx = _patch.emulated_ops()   # noqa: F821
bench._emu_ctx.__enter__()               # noqa: F821
