"""The GPU kernel parity suite (tests/test_kernels_gpu.py) re-run on CPU tensors through the host
emulation of the HIP execution model (tests/hipemu): the product's HIP translation unit compiled
unchanged as host C++, threads as fibers, wave64 shuffles and __syncthreads as yield points.

What this proves in the GPU-less build container: every kernel's index arithmetic, LDS tiling, halo
handling, shuffle/barrier structure, launch geometry and the Python host wrappers agree with the CPU
oracle at the same tolerances as on the device.  What it cannot prove: anything about the gfx950
code generation, memory model or performance — the `-m gpu` suite remains the parity gate.
The emulation library is test infrastructure; the product never loads it.
"""
import importlib.util
import os

import pytest

from tests_hipemu import patch as emu_patch  # noqa: E402  (registered in conftest.py)

HERE = os.path.dirname(os.path.abspath(__file__))

if emu_patch.build_emu.host_compiler() is None:
    pytest.skip("no host clang++ for the HIP emulation build", allow_module_level=True)

_spec = importlib.util.spec_from_file_location("_kernels_gpu_on_emu", os.path.join(HERE, "test_kernels_gpu.py"))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
_mod.DEV = "cpu"

# (test name, parametrize id) too slow for a fiber-per-thread emulation, covered by smaller ids of the same test
SLOW = {
}


@pytest.fixture(autouse=True)
def _emulated():
    with emu_patch.emulated_ops():
        yield


for _name in dir(_mod):
    if _name.startswith("test_"):
        globals()[_name] = getattr(_mod, _name)
