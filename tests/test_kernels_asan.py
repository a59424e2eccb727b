"""Memory checking of the HIP kernels (the role compute-sanitizer / rocgdb memcheck play on a device; SURVEY §4's
"race detection, failure detection" aux row): the product's HIP translation unit, compiled unchanged as host C++ for the
emulation (tests/hipemu), is built with AddressSanitizer and the kernel parity suite re-runs under it in a child
process — with the emulator's thread and block schedule REVERSED (schedule fuzzing: a kernel whose result depends on the
order in which the threads between two barriers, or the blocks of a launch, execute — a missing `__syncthreads()` around
an LDS tile, an inter-block dependency — then fails its parity test; a second canary shows that).  Every `__shared__` array, every local array and — through ASan's malloc interceptor — every tensor carries red
zones, so an out-of-bounds LDS tile access, a halo that runs past the image or a store past the end of an output
buffer aborts the run.  A canary proves the set-up catches such a store by one of the product's kernels.

TEST INFRASTRUCTURE ONLY: nothing here is loaded by the product."""
import os
import subprocess
import sys

import pytest

from tests_hipemu import build_emu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
RUNTIME = build_emu.asan_runtime()

if build_emu.host_compiler() is None or RUNTIME is None:
    pytest.skip("no host clang++ / shared AddressSanitizer runtime", allow_module_level=True)


def _env():
    # HIPEMU_ORDER=1: the same child run also executes the threads of every block in DESCENDING order and the blocks of
    # every launch in reverse (schedule fuzzing, see below) — the plain emulation suites run ascending.  HIPEMU_POISON=1:
    # torch.empty / empty_like hand out NaN / sentinels, so an output element a kernel never writes fails the comparison
    env = dict(os.environ, DORPATCH_EMU_SANITIZE="1", LD_PRELOAD=RUNTIME, PYTHONPATH=ROOT, HIPEMU_ORDER="1", HIPEMU_POISON="1",
               ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:verify_asan_link_order=0:halt_on_error=1:"
                            "abort_on_error=0:exitcode=86")
    return env


CANARY = r'''
import ctypes, sys, torch
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(tests)r)
import importlib.util, os
spec = importlib.util.spec_from_file_location("tests_hipemu", os.path.join(%(tests)r, "hipemu", "__init__.py"),
                                              submodule_search_locations=[os.path.join(%(tests)r, "hipemu")])
mod = importlib.util.module_from_spec(spec); sys.modules["tests_hipemu"] = mod; spec.loader.exec_module(mod)
from tests_hipemu import patch
lib = patch.emu_lib()
x = torch.rand(4, 8, 8)
y = torch.empty(4 * 4 * 4 - %(short)d)            # dp_subsample2 writes 4 x 4 x 4 floats
rc = lib.dp_subsample2(x.data_ptr(), 4, 8, 8, y.data_ptr(), None)
print("returned", rc)
'''


def _run_canary(short):
    code = CANARY % dict(root=ROOT, tests=HERE, short=short)
    return subprocess.run([sys.executable, "-c", code], env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=600)


def test_asan_catches_a_kernel_store_past_its_output_buffer():
    ok = _run_canary(0)
    assert ok.returncode == 0 and "returned 0" in ok.stdout, ok.stderr[-2000:]
    bad = _run_canary(8)                                   # the last 8 floats of the output do not exist
    assert bad.returncode != 0 and "AddressSanitizer" in bad.stderr and "heap-buffer-overflow" in bad.stderr, \
        (bad.returncode, bad.stderr[-2000:])
    assert "k_subsample2" in bad.stderr                    # the report names the kernel


def test_kernel_suite_is_clean_under_address_sanitizer():
    """Default: the kernel suite minus its slowest cases (the whole-network test and the two largest residual-GroupNorm
    shapes; the smaller shapes of the same kernels stay in).  DORPATCH_ASAN_FULL=1: everything, plus the selected-sample backward (dp_gn_relu_bwd_gather)
    and the affine-placement kernels — run clean in the round-2 build container."""
    full = os.environ.get("DORPATCH_ASAN_FULL", "0") == "1"
    files = [os.path.join(HERE, "test_kernels_emu.py")]
    select = []
    if full:
        files += [os.path.join(HERE, "test_taped_emu.py"), os.path.join(HERE, "test_placement_emu.py")]
    else:
        # ... and, of the matrix-core convolution kernels (a 448-pixel x 64-channel MFMA tile costs the fibre emulation under
        # ASan a minute), ONE emulation-sized case per kernel: the others run in the plain emulation suite and on the GPU
        # (round 6: the pixel-tile bit-identity sweeps launch every tile of every mode — the tiles themselves are covered here by
        # the kept cases below, whose `auto` / `64px` parameters run the small-tile instantiations; one Winograd case, the odd side)
        mfma = ("on_the_matrix_cores or conv3x3_flat_kernel or conv3x3_with_folded or conv1x1_with_folded or "
                "conv1x1_launch_variants or pixel_tiles")
        keep = ("(conv3x3_on_the_matrix_cores and 10-8-128-7) or (conv3x3_flat_kernel and 11-8-64-7) or "
                "(stride2_input_gradient and 2-16-64-14) or (stem_convolution and 2-6-True) or "
                "(conv3x3_stride2_on and 10-8-128-14) or (conv1x1_on_the_matrix_cores and 1-16-64-28) or "
                "(conv1x1_with_folded and 1-64-64-28) or (winograd and 3-8-64-7)")
        select = ["-k", "not resnetv2_fused and not (add_gn_relu_fusion and (shape0 or shape1)) and (not (%s) or %s)"
                        % (mfma, keep)]
    res = subprocess.run([sys.executable, "-m", "pytest"] + files + ["-q", "-x", "-p", "no:cacheprovider"] + select,
                         env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=2400)
    tail = (res.stdout + res.stderr)[-3000:]
    assert res.returncode == 0 and "AddressSanitizer" not in res.stdout + res.stderr, tail
    assert " passed" in res.stdout and "failed" not in res.stdout, tail


RACY = r"""
#include <hip/hip_runtime.h>
#include <cstdio>
template <bool SYNC>
__global__ void k_rotate(const int *in, int *out) {
  __shared__ int tile[128];
  tile[threadIdx.x] = in[threadIdx.x];
  if (SYNC) __syncthreads();                       // without it the read below races with the neighbour's write
  out[threadIdx.x] = tile[(threadIdx.x + 1) & 127];
}
int main(int argc, char **argv) {
  int in[128], out[128];
  for (int i = 0; i < 128; ++i) { in[i] = i + 1; out[i] = -1; }
  const int *pin = in;
  int *pout = out;
  if (argv[1][0] == 'r') hipLaunchKernelGGL(k_rotate<false>, dim3(1), dim3(128), 0, nullptr, pin, pout);
  else hipLaunchKernelGGL(k_rotate<true>, dim3(1), dim3(128), 0, nullptr, pin, pout);
  long sum = 0;
  for (int i = 0; i < 128; ++i) sum = sum * 31 + out[i];
  printf("%ld\n", sum);
  return 0;
}
"""


def test_schedule_fuzzing_exposes_a_missing_barrier(tmp_path):
    """A kernel with a missing __syncthreads() gives different results under the ascending and the reversed schedule;
    the same kernel with the barrier does not."""
    src = tmp_path / "racy.cpp"
    src.write_text(RACY)
    exe = str(tmp_path / "racy")
    subprocess.run([build_emu.host_compiler(), "-x", "c++", "-std=c++17", "-O1", "-w",
                    "-I", os.path.join(HERE, "hipemu"), str(src), "-o", exe], check=True, capture_output=True)

    def run(which, order):
        return subprocess.run([exe, which], env=dict(os.environ, HIPEMU_ORDER=str(order)), capture_output=True, text=True,
                              check=True).stdout.strip()
    assert run("s", 0) == run("s", 1) == run("s", 2)
    assert len({run("r", 0), run("r", 1), run("r", 2)}) > 1
