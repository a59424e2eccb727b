"""Build tests/hipemu/libdorpatch_emu.so: the product's HIP translation unit compiled UNCHANGED as host
C++ against the hipemu shim (tests/hipemu/hip/hip_runtime.h).  TEST INFRASTRUCTURE ONLY — lets
`pytest -m "not gpu"` check every kernel's indexing / LDS / shuffle logic against the CPU oracle in the
GPU-less build container.  The product never loads this library (dorpatch_amd/_lib.py opens only
libdorpatch_hip.so, and dorpatch_amd.ops rejects CPU tensors).

The only source transformation: `extern __shared__ <type> name[];` (dynamic LDS, which has no host
equivalent) becomes a pointer to the emulator's per-launch dynamic-LDS buffer.
"""
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(ROOT, "dorpatch_amd", "csrc", "dorpatch_hip.hip")
SANITIZE = os.environ.get("DORPATCH_EMU_SANITIZE", "0") == "1"      # AddressSanitizer build (tests/test_kernels_asan.py)
OUT = os.path.join(HERE, "libdorpatch_emu_asan.so" if SANITIZE else "libdorpatch_emu.so")
GEN = os.path.join(HERE, "_dorpatch_emu_asan_gen.cpp" if SANITIZE else "_dorpatch_emu_gen.cpp")
CSRC = os.path.dirname(SRC)
DEPS = sorted(os.path.join(CSRC, n) for n in os.listdir(CSRC) if n.endswith((".hip", ".inc"))) + [
    os.path.join(ROOT, "include", "dorpatch_hip.h"), os.path.join(HERE, "hip", "hip_runtime.h"), os.path.abspath(__file__)]

_INC = re.compile(r'^#include "(\w+\.inc)".*$', re.M)
_DYN = re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?(\w+)\s+(\w+)\[\];")


def host_compiler():
    for exe in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++")):
        if exe and os.path.exists(exe):
            return exe
    return None


def build(force=False):
    """-> path of the emulation library, or None when no host clang++ exists (tests then skip)."""
    cxx = host_compiler()
    if cxx is None:
        return None
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in DEPS):
        return OUT
    with open(SRC) as f:
        text = f.read()

    def family(m):      # the kernel families (csrc/*.inc) are spliced in so that the LDS rewrite below reaches them
        path = os.path.join(CSRC, m.group(1))
        with open(path) as g:
            return '#line 1 "%s"\n%s\n#line %d "%s"' % (path, g.read(), text.count("\n", 0, m.end()) + 2, SRC)
    text = _INC.sub(family, text)
    text = _DYN.sub(lambda m: "%s *%s = static_cast<%s *>(hipemu::dyn_lds_ptr());" % (m.group(1), m.group(2), m.group(1)),
                    text)
    # per-process scratch names: the ranks of a multi-process test may find the library stale at the same moment
    gen, tmp = "%s.%d.cpp" % (GEN[:-4], os.getpid()), "%s.tmp.%d" % (OUT, os.getpid())
    with open(gen, "w") as f:
        f.write('#line 1 "%s"\n' % SRC)
        f.write(text)
    # same FP contract as the product build (dorpatch_amd/build.py): no fused multiply-add, no fast-math
    cmd = [cxx, "-x", "c++", "-std=c++17", "-O1", "-g0", "-ffp-contract=off", "-fPIC", "-shared", "-w",
           "-I", HERE, "-I", os.path.join(ROOT, "include"), "-I", os.path.dirname(SRC), gen, "-o", tmp]
    if SANITIZE:     # every LDS array, local array and (through the malloc interceptor) every tensor gets red zones
        cmd[cmd.index("-g0")] = "-g"
        cmd[5:5] = ["-fsanitize=address", "-shared-libasan", "-fno-omit-frame-pointer"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    os.remove(gen)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipemu build failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, OUT)
    return OUT


def asan_runtime():
    """Path of clang's shared AddressSanitizer runtime (to LD_PRELOAD into the python process), or None."""
    cxx = host_compiler()
    if cxx is None:
        return None
    res = subprocess.run([cxx, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    path = res.stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


if __name__ == "__main__":
    print(build(force=True))
