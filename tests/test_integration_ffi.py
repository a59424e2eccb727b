"""INTEGRATION.md §2 shows the ctypes stub a reference maintainer would add (``dorpatch_ffi.py``).  The block is taken
from the document VERBATIM and (GPU) executed in a fresh interpreter — no dorpatch_amd import, only torch + ctypes +
libdorpatch_hip.so on LD_LIBRARY_PATH — and its two functions are compared with the CPU oracle; (CPU) its prototypes
are compared with the product's own table, so the document cannot drift from the header."""
import os
import re
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def ffi_block():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(# dorpatch_ffi\.py.*?)```", text, flags=re.S)
    assert len(blocks) == 1
    return blocks[0]


def test_documented_stub_declares_the_products_prototypes():
    import ctypes
    from dorpatch_amd import _lib
    src = ffi_block()
    compile(src, "dorpatch_ffi.py", "exec")
    names = set(re.findall(r"_lib\.(dp_\w+)\.argtypes", src))
    assert names == {"dp_apply_fwd", "dp_sumsq_nchunk", "dp_sumsq_partials", "dp_blend"}

    class FakeLib(object):          # records what the stub assigns, instead of dlopen()ing a GPU library
        def __getattr__(self, name):
            fn = type("Fn", (), {})()
            object.__setattr__(self, name, fn)
            return fn
    fake = FakeLib()
    ns = {}
    real = ctypes.CDLL
    ctypes.CDLL = lambda path: fake
    try:
        exec(compile(src, "dorpatch_ffi.py", "exec"), ns)
    finally:
        ctypes.CDLL = real
    for name in names:
        restype, argtypes = _lib.PROTOTYPES[name]
        got = getattr(fake, name)
        assert got.restype is restype, name
        assert len(got.argtypes) == len(argtypes), name
        for a, b in zip(got.argtypes, argtypes):
            same_struct = (isinstance(getattr(a, "_type_", None), type) and isinstance(getattr(b, "_type_", None), type)
                           and issubclass(a._type_, ctypes.Structure) and issubclass(b._type_, ctypes.Structure)
                           and [f[0] for f in a._type_._fields_] == [f[0] for f in b._type_._fields_])
            assert a is b or same_struct, (name, a, b)


RUNNER = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import dorpatch_ffi as F                      # the block from INTEGRATION.md, verbatim
assert "dorpatch_amd" not in sys.modules
out = {}
g = torch.Generator().manual_seed(3)
for H in (56, 224):
    B, S = 2, 5
    x = torch.rand(B, 3, H, H, generator=g)
    m, p = torch.rand(B, 1, H, H, generator=g), torch.rand(B, 3, H, H, generator=g)
    table = torch.from_numpy(np.load(sys.argv[2] + "/table_%d.npy" % H)).cuda()
    idx = torch.from_numpy(np.load(sys.argv[2] + "/idx_%d.npy" % H)).cuda()
    adv = F.clip(m.cuda(), p.cuda(), x.cuda(), 4.0)
    occ = F.occlude(x.cuda().contiguous(), table, idx)
    torch.cuda.synchronize()
    np.save(sys.argv[2] + "/x_%d.npy" % H, x.numpy()); np.save(sys.argv[2] + "/m_%d.npy" % H, m.numpy())
    np.save(sys.argv[2] + "/p_%d.npy" % H, p.numpy())
    np.save(sys.argv[2] + "/adv_%d.npy" % H, adv.cpu().numpy()); np.save(sys.argv[2] + "/occ_%d.npy" % H, occ.cpu().numpy())
print("ok")
'''


@pytest.mark.gpu
def test_documented_stub_runs_verbatim_and_matches_the_oracle(tmp_path):
    import numpy as np
    import torch
    from dorpatch_amd import masks
    from dorpatch_amd.build import LIB_DIR
    from oracle import restatement as R
    (tmp_path / "dorpatch_ffi.py").write_text(ffi_block())
    (tmp_path / "runner.py").write_text(RUNNER)
    idx = {}
    for H in (56, 224):
        np.save(tmp_path / ("table_%d.npy" % H), masks.universe_rects(H, 2).astype(np.int32))
        idx[H] = np.random.RandomState(H).choice(2520, 5, replace=False).astype(np.int32)
        np.save(tmp_path / ("idx_%d.npy" % H), idx[H])
    env = dict(os.environ, LD_LIBRARY_PATH=LIB_DIR + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    res = subprocess.run([sys.executable, str(tmp_path / "runner.py"), str(tmp_path), str(tmp_path)], env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and res.stdout.strip().endswith("ok"), res.stderr[-3000:]
    for H in (56, 224):
        load = lambda n: torch.from_numpy(np.load(tmp_path / ("%s_%d.npy" % (n, H))))
        x, m, p = load("x"), load("m"), load("p")
        want_adv = x + R.clip(m, p, x, 4.0)                               # dp_blend(add_x = ...) in the stub: see below
        got_adv = load("adv")
        # the stub calls dp_blend with add_x = 0: it returns the clipped delta, i.e. utils.clip itself (utils.py:105-110)
        np.testing.assert_allclose(got_adv.numpy(), (want_adv - x).numpy(), rtol=0, atol=1e-6)
        keep = R.mask_universe(H, 2)[torch.from_numpy(idx[H].astype(np.int64))]          # (S,1,H,W) bool, True = keep
        want_occ = (x[:, None] * keep[None] + (~keep[None]) * 0.5).reshape(-1, 3, H, H)  # attack.py:204-206
        assert torch.equal(load("occ"), want_occ)
