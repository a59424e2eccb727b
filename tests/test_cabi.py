"""The C-ABI shared library loads and exports exactly what include/dorpatch_hip.h declares
(no compute calls: this runs on the GPU-less build box)."""
import os
import re

import pytest

from dorpatch_amd import _lib, build

HEADER = os.path.join(build.REPO_ROOT, "include", "dorpatch_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dp_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    build.build_extension()
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    assert sorted(_lib.PROTOTYPES) == names, "ctypes prototype table out of sync with the header"


def test_host_side_queries():
    lib = _lib.load()
    assert lib.dp_abi_version() == _lib.DP_ABI_VERSION
    assert lib.dp_sumsq_nchunk(224 * 224) == 13 and lib.dp_sumsq_nchunk(384 * 384) == 36
    assert lib.dp_struct_ntile(224, 224) == 7 * 28
    # S-slab partition shared by dp_apply_bwd / dp_stem_dgrad_reduce / dp_apply_affine_bwd: ~4096 workgroups, >= 2 samples per slab
    assert lib.dp_apply_bwd_nslab(64, 32, 224 * 224) == 2 and lib.dp_apply_bwd_nslab(512, 32, 224 * 224) == 1
    n = lib.dp_apply_bwd_nslab(1, 128, 224 * 224)
    assert 1 < n <= 64 and lib.dp_apply_bwd_nslab(1, 2, 224 * 224) == 1
    assert lib.dp_error_string(1) is not None


def test_ops_refuse_cpu_tensors():
    """No CPU fallback: the product path fails loudly without a GPU tensor."""
    import torch
    from dorpatch_amd import ops, utils
    from dorpatch_amd.attack import DorPatch
    x = torch.rand(1, 3, 56, 56)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.local_variance(x)
    with pytest.raises(RuntimeError, match="GPU"):
        utils.clip(torch.rand(1, 1, 56, 56), x, x, 4.0)
    with pytest.raises(RuntimeError, match="GPU"):
        DorPatch(verbose=False).generate(torch.nn.Identity(), x, 0.12, 10, "a/b/c", 0)


def test_product_never_imports_oracle():
    pkg = os.path.join(build.REPO_ROOT, "dorpatch_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
