"""dorpatch_amd.conv1x1: both library routes of a frozen 1x1 convolution compute the same thing as
F.conv2d and its autograd; the default routing is the committed gfx950 table (deterministic), "auto"
(opt-in) caches one measured choice per (direction, shape)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dorpatch_amd import conv1x1


@pytest.fixture(autouse=True)
def _restore():
    mode = conv1x1.MODE
    conv1x1.reset()
    yield
    conv1x1.MODE = mode
    conv1x1.reset()


@pytest.mark.parametrize("mode", ["table", "gemm", "miopen", "auto"])
@pytest.mark.parametrize("N,C,O,H", [(3, 64, 256, 14), (2, 512, 128, 7), (1, 8, 8, 5)])
def test_matches_conv2d_and_its_input_gradient(mode, N, C, O, H):
    conv1x1.MODE = mode
    g = torch.Generator().manual_seed(N * 1000 + C)
    x = torch.randn(N, C, H, H, generator=g)
    w = torch.randn(O, C, 1, 1, generator=g) / np.sqrt(C)
    dy = torch.randn(N, O, H, H, generator=g)
    xr = x.clone().requires_grad_(True)
    want = F.conv2d(xr, w)
    (gx_want,) = torch.autograd.grad(want, xr, dy)
    xa = x.clone().requires_grad_(True)
    got = conv1x1.Conv1x1Function.apply(xa, w)
    (gx,) = torch.autograd.grad(got, xa, dy)
    assert got.shape == want.shape and got.is_contiguous() and gx.is_contiguous()
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gx, gx_want, rtol=1e-5, atol=1e-5)


def test_auto_calibrates_once_per_shape_and_direction():
    conv1x1.MODE = "auto"
    w = torch.randn(16, 8, 1, 1)
    for _ in range(3):
        x = torch.randn(2, 8, 6, 6, requires_grad=True)
        conv1x1.Conv1x1Function.apply(x, w).sum().backward()
    x = torch.randn(4, 8, 6, 6)                       # another batch size: its own entry, forward only
    with torch.no_grad():
        conv1x1.Conv1x1Function.apply(x, w)
    rep = conv1x1.report()
    assert rep["fwd"]["gemm"] + rep["fwd"]["miopen"] == 2 and rep["bwd"]["gemm"] + rep["bwd"]["miopen"] == 1
    assert set(k[0] for k in conv1x1._choice) == {"fwd", "bwd"}


def test_backbone_uses_it_only_for_frozen_folded_1x1_stride1_gpu_tensors():
    from dorpatch_amd.resnetv2 import StdConv2d
    conv = StdConv2d(8, 8, 1)
    x = torch.randn(1, 8, 4, 4)

    class Cuda(object):      # stand-in with the attributes `applicable` reads
        is_cuda, dtype = True, torch.float32
        dim = staticmethod(lambda: 4)
        is_contiguous = staticmethod(lambda: True)
    assert not conv1x1.applicable(conv, x)                              # CPU tensor
    assert not conv1x1.applicable(conv, Cuda)                           # weight still trainable
    conv.weight.requires_grad_(False)
    assert conv1x1.applicable(conv, Cuda)
    assert not conv1x1.applicable(StdConv2d(8, 8, 1, stride=2).requires_grad_(False), Cuda)
    assert not conv1x1.applicable(StdConv2d(8, 8, 3, padding=1).requires_grad_(False), Cuda)
    conv1x1.MODE = "miopen"
    assert not conv1x1.applicable(conv, Cuda)


def test_default_is_the_committed_table_and_it_is_deterministic():
    """no timing race decides which kernel runs.  The table lists every stride-1 1x1
    shape of ResNetV2-50 at 224x224 (incl. the subsampled downsample convolutions of DualConv1x1Function), both
    directions; anything else goes to MIOpen."""
    import importlib
    import os
    assert os.environ.get("DORPATCH_CONV1X1") is None and importlib.reload(conv1x1).MODE == "table"
    from scripts.conv1x1_table import downsample_shapes, shapes
    want = {(d, C, O, HW) for (C, O, HW) in list(shapes(224)) + list(downsample_shapes(224)) for d in ("fwd", "bwd")}
    assert set(conv1x1.TABLE) == want and set(conv1x1.TABLE.values()) <= {"gemm", "miopen"}
    assert set(conv1x1.TABLE_TUNED) == want
    w = torch.randn(256, 64, 1, 1)
    x = torch.randn(2, 64, 56, 56)
    picks = [conv1x1._pick("fwd", x, w, None) for _ in range(3)]
    assert picks == [conv1x1.TABLE[("fwd", 64, 256, 3136)]] * 3 and not conv1x1._timings      # looked up, never timed
    assert conv1x1._pick("bwd", torch.randn(2, 24, 5, 5), torch.randn(24, 12, 1, 1), None) == "miopen"


def test_share_choices_freezes_auto_mode():
    """Under a process group every rank adopts rank 0's choices; shapes seen later fall back to the table."""
    conv1x1.MODE = "auto"
    conv1x1.share_choices(None)                        # no group: nothing changes
    assert not conv1x1._frozen
    w, x = torch.randn(16, 8, 1, 1), torch.randn(2, 8, 6, 6)
    conv1x1._pick("fwd", x, w, None)
    assert len(conv1x1._choice) == 1

    class OneRank(object):
        pass
    import dorpatch_amd.dist as dp_dist
    orig = dp_dist.broadcast_object
    dp_dist.broadcast_object = lambda obj, pg: {("fwd", 2, 8, 16, 36): "gemm"}       # what rank 0 sent
    try:
        conv1x1.share_choices(OneRank())
    finally:
        dp_dist.broadcast_object = orig
    assert conv1x1._frozen and conv1x1._pick("fwd", x, w, None) == "gemm"
    n = len(conv1x1._timings)
    assert conv1x1._pick("bwd", torch.randn(2, 16, 6, 6), w, x) == "miopen" and len(conv1x1._timings) == n


def test_tuned_gemm_solution_file_and_its_route_table():
    """tunableop_gfx950.csv (PyTorch TunableOp solutions for the GEMM route, tuning done offline) + the route table
    column that goes with it: same keys as the plain table, only GEMM entries for ResNetV2-50's shapes at 512
    samples, validators present (TunableOp ignores the file on any version / architecture mismatch).  On a box
    without a GPU nothing is loaded and the plain table applies."""
    assert set(conv1x1.TABLE_TUNED) == set(conv1x1.TABLE)
    assert sum(conv1x1.TABLE_TUNED[k] != conv1x1.TABLE[k] for k in conv1x1.TABLE) >= 1
    rows = open(conv1x1.TUNABLEOP_FILE).read().strip().splitlines()
    validators = [r for r in rows if r.startswith("Validator,")]
    assert {r.split(",")[1] for r in validators} >= {"PT_VERSION", "GCN_ARCH_NAME", "ROCBLAS_VERSION", "HIPBLASLT_VERSION"}
    assert any("gfx950" in r for r in validators)
    gemms = [r for r in rows if not r.startswith("Validator,")]
    assert gemms and all(r.startswith("GemmStridedBatchedTunableOp_float_") for r in gemms)
    batches = {int(r.split("_B_")[1].split("_")[0]) for r in gemms}
    assert batches == set(conv1x1.TUNED) and 512 in batches          # one route column per tuned GEMM batch
    assert set(conv1x1.PLAIN) == set(conv1x1.TUNED)
    if not torch.cuda.is_available():
        assert conv1x1.activate(None, device_is_cuda=False) is False       # nothing is loaded without a GPU
        assert conv1x1.tuned_gemms_active(False) is False and "default" in conv1x1.report_tuned()


def test_tuned_scope_is_a_query_outside_activate(monkeypatch):
    """the tuned solutions are in effect only between activate() and deactivate(); the verdict is taken once
    per process, rank 0's self-test is what every rank adopts, and a refusal on any rank is every rank's."""
    import dorpatch_amd.dist as dp_dist

    class FakeTun(object):
        def __init__(self):
            self.enabled, self.tuning, self.calls = False, True, []
        def is_enabled(self): return self.enabled
        def tuning_is_enabled(self): return self.tuning
        def enable(self, v): self.enabled = bool(v); self.calls.append(("enable", bool(v)))
        def tuning_enable(self, v): self.tuning = bool(v)
        def record_untuned_enable(self, v): pass
        def read_file(self, path): self.calls.append(("read", path)); return True
    fake = FakeTun()
    monkeypatch.setattr(torch.cuda, "tunable", fake, raising=False)
    import sys
    monkeypatch.setitem(sys.modules, "torch.cuda.tunable", fake)
    monkeypatch.setattr(conv1x1, "_tuned_verdict", None)
    monkeypatch.setattr(conv1x1, "_tuned_scope", 0)
    monkeypatch.setattr(conv1x1, "_tuned_prev", None)
    ran = []
    monkeypatch.setattr(conv1x1, "_selftest_tuned", lambda: ran.append(1) or True)
    monkeypatch.setattr(dp_dist, "all_true", lambda flag, pg: bool(flag))
    assert not conv1x1.tuned_gemms_active(True)
    assert conv1x1.activate(None, True) is True and ran == [1]
    assert conv1x1.tuned_gemms_active(True) and fake.enabled and not fake.tuning
    assert conv1x1.activate(None, True) is True and ran == [1]            # nested scope, no second self-test
    conv1x1.deactivate()
    assert conv1x1.tuned_gemms_active(True) and fake.enabled
    conv1x1.deactivate()
    assert not conv1x1.tuned_gemms_active(True) and not fake.enabled and fake.tuning     # the caller's state is back
    assert "tuned" in conv1x1.report_tuned()
    # one rank's refusal is every rank's: with a process group the first activate() runs the agreement, whatever this
    # rank has cached (here: its own True from above)
    pg = object()
    monkeypatch.setattr(dp_dist, "world_rank", lambda g: (2, 0))
    monkeypatch.setattr(dp_dist, "broadcast_object", lambda obj, g: obj)
    monkeypatch.setattr(dp_dist, "all_true", lambda flag, g: False)
    assert conv1x1._tuned_verdict is True
    assert conv1x1.activate(pg, True) is False and not conv1x1.tuned_gemms_active(True) and not fake.enabled
    assert conv1x1._tuned_verdict is False


@pytest.mark.gpu
def test_tuned_gemm_solutions_load_on_the_gpu_box_and_compute_the_same_convolution(request):
    conv1x1.MODE = "table"
    if not (conv1x1.TUNABLEOP and conv1x1.activate(None, True)):
        # a different PyTorch / rocBLAS / hipBLASLt build or GPU stepping: TunableOp ignores the file and the library
        # defaults + the plain route column apply — slower (433 vs 409 ms/step), not wrong
        pytest.skip("tunableop_gfx950.csv was rejected by TunableOp's validators on this box: %r" % (conv1x1.selftest_report(),))
    rep = conv1x1.selftest_report()
    if rep and not rep.get("inherited"):          # the numeric self-test ran in this process tree: every tuned GEMM vs default
        assert rep["ok"] and rep["gemms"] >= 20 and rep["max_rel_err"] <= conv1x1.SELFTEST_RTOL, rep
    request.addfinalizer(conv1x1.deactivate)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(512, 64, 56, 56, generator=g).cuda()
    w = (torch.randn(64, 64, 1, 1, generator=g) / 8).cuda()
    got = conv1x1._IMPL[("fwd", "gemm")](x, w, None)
    want = F.conv2d(x, w)
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4)
    got_b = conv1x1._IMPL[("bwd", "gemm")](want, w, x)
    want_b = torch.ops.aten.convolution_backward(want, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1,
                                                 (True, False, False))[0]
    torch.testing.assert_close(got_b, want_b, rtol=1e-4, atol=1e-3)
