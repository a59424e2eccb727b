"""The product's multi-rank hot loop itself — ``HotLoop.step`` with a process group, the sharded
``collect_failure`` sweep and the image-sharded evaluation driver — on world_size-2 ``gloo``, with the
HIP kernels running through the host emulation (tests/hipemu) on CPU tensors.

tests/test_dist_gloo.py checks the sharding arithmetic with the oracle standing in for the kernels;
here the code under test is exactly what runs per GPU under ``torch.distributed.run`` (same-state
draws on every rank, S-slice per rank, ONE all-reduce carrying the patch gradient + the loss / prediction
slabs + a draw checksum, failure bitmap MAX-reduce), compared with the same problem on a single rank.
"""
import importlib.util
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _emu_patch():
    """tests/hipemu as the package `tests_hipemu` (conftest.py does this in the pytest process; spawned
    ranks do it themselves)."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    if "tests_hipemu" not in sys.modules:
        spec = importlib.util.spec_from_file_location("tests_hipemu", os.path.join(HERE, "hipemu", "__init__.py"),
                                                      submodule_search_locations=[os.path.join(HERE, "hipemu")])
        mod = importlib.util.module_from_spec(spec)
        sys.modules["tests_hipemu"] = mod
        spec.loader.exec_module(mod)
    from tests_hipemu import patch
    return patch


if _emu_patch().build_emu.host_compiler() is None:
    pytest.skip("no host clang++ for the HIP emulation build", allow_module_level=True)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class FixedDraw(object):
    def __init__(self, rows):
        self.rows = list(rows)

    def choice(self, a, n, replace=False):
        return np.asarray(self.rows.pop(0)).copy()


H, S, B, N_STEPS = 56, 8, 2, 3
N_MASK = 144          # dropout = 1: the 4 x 36 single-window universe keeps the sharded failure sweep cheap


def _problem():
    from oracle import toy_models
    g = torch.Generator().manual_seed(3)
    x, m, p = torch.rand(B, 3, H, H, generator=g), torch.rand(B, 1, H, H, generator=g), torch.rand(B, 3, H, H, generator=g)
    y = torch.tensor([2, 7])
    rows = [[np.random.RandomState(10 * b + k).choice(N_MASK, S, replace=False) for k in range(N_STEPS)] for b in range(B)]
    net = toy_models.NormModel(toy_models.make_toy(gain=3.0), toy_models.Normalize())
    return x, m, p, y, rows, net


def _run_loop(pg, rank):
    """N_STEPS steps of the product's HotLoop (step 0 includes the collect_failure sweep)."""
    from dorpatch_amd.attack import DorPatch, HotLoop
    x, m, p, y, rows, net = _problem()
    if rank != 0:                     # only rank 0's init may matter (broadcast at construction): poison the others.
        m, p = torch.zeros_like(m), torch.zeros_like(p)      # The draws are NOT exchanged: same `rngs` on every rank.
    seen = []
    hook = lambda d: seen.append(dict(idx=d["idx"].copy(), loss_adv=d["loss_adv"].copy(), g_adv=d["g_adv"].clone(),
                                      grad_mask=d["grad_mask"].clone(), lr=d["lr"].copy()))
    # verbose=True: the progress line (attack.py:318-330) must not involve a collective only rank 0 enters
    loop = HotLoop(DorPatch(micro_batch=6, process_group=pg, verbose=True), net, x, 0.12, 10, "t/cfg/sub", 0, y,
                   True, 1e-2, 1e-1, 0, 1, 10 ** 6, 7, 'topk', 1, S, 1e-3, 1e-3, 4.0, False,
                   dict(init_mask=m, init_pattern=p, rngs=[FixedDraw(rows[b]) for b in range(B)], step_hook=hook))
    assert loop.n_mask == N_MASK
    for i in range(N_STEPS):
        loop.step(i)
    out = dict(seen=seen, pattern=loop.adv_pattern.clone(), mask=loop.adv_mask.clone(),
               failed=[list(st.failed_idxs) for st in loop.img], s_local=loop.S_local)
    loop.close()
    return out


def _loop_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with _emu_patch().emulated_ops():
            out = _run_loop(dist.group.WORLD, rank)
        torch.save(out, os.path.join(out_dir, "rank%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_hot_loop_two_ranks_equals_one(world, tmp_path):
    """world = 4 as well: the slab layout of the all-reduce buffer, the mask-universe slices of the sharded failure sweep
    and the draw checksums beyond two ranks (the driver's 8-GPU scaling run is the first time RCCL sees this code)."""
    mp.spawn(_loop_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    with _emu_patch().emulated_ops():
        want = _run_loop(None, 0)
    outs = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=False) for r in range(world)]
    assert want["s_local"] == S and all(o["s_local"] == S // world for o in outs)
    for o in outs:
        assert o["failed"] == want["failed"]                       # sharded sweep + MAX-reduce == full sweep
        for k in range(N_STEPS):
            a, w = o["seen"][k], want["seen"][k]
            assert np.array_equal(a["idx"], w["idx"])                  # the same draw, everywhere
            if k == 0:
                # identical parameters going in: only the S-summation order differs
                np.testing.assert_allclose(a["loss_adv"], w["loss_adv"], rtol=1e-5, atol=1e-6)
                scale = float(w["g_adv"].abs().max())
                np.testing.assert_allclose(a["g_adv"].numpy(), w["g_adv"].numpy(), rtol=1e-4, atol=1e-5 * scale)
                assert torch.equal(torch.isnan(a["grad_mask"]), torch.isnan(w["grad_mask"]))
            else:
                np.testing.assert_allclose(a["loss_adv"], w["loss_adv"], rtol=2e-2, atol=2e-3)
        # the signed update may flip where |grad| ~ ulp (sum order): compare by fraction of differing pixels
        assert ((o["pattern"] - want["pattern"]).abs() > 1e-6).float().mean() < 5e-3
        assert ((o["mask"] - want["mask"]).abs() > 1e-6).float().mean() < 5e-3
    # ranks stay in lock-step: bit-identical reduced gradients, losses and parameters
    for other in outs[1:]:
        for k in range(N_STEPS):
            assert torch.equal(outs[0]["seen"][k]["g_adv"], other["seen"][k]["g_adv"])
            assert np.array_equal(outs[0]["seen"][k]["loss_adv"], other["seen"][k]["loss_adv"])
        assert torch.equal(outs[0]["pattern"], other["pattern"]) and torch.equal(outs[0]["mask"], other["mask"])


# ---------------------------------------------------------------- finished images leave the batch, on two ranks
def _run_retire(pg, rank, retire):
    """4 steps of a 2-image loop whose image 0 is marked finished before step 2 (attack.py:311-316), sweeps at steps 0, 2."""
    from dorpatch_amd.attack import DorPatch, HotLoop
    x, m, p, y, rows, net = _problem()
    rows = [r + [np.random.RandomState(99 + b).choice(N_MASK, S, replace=False)] for b, r in enumerate(rows)]
    seen = []
    hook = lambda d: seen.append(dict(loss_adv=d["loss_adv"].copy(), g_adv=d["g_adv"].clone(), lr=d["lr"].copy()))
    loop = HotLoop(DorPatch(micro_batch=6, process_group=pg, verbose=False), net, x, 0.12, 10, "t/cfg/sub", 0, y,
                   True, 1e-2, 1e-1, 0, 1, 10 ** 6, 7, 'topk', 1, S, 1e-3, 1e-3, 4.0, False,
                   dict(init_mask=m, init_pattern=p, rngs=[FixedDraw(rows[b]) for b in range(B)], step_hook=hook,
                        failure_refresh=2, retire=retire))
    preds = []
    for i in range(4):
        if i == 2:
            loop.img[0].active = False
        loop.step(i)
        preds.append(loop.pred_host.copy())
    out = dict(seen=seen, preds=preds, pattern=loop.adv_pattern.clone(), mask=loop.adv_mask.clone(),
               failed=[list(st.failed_idxs) for st in loop.img], n_fwd=loop.n_forward, swept=loop.swept_images)
    loop.close()
    return out


def _retire_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)          # torch-CPU convolutions are batch-size invariant on one thread only
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with _emu_patch().emulated_ops():
            out = {r: _run_retire(dist.group.WORLD, rank, r) for r in (True, False)}
        torch.save(out, os.path.join(out_dir, "rank%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_retiring_images_on_two_ranks(tmp_path):
    """The dense batch of the running images under EOT-sample sharding: each rank gathers the same rows (the `active`
    flags are host state derived from all-reduced losses, identical everywhere), fills its own loss / prediction slab for
    them, and the one all-reduce of the step carries zero gradient rows for the finished image.  Against `retire=False`
    on the same two ranks the running image's gradients, losses, failure list and parameters are bit-identical, the
    finished image's prediction row keeps its last value, and the ranks stay in lock-step."""
    mp.spawn(_retire_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=False) for r in range(2)]
    for o in outs:
        on, off = o[True], o[False]
        assert on["n_fwd"] == (2 + 2 + 1 + 1) * (S // 2) and off["n_fwd"] == 8 * (S // 2)     # this rank's half of the samples
        assert (on["swept"], off["swept"]) == (2 + 1, 2 + 2)
        for k in range(4):
            a, w = on["seen"][k], off["seen"][k]
            assert np.array_equal(a["lr"], w["lr"]) and (k < 2 or a["lr"][0] == 0)
            assert np.array_equal(a["loss_adv"][1], w["loss_adv"][1]) and torch.equal(a["g_adv"][1], w["g_adv"][1])
            if k >= 2:
                assert not a["g_adv"][0].any()
            assert np.array_equal(on["preds"][k], off["preds"][k])
        assert on["failed"][1] == off["failed"][1]
        assert torch.equal(on["pattern"], off["pattern"]) and torch.equal(on["mask"], off["mask"])
    for key in (True, False):                                   # lock-step across the ranks
        assert torch.equal(outs[0][key]["pattern"], outs[1][key]["pattern"])
        assert all(torch.equal(a["g_adv"], b["g_adv"]) for a, b in zip(outs[0][key]["seen"], outs[1][key]["seen"]))


def _placement_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with _emu_patch().emulated_ops():
            out = _run_placement(dist.group.WORLD, rank)
        torch.save(out, os.path.join(out_dir, "place%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def _run_placement(pg, rank):
    """Two steps with the affine-placement extension: mask draws AND placements come from the per-image generators,
    whose state every rank shares, so nothing but the one all-reduce is exchanged."""
    from dorpatch_amd.attack import DorPatch, HotLoop
    from dorpatch_amd.placement import RandomAffine
    x, m, p, y, _, net = _problem()
    np.random.seed(77 + 1000 * rank)            # ranks arrive with DIFFERENT global state: rank 0's seeds are broadcast
    seen = []
    hook = lambda d: seen.append(dict(idx=d["idx"].copy(), theta=d["theta"].copy(), g_adv=d["g_adv"].clone()))
    loop = HotLoop(DorPatch(micro_batch=6, process_group=pg, verbose=False), net, x, 0.12, 10, "t/cfg/sub", 0, y,
                   True, 1e-2, 1e-1, 0, 1, 10 ** 6, 7, 'topk', 1, S, 1e-3, 1e-3, 4.0, False,
                   dict(init_mask=m, init_pattern=p, step_hook=hook, failure_refresh=10 ** 9,
                        placement=RandomAffine(12.0, (0.9, 1.1), 3.0)))
    for i in (1, 2):
        loop.step(i)
    out = dict(seen=seen, pattern=loop.adv_pattern.clone())
    loop.close()
    return out


def test_placement_extension_two_ranks_stay_in_lock_step(tmp_path):
    world = 2
    mp.spawn(_placement_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "place%d.pt" % r), weights_only=False) for r in range(world)]
    for k in range(2):
        assert np.array_equal(outs[0]["seen"][k]["idx"], outs[1]["seen"][k]["idx"])
        assert np.array_equal(outs[0]["seen"][k]["theta"], outs[1]["seen"][k]["theta"])
        assert torch.equal(outs[0]["seen"][k]["g_adv"], outs[1]["seen"][k]["g_adv"])
    assert torch.equal(outs[0]["pattern"], outs[1]["pattern"])
    assert not np.array_equal(outs[0]["seen"][0]["theta"], outs[0]["seen"][1]["theta"])     # a fresh draw every step


# ---------------------------------------------------------------- opt-in selected-sample backward, sharded
SK_H, SK_S = 112, 4            # the smallest input the fused stem pooling takes; 2 samples per rank


def _run_skip(pg, rank, skip, confidence):
    """One step through a shallow ResNetV2 of the real block types (the taped path needs dorpatch_amd's own network):
    every rank decides locally which of ITS samples carry gradient; the one all-reduce is unchanged."""
    from dorpatch_amd.attack import DorPatch, HotLoop
    from dorpatch_amd.resnetv2 import WELL_CONDITIONED_GN_BIAS, ResNetV2, seeded_init_
    from dorpatch_amd.utils import NormModel, get_normalize
    net = seeded_init_(ResNetV2((1, 1, 1, 1), (256, 512, 1024, 2048), 10), seed=1234,
                       gn_bias=WELL_CONDITIONED_GN_BIAS).fold_weight_standardization().freeze()
    model = NormModel(net, get_normalize("imagenet", "resnetv2")).eval()
    g = torch.Generator().manual_seed(5)
    x, m, p = (torch.rand(1, 3, SK_H, SK_H, generator=g), torch.rand(1, 1, SK_H, SK_H, generator=g),
               torch.rand(1, 3, SK_H, SK_H, generator=g))
    with torch.no_grad():
        y = model(x).topk(2)[1][:, 1].clone()
    idx = np.random.RandomState(5).choice(630, SK_S, replace=False)
    got = {}
    hook = lambda d: got.update(loss_adv=d["loss_adv"].copy(), g_adv=d["g_adv"].clone())
    loop = HotLoop(DorPatch(micro_batch=2, process_group=pg, verbose=False, skip_satisfied=skip, deterministic=False),
                   model, x, 0.12, 10, "t/cfg/sub", 0, y, True, 1e-2, confidence, 0, 1, 10, 7, 'topk', 2, SK_S, 1e-3,
                   1e-3, 4.0, False, dict(init_mask=m, init_pattern=p, rngs=[FixedDraw([idx])], failure_refresh=10 ** 9,
                                          step_hook=hook, backward_ladder=[1, 2], skip_min_fraction=0.0))
    loop.step(1)
    got.update(counts=(loop.n_forward, loop.n_active, loop.n_backward), taped=loop._taped)
    loop.close()
    return got


def _skip_worker(rank, world, port, out_dir, confidence):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with _emu_patch().emulated_ops():
            out = _run_skip(dist.group.WORLD, rank, True, confidence)
        torch.save(out, os.path.join(out_dir, "skip%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_skip_satisfied_two_ranks_equals_one_rank_all_samples(tmp_path):
    world = 2
    with _emu_patch().emulated_ops():
        base = _run_skip(None, 0, False, 0.1)
        d = np.sort(base["loss_adv"].reshape(-1) - 0.1)
        conf = float(-0.5 * (d[0] + d[1]))                   # exactly one of the 4 samples meets its margin
        want = _run_skip(None, 0, False, conf)
    assert int((want["loss_adv"] > 0).sum()) == SK_S - 1 and not want["taped"]
    mp.spawn(_skip_worker, args=(world, _free_port(), str(tmp_path), conf), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "skip%d.pt" % r), weights_only=False) for r in range(world)]
    assert all(o["taped"] for o in outs)
    assert sorted(o["counts"] for o in outs) == [(2, 1, 1), (2, 2, 2)]      # the satisfied sample lives on one rank
    scale = float(want["g_adv"].abs().max())
    for o in outs:
        np.testing.assert_allclose(o["loss_adv"], want["loss_adv"], rtol=1e-5, atol=1e-6)
        assert float((o["g_adv"] - want["g_adv"]).abs().max()) <= 1e-5 * scale
    assert torch.equal(outs[0]["g_adv"], outs[1]["g_adv"])


# ---------------------------------------------------------------- evaluation driver, both shard modes
def _driver_worker(rank, world, port, out_dir, shard):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.chdir(out_dir)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dorpatch_amd import driver
        from oracle import toy_models
        driver.DEFENSE_RATIOS = (0.03,)           # one defence instead of four: the sweeps dominate under emulation
        model = toy_models.NormModel(toy_models.make_toy(gain=2.0), toy_models.Normalize())
        batches = []
        for i in range(2):
            x = torch.rand(1, 3, H, H, generator=torch.Generator().manual_seed(100 + i))
            with torch.no_grad():
                batches.append((x, model(x).argmax(-1)))
        args = driver.build_parser().parse_args(["--num_images", "2", "--max_iterations", "3", "--sampling_size", "4",
                                                 "--img_size", str(H), "--quiet", "--shard", shard])
        with _emu_patch().emulated_ops():
            out = driver.run(args, model=model, dataloader=batches, device="cpu", process_group=dist.group.WORLD,
                             n_classes=10)
        torch.save(out, os.path.join(out_dir, "metrics%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shard", ["images", "samples"])
def test_driver_two_ranks(shard, tmp_path):
    world = 2
    mp.spawn(_driver_worker, args=(world, _free_port(), str(tmp_path), shard), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "metrics%d.pt" % r), weights_only=False) for r in range(world)]
    keys = ("acc_clean", "acc_robust", "acc_PC", "certified_acc_PC", "certified_asr_PC", "n_images")
    assert all(outs[0][k] == outs[1][k] for k in keys)                  # every rank reports the whole job
    assert outs[0]["n_images"] == 2
    rd = os.path.join(str(tmp_path), outs[0]["result_dir"])
    for i in range(2):
        for name in ("adv_mask_%d.pt", "adv_pattern_%d.pt", "adv_PC_%d.pt"):
            assert os.path.exists(os.path.join(rd, name % i)), (shard, name % i)


# ---------------------------------------------------------------- bench.py's multi-rank control flow
def _bench_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.argv = ["bench.py", "--gpus", str(world), "--backend", "gloo", "--batch", "1", "--samples", "2", "--size", "32",
                "--steps", "1", "--warmup", "0", "--micro-batch", "2", "--no-sweep", "--no-cpu-baseline", "--deterministic", "off"]
    import contextlib
    import bench
    bench.DEVICE_OVERRIDE = "cpu"
    with open(os.path.join(out_dir, "stdout%d.txt" % rank), "w") as f, contextlib.redirect_stdout(f):
        with _emu_patch().emulated_ops():
            bench.main()


@pytest.mark.parametrize("world", [2, 8])
def test_bench_two_ranks_prints_one_whole_job_line(world, tmp_path):
    """`python -m torch.distributed.run ... bench.py --gpus N` as the driver launches it for N = 2 and 8 (here: gloo,
    CPU tensors, emulated kernels): rank 0 alone prints the line, for the WHOLE job (weak scaling: 2 masks per
    image per rank -> 2 N per image in total)."""
    import json
    mp.spawn(_bench_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    out0 = open(os.path.join(str(tmp_path), "stdout0.txt")).read().strip().splitlines()
    others = [open(os.path.join(str(tmp_path), "stdout%d.txt" % r)).read().strip() for r in range(1, world)]
    assert len(out0) == 1 and all(o == "" for o in others)
    line = json.loads(out0[0])
    assert line["n_gpus"] == world and line["scaling"] == "weak" and "cpu_baseline" not in line
    assert line["config"]["masks_per_image_per_gpu"] == 2 and line["config"]["masks_per_image_total"] == 2 * world
    assert abs(line["value"] - 1 * 2 * world / (line["ms_per_step"] / 1e3)) <= 0.006 + 1e-3 * line["value"]
