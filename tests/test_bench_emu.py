"""bench.py's own control flow and its one-JSON-line contract, exercised without a GPU: main() runs on
CPU tensors with every dp_* kernel going through the host emulation (tests/hipemu), on a tiny geometry
(1 image x 2 masks @32x32, the real ResNetV2-50x1-BiT).  Guards the driver's round-end bench against
host-side breakage (argument handling, the roofline / cpu_baseline / config objects); says nothing about
speed — the numbers printed here are meaningless and are not checked."""
import json
import sys

import pytest

from tests_hipemu import patch as emu_patch  # noqa: E402  (registered in conftest.py)

if emu_patch.build_emu.host_compiler() is None:
    pytest.skip("no host clang++ for the HIP emulation build", allow_module_level=True)

REQUIRED = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int,
            "ms_per_step": float, "higher_is_better": bool, "scaling": str, "dtype": str, "data": str,
            "config": dict, "roofline": dict}


def _run_bench(monkeypatch, capsys, argv):
    import bench
    monkeypatch.setattr(bench, "DEVICE_OVERRIDE", "cpu")
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    with emu_patch.emulated_ops():
        bench.main()
    lines = [l for l in capsys.readouterr().out.splitlines() if l.strip()]
    assert len(lines) == 1, lines                         # exactly ONE JSON line on stdout
    return json.loads(lines[0])


def test_bench_line_contract(monkeypatch, capsys):
    out = _run_bench(monkeypatch, capsys, ["--batch", "1", "--samples", "2", "--size", "32", "--steps", "1",
                                           "--warmup", "0", "--micro-batch", "2", "--no-sweep", "--no-cpu-baseline", "--deterministic", "off"])
    for k, t in REQUIRED.items():
        assert isinstance(out[k], t), (k, out[k])
    assert out["metric"] == "EOT-samples/sec" and out["unit"] == "EOT-samples/s" and out["higher_is_better"] is True
    assert (out["n_gpus"], out["steps"], out["warmup"]) == (1, 1, 0) and out["scaling"] == "weak"
    assert out["vs_baseline"] is None and out["dtype"] == "f32" and out["data"] == "synthetic"
    # value == B*S / step time (both are rounded to 2-3 decimals in the line)
    assert abs(out["value"] - 1 * 2 / (out["ms_per_step"] / 1e3)) <= 0.006 + 1e-3 * out["value"]
    cfg, roof = out["config"], out["roofline"]
    assert cfg["workload"].startswith("custom:") and "model" not in cfg
    assert cfg["conv1x1"]["mode"] == "table" and cfg["images"] == 1 and cfg["masks_per_image_per_gpu"] == 2
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert roof["algorithmic_bytes_per_launch"] == 1 * 2 * 3 * 32 * 32 * 4
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and roof["traffic"] is None
    assert "cpu_baseline" not in out and "collect_failure_sweep_ms" not in out


def test_presets_name_the_baseline_configs(monkeypatch):
    import bench
    for argv, want in ([], "BASELINE configs[1]"), (["--config", "2"], "BASELINE configs[2]"), (["--samples", "7"], "custom"):
        monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
        assert bench.parse().config_label == want
    monkeypatch.setattr(sys, "argv", ["bench.py", "--config", "2"])
    a = bench.parse()
    assert (a.batch, a.samples, a.size, a.patch_budget) == (1, 64, 384, 0.015625)


def test_live_pmc_traffic_degrades_to_null_without_a_gpu_toolchain(monkeypatch):
    """roofline.traffic is measured in the run (a rocprofv3 --pmc child process) or null with the reason — never a
    number replayed from an earlier session."""
    import bench
    monkeypatch.setattr(bench, "ROOT", "/nonexistent")
    traffic, why = bench.pmc_traffic_live(1, 2, 32)
    assert traffic is None and "missing" in why


def test_cpu_baseline_leg_is_bounded_and_labelled():
    import bench
    out = bench.cpu_baseline(64, n_masks=2, warm=1, timed=2, budget_s=20.0, threads=2)
    assert out["kind"] == "port" and out["unit"] == "EOT-samples/s" and out["cores"] == 2 and out["value"] > 0
    assert "oracle/restatement.eot_step" in out["sample"] and "64x64" in out["sample"]
    # both the reference's real behaviour (weights trainable) and the frozen variant, median of timed steps
    assert out["value_frozen"] > 0 and out["detail"]["as_is"]["timed_steps"] == 2 and out["detail"]["frozen"]["timed_steps"] == 2


# ---------------------------------------------------------------- the plain command, N > 1
TINY = ["--batch", "1", "--size", "32", "--steps", "1", "--warmup", "0", "--no-sweep", "--no-cpu-baseline",
        "--deterministic", "off", "--backend", "gloo"]


def _plain(argv, hook=True, timeout=600):
    """`python bench.py <argv>` as a child process with NO launcher environment — what the driver runs."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    if hook:
        env["DORPATCH_BENCH_RANK_HOOK"] = hook if isinstance(hook, str) else os.path.join(root, "tests", "bench_emu_hook.py")
    else:
        env.pop("DORPATCH_BENCH_RANK_HOOK", None)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, env=env, capture_output=True, text=True,
                          timeout=timeout, cwd=root)


@pytest.mark.parametrize("world", [2] + ([8] if __import__("os").environ.get("DORPATCH_EMU_FULL", "0") == "1" else []))
def test_plain_command_spawns_its_own_ranks(world):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment starts N ranks itself, and the launcher's stdout
    carries exactly rank 0's JSON line for the whole job (weak scaling: 2 masks per image per rank).  N = 2 here, N = 4 in
    the strong-scaling test below; N = 8 with DORPATCH_EMU_FULL=1 (8 emulated ResNetV2-50 ranks: ~40 s; bench.py's rank
    code on 8 gloo ranks is covered by tests/test_dist_emu.py either way)."""
    res = _plain(["--gpus", str(world), "--samples", "2", "--micro-batch", "2"] + TINY)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["scaling"] == "weak"
    assert out["config"]["masks_per_image_per_gpu"] == 2 and out["config"]["masks_per_image_total"] == 2 * world
    assert "process group backend gloo" in out["config"]["parallelism"]


def test_plain_command_strong_scaling_mode():
    """--scaling strong: --samples is the total per image; each of the N ranks takes 1/N.  (BASELINE configs[3] is
    `--config 3 --scaling strong --gpus 8`: 512 in total, 64 per GPU — the preset arithmetic is checked below without
    running it.)"""
    res = _plain(["--gpus", "4", "--samples", "8", "--scaling", "strong", "--micro-batch", "2"] + TINY)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 4 and out["scaling"] == "strong"
    assert out["config"]["masks_per_image_per_gpu"] == 2 and out["config"]["masks_per_image_total"] == 8
    assert abs(out["value"] - 1 * 8 / (out["ms_per_step"] / 1e3)) <= 0.006 + 1e-3 * out["value"]
    import bench
    a = bench.parse(["--config", "3", "--scaling", "strong", "--gpus", "8"])
    assert (a.batch, a.samples, a.size) == (1, 512, 224) and a.config_label.startswith("BASELINE configs[3]")
    a = bench.parse(["--config", "3", "--gpus", "8"])                 # weak: 64 per GPU -> the same 512 at N = 8
    assert (a.batch, a.samples) == (1, 64) and a.config_label == "BASELINE configs[3]"
    with pytest.raises(SystemExit):
        bench.parse(["--samples", "10", "--scaling", "strong", "--gpus", "4"])


def test_the_8_gpu_strong_scaling_command_runs_on_8_ranks_with_identical_replicas(tmp_path):
    """the exact command a driver would run for BASELINE configs[3] on an 8-GPU node —
    `python bench.py --gpus 8 --config 3 --scaling strong --steps K --warmup W` — through bench.py's own launcher on 8 gloo
    ranks (kernels through the emulation; tests/bench_emu_hook_tiny.py swaps in 32 x 32 images and a toy classifier, nothing
    else): 512 EOT samples of one image, 64 per rank; ONE JSON line with n_gpus 8; and every rank ends with the bit-identical
    mask / pattern / all-reduced gradient (the replicas stay in lock-step through the signed update)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hook = os.path.join(root, "tests", "bench_emu_hook_tiny.py")
    env_dir = str(tmp_path)
    os.environ["DORPATCH_BENCH_DIGEST_DIR"] = env_dir
    try:
        res = _plain(["--gpus", "8", "--config", "3", "--scaling", "strong", "--steps", "2", "--warmup", "1", "--no-sweep",
                      "--no-cpu-baseline", "--deterministic", "off", "--backend", "gloo"], hook=hook, timeout=900)
    finally:
        del os.environ["DORPATCH_BENCH_DIGEST_DIR"]
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["steps"] == 2 and out["warmup"] == 1
    assert out["config"]["masks_per_image_per_gpu"] == 64 and out["config"]["masks_per_image_total"] == 512
    assert out["config"]["workload"].startswith("BASELINE configs[3] (strong scaling: 512 EOT samples of one image in total, 64 per GPU)")
    assert abs(out["value"] - 512 / (out["ms_per_step"] / 1e3)) <= 0.006 + 1e-3 * out["value"]
    digests = [open(os.path.join(env_dir, "rank%d.txt" % r)).read() for r in range(8)]
    assert len(set(digests)) == 1, digests


def test_comm_only_mode_times_the_steps_real_message():
    """`bench.py --gpus N --comm-only` = 50 all-reduces of HotLoop's own message (patch gradient + one
    loss / prediction slab + checksum per rank), one JSON line with us per call and the byte count — here 2 gloo ranks."""
    res = _plain(["--gpus", "2", "--samples", "2", "--micro-batch", "2", "--comm-only"] + TINY)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["unit"] == "us per call" and out["calls"] == 50 and out["value"] > 0
    # (1,3,32,32) gradient + per rank: 2 loss + 2 prediction floats + 1 checksum
    assert out["bytes"] == 4 * (3 * 32 * 32 + 2 * (2 * 2 + 1)) and out["higher_is_better"] is False


def test_plain_command_without_enough_gpus_fails_loudly():
    """No GPU in the build container: the launcher must say so and exit non-zero BEFORE starting any rank."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this host really has >= 2 GPUs")
    res = _plain(["--gpus", "2"] + TINY, hook=False, timeout=120)
    assert res.returncode == 2 and res.stdout.strip() == ""
    assert "needs 2 visible GPUs" in res.stderr


def test_a_failing_rank_fails_the_launcher(tmp_path):
    """A rank that dies must surface as a non-zero exit code of the plain command, with no JSON line, and must not leave
    the other ranks waiting in a collective (they are terminated: the command returns promptly)."""
    import os
    bad = tmp_path / "hook.py"
    bad.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\ntime.sleep(600)\n")
    res = _plain(["--gpus", "2", "--samples", "2"] + TINY, hook=str(bad), timeout=120)
    assert res.returncode == 7
    assert "rank 1 exited with code 7" in res.stderr and res.stdout.strip() == ""


def test_whole_attack_mode_reports_the_split():
    """bench.py --whole-attack: the driver (= main.py) on one synthetic batch, with and without retiring finished images;
    here 2 images x 3 iterations per stage @56x56 on a toy classifier through the emulation — the control flow and the
    JSON object, not the speed."""
    import torch
    import bench
    from oracle import toy_models
    args = bench.parse(["--whole-attack", "--attack-iterations", "3", "--attack-batch", "2", "--size", "56",
                        "--micro-batch", "16"])
    model = toy_models.NormModel(toy_models.make_toy(n_classes=100, gain=2.0), toy_models.Normalize())   # 100 classes: the
    with emu_patch.emulated_ops():       # random target (main.py:122) must differ from the label, as the reference asserts
        out = bench.whole_attack(args, torch.device("cpu"), model=model, n_classes=100,
                                 extra_argv=["--sampling_size", "8", "--dropout", "1"])
    assert out["metric"] == "whole-attack seconds per image" and out["higher_is_better"] is False
    for mode in ("retire", "no_retire"):
        v = out["variants"][mode]
        assert v["images"] == 2 and v["stage0_steps"] == 3 and v["stage1_steps"] == 3 and v["sweeps"] == 2
        assert v["stage0_image_steps"] == [3, 3] and v["samples_forward"] == 2 * 8 * 6
        assert v["seconds_per_image"] > 0 and v["stage0_s"] >= v["stage0_sweeps_s"] > 0 and v["patchcleanser_s"] > 0
        assert len(v["certified_asr_PC"]) == 4
    assert "straggler_saving" in out and json.dumps(out)


@pytest.mark.parametrize("stage", [0, 1])
def test_project_update_roofline_entry(stage):
    """The second roofline object of the bench line (dp_project_update past the Infinity Cache): the byte arithmetic and
    the object's shape, through the emulation on one image."""
    import torch
    import bench
    with emu_patch.emulated_ops():
        r = bench.project_update_roofline(torch.device("cpu"), 56, stage)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["traffic"] is None
    assert r["algorithmic_bytes_per_launch"] == 56 * 56 * (72 if stage == 0 else 68) and r["avg_launch_ms"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "k_project_update_v4" in r["kernel"]


def test_rank_local_extra_steps_only_in_single_rank_jobs():
    """bench.py's conv-roofline pass is ONE more HotLoop.step on rank 0 — a step holds the job's all-reduce, so it may only
    run when there is no peer to wait for (found by reading, round 5: the GPU multi-rank runs of the test suite pass
    --no-conv-roofline or run on CPU tensors, where the pass is off anyway; the driver's --gpus N command passes neither)."""
    import bench
    assert bench.conv_roofline_wanted(0, 1, "cuda", False) is True
    assert bench.conv_roofline_wanted(0, 2, "cuda", False) is False and bench.conv_roofline_wanted(0, 8, "cuda", False) is False
    assert bench.conv_roofline_wanted(1, 2, "cuda", False) is False
    assert bench.conv_roofline_wanted(0, 1, "cpu", False) is False and bench.conv_roofline_wanted(0, 1, "cuda", True) is False
    # nothing else between the timed region and the JSON line runs a step on a subset of the ranks
    import inspect
    src = inspect.getsource(bench)
    tail = src[src.index("conv_roof = None"):]
    assert tail.count("loop.step(") == 0 and src.count("= conv_roofline(loop") == 1
