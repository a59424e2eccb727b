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
