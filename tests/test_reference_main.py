"""The boundary's caller: the UNMODIFIED reference driver /root/reference/main.py executed
against this repository's drop-in modules (repo-root ``utils.py`` / ``attack.py`` / ``defenses/PatchCleanser.py``).
Build container only (needs /root/reference and the host emulation of the kernels); tests/run_reference_main.py is the
harness and says exactly what it supplies from outside main.py.  INTEGRATION.md's "main.py runs unchanged" rests on this."""
import json
import os
import pickle
import subprocess
import sys

import pytest

from tests_hipemu import patch as emu_patch  # noqa: E402  (registered in conftest.py)

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

if not os.path.isfile("/root/reference/main.py"):
    pytest.skip("needs /root/reference (build container only)", allow_module_level=True)
if emu_patch.build_emu.host_compiler() is None:
    pytest.skip("no host clang++ for the HIP emulation build", allow_module_level=True)


def _start(work, *flags):
    env = dict(os.environ, OMP_NUM_THREADS="2", DORPATCH_REFMAIN_BATCHES="1")
    return subprocess.Popen([sys.executable, os.path.join(HERE, "run_reference_main.py"), str(work)] + list(flags),
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)


def _finish(proc):
    out, err = proc.communicate(timeout=1500)
    assert proc.returncode == 0, err[-4000:]
    return json.loads(out.strip().splitlines()[-1])


def _run(work, *flags):
    return _finish(_start(work, *flags))


@pytest.fixture(scope="module")
def first_runs(tmp_path_factory):
    """The targeted and the untargeted run of main.py, side by side (each ~75 s under the emulation: 2 x 2520 + 2664 masked
    224 x 224 images through the fibre-emulated occlusion kernel)."""
    dirs = {k: tmp_path_factory.mktemp(k) for k in ("targeted", "untargeted")}
    procs = {"targeted": _start(dirs["targeted"], "--targeted"), "untargeted": _start(dirs["untargeted"])}
    return {k: (dirs[k], _finish(p)) for k, p in procs.items()}


def test_unmodified_reference_main_runs_against_the_drop_in_modules(first_runs):
    tmp_path, out = first_runs["targeted"]
    # main.py's names are bound to the product, not to anything of the reference
    assert out["bound"] == dict(DorPatch="dorpatch_amd.attack", PatchCleanser="dorpatch_amd.patchcleanser",
                                MaskWindow="dorpatch_amd.patchcleanser", clip="dorpatch_amd.utils", NormModel="dorpatch_amd.utils")
    # main.py:128-134 — the keyword arguments it passes arrive as the reference passes them
    (call,) = out["calls"]
    top = "results/dataset=imagenet_base_arch=resnetv2_targeted=True_attack=DorPatch_dropout=2_density=0.001_structured=0.001"
    sub = top + "/num_patch=-1_patch_budget=0.12"
    assert call == dict(targeted=True, y="tensor(1,)", lr=0.01, num_patch=-1, dropout=2, density=0.001, structured=0.001,
                        save_dir=sub, batch_id=0, eps=4.0, n_positional=4)
    # files: stage-0 cache one level up (attack.py:351-356), final tensors + PatchCleanser records (main.py:135-153)
    assert out["files"] == sorted([top + "/adv_mask_0.pt", top + "/adv_pattern_0.pt", sub + "/adv_mask_0.pt",
                                   sub + "/adv_pattern_0.pt", sub + "/adv_PC_0.pt"])
    import torch
    mask = torch.load(os.path.join(str(tmp_path), sub, "adv_mask_0.pt"))
    pattern = torch.load(os.path.join(str(tmp_path), sub, "adv_pattern_0.pt"))
    assert tuple(mask.shape) == (1, 1, 224, 224) and tuple(pattern.shape) == (1, 3, 224, 224)
    assert set(mask.unique().tolist()) <= {0.0, 1.0} and 0 < int(mask.sum()) <= 0.12 * 224 * 224
    sys.path.insert(0, ROOT)
    try:
        with open(os.path.join(str(tmp_path), sub, "adv_PC_0.pt"), "rb") as f:
            records = pickle.load(f)                 # names defenses.PatchCleanser.PatchCleanserRecord (main.py:4)
    finally:
        sys.path.remove(ROOT)
    assert len(records) == 1 and len(records[0]) == 4 and out["record_type"] == "defenses.PatchCleanser.PatchCleanserRecord"
    assert all(hasattr(r, "prediction") and hasattr(r, "certification") for r in records[0])
    # the metric line of main.py:186-187, printed by main.py itself
    last = out["stdout"].strip().splitlines()[-1]
    assert last.startswith("clean accuracy: 100.00%, robust accuracy:") and "certified_ASR@PC:" in last
    assert "============= Stage 0 =============" in out["stdout"] and "============= Stage 1 =============" in out["stdout"]



def test_unmodified_reference_main_resumes_from_its_files(first_runs):
    """Untargeted (main.py's default; its resume branch for --targeted re-derives the target from the stage-0 files and
    asserts the attack had succeeded, main.py:112-118 — not after 3 iterations).  A second invocation in the same
    directory takes main.py:102-104 / 144-147: no generate() call, the PatchCleanser records are unpickled — through the
    drop-in ``defenses.PatchCleanser`` — and the same metric line comes out."""
    tmp_path, first = first_runs["untargeted"]
    assert len(first["calls"]) == 1 and first["calls"][0]["targeted"] is False and first["calls"][0]["y"] is None
    last = first["stdout"].strip().splitlines()[-1]
    assert last.startswith("clean accuracy: 100.00%") and "certified_ASR@PC:" in last
    again = _run(tmp_path)
    assert again["calls"] == [] and again["files"] == first["files"]
    assert again["stdout"].strip().splitlines()[-1] == last
