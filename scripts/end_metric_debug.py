#!/usr/bin/env python
"""Diagnostic: the 32-problem end-metric run of tests/test_end_metric_gpu.py, batched (4 x 8 images) AND one image at a
time, failure counts side by side with the reference null (round 3: the batched product total was 6 % below the null)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import conftest  # noqa
from dorpatch_amd.attack import DorPatch
from dorpatch_amd import masks, ops
from oracle import toy_models

g = conftest.load_golden("end_metric_null_56.npz")
dev = "cuda:0"
H, S, n_it, eps = int(g["H"]), int(g["S"]), int(g["max_iterations"]), float(g["eps"])
table = ops.upload_table(masks.universe_rects(H, 2), dev)
os.chdir(tempfile.mkdtemp())
gains = g["gains"]
n = len(gains)
res = {"batched": np.zeros(n, int), "single": np.zeros(n, int)}
for mode in ("batched", "single"):
    for gi, gain in enumerate(sorted(set(gains.tolist()))):
        model = toy_models.NormModel(toy_models.make_toy(gain=float(gain)), toy_models.Normalize()).to(dev)
        ks_all = np.flatnonzero(gains == gain)
        for ks in ([ks_all] if mode == "batched" else [ks_all[i:i + 1] for i in range(len(ks_all))]):
            x = torch.from_numpy(g["x"][ks]).to(dev); y = torch.from_numpy(g["target"][ks]).to(dev)
            inits = []
            for k in ks:
                torch.manual_seed(1234 + int(k)); inits.append((torch.rand([1, 1, H, H]), torch.rand((1, 3, H, H))))
            atk = DorPatch(verbose=False)
            mask, pattern = atk.generate(model, x, float(g["patch_budget"]), 10, "%s%d_%d/cfg/sub" % (mode, gi, int(ks[0])), 0, y=y,
                                         targeted=True, sampling_size=S, max_iterations=n_it, eps=eps,
                                         init_mask=torch.cat([m for m, _ in inits]), init_pattern=torch.cat([p for _, p in inits]),
                                         rngs=[np.random.RandomState(1234 + int(k)) for k in ks])
            adv = x + ops.blend(mask, pattern, x, eps, add_x=False)[0]
            for j, k in enumerate(ks):
                res[mode][k] = len(atk.collect_failure(adv[j:j + 1], y[j:j + 1], table, True, model))
print("gain      ", [round(float(v), 2) for v in gains])
print("null mean ", g["n_fail"].mean(0).round(0).astype(int).tolist())
print("null run0 ", g["n_fail"][0].tolist())
print("batched   ", res["batched"].tolist(), res["batched"].sum())
print("single    ", res["single"].tolist(), res["single"].sum())
print("null totals", g["n_fail"].sum(1).tolist())
