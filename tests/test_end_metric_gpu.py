"""End-metric parity: the figures the reference prints at the end of a run
(``main.py:168-184``: certified-ASR / certified-ACC per PatchCleanser mask ratio) and the failure count over
the 2520-mask universe, for 8 single-image problems, against ``tests/golden/end_metric_56.npz`` — recorded
from the UNMODIFIED reference (``oracle/gen_golden.py::make_end_metric_fixture``: both stages of
``DorPatch.generate``, 300 iterations each, then the reference's own ``PatchCleanser.robust_predict``).

The product runs the same 8 problems on the GPU from the same seeds (identical RNG consumption: the mask
draws are index-for-index those of the reference run) through ``attack.DorPatch`` + ``PatchCleanser``.
The two 600-step trajectories are NOT bit-comparable: the update is ``p -= lr * sign(g)``, so a gradient
component at the fp32 noise level flips a +-lr step and the runs decorrelate pixel-wise; what
the reference's user sees — and what is compared here — is the end metric.  The 8 toy classifiers sweep
the gain range in which the attack goes from certifiably succeeding (image 0) to failing (image 6), so
the metric is neither all-zero nor all-one.

Bands (stated, not tuned per run):
 * certified-ASR and certified-ACC per ratio: within 1 image of 8 (12.5 points) of the reference;
 * per image, "is the target reached on the clean adversarial image" agrees for >= 7 of 8 images;
 * failure counts: an image the reference fully breaks (< 5 % of the universe failing) is broken by the product
   (< 15 %), one it cannot break (> 85 %) stays unbroken (> 70 %); every image within 15 % of the universe
   (378 masks) of the reference's count — the in-between images sit on the tipping point, where the count
   swings by a few hundred masks between two equally valid runs (measured: the product's kernels run through
   the CPU emulation differ from the reference by 0 ... 161 masks on these 8 images, the GPU by a similar
   amount, see profiles/README.md).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from dorpatch_amd.attack import DorPatch  # noqa: E402
from dorpatch_amd import masks, ops  # noqa: E402
from dorpatch_amd.patchcleanser import MaskWindow, PatchCleanser  # noqa: E402
from oracle import toy_models  # noqa: E402

DEV = "cuda:0"


def _asr_acc(pred, cert, target, clean):
    """main.py:176-184 for one ratio: % certified predictions of the target / of the clean label."""
    asr = ((pred == target) & cert).mean(0) * 100
    acc = ((pred == clean) & cert).mean(0) * 100
    return asr, acc


def run_product(g, dev, tmp_path, monkeypatch, images=None):
    H, S, n_it, eps = int(g["H"]), int(g["S"]), int(g["max_iterations"]), float(g["eps"])
    monkeypatch.chdir(tmp_path)
    table = ops.upload_table(masks.universe_rects(H, 2), dev)
    pred, cert, n_fail, adv_pred = [], [], [], []
    images = range(len(g["gains"])) if images is None else images
    for k in images:
        model = toy_models.NormModel(toy_models.make_toy(gain=float(g["gains"][k])), toy_models.Normalize()).to(dev)
        x = torch.from_numpy(g["x"][k:k + 1]).to(dev)
        y = torch.tensor([int(g["target"][k])], device=dev)
        torch.manual_seed(1234 + k)                   # gen_golden.run_reference(seed=1234 + k)
        np.random.seed(1234 + k)
        atk = DorPatch(verbose=False)
        mask, pattern = atk.generate(model, x, float(g["patch_budget"]), 10, "res%d/cfg/sub" % k, 0, y=y, targeted=True,
                                     sampling_size=S, max_iterations=n_it, eps=eps)
        adv = x + ops.blend(mask, pattern, x, eps, add_x=False)[0]                       # main.py:140-141
        n_fail.append(len(atk.collect_failure(adv, y, table, True, model)))
        recs = [PatchCleanser(MaskWindow(H, float(r), 1), model).robust_predict(adv[0], True) for r in g["ratios"]]
        pred.append([r.prediction for r in recs])
        cert.append([r.certification for r in recs])
        with torch.no_grad():
            adv_pred.append(int(model(adv).argmax(-1)))
    return np.array(pred), np.array(cert, dtype=bool), np.array(n_fail), np.array(adv_pred)


def check(g, pred, cert, n_fail, adv_pred):
    target, clean = g["target"][:, None], g["clean"][:, None]
    asr, acc = _asr_acc(pred, cert, target, clean)
    asr_ref, acc_ref = _asr_acc(g["pc_pred"], g["pc_cert"].astype(bool), target, clean)
    print("certified ASR  product %s  reference %s" % (asr.tolist(), asr_ref.tolist()))
    print("certified ACC  product %s  reference %s" % (acc.tolist(), acc_ref.tolist()))
    print("failures       product %s  reference %s" % (n_fail.tolist(), g["n_fail"].tolist()))
    one_image = 100.0 / len(target) + 1e-9
    assert (np.abs(asr - asr_ref) <= one_image).all(), (asr, asr_ref)
    assert (np.abs(acc - acc_ref) <= one_image).all(), (acc, acc_ref)
    assert ((adv_pred == g["target"]) == (g["adv_pred"] == g["target"])).sum() >= len(target) - 1
    n_mask = 2520
    assert (np.abs(n_fail - g["n_fail"]) <= 0.15 * n_mask).all(), (n_fail, g["n_fail"])
    for k, ref in enumerate(g["n_fail"]):
        if ref < 0.05 * n_mask:
            assert n_fail[k] < 0.15 * n_mask, (k, n_fail[k], ref)
        elif ref > 0.85 * n_mask:
            assert n_fail[k] > 0.70 * n_mask, (k, n_fail[k], ref)


def test_certified_asr_matches_reference(tmp_path, monkeypatch):
    from conftest import load_golden
    g = load_golden("end_metric_56.npz")
    check(g, *run_product(g, DEV, tmp_path, monkeypatch))


# ---------------------------------------------------------------------------------------------------------------------
# The same question WITH A NULL DISTRIBUTION.  tests/golden/end_metric_null_56.npz holds, for 32 toy
# problems, 8 full runs of the UNMODIFIED reference each: run 0 as is, runs 1-7 with noise of 2 ulp of the typical entry
# added to the gradients its backward produces (oracle/gen_golden.py::make_end_metric_null_fixture) — the spread of
# reference-vs-reference under rounding-level perturbation.  Measured there: certified ASR moves by +-1 image of 32 per
# ratio between runs, an image's failure count by up to ~100 masks (standard deviation) on the tipping-point images and by
# 0 on the clearly broken / unbroken ones.  The product must be a plausible draw from that distribution:
#   * certified ASR / certified ACC per ratio, the number of images whose clean adversarial image reaches the target, and
#     the total failure count: inside the 99 % prediction interval for one more draw given the recorded runs
#     (Student t: 3.7 sample standard deviations for 8 runs, 6.5 for 4) widened by two images (a count over 32 images
#     whose sample deviation over a handful of runs is often exactly 0);
#   * per (image, ratio) cell: where ALL recorded runs agree on "certified attack success" (resp. "certified clean label"),
#     ~120 of the 128 cells, the product agrees on all but <= 5 (a reference run judged against the other seven
#     disagrees on 0-4) — this is the sensitive part: 6 certificates lost or gained on stable images fail it;
#   * each image's failure count: at most 250 masks (10 % of the universe) outside the range the recorded runs span, and
#     at most 3 images more than 100 outside (leave-one-out: worst 175, at most 2 images).
# Leave-one-out over the recorded runs (each reference run judged against the others) passes 8 of 8.
# The 32 problems share 4 toy classifiers, so the product runs them as 4 batched generate() calls of 8 independent
# single-image problems each (own init, own RNG stream = the seeds of the recorded runs: index-for-index identical draws).
def _init_like_reference(k, H):
    """attack.py:59-60 after gen_golden.run_reference's torch.manual_seed(1234 + k): mask first, then pattern."""
    torch.manual_seed(1234 + k)
    return torch.rand([1, 1, H, H]), torch.rand((1, 3, H, H))


def run_product_batched(g, dev, tmp_path, monkeypatch, groups, n_classes):
    """groups: [(model, [image indices])].  -> pred (n,4), cert (n,4), n_fail (n,), adv_pred (n,) in image order."""
    H, S, n_it, eps = int(g["H"]), int(g["S"]), int(g["max_iterations"]), float(g["eps"])
    monkeypatch.chdir(tmp_path)
    table = ops.upload_table(masks.universe_rects(H, 2), dev)
    n = len(g["target"])
    pred, cert = np.zeros((n, len(g["ratios"])), np.int64), np.zeros((n, len(g["ratios"])), bool)
    n_fail, adv_pred = np.zeros(n, np.int64), np.zeros(n, np.int64)
    for gi, (model, ks) in enumerate(groups):
        x = torch.from_numpy(g["x"][ks]).to(dev)
        y = torch.from_numpy(g["target"][ks]).to(dev)
        inits = [_init_like_reference(int(k), H) for k in ks]
        atk = DorPatch(verbose=False)
        mask, pattern = atk.generate(model, x, float(g["patch_budget"]), n_classes, "grp%d/cfg/sub" % gi, 0, y=y,
                                     targeted=True, sampling_size=S, max_iterations=n_it, eps=eps,
                                     init_mask=torch.cat([m for m, _ in inits]), init_pattern=torch.cat([p for _, p in inits]),
                                     rngs=[np.random.RandomState(1234 + int(k)) for k in ks])
        adv = x + ops.blend(mask, pattern, x, eps, add_x=False)[0]                       # main.py:140-141
        with torch.no_grad():
            adv_pred[ks] = model(adv).argmax(-1).cpu().numpy()
        for j, k in enumerate(ks):
            n_fail[k] = len(atk.collect_failure(adv[j:j + 1], y[j:j + 1], table, True, model))
            recs = [PatchCleanser(MaskWindow(H, float(r), 1), model).robust_predict(adv[j], True) for r in g["ratios"]]
            pred[k] = [r.prediction for r in recs]
            cert[k] = [r.certification for r in recs]
    return pred, cert, n_fail, adv_pred


def check_against_null(g, pred, cert, n_fail, adv_pred, per_image=True):
    target, clean = g["target"], g["clean"]
    n = len(target)
    one_image = 100.0 / n
    null_asr = ((g["pc_pred"] == target[None, :, None]) & g["pc_cert"]).mean(1) * 100        # (runs, ratios)
    null_acc = ((g["pc_pred"] == clean[None, :, None]) & g["pc_cert"]).mean(1) * 100
    asr, acc = _asr_acc(pred, cert, target[:, None], clean[:, None])
    report, bad = [], []
    runs = g["pc_pred"].shape[0]
    from scipy import stats
    t99 = float(stats.t.ppf(0.995, runs - 1)) * np.sqrt(1.0 + 1.0 / runs)     # one more draw, mean and sd estimated from `runs`
    for name, got, null in (("certified ASR", asr, null_asr), ("certified ACC", acc, null_acc)):
        mu, sd = null.mean(0), null.std(0, ddof=1)
        report.append("%s  product %s  null mean %s sd %s  [min %s max %s]" % (
            name, got.round(2).tolist(), mu.round(2).tolist(), sd.round(2).tolist(), null.min(0).tolist(), null.max(0).tolist()))
        if not (np.abs(got - mu) <= t99 * sd + 2 * one_image + 1e-9).all():
            bad.append((name, got.tolist(), mu.tolist(), sd.tolist()))
    null_hit = (g["adv_pred"] == target[None]).sum(1)
    hit = int((adv_pred == target).sum())
    report.append("clean adversarial image reaches the target: product %d of %d, null %s" % (hit, n, null_hit.tolist()))
    if not abs(hit - null_hit.mean()) <= t99 * null_hit.std(ddof=1) + 2.0 + 1e-9:
        bad.append(("target reached", hit, null_hit.tolist()))
    tot, null_tot = int(n_fail.sum()), g["n_fail"].sum(1)
    if not abs(tot - null_tot.mean()) <= t99 * null_tot.std(ddof=1) + 0.005 * 2520 * n:
        bad.append(("total failures", tot, null_tot.tolist()))
    for name, mine, null in (("certified attack success", (pred == target[:, None]) & cert,
                              (g["pc_pred"] == target[None, :, None]) & g["pc_cert"].astype(bool)),
                             ("certified clean label", (pred == clean[:, None]) & cert,
                              (g["pc_pred"] == clean[None, :, None]) & g["pc_cert"].astype(bool))):
        unanimous = null.all(0) | (~null).all(0)
        wrong = int(((mine != null[0]) & unanimous).sum())
        report.append("%s: %d of %d (image, ratio) cells unanimous in the null, product disagrees on %d"
                      % (name, int(unanimous.sum()), unanimous.size, wrong))
        if wrong > 5:
            bad.append((name, wrong, np.argwhere((mine != null[0]) & unanimous).tolist()))
    lo, hi = g["n_fail"].min(0), g["n_fail"].max(0)
    outside = np.maximum(np.maximum(lo - n_fail, n_fail - hi), 0)
    report.append("failures  product total %d  null totals %s; per image outside the null's range by at most %d (image %d)"
                  % (tot, null_tot.tolist(), int(outside.max()), int(outside.argmax())))
    report.append("failures per image  product %s\n                    null min %s\n                    null max %s"
                  % (n_fail.tolist(), lo.tolist(), hi.tolist()))
    print("\n".join(report))
    if per_image and not (outside.max() <= 250 and int((outside > 100).sum()) <= 3):
        bad.append(("per-image failures", n_fail.tolist(), lo.tolist(), hi.tolist()))
    assert not bad, bad


def test_end_metric_is_a_plausible_draw_from_the_reference_null(tmp_path, monkeypatch):
    from conftest import load_golden
    g = load_golden("end_metric_null_56.npz")
    gains = g["gains"]
    groups = []
    for gain in sorted(set(gains.tolist())):
        model = toy_models.NormModel(toy_models.make_toy(gain=float(gain)), toy_models.Normalize()).to(DEV)
        groups.append((model, np.flatnonzero(gains == gain)))
    check_against_null(g, *run_product_batched(g, DEV, tmp_path, monkeypatch, groups, int(g["n_classes"])))


def test_end_metric_through_resnetv2_is_a_plausible_draw_from_the_reference_null(tmp_path, monkeypatch):
    """The same through the REAL backbone at reduced resolution: ResNetV2-50x1-BiT, well-conditioned
    seeded weights, 56 x 56 (the reference needs a multiple of 7; 64 is not), S = 8, 300 iterations per stage, 8 images
    x (1 + 3) runs of the unmodified reference on the CPU (tests/golden/end_metric_bit_56.npz,
    gen_golden.make_end_metric_bit_fixture).  The product runs the 8 problems as one batched generate() through the
    hand-written backbone kernels + routed library convolutions.

    What this fixture can and cannot say.  With random weights and 1000 near-tied classes the trajectories are far more
    chaotic than on the toy nets: between two reference runs that differ by 2 ulp of gradient noise, an image's failure
    count moves by up to ~2000 of 2520 masks (image 1: 481 / 2425 / 818 / 774) and PatchCleanser never certifies anything
    (certified ASR 0 in all 4 x 8 x 4 recorded cells, certified ACC 0 except one run).  So the per-image failure-count
    criterion is meaningless here and is switched off; what is held: the certified cells agree with the unanimous zeros,
    the number of clean adversarial images that reach the target (4-7 of 8 in the null) and the total failure count lie in
    the null's prediction interval, and — the point of the exercise — the complete two-stage run + failure sweep +
    PatchCleanser executes through ResNetV2-50 on the GPU and lands where the reference lands."""
    from conftest import load_golden
    from dorpatch_amd.resnetv2 import WELL_CONDITIONED_GN_BIAS, resnetv2_50x1_bit, seeded_init_
    from dorpatch_amd.utils import NormModel, get_normalize
    g = load_golden("end_metric_bit_56.npz")
    net = seeded_init_(resnetv2_50x1_bit(1000), seed=1234, gn_bias=WELL_CONDITIONED_GN_BIAS).fold_weight_standardization().freeze()
    model = NormModel(net, get_normalize("imagenet", "resnetv2")).eval().to(DEV)
    check_against_null(g, *run_product_batched(g, DEV, tmp_path, monkeypatch, [(model, np.arange(len(g["target"])))], 1000),
                       per_image=False)


def test_end_metric_at_224_through_resnetv2_is_a_plausible_draw_from_the_reference_null(tmp_path, monkeypatch):
    """The end metric AT THE SIZE IT IS QUOTED ON: 224 x 224 through ResNetV2-50x1-BiT with a 10-class
    head (so that PatchCleanser has something to certify), well-conditioned seeded weights, S = 32, 100 iterations per stage
    — tests/golden/end_metric_bit_224.npz: 2 (round 4) or 6 (round 5) images x (1 + 3) full runs of the UNMODIFIED reference on the CPU
    (gen_golden.make_end_metric_bit224_fixture; runs 1-3 with 2 ulp of gradient noise = the reference-vs-reference spread).
    Image 0 keeps the network's natural margin between the clean class and the target, image 1 has the target's head bias
    raised (``gains`` = the shift) so that the margin is 0.15: one problem on each side of the tipping point.
    The product runs the two problems through DorPatch.generate -> collect_failure -> PatchCleanser on the GPU from the
    reference runs' seeds (B = 1: the same global RNG streams, index-for-index identical mask draws)."""
    from conftest import load_golden
    from dorpatch_amd.resnetv2 import WELL_CONDITIONED_GN_BIAS, resnetv2_50x1_bit, seeded_init_
    from dorpatch_amd.utils import NormModel, get_normalize
    import os
    from conftest import GOLDEN
    if not os.path.exists(os.path.join(GOLDEN, "end_metric_bit_224.npz")):
        pytest.skip("tests/golden/end_metric_bit_224.npz not generated yet (python -m oracle.gen_golden --only-end-metric-bit224)")
    g = load_golden("end_metric_bit_224.npz")
    H, S, n_it, eps = int(g["H"]), int(g["S"]), int(g["max_iterations"]), float(g["eps"])
    assert (H, S, n_it, int(g["n_classes"])) == (224, 32, 100, 10)
    monkeypatch.chdir(tmp_path)
    table = ops.upload_table(masks.universe_rects(H, 2), DEV)
    n = len(g["target"])
    pred, cert = np.zeros((n, 4), np.int64), np.zeros((n, 4), bool)
    n_fail, adv_pred = np.zeros(n, np.int64), np.zeros(n, np.int64)
    for k in range(n):
        net = seeded_init_(resnetv2_50x1_bit(10), seed=1234, gn_bias=WELL_CONDITIONED_GN_BIAS).fold_weight_standardization().freeze()
        with torch.no_grad():
            net.head.fc.bias[int(g["target"][k])] += float(g["gains"][k])          # gen_golden.bit224_problem
        model = NormModel(net, get_normalize("imagenet", "resnetv2")).eval().to(DEV)
        x = torch.from_numpy(g["x"][k:k + 1]).to(DEV)
        y = torch.tensor([int(g["target"][k])], device=DEV)
        with torch.no_grad():
            assert int(model(x).argmax(-1)) == int(g["clean"][k])
        torch.manual_seed(1234 + k)                   # gen_golden.run_reference(seed=1234 + k)
        np.random.seed(1234 + k)
        atk = DorPatch(verbose=False)
        mask, pattern = atk.generate(model, x, float(g["patch_budget"]), 10, "res%d/cfg/sub" % k, 0, y=y, targeted=True,
                                     sampling_size=S, max_iterations=n_it, eps=eps)
        adv = x + ops.blend(mask, pattern, x, eps, add_x=False)[0]                       # main.py:140-141
        n_fail[k] = len(atk.collect_failure(adv, y, table, True, model))
        recs = [PatchCleanser(MaskWindow(H, float(r), 1), model).robust_predict(adv[0], True) for r in g["ratios"]]
        pred[k], cert[k] = [r.prediction for r in recs], [r.certification for r in recs]
        with torch.no_grad():
            adv_pred[k] = int(model(adv).argmax(-1))
    print("224 end metric: product pred %s cert %s n_fail %s adv_pred %s" % (pred.tolist(), cert.tolist(), n_fail.tolist(),
                                                                            adv_pred.tolist()))
    print("               reference runs: pred %s cert %s n_fail %s adv_pred %s" % (
        g["pc_pred"].tolist(), g["pc_cert"].tolist(), g["n_fail"].tolist(), g["adv_pred"].tolist()))
    # The four recorded runs are UNANIMOUS on everything (2 ulp of gradient noise moves nothing here): image 0 is not
    # broken at all (2520 of 2520 masks fail, PatchCleanser certifies the clean class at all four ratios), image 1 is
    # broken (1 failing mask of 2520; PatchCleanser returns the TARGET at all four ratios, certified at 0.015 / 0.03 / 0.06
    # and not at 0.12) -> certified ASR 50 / 50 / 50 / 0 %, certified ACC 50 / 50 / 50 / 50 %.  With a two-image fixture
    # the interval tests of check_against_null are vacuous (their slack is two images), so the product is held to the
    # cells directly: at most ONE of the 8 (image, ratio) cells may differ for "certified attack success" and for
    # "certified clean label", the clean adversarial images must land where the reference's do, and each failure count
    # must lie within 10 % of the universe (250 masks) of the recorded range.
    if n > 2:
        # The extended fixture (round 5): 6 images whose clean-vs-target margins straddle the tipping point
        # (none, 0.15, 0.05, 0.22, 0.30, 0.38), 1 + 3 reference runs each.  What the reference recorded: the four runs are
        # UNANIMOUS on every cell again (2 ulp of gradient noise moves nothing at 100 iterations per stage) — images 1, 2
        # broken (1 / 0 failing masks of 2520, the target certified at 0.015 - 0.06), images 0, 4, 5 not (2520 / 2519 / 2520
        # failing masks, the clean class certified), image 3 IN BETWEEN: 1110 failing masks, PatchCleanser returns the target
        # at three ratios and the clean class at 0.12, nothing certified.  Criteria, fixed before the product was run on it:
        # the five decided images must agree with the reference cell for cell on "certified attack success" and "certified
        # clean label" up to ONE cell, land where the reference's clean adversarial images land, and keep their failure
        # counts within 10 % of the universe (250 masks); the in-between image may fall either way but must stay
        # un-certified-as-clean-AND-as-target in at most the pattern of one side, i.e. its failure count is only reported.
        target, clean = g["target"], g["clean"]
        ref_asr = (g["pc_pred"] == target[None, :, None]) & g["pc_cert"].astype(bool)
        ref_acc = (g["pc_pred"] == clean[None, :, None]) & g["pc_cert"].astype(bool)
        assert (ref_asr == ref_asr[0]).all() and (ref_acc == ref_acc[0]).all() and (g["n_fail"] == g["n_fail"][0]).all()
        asr, acc = (pred == target[:, None]) & cert, (pred == clean[:, None]) & cert
        print("certified ASR per ratio: product %s reference %s; certified ACC: product %s reference %s" % (
            (asr.mean(0) * 100).round(1).tolist(), (ref_asr[0].mean(0) * 100).round(1).tolist(),
            (acc.mean(0) * 100).round(1).tolist(), (ref_acc[0].mean(0) * 100).round(1).tolist()))
        ref_fail = g["n_fail"][0]
        decided = np.flatnonzero((ref_fail <= 2) | (ref_fail >= 2518))          # images 0, 1, 2, 4, 5
        between = np.setdiff1d(np.arange(n), decided)
        print("decided images %s, in between %s: product failures %s vs reference %s" % (
            decided.tolist(), between.tolist(), n_fail[between].tolist(), ref_fail[between].tolist()))
        assert len(decided) == 5
        assert int((asr[decided] != ref_asr[0][decided]).sum()) <= 1, (asr.tolist(), ref_asr[0].tolist())
        assert int((acc[decided] != ref_acc[0][decided]).sum()) <= 1, (acc.tolist(), ref_acc[0].tolist())
        assert ((adv_pred == target) == (g["adv_pred"][0] == target))[decided].all(), (adv_pred.tolist(), g["adv_pred"][0].tolist())
        assert (np.abs(n_fail - ref_fail)[decided] <= 250).all(), (n_fail.tolist(), ref_fail.tolist())
        # certified ASR / ACC of the whole set: within one image (16.7 %) of the reference at every ratio
        assert (np.abs(asr.mean(0) - ref_asr[0].mean(0)) <= 1.0 / n + 1e-9).all()
        assert (np.abs(acc.mean(0) - ref_acc[0].mean(0)) <= 1.0 / n + 1e-9).all()
        return
    target, clean = g["target"], g["clean"]
    ref_asr = (g["pc_pred"] == target[None, :, None]) & g["pc_cert"].astype(bool)
    ref_acc = (g["pc_pred"] == clean[None, :, None]) & g["pc_cert"].astype(bool)
    assert (ref_asr == ref_asr[0]).all() and (ref_acc == ref_acc[0]).all() and (g["n_fail"] == g["n_fail"][0]).all()
    asr, acc = (pred == target[:, None]) & cert, (pred == clean[:, None]) & cert
    print("certified ASR per ratio: product %s reference %s; certified ACC: product %s reference %s" % (
        (asr.mean(0) * 100).tolist(), (ref_asr[0].mean(0) * 100).tolist(), (acc.mean(0) * 100).tolist(),
        (ref_acc[0].mean(0) * 100).tolist()))
    assert int((asr != ref_asr[0]).sum()) <= 1, (asr.tolist(), ref_asr[0].tolist())
    assert int((acc != ref_acc[0]).sum()) <= 1, (acc.tolist(), ref_acc[0].tolist())
    assert ((adv_pred == target) == (g["adv_pred"][0] == target)).all(), (adv_pred.tolist(), g["adv_pred"][0].tolist())
    lo, hi = g["n_fail"].min(0), g["n_fail"].max(0)
    assert (np.maximum(np.maximum(lo - n_fail, n_fail - hi), 0) <= 250).all(), (n_fail.tolist(), lo.tolist(), hi.tolist())
