"""Host logic of the evaluation driver (reference main.py:8-51, 161-184) that needs no kernels:
the command line, the result-directory mangling and the metric block."""
import os

import numpy as np

from dorpatch_amd import driver
from dorpatch_amd.patchcleanser import PatchCleanser, PatchCleanserRecord

# (dest, default) of every flag the reference's parser defines — main.py:11-44
REFERENCE_FLAGS = dict(device='0', dataset='imagenet', data_dir='/home/data/data', model_dir='pretrained_models/',
                       base_arch='resnetv2', targeted=False, patch_budget=0.12, attack='DorPatch', batch_size=1,
                       epsilon=4., lr=0.01, num_patch=-1, dropout=2, density=1e-3, structured=1e-3)


def test_parser_keeps_the_reference_flags_and_defaults():
    import main as root_main                         # the drop-in module exposes `parser` like the reference
    for parser in (driver.build_parser(), root_main.parser):
        ns = vars(parser.parse_args([]))
        for k, v in REFERENCE_FLAGS.items():
            assert ns[k] == v and type(ns[k]) is type(v), k
        assert set(driver.REFERENCE_KEYS) == set(REFERENCE_FLAGS)
        # extras default to the reference's hard-coded values
        assert (ns["num_images"], ns["max_iterations"], ns["sampling_size"], ns["img_size"]) == (10, 5000, 128, 224)
    # the reference's short options and aliases
    ns = driver.build_parser().parse_args(["-d", "cifar10", "-ba", "resnetv2", "-t", "-a", "DorPatch", "-b", "4",
                                           "-e", "2.5", "--learning-rate", "0.1"])
    assert (ns.dataset, ns.targeted, ns.batch_size, ns.epsilon, ns.lr) == ("cifar10", True, 4, 2.5, 0.1)


def test_result_path_is_the_reference_path_and_ignores_extras(tmp_path, monkeypatch):
    """utils.py:24-44 on main.py's vars(args): extras never leak into the directory name."""
    from dorpatch_amd import utils as U
    monkeypatch.chdir(tmp_path)
    args = driver.build_parser().parse_args(["--targeted", "--patch_budget", "0.06", "--num_images", "3",
                                             "--synthetic", "--max_iterations", "7"])
    path = U.generate_saving_path({k: getattr(args, k) for k in driver.REFERENCE_KEYS})
    assert path == os.path.join("results", "dataset=imagenet_base_arch=resnetv2_targeted=True_attack=DorPatch_"
                                "dropout=2_density=0.001_structured=0.001", "num_patch=-1_patch_budget=0.06")
    assert os.path.isdir(path)
    # attack.py:103 derives the stage-0 cache directory from it
    assert os.path.join(*path.split('/')[:-1]) == os.path.dirname(path)


def test_summarize_is_the_reference_metric_block():
    """main.py:161-184 on hand-made records: 3 images x 2 defences."""
    rec = lambda p, c: PatchCleanserRecord(p, c, np.zeros(36, dtype=np.int64), np.ones(630, dtype=bool))
    records = [[rec(1, True), rec(1, False)], [rec(5, True), rec(2, True)], [rec(3, False), rec(9, True)]]
    defense = [PatchCleanser(None, None), PatchCleanser(None, None)]
    y = np.array([1, 2, 3])
    preds, preds_adv, target = np.array([1, 2, 3]), np.array([1, 5, 9]), np.array([5, 5, 9])
    out = driver.summarize(defense, records, preds, y, preds_adv, target)
    assert out["acc_clean"] == 100.0 and abs(out["acc_robust"] - 100 / 3) < 1e-9
    np.testing.assert_allclose(out["acc_PC"], [200 / 3, 200 / 3])
    np.testing.assert_allclose(out["certified_acc_PC"], [100 / 3, 100 / 3])
    np.testing.assert_allclose(out["certified_asr_PC"], [100 / 3, 100 / 3])      # p == target & certified
    out_u = driver.summarize(defense, records, preds, y, preds_adv, None)
    np.testing.assert_allclose(out_u["certified_asr_PC"], [100 / 3, 100 / 3])    # p != y & certified
    assert defense[0].result.predictions.tolist() == [1, 5, 3]
