"""Evaluation driver — the reference's ``main.py`` (``main.py:8-187``) on the MI355X path
(the caller of ``DorPatch.generate`` and the on-disk formats either side of it).

Same command-line flags, defaults, result-directory mangling (``utils.generate_saving_path``),
file formats and metric line as the reference:

    results/<k=v_...>/<num_patch=.._patch_budget=..>/adv_mask_{i}.pt      (B,1,H,W) device tensor
                                                     adv_pattern_{i}.pt   (B,3,H,W) device tensor
                                                     adv_PC_{i}.pt        pickle: per image, one
                                                                          PatchCleanserRecord per defence
    results/<k=v_...>/adv_{mask,pattern}_{i}.pt                           stage-0 cache (attack.py:351-356)

and the same resume rule (an existing ``adv_mask_{i}.pt`` / ``adv_PC_{i}.pt`` is loaded instead of
recomputed, ``main.py:100-118, 144-153``).  What changes is the machinery:

* ``DorPatch.generate`` / ``PatchCleanser`` are the HIP-kernel implementations; the four defences'
  mask sweeps of a batch run batched (``robust_predict_batch``) instead of image by image;
* ``nn.DataParallel`` (``main.py:53``) is replaced by one process per GPU
  (``python -m torch.distributed.run --nproc-per-node N main.py ...``) with ``--shard``:
  ``images`` — batch ``i`` goes to rank ``i % world`` (BASELINE configs[4]: 1000 images streamed over
  8 GPUs, no data-path collective at all, records gathered once at the end), or ``samples`` — every
  rank works on the same image and the EOT samples are sharded (configs[3]: one all-reduce of the
  patch gradient per step);
* extra flags (never part of the result path, so directories stay interchangeable with the
  reference's): ``--num_images`` (the reference hard-codes 10, ``main.py:85``), ``--max_iterations``,
  ``--sampling_size``, ``--img_size``, ``--miopen_find``, ``--synthetic`` (seeded-random weights + ``torch.rand`` images:
  neither ImageNet nor the checkpoint can be fetched offline), ``--shard``, ``--micro_batch``, ``--skip_satisfied``,
  ``--no_retire``.
"""
import argparse
import os
import pickle
import time

import numpy as np
import torch

from . import dist as dp_dist
from . import utils as U
from .attack import DorPatch
from .patchcleanser import MaskWindow, PatchCleanser

DEFENSE_RATIOS = (0.015, 0.03, 0.06, 0.12)          # main.py:61
REFERENCE_KEYS = ("device", "dataset", "data_dir", "model_dir", "base_arch", "targeted", "patch_budget",
                  "attack", "batch_size", "epsilon", "lr", "num_patch", "dropout", "density", "structured")


def build_parser():
    """The reference's parser (``main.py:8-44``: same flags, defaults, choices) + the extras above."""
    parser = argparse.ArgumentParser(description='set parameters for patch generation')
    parser.add_argument('--device', default='0', type=str, metavar='DEVICE', help='gpu device id')
    parser.add_argument('--dataset', '-d', default='imagenet', type=str, metavar='DATASET', help='dataset',
                        choices=['cifar10', 'imagenet', 'cifar100'])
    parser.add_argument('--data_dir', default='/home/data/data', help='path to dataset')
    parser.add_argument('--model_dir', default='pretrained_models/', help='path to model')
    parser.add_argument('--base_arch', '-ba', metavar='BARCH', default='resnetv2', choices=['resnetv2'],
                        help='base model architecture for patch generation (default: resnetv2)')
    parser.add_argument('--targeted', '-t', action='store_true', help='targeted attack or not')
    parser.add_argument('--patch_budget', default=0.12, type=float, help='patch budget')
    parser.add_argument('--attack', '-a', default='DorPatch', type=str, metavar='ATTACK', help='atttack method',
                        choices=['DorPatch'])
    parser.add_argument('-b', '--batch-size', default=1, type=int, metavar='N',
                        help='mini-batch size (the reference only works with 1; any B works here: B independent '
                             'single-image problems)')
    parser.add_argument('-e', '--epsilon', default=4., type=float, metavar='E',
                        help='epsilon to bound the perturbation (l2 norm)')
    parser.add_argument('--lr', '--learning-rate', default=0.01, type=float, metavar='LR', help='initial learning rate')
    parser.add_argument('--num_patch', default=-1, type=int, help='number of patches (default: -1 as unconstrained)')
    parser.add_argument('--dropout', default=2, type=int,
                        help='using how many rounds of image dropout (for robustness to occlusion)')
    parser.add_argument('--density', default=1e-3, type=float,
                        help='the coeff of density regularization (for distributed property) or not')
    parser.add_argument('--structured', default=1e-3, type=float, help='the coeff of structured loss')
    # ---- extras (not in the reference; excluded from the result path)
    extra = parser.add_argument_group("MI355X driver extras")
    extra.add_argument('--num_images', default=10, type=int, help='batches to attack (reference: 10, main.py:85)')
    extra.add_argument('--max_iterations', default=5000, type=int, help='per stage (reference: 5000, attack.py:52)')
    extra.add_argument('--sampling_size', default=128, type=int, help='EOT masks per step (reference: 128)')
    extra.add_argument('--img_size', default=224, type=int, help='input side (reference: 224)')
    extra.add_argument('--synthetic', action='store_true',
                       help='seeded-random ResNetV2-50x1-BiT weights and torch.rand images (no dataset / checkpoint)')
    extra.add_argument('--shard', default='images', choices=['images', 'samples'],
                       help='multi-process work split: whole batches per rank, or EOT samples of every batch')
    extra.add_argument('--micro_batch', default=512, type=int, help='EOT samples per backbone forward/backward')
    extra.add_argument('--miopen_find', action='store_true',
                       help="keep the reference's cudnn.benchmark=True (utils.py:17): on ROCm that is MIOpen's exhaustive "
                            "find, minutes per new batch shape for < 2 %% (profiles/README.md); default: immediate mode")
    extra.add_argument('--skip_satisfied', action='store_true',
                       help='back-propagate only the EOT samples whose CW hinge is still active (DorPatch(skip_satisfied=True); '
                            'off = the reference: every sample)')
    extra.add_argument('--no_retire', action='store_true',
                       help='keep early-stopped images of a batch in the EOT pass and the failure sweep until the last image '
                            'stops (the round-3 behaviour; default: they leave the batch, generate(retire=True))')
    extra.add_argument('--quiet', action='store_true', help='no per-iteration progress lines')
    return parser


class SyntheticLoader(object):
    """Deterministic stand-in for the ImageNet-val loader: batch ``i`` is ``torch.rand`` from seed
    ``seed + i`` and is labelled with the model's own clean prediction (so the reference's "keep only
    correctly classified images" filter, ``main.py:91-100``, keeps every image)."""

    def __init__(self, model, n_batches, batch_size, img_size, device, seed=1234):
        self.model, self.n, self.b, self.h, self.dev, self.seed = model, n_batches, batch_size, img_size, device, seed

    def __iter__(self):
        for i in range(self.n):
            x = torch.rand(self.b, 3, self.h, self.h, generator=torch.Generator().manual_seed(self.seed + i))
            with torch.no_grad():
                y = self.model(x.to(self.dev)).argmax(-1).cpu()
            yield x, y


def synthetic_model(n_classes):
    from .resnetv2 import resnetv2_50x1_bit, seeded_init_
    return seeded_init_(resnetv2_50x1_bit(n_classes), seed=1234)


def _dist_env():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return world, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def _load(path, device):
    # the reference saves DEVICE tensors (main.py:135-138); map them to wherever this rank lives
    return torch.load(path, map_location=device)


def _atomic_write(path, write):
    """``write(file_object)`` into a temporary sibling, then rename: a reader (another rank, a resumed run)
    sees either no file or a complete one, never a torn pickle."""
    tmp = "%s.tmp.%d" % (path, os.getpid())
    with open(tmp, "wb") as f:
        write(f)
    os.replace(tmp, path)


def run(args, model=None, dataloader=None, device=None, process_group=None, n_classes=None):
    """``main(args)`` of the reference (``main.py:47-184``).  ``model`` / ``dataloader`` / ``device``
    default to what the reference builds; tests and ``--synthetic`` inject their own.
    Returns the metric dict that is also printed in the reference's format."""
    world, rank = dp_dist.world_rank(process_group)
    local_rank = _dist_env()[2] if world > 1 else 0
    if device is None:
        device = torch.device("cuda", local_rank if world > 1 else 0)
    device = torch.device(device)
    if device.type == "cuda":
        torch.cuda.set_device(device)
    U.set_random_seed()                                                        # main.py:49
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    n_classes = n_classes or U.NUM_CLASSES_DICT[args.dataset]
    ref_cfg = {k: getattr(args, k) for k in REFERENCE_KEYS}
    if rank == 0:
        result_dir = U.generate_saving_path(dict(ref_cfg))                     # main.py:51
    else:                                             # same path on every rank (makedirs is exist_ok), printed once
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            result_dir = U.generate_saving_path(dict(ref_cfg))

    if model is None:                                                          # main.py:53-57
        net = synthetic_model(n_classes) if args.synthetic else U.get_model(args.dataset, args.base_arch, args.model_dir)
        if hasattr(net, "fold_weight_standardization"):
            net.fold_weight_standardization()       # frozen for the whole evaluation: standardise once
        model = U.NormModel(net, U.get_normalize(args.dataset, args.base_arch))
    model = model.to(device).eval()
    for p in model.parameters():
        p.requires_grad_(False)
    if dataloader is None:                                                     # main.py:58-59
        if args.synthetic:
            dataloader = SyntheticLoader(model, args.num_images, args.batch_size, args.img_size, device)
        else:
            dataloader = U.get_dataset(args.dataset, data_dir=args.data_dir, batch_size=args.batch_size)

    shard_samples = world > 1 and args.shard == "samples"
    attack = DorPatch(micro_batch=args.micro_batch, process_group=process_group if shard_samples else None,
                      verbose=not args.quiet, skip_satisfied=args.skip_satisfied)
    defense = [PatchCleanser(MaskWindow(args.img_size, r, 1), model) for r in DEFENSE_RATIOS]   # main.py:61
    owns_files = rank == 0 or not shard_samples       # sample-sharded ranks compute the same tensors: rank 0 writes

    def exists(path):
        """Resume-or-compute is decided ONCE: sample-sharded ranks follow rank 0's view of the file system (a
        slower rank must not take the resume branch on a file rank 0 has just written while rank 0 itself
        computes: the collectives inside generate() would then mismatch)."""
        found = os.path.exists(path)
        return bool(dp_dist.broadcast_object(found, process_group)) if shard_samples else found

    per_batch = {}                                    # i -> dict of numpy results (gathered over ranks at the end)
    breakdown = []                                    # this rank's generate() calls: where the seconds went
    t_attack = t_defense = 0.0
    with torch.no_grad():
        for i, (x, y) in enumerate(dataloader):
            if i == args.num_images:                                           # main.py:85-86
                break
            if world > 1 and not shard_samples:
                if i % world != rank:
                    continue
                U.set_random_seed(1234 + i)           # image-sharded: draws do not depend on the world size
                torch.backends.cudnn.benchmark = bool(args.miopen_find)
            x, y = x.to(device), y.to(device)
            logits = model(x)                                                  # main.py:91-100
            preds = logits.argmax(-1)
            correct = preds == y
            if correct.sum() == 0:
                continue
            x, y, preds = x[correct].contiguous(), y[correct], preds[correct]
            mpath = os.path.join(result_dir, "adv_mask_%d.pt" % i)
            ppath = os.path.join(result_dir, "adv_pattern_%d.pt" % i)
            target = None
            if exists(mpath):                                                  # main.py:102-118
                adv_mask, adv_pattern = _load(mpath, device), _load(ppath, device)
                if args.targeted:       # recover the target label from stage 0
                    dir_0 = os.path.join(*result_dir.split('/')[:-1])
                    m0 = _load(os.path.join(dir_0, "adv_mask_%d.pt" % i), device)
                    p0 = _load(os.path.join(dir_0, "adv_pattern_%d.pt" % i), device)
                    adv_x_0 = x + U.clip(m0, p0, x, args.epsilon)
                    target = model(adv_x_0).argmax(-1)
                    assert (target != y).all()
            else:
                if args.targeted:                                              # main.py:120-124
                    target = torch.randint(0, n_classes, x.shape[:1]).to(device)
                    assert (target != y).all()
                t0 = time.perf_counter()
                with torch.enable_grad():                                      # main.py:128-134
                    adv_mask, adv_pattern = attack.generate(
                        model, x, args.patch_budget, n_classes, targeted=args.targeted,
                        y=target if args.targeted else None, lr=args.lr, num_patch=args.num_patch,
                        dropout=args.dropout, density=args.density, structured=args.structured,
                        save_dir=result_dir, batch_id=i, eps=args.epsilon,
                        max_iterations=args.max_iterations, sampling_size=args.sampling_size,
                        retire=not args.no_retire)
                if device.type == "cuda":
                    torch.cuda.synchronize()
                t_attack += time.perf_counter() - t0
                run_stats = attack.last_run
                breakdown.append(dict(batch=i, images=int(x.shape[0]), seconds=time.perf_counter() - t0,
                                      samples_forward=run_stats.n_forward, samples_back_propagated=run_stats.n_backward,
                                      swept_images=run_stats.swept_images, **run_stats.timing))
                if owns_files:                                                 # main.py:135-138
                    _atomic_write(ppath, lambda f: torch.save(adv_pattern, f))
                    _atomic_write(mpath, lambda f: torch.save(adv_mask, f))    # the mask last: its presence means "done"
            adv_x = x + U.clip(adv_mask, adv_pattern, x, args.epsilon)         # main.py:140-141

            pc_path = os.path.join(result_dir, "adv_PC_%d.pt" % i)             # main.py:143-153
            if exists(pc_path):
                with open(pc_path, 'rb') as f:
                    records_batch = pickle.load(f)
            else:
                t0 = time.perf_counter()
                by_defense = [d.robust_predict_batch(adv_x, True) for d in defense]
                records_batch = [[recs[b] for recs in by_defense] for b in range(adv_x.shape[0])]
                t_defense += time.perf_counter() - t0
                if owns_files:
                    _atomic_write(pc_path, lambda f: pickle.dump(records_batch, f))
            per_batch[i] = dict(preds=preds.cpu().numpy(), y=y.cpu().numpy(),
                                preds_adv=model(adv_x).argmax(-1).cpu().numpy(),            # main.py:158-159
                                target=None if target is None else target.cpu().numpy(), records=records_batch)

    if world > 1 and not shard_samples:               # image-sharded: one gather of the per-batch results
        import torch.distributed as dist
        parts = [None] * world
        dist.all_gather_object(parts, per_batch, group=process_group)
        per_batch = {i: r for part in parts for i, r in part.items()}
    order = sorted(per_batch)
    if not order:
        raise RuntimeError("no correctly classified image in the first %d batches" % args.num_images)
    cat = lambda key: np.concatenate([per_batch[i][key] for i in order])
    preds_list, y_list, preds_adv_list = cat("preds"), cat("y"), cat("preds_adv")
    target_list = cat("target") if args.targeted else None
    records = [r for i in order for r in per_batch[i]["records"]]

    out = summarize(defense, records, preds_list, y_list, preds_adv_list, target_list)     # main.py:161-184
    out.update(result_dir=result_dir, n_images=int(len(y_list)), attack_seconds=t_attack, defense_seconds=t_defense,
               attack_breakdown=breakdown)
    if rank == 0:
        print("clean accuracy: {:.2f}%, robust accuracy:{:.2f}%, acc@PC:{:s}%, certified_ACC@PC:{:s}%, "
              "certified_ASR@PC:{:s}%".format(out["acc_clean"], out["acc_robust"],
                                              U.convert_float_list_to_str(out["acc_PC"]),
                                              U.convert_float_list_to_str(out["certified_acc_PC"]),
                                              U.convert_float_list_to_str(out["certified_asr_PC"])))
    return out


def summarize(defense, records, preds_list, y_list, preds_adv_list, target_list=None):
    """The metric block of ``main.py:166-184``."""
    acc_clean = float((preds_list == y_list).mean() * 100)
    acc_robust = float((preds_adv_list == y_list).mean() * 100)
    for k, d in enumerate(defense):
        d.collect([r[k] for r in records])
    pred_prov = [d.result.predictions for d in defense]
    certifiable = [d.result.certifications for d in defense]
    acc_PC = [float((p == y_list).mean() * 100) for p in pred_prov]
    certified_acc_PC = [float(((p == y_list) & c).mean() * 100) for p, c in zip(pred_prov, certifiable)]
    if target_list is not None:
        certified_asr_PC = [float(((p == target_list) & c).mean() * 100) for p, c in zip(pred_prov, certifiable)]
    else:
        certified_asr_PC = [float(((p != y_list) & c).mean() * 100) for p, c in zip(pred_prov, certifiable)]
    return dict(acc_clean=acc_clean, acc_robust=acc_robust, acc_PC=acc_PC, certified_acc_PC=certified_acc_PC,
                certified_asr_PC=certified_asr_PC)


def main(argv=None):
    args = build_parser().parse_args(argv)
    world, rank, local_rank = _dist_env()
    pg = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm
        pg = dist.group.WORLD
    else:
        U.set_device(args.device)                                              # main.py:48
    try:
        return run(args, process_group=pg)
    finally:
        if pg is not None:
            import torch.distributed as dist
            dist.destroy_process_group()
