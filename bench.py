#!/usr/bin/env python
"""bench.py — EOT-samples/sec of the DorPatch hot loop on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (reference attack.py:184-342, stage 0) over one batch of
synthetic input: blend/L2-project -> sample masks -> fused occlude+normalise (dp_apply_fwd) ->
frozen ResNetV2-50x1-BiT forward + input-gradient backward (fp32, MIOpen) -> CW loss ->
S-reduction of the input gradients (dp_apply_bwd) -> structural / density / group-lasso terms ->
bookkeeping -> signed update (dp_project_update).  Workload = BASELINE.json configs[1]:
64 images x 32 sampled double-masks = 2048 EOT samples per step per GPU at 224x224.  With N > 1
ranks the per-GPU work is fixed (weak scaling: S = 32*N masks per image, 32 per rank) and the ranks
exchange one all-reduce of the (64,3,224,224) patch gradient per step (RCCL).

Rank 0 prints ONE JSON line (contract in the task statement) including
  "roofline":     dp_apply_fwd, algorithmic bytes (602 112 B/sample @224) / HIP-event time, vs 8 TB/s
  "cpu_baseline": the CPU oracle (a port of the reference step) timed on this host's cores on a
                  bounded sample (1 image x 16 masks per step), N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

DEVICE_OVERRIDE = None     # test hook (tests/test_bench_emu.py runs main() on CPU tensors through the HIP emulation)
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


# BASELINE.json `configs` entries that fit one GPU: (images, masks per image per GPU, side, budget)
PRESETS = {0: (8, 4, 224, 0.0204),          # configs[0]: the reference's own CPU-runnable case
           1: (64, 32, 224, 0.0204),        # configs[1]: the configuration the metric is quoted on (default)
           2: (1, 64, 384, 0.015625),       # configs[2]: 384x384, 48x48 budget, 64 EOT samples
           3: (1, 64, 224, 0.0204)}         # configs[3]: per-GPU share (64 of 512 samples of one image)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", type=int, default=None, choices=sorted(PRESETS),
                    help="index into BASELINE.json `configs`: sets --batch/--samples/--size/--patch-budget")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="images per step (config 2: 64)")
    ap.add_argument("--samples", type=int, default=32, help="sampled masks per image per GPU (config 2: 32)")
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--micro-batch", type=int, default=512, help="EOT samples per backbone fwd/bwd")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="time only the CPU oracle leg (no GPU needed)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads for the CPU baseline (0: min(cores, 32))")
    ap.add_argument("--patch-budget", type=float, default=0.0204, help="32x32 px @224 (SURVEY §0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="skip the separately-timed collect_failure sweep")
    ap.add_argument("--stage", type=int, default=0)
    ap.add_argument("--find", type=int, default=0,
                    help="1: torch.backends.cudnn.benchmark=True (MIOpen exhaustive find: minutes on a fresh box); "
                         "0: MIOpen immediate mode")
    ap.add_argument("--no-fused-gn", action="store_true", help="eager GroupNorm+ReLU instead of dp_gn_relu_*")
    ap.add_argument("--conv1x1", default="table", choices=["table", "auto", "gemm", "miopen"],
                    help="library route of the backbone's frozen 1x1/1 convolutions: the committed per-shape gfx950 "
                         "table (deterministic, default), measured per shape at first use (auto), always the batched "
                         "GEMM, or always MIOpen (dorpatch_amd/conv1x1.py)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for functional "
                                                       "multi-rank tests on a single GPU)")
    ap.add_argument("--same-device", action="store_true",
                    help="all ranks use cuda:0 (functional test of the multi-rank path on a 1-GPU box; gloo only)")
    args = ap.parse_args()
    if args.config is not None:
        args.batch, args.samples, args.size, args.patch_budget = PRESETS[args.config]
    args.config_label = next(("BASELINE configs[%d]" % k for k, v in PRESETS.items()
                              if v == (args.batch, args.samples, args.size, args.patch_budget)), "custom")
    return args


def build_model(device):
    from dorpatch_amd.resnetv2 import resnetv2_50x1_bit, seeded_init_
    from dorpatch_amd.utils import NormModel, get_normalize
    net = seeded_init_(resnetv2_50x1_bit(1000), seed=1234).fold_weight_standardization().freeze()
    return NormModel(net, get_normalize("imagenet", "resnetv2")).to(device).eval()


def cpu_baseline(size, n_masks=16, max_steps=3, budget_s=25.0, threads=None):
    """The reference step restated on the CPU (oracle/restatement.py), weights left trainable as the
    reference leaves them, B = 1 (the only batch the reference supports).  Bounded: stops
    after `max_steps` timed steps or `budget_s` seconds, whichever comes first.  `threads` defaults to
    min(host cores, 32): oneDNN convolutions at batch 16 do not scale past one CCD-group of a big
    2-socket host (256 threads measured 0.13 samples/s on a 2x64-core EPYC 9575F)."""
    from oracle import restatement as R
    cores = os.cpu_count() or 1
    threads = int(threads or min(cores, 32))
    torch.set_num_threads(threads)
    model = build_model("cpu")
    for p in model.parameters():
        p.requires_grad_(True)            # as-is: the reference never freezes the backbone
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(1, 3, size, size, generator=g)
    mask, pattern = torch.rand(1, 1, size, size, generator=g), torch.rand(1, 3, size, size, generator=g)
    y = torch.randint(0, 1000, (1,), generator=g)
    universe = R.mask_universe(size, 2)
    rng = np.random.RandomState(1234)
    lvx = R.local_variance(x)[0].mean(1)

    def one():
        keep = universe[torch.from_numpy(rng.choice(universe.shape[0], n_masks, replace=False))]
        R.eot_step(model, x, mask, pattern, y, keep, stage=0, targeted=True, n_classes=1000, lr=0.01,
                   local_var_x=lvx)
        model.zero_grad(set_to_none=True)
    t0 = time.perf_counter()
    one()                                  # warm-up (oneDNN primitive creation)
    t_warm = time.perf_counter() - t0
    done, t0 = 0, time.perf_counter()
    while done < max_steps and (time.perf_counter() - t0) + t_warm < budget_s:
        one()
        done += 1
    dt = time.perf_counter() - t0
    if done == 0:                          # the warm-up alone exhausted the budget: report it
        done, dt = 1, t_warm
    return {"value": round(n_masks * done / dt, 3), "unit": "EOT-samples/s", "cores": threads,
            "host_cores": cores, "kind": "port",
            "sample": "oracle/restatement.eot_step (reference step, backbone weights trainable as in the "
                      "reference), B=1 x %d masks x %d steps @%dx%d fp32, %d threads, 1 warm-up step "
                      "(%.1f s)" % (n_masks, done, size, size, threads, t_warm)}


def pmc_traffic(B, S, H):
    """HBM bytes per dp_apply_fwd launch from the committed rocprofv3 PMC passes (separate --pmc FETCH_SIZE /
    --pmc WRITE_SIZE runs of tools/kbench; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM, WRITE_SIZE
    verified exact against the fill/copy calibration kernels).  None when no pass matches this geometry."""
    path = os.path.join(ROOT, "profiles", "pmc_apply_fwd.json")
    try:
        with open(path) as f:
            for rec in json.load(f)["records"]:
                if (rec["B"], rec["S"], rec["H"]) == (B, S, H):
                    return rec["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def note(msg):
    """Progress on stderr (stdout carries exactly one JSON line)."""
    print("[bench %7.1fs] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.perf_counter()


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.size, threads=args.cpu_threads or None)))
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node == --gpus"
    if DEVICE_OVERRIDE is None:
        assert torch.cuda.is_available(), "bench.py needs a ROCm GPU (no CPU fallback for the product path)"
        if args.same_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device(DEVICE_OVERRIDE)
    pg = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(args.backend)
        pg = dist.group.WORLD

    from dorpatch_amd.attack import DorPatch, HotLoop
    torch.backends.cudnn.benchmark = bool(args.find)   # reference utils.py:17 sets True (MIOpen find)
    if args.no_fused_gn:
        from dorpatch_amd.resnetv2 import GroupNormAct
        GroupNormAct.fused = False
    from dorpatch_amd import conv1x1
    conv1x1.MODE = args.conv1x1
    B, S_local, H = args.batch, args.samples, args.size
    S = S_local * world                          # weak scaling: fixed per-GPU work
    torch.manual_seed(1234)
    np.random.seed(1234)
    model = build_model(dev)
    x = torch.rand(B, 3, H, H, generator=torch.Generator().manual_seed(1234)).to(dev)
    with torch.no_grad():
        clean = torch.cat([model(x[i:i + 64]).argmax(-1) for i in range(0, B, 64)])
    y = (clean + 1 + torch.randint(0, 998, (B,), generator=torch.Generator().manual_seed(7)).to(dev)) % 1000
    owner = DorPatch(micro_batch=args.micro_batch, process_group=pg, verbose=False)
    loop = HotLoop(owner, model, x, args.patch_budget, 1000, "bench_out/cfg/sub", 0, y, True, 1e-2, 1e-1,
                   0, 1, 10 ** 9, 7, 'topk', 2, S, 1e-3, 1e-3, 4.0, False, {})
    loop.stage = args.stage

    def barrier():
        if dev.type == "cuda":
            torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize()

    note("model + loop ready (B=%d S=%d H=%d world=%d)" % (B, S, H, world))
    i = 1                                        # i % 100 != 0: the periodic failure sweep is timed apart
    for _ in range(args.warmup):
        loop.step(i)
        i += 1
    note("warm-up done")
    loop.kernel_events = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loop.step(i)
        i += 1
    barrier()
    dt = time.perf_counter() - t0
    note("timed region done: %.3f s for %d steps" % (dt, args.steps))
    events = loop.kernel_events
    loop.kernel_events = None
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # the every-100-steps collect_failure sweep (2520 forward-only samples per image), timed apart
    dt_sweep = None
    if not args.no_sweep:
        barrier()
        t1 = time.perf_counter()
        loop._refresh_failures()
        barrier()
        dt_sweep = time.perf_counter() - t1
        note("collect_failure sweep done: %.3f s" % dt_sweep)
    loop.close()

    if rank == 0:
        P = H * H
        apply_ms = float(np.mean([t.ms() for t in events]))   # kernel-begin -> kernel-end (dp_apply_fwd_timed)
        for t in events:
            t.close()
        algo_bytes = B * S_local * 3 * P * 4          # 3*P*4 B written per EOT sample
        achieved = algo_bytes / (apply_ms * 1e-3) / 1e9
        value = B * S * args.steps / dt
        out = {
            "metric": "EOT-samples/sec", "value": round(value, 2), "unit": "EOT-samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d images x %d sampled PatchCleanser double-masks per "
                                   "image per GPU = %d EOT samples/step/GPU, %dx%d, ResNetV2-50x1-BiT "
                                   "(seeded random weights, frozen, fp32), stage-%d step of DorPatch.generate, "
                                   "patch_budget %.4f" % (args.config_label, B, S_local, B * S_local, H, H, args.stage,
                                                          args.patch_budget),
                       "images": B, "masks_per_image_per_gpu": S_local, "masks_per_image_total": S,
                       "image_size": H, "micro_batch": args.micro_batch, "miopen_find": bool(args.find),
                       "fused_gn_relu": not args.no_fused_gn,
                       "conv1x1": dict(mode=args.conv1x1, **conv1x1.report()),
                       "parallelism": "eot-sample sharding x%d, 1 all-reduce of the patch gradient per step" % world},
            "roofline": {"kernel": "k_apply_fwd (dp_apply_fwd)", "bound": "hbm",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(B, S_local, H),
                         "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": round(apply_ms, 4)},
        }
        if dt_sweep is not None:
            out["collect_failure_sweep_ms"] = round(dt_sweep * 1e3, 1)
            out["value_with_sweep_amortised"] = round(B * S * 100 / (100 * dt / args.steps + dt_sweep), 2)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(H, threads=args.cpu_threads or None)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
