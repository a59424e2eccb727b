#!/usr/bin/env python
"""bench.py — EOT-samples/sec of the DorPatch hot loop on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus N --steps K --warmup W          # no launcher: spawns its own N ranks (rank r <-> GPU r)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --config 3 --scaling strong --gpus 8   # BASELINE configs[3]: 512 EOT samples of one image, 64 per GPU
    python bench.py --config {0,2,3}            # the other single-GPU BASELINE configs (parity-test cases, also timed)
    python bench.py --whole-attack              # NOT the headline: seconds per image of a whole attack as main.py runs it
                                                # (stage 0 + stage 1 + failure sweeps + PatchCleanser), retiring finished
                                                # images vs not; library batch shapes warmed first
    DORPATCH_TRACE=1: roctx ranges around the step's phases (for `rocprofv3 --marker-trace --kernel-trace`).
    A/B switches: --deterministic {auto,on,off}, --conv1x1 {table,auto,gemm,miopen}, --conv3x3 {table,on,off}, DORPATCH_TUNABLEOP=0, --stem-split, --skip-satisfied {on,off},
                  --satisfied F (what-if),
                  --no-fused-gn, --micro-batch N, --find 1

One "step" = one pass of the hot path (reference attack.py:184-342, stage 0) over one batch of
synthetic input: blend/L2-project -> sample masks -> fused occlude+normalise (dp_apply_fwd) ->
frozen ResNetV2-50x1-BiT forward + input-gradient backward (fp32: MIOpen / hipBLASLt, the stride-1 3x3 convolutions
on the hand-written fp32-MFMA kernel dp_conv3x3_fwd where the committed table routes them) -> CW loss ->
S-reduction of the input gradients (dp_apply_bwd) -> structural / density / group-lasso terms ->
bookkeeping -> signed update (dp_project_update).  Workload = BASELINE.json configs[1]:
64 images x 32 sampled double-masks = 2048 EOT samples per step per GPU at 224x224.  With N > 1
ranks the per-GPU work is fixed (weak scaling: S = 32*N masks per image, 32 per rank) and the ranks
exchange one all-reduce of the (64,3,224,224) patch gradient per step (RCCL).  `--scaling strong` fixes the
TOTAL instead: --samples is then the number of masks per image over all ranks (each rank takes 1/N of them).
Launch: under torch.distributed.run (RANK / WORLD_SIZE in the environment) this process is one rank; started
plainly with --gpus N > 1 it is the launcher — it checks that N GPUs are visible (exit code 2 with a message
otherwise), starts N copies of itself with RANK = LOCAL_RANK = r on 127.0.0.1, passes rank 0's JSON line
through, and exits non-zero if any rank fails (the others are terminated).  The reference's mechanism for
more than one GPU is nn.DataParallel inside one process (main.py:53).

Rank 0 prints ONE JSON line (contract in the task statement) including
  "roofline":     dp_apply_fwd, algorithmic bytes (602 112 B/sample @224) / HIP-event time, vs 8 TB/s
                  ("roofline_project_update": the same for dp_project_update on a 256-image working set);
                  "traffic" = HBM bytes per launch from rocprofv3 PMC passes run live (a child process
                  replays the same launch through tools/kbench under `rocprofv3 --pmc`), or null
  "cpu_baseline": the CPU oracle (a port of the reference step) timed on this host's cores on a
                  bounded sample (1 image x 32 masks per step, >= 3 warm-up + >= 5 timed
                  steps, backbone weights trainable as the reference leaves them AND frozen), N = 1 only.
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
SIZE_OVERRIDE = None       # test hook (tests/bench_emu_hook_tiny.py): image side in place of the preset's — the exact driver
                           # command line of an 8-GPU configuration at a size 8 emulated ranks finish in a minute
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
F32_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: fp32 matrix (v_mfma_f32_32x32x2_f32) = fp32 vector peak; 155 measured


# BASELINE.json `configs` entries that fit one GPU: (images, masks per image per GPU, side, budget)
PRESETS = {0: (8, 4, 224, 0.0204),          # configs[0]: the reference's own CPU-runnable case
           1: (64, 32, 224, 0.0204),        # configs[1]: the configuration the metric is quoted on (default)
           2: (1, 64, 384, 0.015625),       # configs[2]: 384x384, 48x48 budget, 64 EOT samples
           3: (1, 64, 224, 0.0204)}         # configs[3]: per-GPU share (64 of 512 samples of one image)


CONFIG3_TOTAL_SAMPLES = 512              # configs[3]: "512 EOT samples sharded 64/GPU" on 8 GPUs


def conv_roofline_wanted(rank, world, dev_type, disabled):
    """The extra, untimed step of `conv_roofline` runs on ONE rank — so only in a single-rank job: a step contains the
    all-reduce of the patch gradient, and rank 0 entering it alone would wait for its peers for ever (the per-GPU launches
    are the same at every N: like the PMC traffic, the entry is measured at --gpus 1)."""
    return rank == 0 and world == 1 and dev_type == "cuda" and not disabled


def conv_roofline(loop, i, ms_per_step, samples):
    """One extra, UNTIMED step with an event pair around every matrix-core convolution launch (ops.CONV_EVENTS) ->
    the dominant launch class (kernel, shape, fold / add) as a roofline entry + a per-kernel summary.  Bound: the fp32
    matrix peak (157.3 TFLOP/s); `achieved` = the class's algorithmic flop / its summed launch time."""
    from dorpatch_amd import ops
    ops.CONV_EVENTS = []
    streams = loop.o.streams
    loop.o.streams = 1          # one stream for this pass: an event pair then brackets its own launch only
    try:
        loop.step(i)
        torch.cuda.synchronize()
        ev = ops.CONV_EVENTS
    finally:
        ops.CONV_EVENTS = None
        loop.o.streams = streams
    if not ev:
        return None
    agg, per_kernel = {}, {}
    for kernel, key, flop, a, b in ev:
        ms = a.elapsed_time(b)
        e = agg.setdefault((kernel, key), [0, 0.0, 0.0])
        e[0] += 1; e[1] += ms; e[2] += flop
        k = per_kernel.setdefault(kernel, [0, 0.0, 0.0])
        k[0] += 1; k[1] += ms; k[2] += flop
    (kernel, key), (n, ms, flop) = max(agg.items(), key=lambda kv: kv[1][1])
    N, C, O, HW, fold, add = key
    tf = flop / (ms * 1e-3) / 1e12
    total_ms = sum(v[1] for v in per_kernel.values())
    return {"kernel": "%s (%s)" % (kernel, "dp_conv1x1_fwd" if "1x1" in kernel else "dp_conv3x3_fwd"),
            "launch_class": "N=%d %d->%d @%d pixels%s%s" % (N, C, O, HW, ", GroupNorm folded" if fold else "",
                                                           ", epilogue add" if add else ""),
            "bound": "mfma_f32", "achieved": round(tf, 1), "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / F32_PEAK_TFLOPS, 4), "avg_launch_ms": round(ms / n, 4), "launches_per_step": n,
            "class_ms_per_step": round(ms, 2), "timing": "torch events around each launch on the launch stream, one extra "
                                                         "untimed step on ONE stream (in the step's own order and cache state; "
                                                         "the timed steps run on config.streams streams)",
            "own_conv_kernels": {k: {"launches_per_step": v[0], "ms_per_step": round(v[1], 2),
                                     "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1),
                                     "frac_of_peak": round(v[2] / (v[1] * 1e-3) / 1e12 / F32_PEAK_TFLOPS, 4)}
                                 for k, v in sorted(per_kernel.items())},
            "own_conv_share_of_step": round(total_ms / ms_per_step, 4),
            "classes": [{"kernel": k[0], "N": k[1][0], "C": k[1][1], "O": k[1][2], "pixels": k[1][3], "fold": bool(k[1][4]),
                         "add": bool(k[1][5]), "launches": v[0], "ms": round(v[1], 3), "avg_ms": round(v[1] / v[0], 4),
                         "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1)}
                        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])]}


def conv_class_traffic(cls):
    """HBM bytes of ONE launch of a 1x1 launch class of `roofline_conv.classes` (round 6): the same
    launch replayed by tools/kbench (same kernel, shape, fold / residual flags, the launcher's own tile) under rocprofv3 --pmc,
    WRITE_SIZE + 2 x FETCH_SIZE in separate passes exactly as `roofline.traffic`.  Algorithmic bytes next to it:
    4 (C + O [+ O with the residual]) N HW.  -> dict for the JSON line."""
    N, C, O, HW = cls["N"], cls["C"], cls["O"], cls["pixels"]
    side = int(round(HW ** 0.5))
    mode = (1 if cls["fold"] else 0) + (2 if cls["add"] else 0)
    env = {"DP_C1_SHAPES": "%d:%d:%d" % (C, O, side), "DP_C1_MODES": str(mode), "DP_C1_VARIANTS": "0"}
    traffic, note = pmc_traffic_live(N, 1, 224, bench_filter="conv1x1", kernel="k_conv1x1_mfma", extra_env=env)
    alg = 4 * N * HW * (C + O + (O if cls["add"] else 0))
    out = {"launch_class": "N=%d %d->%d @%d pixels%s%s" % (N, C, O, HW, ", GroupNorm folded" if cls["fold"] else "",
                                                          ", epilogue add" if cls["add"] else ""),
           "tflops_in_step": cls["tflops"], "avg_ms_in_step": cls["avg_ms"], "algorithmic_bytes_per_launch": alg,
           "traffic": traffic, "traffic_source": note}
    if traffic:
        out["traffic_over_algorithmic"] = round(traffic / alg, 3)
        out["hbm_GBs_in_step"] = round(traffic / (cls["avg_ms"] * 1e-3) / 1e9, 1)
        out["hbm_frac_of_8TBs"] = round(traffic / (cls["avg_ms"] * 1e-3) / 8e12, 4)
    return out


def comm_only(loop, pg, world, rank, json_fd, reps=50):
    """bench.py --comm-only: the step's one collective, alone."""
    from dorpatch_amd import dist as dp_dist
    view = loop._comm[:loop._n_g + loop._n_tail]
    for _ in range(5):
        dp_dist.allreduce_sum_(view, pg)
    if view.is_cuda:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        dp_dist.allreduce_sum_(view, pg)
    if view.is_cuda:
        torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    if pg is not None:
        import torch.distributed as dist
        t = torch.tensor([us], dtype=torch.float64, device=view.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        us = float(t.item())
    if rank == 0:
        nbytes = view.numel() * 4
        out = {"metric": "all-reduce of the step's message", "value": round(us, 1), "unit": "us per call", "n_gpus": world,
               "higher_is_better": False, "calls": reps, "bytes": nbytes,
               "algbw_GBs": round(nbytes / (us * 1e-6) / 1e9, 2),
               "message": "HotLoop._comm[:n_g + n_tail] = patch gradient (B,3,H,W) + one loss slab + one prediction slab + one "
                          "draw checksum per rank, fp32, SUM",
               "backend": "none (single rank: the collective is skipped)" if pg is None else "process group"}
        if json_fd is None:
            print(json.dumps(out))
        else:
            os.write(json_fd, (json.dumps(out) + "\n").encode())


def parse(argv=None):
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
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 PMC passes (roofline.traffic = null)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the separately-timed collect_failure sweep")
    ap.add_argument("--no-conv-roofline", action="store_true",
                    help="skip the extra untimed step that stamps every matrix-core convolution launch with events "
                         "(roofline_conv)")
    ap.add_argument("--comm-only", action="store_true",
                    help="no steps: build the loop, then time 50 all-reduces of the step's REAL message (HotLoop._comm: patch "
                         "gradient + loss / prediction slabs) on the process group and report us per call and bytes — the "
                         "collective's cost next to the step, for the first multi-GPU lease")
    ap.add_argument("--no-update-roofline", action="store_true",
                    help="skip the dp_project_update roofline pass after the timed steps (for kernel-trace runs: its 13 "
                         "launches on a 256-image working set would mix into the step's per-kernel statistics)")
    ap.add_argument("--stage", type=int, default=0)
    ap.add_argument("--find", type=int, default=0,
                    help="1: torch.backends.cudnn.benchmark=True (MIOpen exhaustive find: minutes on a fresh box); "
                         "0: MIOpen immediate mode")
    ap.add_argument("--no-fused-gn", action="store_true", help="eager GroupNorm+ReLU instead of dp_gn_relu_*")
    ap.add_argument("--streams", type=int, default=None,
                    help="DorPatch(streams=N): the step's independent micro-batches enqueued round-robin on N HIP streams "
                         "(default: the product's — DORPATCH_STREAMS, else 2; each extra stream keeps one more micro-batch of saved "
                         "activations live in its own allocator pool: DORPATCH_STREAMS=1 where memory is tight)")
    ap.add_argument("--conv3x3-kernel", default="default", choices=["default", "rows", "flat"],
                    help="A/B: dp_debug_set(DP_DEBUG_CONV3X3_VARIANT): which of the two stride-1 3x3 MFMA kernels runs "
                         "(rows: k_conv3x3_mfma wherever it applies; flat: k_conv3x3_flat everywhere; default: per side)")
    ap.add_argument("--gn-fold", default=None, choices=["on", "off"],
                    help="round 5: GroupNorm-apply + ReLU folded into the consuming convolution's operand staging and the "
                         "residual add into the producing convolution's epilogue (default: on, or DORPATCH_GNFOLD=0); off = "
                         "the round-4 graph (A/B)")
    ap.add_argument("--conv1x1", default="table", choices=["table", "auto", "gemm", "miopen", "mfma"],
                    help="library route of the backbone's frozen 1x1/1 convolutions: the committed per-shape gfx950 "
                         "table (deterministic, default), measured per shape at first use (auto), always the batched "
                         "GEMM, or always MIOpen (dorpatch_amd/conv1x1.py)")
    ap.add_argument("--conv3x3", default=None, choices=["table", "on", "off"],
                    help="route of the backbone's stride-1 3x3 convolutions: the committed per-shape table (default: "
                         "dp_conv3x3_fwd, the hand-written fp32-MFMA implicit GEMM, where it measured faster than MIOpen's "
                         "Winograd), every supported shape on dp_conv3x3_fwd, or MIOpen only (dorpatch_amd/libconv.py)")
    ap.add_argument("--deterministic", default="auto", choices=["auto", "on", "off"],
                    help="DorPatch(deterministic=...): auto = verify on the first micro-batch that the library "
                         "convolutions are bit-reproducible and only otherwise force deterministic kernels (default); "
                         "on = always force them (5 %% slower at configs[1], where they change nothing); off = never")
    ap.add_argument("--skip-satisfied", default=None, choices=["on", "off"],
                    help="DorPatch(skip_satisfied=...).  off (default, also the product's default): every EOT sample "
                         "completes forward AND backward — the metric's definition (SURVEY §8d).  on: back-propagate only "
                         "the samples whose CW hinge is active (opt-in; with the benchmark's inputs ~9 %% of the hinges are "
                         "met after 10 steps) — NOT the headline; config.backward reports the counts either way")
    ap.add_argument("--satisfied", type=float, default=None,
                    help="WHAT-IF, not the headline: after the first warm-up step lower the CW confidence so that about "
                         "this fraction of the EOT samples already meets its margin (what a partly successful attack "
                         "looks like); shows what --skip-satisfied buys.  The workload label says so")
    ap.add_argument("--stem-split", action="store_true",
                    help="dp_stem_dgrad_reduce (stem input gradient + S-reduction in one launch) instead of autograd down "
                         "to the masked input + dp_apply_bwd (A/B; bit-identical, measured 0.6 %% slower)")
    ap.add_argument("--placement", action="store_true",
                    help="EXTENSION workload (not the headline): every EOT sample sees the patch under its own random affine "
                         "placement (dorpatch_amd.placement.RandomAffine defaults: +-10 deg, scale 0.9-1.1, +-8 px); the "
                         "roofline object then describes dp_apply_affine_fwd, the fused apply-patch-with-transform-and-"
                         "occlusion kernel north_star names (same algorithmic bytes: 3*P*4 B written per EOT sample)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for functional "
                                                       "multi-rank tests on a single GPU)")
    ap.add_argument("--force-pg", action="store_true",
                    help="with --gpus 1: still create a (world-size-1) process group of --backend, so that every collective "
                         "of the multi-rank path — init_process_group(nccl, device_id), broadcast, broadcast_object_list, "
                         "the SUM all-reduce of the step's _comm buffer, the int32 MAX-reduce of the failure bitmap — "
                         "executes on RCCL on a 1-GPU box (they are not short-circuited when world == 1)")
    ap.add_argument("--whole-attack", action="store_true",
                    help="NOT the headline metric: time a WHOLE attack the way main.py runs it (reference main.py:82-153) — "
                         "`main.py --synthetic --targeted --max_iterations K -b B --sampling_size 128`, one batch: stage 0 + "
                         "stage 1 + the collect_failure sweeps + PatchCleanser at 4 ratios — once with finished images "
                         "leaving the batch (default) and once with --no_retire; prints seconds per image and the split")
    ap.add_argument("--attack-iterations", type=int, default=1000, help="--whole-attack: max_iterations per stage")
    ap.add_argument("--attack-batch", type=int, default=4, help="--whole-attack: images in the batch")
    ap.add_argument("--attack-modes", default="retire,no_retire", help="--whole-attack: which variants to run")
    ap.add_argument("--no-attack-warmup", action="store_true",
                    help="--whole-attack: skip the two throw-away 2-iteration attacks that warm the library's batch shapes")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --samples masks per image PER GPU, the job grows with --gpus; strong: --samples "
                         "masks per image IN TOTAL, each of the N ranks takes 1/N of them (with --config 3: 512 in total = "
                         "BASELINE configs[3], 64 per GPU at N = 8)")
    ap.add_argument("--same-device", action="store_true",
                    help="all ranks use cuda:0 (functional test of the multi-rank path on a 1-GPU box; gloo only)")
    args = ap.parse_args(argv)
    if args.skip_satisfied is None:          # the what-if only makes sense with the selected-sample backward
        args.skip_satisfied = "on" if args.satisfied is not None else "off"
    if args.config is not None:
        args.batch, args.samples, args.size, args.patch_budget = PRESETS[args.config]
        if args.config == 3 and args.scaling == "strong":
            args.samples = CONFIG3_TOTAL_SAMPLES
    if args.scaling == "strong" and args.samples % max(1, args.gpus):
        ap.error("--scaling strong: --samples (%d, the total per image) must be divisible by --gpus (%d)"
                 % (args.samples, args.gpus))
    args.config_label = next(("BASELINE configs[%d]" % k for k, v in PRESETS.items()
                              if v == (args.batch, args.samples, args.size, args.patch_budget)), "custom")
    if args.config == 3 and args.scaling == "strong":
        args.config_label = "BASELINE configs[3] (strong scaling: %d EOT samples of one image in total, %d per GPU)" % (
            args.samples, args.samples // max(1, args.gpus))
    return args


def build_model(device):
    from dorpatch_amd.resnetv2 import resnetv2_50x1_bit, seeded_init_
    from dorpatch_amd.utils import NormModel, get_normalize
    net = seeded_init_(resnetv2_50x1_bit(1000), seed=1234).fold_weight_standardization().freeze()
    return NormModel(net, get_normalize("imagenet", "resnetv2")).to(device).eval()


def cpu_baseline(size, n_masks=32, warm=3, timed=5, budget_s=40.0, threads=None, sweep=(32, 64, 128)):
    """the reference step restated on the CPU (oracle/restatement.eot_step), B = 1 (the only
    batch the reference supports), S = `n_masks` sampled double-masks, `warm` warm-up steps discarded, then
    `timed` steps, MEDIAN step time; two variants: "as-is" (backbone weights keep requires_grad=True, the
    reference's real behaviour — it computes and discards their gradients every step) and "frozen".
    `value` is the as-is figure.  Bounded: a variant stops early once `budget_s` / 2 is spent (never below one
    timed step; the sample string says what was actually run).
    Threads: BASELINE.md §3 says "all cores", but oneDNN convolutions at batch 32 do not scale across a big 2-socket
    host (256 threads measured 0.13 samples/s on a 2x64-core EPYC 9575F), so unless `threads` is given a short scan
    (1 warm-up + 2 timed as-is steps per candidate) over `sweep` (those <= the host's cores, plus the core count itself
    when it is smaller) picks the FASTEST thread count, and the protocol above then runs with it — the reported
    baseline is the best this host does, and the scan is part of the record.
    profiles/r02_cpu_reference_vs_port.json shows this port steps 1.4-1.5x faster than the UNMODIFIED reference
    (through oracle/ref_shim.py) in the build container."""
    from oracle import restatement as R
    cores = os.cpu_count() or 1
    model = build_model("cpu")
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(1, 3, size, size, generator=g)
    mask, pattern = torch.rand(1, 1, size, size, generator=g), torch.rand(1, 3, size, size, generator=g)
    y = torch.randint(0, 1000, (1,), generator=g)
    universe = R.mask_universe(size, 2)
    lvx = R.local_variance(x)[0].mean(1)
    rng = np.random.RandomState(1234)

    def one():
        keep = universe[torch.from_numpy(rng.choice(universe.shape[0], n_masks, replace=False))]
        t0 = time.perf_counter()
        R.eot_step(model, x, mask, pattern, y, keep, stage=0, targeted=True, n_classes=1000, lr=0.01,
                   local_var_x=lvx)
        model.zero_grad(set_to_none=True)
        return time.perf_counter() - t0

    scan = {}
    if threads is None:
        cands = sorted({t for t in sweep if t <= cores} | ({cores} if cores < min(sweep) else set()))
        for p in model.parameters():
            p.requires_grad_(True)
        t_scan = time.perf_counter()
        for t in cands:
            torch.set_num_threads(t)
            one()
            scan[t] = round(n_masks / min(one(), one()), 2)
            if time.perf_counter() - t_scan > budget_s:      # a hopeless candidate must not eat the bench's minutes
                break
        threads = max(scan, key=scan.get)
    threads = int(threads)
    torch.set_num_threads(threads)
    out = {}
    for variant, trainable in (("as_is", True), ("frozen", False)):
        for p in model.parameters():
            p.requires_grad_(trainable)
        t_begin = time.perf_counter()
        warm_t = [one() for _ in range(warm)]
        steps = []
        while len(steps) < timed and (not steps or time.perf_counter() - t_begin < budget_s / 2):
            steps.append(one())
        out[variant] = dict(samples_per_s=round(n_masks / float(np.median(steps)), 3), timed_steps=len(steps),
                            median_step_s=round(float(np.median(steps)), 3), warmup_s=round(sum(warm_t), 2))
    return {"value": out["as_is"]["samples_per_s"], "unit": "EOT-samples/s", "cores": threads, "host_cores": cores,
            "kind": "port", "value_frozen": out["frozen"]["samples_per_s"],
            "thread_scan_samples_per_s": {str(k): v for k, v in scan.items()},
            "sample": "oracle/restatement.eot_step (the reference step, attack.py:184-342), B=1 x %d masks @%dx%d fp32, "
                      "%d threads (%s), %d warm-up steps discarded, median of %d (as-is: backbone weights trainable as in "
                      "the reference) / %d (frozen) timed steps; collect_failure sweep excluded"
                      % (n_masks, size, size, threads,
                         "fastest of a scan over %s" % sorted(scan) if scan else "as requested", warm,
                         out["as_is"]["timed_steps"], out["frozen"]["timed_steps"]),
            "detail": out}


def pmc_traffic_live(B, S, H, timeout_s=150, bench_filter="dp_apply_fwd (default", kernel="k_apply_fwd", extra_env=None):
    """HBM bytes of ONE launch of `kernel` (default: dp_apply_fwd at this run's geometry), measured now: a child process replays
    the launch (tools/kbench, same kernel, same grid) under `rocprofv3 --pmc WRITE_SIZE` and, in a second
    pass, `--pmc FETCH_SIZE` (MI355X_MICROARCH.md: the two do not fit one pass).  Corrections per that guide's
    HBM section: both counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of a wide coalesced
    read at 64 B, so it is doubled; WRITE_SIZE is exact (calibrated on fill/copy kernels of known size,
    profiles/r01b_pmc_kbench_cfg2_raw.json).  -> (bytes or None, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "tools", "kbench")
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not (os.path.exists(exe) and os.path.exists(rocprof)):
        return None, "tools/kbench or rocprofv3 missing"
    kib = {}
    for ctr in ("WRITE_SIZE", "FETCH_SIZE"):
        d = tempfile.mkdtemp(prefix="dp_pmc_", dir="/tmp")
        cmd = [rocprof, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "kb", "--",
               exe, str(B), str(S), str(H), "2", bench_filter]
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", **(extra_env or {})), timeout=timeout_s,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            vals = []
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        if kernel in row.get("Kernel_Name", "") and row.get("Counter_Name") == ctr:
                            vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, "no %s rows for %s in the rocprofv3 output" % (ctr, kernel)
            kib[ctr] = float(np.mean(vals))
        except (subprocess.SubprocessError, OSError, ValueError, KeyError) as e:
            return None, "%s pass failed: %r" % (ctr, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    total = int(round(kib["WRITE_SIZE"] * 1024 + 2 * kib["FETCH_SIZE"] * 1024))
    return total, ("rocprofv3 --pmc, separate passes, this run: WRITE_SIZE %.1f KiB + 2 x FETCH_SIZE %.1f KiB per launch "
                   "(tools/kbench replay of the same launch)" % (kib["WRITE_SIZE"], kib["FETCH_SIZE"]))


def project_update_roofline(dev, H, stage, images=256, iters=10):
    """Second roofline entry: dp_project_update (chain rule through utils.clip + structural / density / group-lasso
    gradients + signed update, reference attack.py:247, 333-342 — the other kernel `north_star` names) on a working set
    past the 256 MiB Infinity Cache: `images` = 256 images @224 = 0.9 GB per launch (the step's own 64 images = 231 MB
    would sit in the cache and flatter it).  Algorithmic bytes per pixel: reads x 12 + adv_x 12 + lv_x 4 +
    g_adv 12 + pattern 12 + mask 4, writes pattern 12 (+ mask 4 in stage 0) = 72 / 68 B.  Timed with events on the
    stream the kernel is launched on (torch's current stream), average of `iters` launches after 3 warm-ups."""
    from dorpatch_amd import ops
    emu = dev.type != "cuda"
    B2 = 1 if emu else images
    g = torch.Generator().manual_seed(5)
    r = lambda *shape: torch.rand(*shape, generator=g).to(dev)
    x, adv, pat, gadv = r(1, 3, H, H).expand(B2, 3, H, H).contiguous(), r(1, 3, H, H).expand(B2, 3, H, H).contiguous(), \
        r(1, 3, H, H).expand(B2, 3, H, H).contiguous(), (r(1, 3, H, H) - 0.5).expand(B2, 3, H, H).contiguous()
    mask, lv = r(1, 1, H, H).expand(B2, 1, H, H).contiguous(), r(1, H, H).expand(B2, H, H).contiguous()
    unit, win = 7, H // 8
    cell, wsum, _, _ = ops.mask_stats(mask, unit, win)
    ones = torch.ones(B2, device=dev)
    kw = dict(stage=stage, lr=ones * 1e-2, coeff_gl=ones * 1e-5, cell_sumsq=cell, win_sum=wsum, unit=unit, win=win,
              density=1e-3, do_update=True)
    run = lambda: ops.project_update(x, adv, lv, gadv, ones, ones * 1e-3, pat, mask, **kw)
    for _ in range(3):
        run()
    if emu:
        t0 = time.perf_counter()
        run()
        ms = (time.perf_counter() - t0) * 1e3
    else:
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        start.record()
        for _ in range(iters):
            run()
        stop.record()
        stop.synchronize()
        ms = start.elapsed_time(stop) / iters
    bpp = 72 if stage == 0 else 68
    algo = B2 * H * H * bpp
    achieved = algo / (ms * 1e-3) / 1e9
    return {"kernel": "k_project_update_v4 (dp_project_update), stage %d" % stage, "bound": "hbm", "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "algorithmic_bytes_per_launch": algo, "avg_launch_ms": round(ms, 4),
            "working_set": "%d images x %dx%d (%.0f MB per launch: past the 256 MiB Infinity Cache)" % (B2, H, H, algo / 1e6)}


def note(msg):
    """Progress on stderr (stdout carries exactly one JSON line)."""
    print("[bench %7.1fs] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.perf_counter()


def whole_attack(args, dev, model=None, n_classes=None, extra_argv=()):
    """`--whole-attack`: dorpatch_amd.driver.run (= the reference's main(args), main.py:47-184) on one synthetic batch,
    per variant.  -> the JSON object.  Seconds are wall clock with the device drained at the boundaries; `stage{0,1}_s`
    include that stage's failure sweeps (`stage{0,1}_sweeps_s`), `patchcleanser_s` is the 4-ratio certification of the
    batch, `other_s` what is left of the driver call (model construction, clean forward, file writes)."""
    import shutil
    import tempfile
    from dorpatch_amd import driver
    B, K, S = args.attack_batch, args.attack_iterations, 128
    res = {}
    modes = [m for m in args.attack_modes.split(",") if m]
    if not args.no_attack_warmup:
        # The first use of a batch SHAPE costs MIOpen tens of seconds of kernel loading plus the determinism probes
        # (profiles/README.md) — once per process, whichever variant meets the shape first.  A batch whose images retire
        # presents 128-row micro-batches (one image) that the non-retiring run never sees, so an unwarmed comparison bills
        # that one-off to `retire` (measured: rounds r04a-c).  Two throw-away 2-iteration attacks — one image (128 rows)
        # and the full batch (B x 128 rows) — warm every shape either variant will use; nothing of them is timed.
        modes = ["warmup1", "warmupB"] + modes
    for mode in modes:
        tmp = tempfile.mkdtemp(prefix="dp_whole_", dir="/tmp")
        cwd = os.getcwd()
        os.chdir(tmp)                       # the reference's result paths are relative (attack.py:103)
        try:
            warm = mode.startswith("warmup")
            argv = ["--synthetic", "--targeted", "-b", str(1 if mode == "warmup1" else B), "--num_images", "1",
                    "--max_iterations", str(2 if warm else K),
                    "--sampling_size", str(S), "--img_size", str(args.size), "--micro_batch", str(args.micro_batch),
                    "--quiet"] + (["--no_retire"] if mode == "no_retire" else []) + list(extra_argv)
            dargs = driver.build_parser().parse_args(argv)
            if dev.type == "cuda":
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = driver.run(dargs, device=dev, model=model, n_classes=n_classes)   # model: tests only (a toy net)
            if dev.type == "cuda":
                torch.cuda.synchronize()
            total = time.perf_counter() - t0
        finally:
            os.chdir(cwd)
            shutil.rmtree(tmp, ignore_errors=True)
        if mode.startswith("warmup"):
            note("whole attack: shapes warmed (%s, %.1f s, not timed)" % (mode, total))
            continue
        bd = out["attack_breakdown"][0]
        n = max(1, bd["images"])
        res[mode] = {
            "seconds_per_image": round(total / n, 3), "total_s": round(total, 3), "images": n,
            "stage0_s": round(bd.get("stage0_s", 0.0), 3), "stage1_s": round(bd.get("stage1_s", 0.0), 3),
            "stage0_sweeps_s": round(bd.get("stage0_sweeps_s", 0.0), 3), "stage1_sweeps_s": round(bd.get("stage1_sweeps_s", 0.0), 3),
            "sweeps": bd["sweeps"], "swept_images": bd["swept_images"],
            "patchcleanser_s": round(out["defense_seconds"], 3),
            "other_s": round(total - out["attack_seconds"] - out["defense_seconds"], 3),
            "stage0_steps": bd.get("stage0_steps"), "stage1_steps": bd.get("stage1_steps"),
            "stage0_image_steps": bd.get("stage0_image_steps"), "stage1_image_steps": bd.get("stage1_image_steps"),
            "samples_forward": bd["samples_forward"], "samples_back_propagated": bd["samples_back_propagated"],
            "certified_asr_PC": out["certified_asr_PC"], "acc_robust": out["acc_robust"]}
        note("whole attack (%s): %.1f s for %d images" % (mode, total, n))
    first = next(iter(res.values()))
    line = {"metric": "whole-attack seconds per image", "value": first["seconds_per_image"], "unit": "s/image",
            "n_gpus": 1, "higher_is_better": False, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "NOT the headline: main.py --synthetic --targeted --max_iterations %d -b %d --sampling_size %d "
                                   "@%dx%d (reference defaults otherwise: patch_budget 0.12, dropout 2, eps 4), ResNetV2-50x1-BiT "
                                   "seeded random weights: stage 0 + stage 1 + collect_failure sweeps + PatchCleanser x4 ratios "
                                   "(reference main.py:82-153)" % (K, B, S, args.size, args.size)},
            "variants": res, "shapes_warmed_before_timing": not args.no_attack_warmup}
    if "retire" in res and "no_retire" in res:
        line["straggler_saving"] = {
            "seconds_per_image": round(res["no_retire"]["seconds_per_image"] - res["retire"]["seconds_per_image"], 3),
            "fraction": round(1.0 - res["retire"]["total_s"] / res["no_retire"]["total_s"], 4),
            "samples_forward_ratio": round(res["retire"]["samples_forward"] / max(1, res["no_retire"]["samples_forward"]), 4)}
    return line


RANK_HOOK_ENV = "DORPATCH_BENCH_RANK_HOOK"     # TEST HOOK: a Python file exec'd in every rank process before main()
                                               # (tests/bench_emu_hook.py routes the kernels through the CPU emulation)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N rank processes of this file (RANK = LOCAL_RANK = r,
    WORLD_SIZE = N, rendezvous on 127.0.0.1:<free port>), rank r bound to GPU r by main().  Rank 0 inherits this
    process's stdout (its ONE JSON line is the launcher's), every rank inherits stderr.  Returns the exit code: 0 only
    if every rank exited 0; the first failure terminates the others (a dead rank would leave them in a collective)."""
    import subprocess
    n = args.gpus
    if os.environ.get(RANK_HOOK_ENV) is None and not args.same_device:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            print("bench.py: --gpus %d needs %d visible GPUs, this host shows %d (HIP_VISIBLE_DEVICES=%r); nothing was run"
                  % (n, n, have, os.environ.get("HIP_VISIBLE_DEVICES")), file=sys.stderr, flush=True)
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    note("launcher: %d ranks started (pids %s), rendezvous 127.0.0.1:%d" % (n, [p.pid for p in procs], port))
    rc = 0
    try:
        live = set(range(n))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print("bench.py: rank %d exited with code %d; terminating the other ranks" % (r, code),
                          file=sys.stderr, flush=True)
                    for q in live:
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse(argv)
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.size, threads=args.cpu_threads or None)))
        return 0
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:       # started without a launcher: be the launcher
        return launch_ranks(args, argv)
    hook = os.environ.get(RANK_HOOK_ENV)
    if hook:                                                    # test infrastructure only, see RANK_HOOK_ENV
        with open(hook) as f:
            exec(compile(f.read(), hook, "exec"), {"__name__": "bench_rank_hook", "__file__": hook, "bench": sys.modules[__name__]})
    if SIZE_OVERRIDE is not None:                               # test infrastructure only, see SIZE_OVERRIDE
        args.size = int(SIZE_OVERRIDE)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("bench.py: --gpus %d but the launcher's WORLD_SIZE is %d (torch.distributed.run --nproc-per-node must equal "
              "--gpus)" % (args.gpus, world), file=sys.stderr, flush=True)
        return 2
    if DEVICE_OVERRIDE is None:
        assert torch.cuda.is_available(), "bench.py needs a ROCm GPU (no CPU fallback for the product path)"
        if args.same_device:
            local_rank = 0
        if local_rank >= torch.cuda.device_count():
            print("bench.py: rank %d wants GPU %d, only %d visible" % (rank, local_rank, torch.cuda.device_count()),
                  file=sys.stderr, flush=True)
            return 2
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device(DEVICE_OVERRIDE)
    if args.whole_attack:
        if world != 1:
            print("bench.py: --whole-attack is a single-GPU measurement", file=sys.stderr, flush=True)
            return 2
        print(json.dumps(whole_attack(args, dev)))
        return 0
    pg = None
    json_fd = None
    if world > 1 or args.force_pg:
        # RCCL prints a version banner ("RCCL version : ...", 5 lines) on the C stdout of rank 0 (gloo: "[Gloo] Rank 0 is
        # connected to ..."); stdout must carry exactly one JSON line, so file descriptor 1 is pointed at stderr for the
        # rest of the run and the JSON line goes to a duplicate of the original descriptor (not under the in-process tests,
        # whose stdout is a Python-level capture)
        if DEVICE_OVERRIDE is None or os.environ.get(RANK_HOOK_ENV):
            sys.stdout.flush()
            json_fd = os.dup(1)
            os.dup2(2, 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        kw = dict(rank=rank, world_size=world) if "RANK" not in os.environ else {}
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, **kw)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(args.backend, **kw)
        pg = dist.group.WORLD

    from dorpatch_amd.attack import DorPatch, HotLoop
    torch.backends.cudnn.benchmark = bool(args.find)   # reference utils.py:17 sets True (MIOpen find)
    from dorpatch_amd.resnetv2 import GroupNormAct
    if args.no_fused_gn:
        GroupNormAct.fused = False
    if args.gn_fold is not None:
        GroupNormAct.fold = args.gn_fold == "on"
    from dorpatch_amd import conv1x1, libconv as _libconv
    conv1x1.MODE = args.conv1x1
    if args.conv3x3 is not None:
        _libconv.CONV3X3 = args.conv3x3
    B, H = args.batch, args.size
    if args.scaling == "strong":                 # fixed total: --samples masks per image over all ranks
        S, S_local = args.samples, args.samples // world
    else:                                        # weak scaling: fixed per-GPU work
        S_local = args.samples
        S = S_local * world
    torch.manual_seed(1234)
    np.random.seed(1234)
    if args.conv3x3_kernel != "default":
        from dorpatch_amd import _lib as _dp_lib, ops as _dp_ops
        _dp_ops.debug_set(_dp_lib.DP_DEBUG_CONV3X3_VARIANT, {"rows": 1, "flat": 2}[args.conv3x3_kernel])
    for kv in filter(None, os.environ.get("DORPATCH_BENCH_DEBUG_SET", "").split(",")):     # A/B only: "5=16,6=16" = dp_debug_set(5, 16) ...
        from dorpatch_amd import ops as _dp_ops
        _dp_ops.debug_set(int(kv.split("=")[0]), int(kv.split("=")[1]))
    model = build_model(dev)
    x = torch.rand(B, 3, H, H, generator=torch.Generator().manual_seed(1234)).to(dev)
    with torch.no_grad():
        clean = torch.cat([model(x[i:i + 64]).argmax(-1) for i in range(0, B, 64)])
    y = (clean + 1 + torch.randint(0, 998, (B,), generator=torch.Generator().manual_seed(7)).to(dev)) % 1000
    owner = DorPatch(micro_batch=args.micro_batch, process_group=pg, verbose=False, streams=args.streams,
                     deterministic={"auto": "auto", "on": True, "off": False}[args.deterministic])
    # failure_refresh: the every-100-steps collect_failure sweep is timed apart below, never inside the timed steps
    extras = dict(failure_refresh=10 ** 12, stem_split=args.stem_split, skip_satisfied=args.skip_satisfied == "on")
    if args.placement:
        from dorpatch_amd.placement import RandomAffine
        extras["placement"] = RandomAffine()
        args.config_label = "custom (EXTENSION: random affine placement per EOT sample; otherwise %s)" % args.config_label
    loop = HotLoop(owner, model, x, args.patch_budget, 1000, "bench_out/cfg/sub", 0, y, True, 1e-2, 1e-1,
                   0, 1, 10 ** 9, 7, 'topk', 2, S, 1e-3, 1e-3, 4.0, False, extras)
    loop.stage = args.stage

    def barrier():
        if dev.type == "cuda":
            torch.cuda.synchronize()
        if pg is not None:
            import torch.distributed as dist
            dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize()

    note("model + loop ready (B=%d S=%d H=%d world=%d)" % (B, S, H, world))
    if args.comm_only:
        comm_only(loop, pg, world, rank, json_fd)
        loop.close()
        if pg is not None:
            import torch.distributed as dist
            dist.destroy_process_group()
        return 0
    i = 1
    for k in range(args.warmup):
        loop.step(i)
        i += 1
        if k == 0 and args.satisfied is not None and world == 1:       # what-if: margin = conf + (other - real); shift conf to the quantile
            margin = loop._own_loss.detach().float().cpu().numpy()
            gap = margin[margin > 0] - loop.confidence
            loop.confidence = float(-np.quantile(gap, args.satisfied)) if gap.size else loop.confidence
            args.config_label = "custom (what-if: CW confidence lowered to %.4f so that ~%d %% of the samples meet their " \
                                "margin)" % (loop.confidence, round(100 * args.satisfied))
    note("warm-up done")
    loop.kernel_events = []
    loop.n_forward = loop.n_active = loop.n_backward = 0
    barrier()
    t0 = time.perf_counter()
    marks, active_each = [], []
    for _ in range(args.steps):
        loop.step(i)            # ends with the step's one D2H sync, so the marks below are per-step times (no extra sync)
        marks.append(time.perf_counter())
        active_each.append(loop.n_active)
        i += 1
    barrier()
    dt = time.perf_counter() - t0
    step_ms = [round(1e3 * (b - a), 1) for a, b in zip([t0] + marks[:-1], marks)]
    active_each = [b - a for a, b in zip([0] + active_each[:-1], active_each)]
    note("timed region done: %.3f s for %d steps; per step (ms): %s; samples with gradient per step: %s"
         % (dt, args.steps, step_ms, active_each))
    events = loop.kernel_events
    loop.kernel_events = None
    digest_dir = os.environ.get("DORPATCH_BENCH_DIGEST_DIR")     # TEST HOOK: every rank records a digest of the optimised state
    if digest_dir:                                              # after the timed steps (replicas must be bit-identical)
        import hashlib
        h = hashlib.sha256()
        for t in (loop.adv_mask, loop.adv_pattern, loop.g_adv):
            h.update(t.detach().cpu().contiguous().numpy().tobytes())
        with open(os.path.join(digest_dir, "rank%d.txt" % rank), "w") as f:
            f.write(h.hexdigest())
    conv_roof = None
    if conv_roofline_wanted(rank, world, dev.type, args.no_conv_roofline):
        conv_roof = conv_roofline(loop, i, dt / args.steps * 1e3, B * S_local)
        i += 1
        note("matrix-core convolution pass done: %s" % (conv_roof or {}).get("kernel"))
    apply_ms = float(np.mean([t.ms() for t in events]))   # kernel-begin -> kernel-end (dp_apply_fwd_timed)
    for t in events:
        t.close()
    if pg is not None:
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
    roof2 = None
    if rank == 0 and H % 8 == 0 and H >= 56 and not args.no_update_roofline:
        roof2 = project_update_roofline(dev, H, args.stage)
        note("dp_project_update roofline pass done: %.4f ms" % roof2["avg_launch_ms"])
    det_report = loop.deterministic_in_effect          # read before close() restores the caller's settings
    from dorpatch_amd import libconv
    det_forced = libconv.summary()["forced_list"]
    loop.close()

    if rank == 0:
        P = H * H
        algo_bytes = B * S_local * 3 * P * 4          # 3*P*4 B written per EOT sample
        achieved = algo_bytes / (apply_ms * 1e-3) / 1e9
        value = B * S * args.steps / dt
        if args.no_pmc or DEVICE_OVERRIDE is not None:
            traffic, traffic_note = None, "skipped (--no-pmc)"
        elif args.placement:
            traffic, traffic_note = None, "not collected for the placement extension"
        elif world > 1:      # the per-GPU launch is the same at every N; the counter passes run in the N = 1 bench only
            traffic, traffic_note = None, "measured at --gpus 1 only"
        else:
            traffic, traffic_note = pmc_traffic_live(B, S_local, H)
            if roof2 is not None and args.stage == 0:      # the same two passes for the second roofline kernel
                roof2["traffic"], roof2["traffic_source"] = pmc_traffic_live(
                    256, 1, H, bench_filter="dp_project_update stage 0", kernel="k_project_update_v4")
        note("PMC passes done: %s" % traffic_note)
        out = {
            "metric": "EOT-samples/sec", "value": round(value, 2), "unit": "EOT-samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d images x %d sampled PatchCleanser double-masks per "
                                   "image per GPU = %d EOT samples/step/GPU, %dx%d, ResNetV2-50x1-BiT "
                                   "(seeded random weights, frozen, fp32), stage-%d step of DorPatch.generate, "
                                   "patch_budget %.4f" % (args.config_label, B, S_local, B * S_local, H, H, args.stage,
                                                          args.patch_budget),
                       "images": B, "masks_per_image_per_gpu": S_local, "masks_per_image_total": S,
                       "image_size": H, "micro_batch": args.micro_batch, "streams": owner.streams, "miopen_find": bool(args.find),
                       "fused_gn_relu": not args.no_fused_gn, "gn_fold": bool(GroupNormAct.fold), "trace": loop.phases.mode, "deterministic": "%s: %s" % (args.deterministic, det_report), "deterministic_forced_problems": det_forced,
                       "backward": {"skip_satisfied": args.skip_satisfied == "on", "explicit_tape": bool(loop._taped),
                                    "samples_forward": loop.n_forward, "samples_with_gradient": loop.n_active,
                                    "samples_back_propagated": loop.n_backward, "tape_micro_batches": loop._tape_tabs,
                                    "step_ms": step_ms, "samples_with_gradient_each_step": active_each},
                       "conv3x3": libconv.report_conv3x3(),
                       "conv3x3_winograd": {"mode": libconv.CONV3X3_WINO, "problems": len(libconv._used_wino),
                                            "note": "stride-1 3x3 problems on own kernels run as Winograd F(2x2,3x3) on the "
                                                    "matrix cores (dp_conv3x3_wino_fwd): fp32, fixed order; effective TFLOP/s of "
                                                    "k_conv3x3_wino in roofline_conv count the direct form's 18 N HW C O flop"},
                       "conv1x1": dict(mode=args.conv1x1, gemm_solutions=conv1x1.report_tuned(),
                                       tuned_selftest=conv1x1.selftest_report(), **conv1x1.report()),
                       "parallelism": "eot-sample sharding x%d, 1 all-reduce per step (patch gradient + loss slabs)%s"
                                      % (world, "" if pg is None else "; process group backend %s%s" % (
                                          args.backend, " (world-size-1 group forced: collectives executed, not skipped)"
                                          if world == 1 else ""))},
            "roofline": {"kernel": "k_apply_affine_fwd (dp_apply_affine_fwd)" if args.placement else "k_apply_fwd (dp_apply_fwd)",
                         "bound": "hbm",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_note,
                         "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": round(apply_ms, 4)},
        }
        if roof2 is not None:
            out["roofline_project_update"] = roof2
        # what the step IS: the frozen backbone's forward + input-gradient backward, 16.36 GFLOP per 224 x 224 sample
        # (x (H / 224)^2), against the fp32 matrix / vector peak of the MI355X (the same 157.3 TFLOP/s)
        step_tflops = B * S_local * 16.36e9 * (H / 224.0) ** 2 / (dt / args.steps) / 1e12
        out["step_tflops"] = round(step_tflops, 1)
        out["step_frac_of_peak"] = round(step_tflops / F32_PEAK_TFLOPS, 4)
        if conv_roof is not None:
            if traffic is not None:      # the PMC passes are on in this run: HBM bytes of the 1x1 classes that matter most
                c1 = [c for c in conv_roof["classes"] if c["kernel"] == "k_conv1x1_mfma" and c["launches"] >= 4]
                picks = {}
                if c1:
                    picks["largest_share"] = max(c1, key=lambda c: c["ms"])
                    picks["slowest"] = min(c1, key=lambda c: c["tflops"])
                    picks["fastest"] = max(c1, key=lambda c: c["tflops"])
                seen, conv_roof["traffic_classes"] = {}, {}
                for name, cls in picks.items():
                    key = (cls["C"], cls["O"], cls["pixels"], cls["fold"], cls["add"])
                    if key not in seen:
                        seen[key] = conv_class_traffic(cls)
                    conv_roof["traffic_classes"][name] = seen[key]
                dom = conv_roof["traffic_classes"].get("largest_share")
                conv_roof["traffic"] = dom["traffic"] if dom and conv_roof["kernel"].startswith("k_conv1x1") else None
                note("conv PMC passes done")
            out["roofline_conv"] = conv_roof
        if dt_sweep is not None:
            out["collect_failure_sweep_ms"] = round(dt_sweep * 1e3, 1)
            out["value_with_sweep_amortised"] = round(B * S * 100 / (100 * dt / args.steps + dt_sweep), 2)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(H, threads=args.cpu_threads or None)
        if json_fd is None:
            print(json.dumps(out))
        else:
            os.write(json_fd, (json.dumps(out) + "\n").encode())
    if pg is not None:
        import torch.distributed as dist
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
