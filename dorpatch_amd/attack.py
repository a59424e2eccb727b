"""MI355X-native DorPatch optimiser — drop-in for the reference ``attack.DorPatch``.

Boundary: ``DorPatch().generate(...)`` keeps the reference signature
(``attack.py:51-53``), return values (``attack.py:361``) and stage-0 cache files
(``attack.py:134-141, 351-356``).  Inside, the per-step hot loop
(``attack.py:167-342``) is a PyTorch-ROCm host that launches the hand-written
HIP kernels of ``csrc/dorpatch_hip.hip`` through the C ABI:

    dp_sumsq_partials, dp_blend   utils.clip + adv_x          (utils.py:105-110, attack.py:184-185)
    dp_struct_loss                loss_struc                   (attack.py:33-45, 227-228)
    dp_mask_stats                 density + group lasso        (attack.py:237-245)
    dp_apply_fwd                  mask sampling apply (+Norm)  (attack.py:204-220, utils.py:77-78)
    backbone fwd/bwd              torch / MIOpen / hipBLASLt convolutions (frozen), with dorpatch_amd/resnetv2.py also
                                  dp_gn_relu_*, dp_pad_maxpool_*, dp_stem_dgrad, dp_subsample2* (attack.py:222, 247)
    dp_cw_loss                    CW loss + dlogits            (attack.py:16-23, 224-230)
    dp_apply_bwd, dp_sum_slabs    sum_S of input grads         (autograd of attack.py:206-220)
    dp_project_update             chain rule, TV/sparsity grads, signed update (attack.py:247, 333-342)

Differences from the reference, all deliberate and documented in DESIGN.md:

* a batch of B images is **B independent single-image problems** (own lr, own
  coefficient schedules, own failure set, own RNG stream); the reference only
  works for B = 1 (``attack.py:98`` raises for B > 1) and for B = 1 this
  implementation consumes the same global numpy / torch RNG streams;
* backbone parameters are frozen while ``generate`` runs (the reference wastes a
  third of the backward on weight gradients it never reads);
* one device->host synchronisation per step instead of >= 8;
* EOT samples shard across ranks (``process_group``) with ONE all-reduce per step: the
  (B,3,H,W) patch gradient with every rank's loss / prediction columns riding in its tail
  (``HotLoop._comm``) — the reference's only multi-GPU mechanism is ``nn.DataParallel``
  (``main.py:53``);
* two latent reference bugs on the untargeted path (``set_target`` called with a
  missing argument, ``attack.py:155, 359``) are implemented as evidently intended.
"""
import contextlib
import os

import numpy as np
import torch

from . import dist as dp_dist
from . import masks, ops
from ._lib import DP_MAX_RECTS

SCALE_UP = 1.2                              # attack.py:88
SCALE_DOWN = np.sqrt(SCALE_UP ** 3)         # attack.py:89
PATIENCE = 200                              # attack.py:65
SUCCESS_THRESHOLD = np.float32(1e-1)        # attack.py:255
IMPROVE_THRESHOLD = np.float32(-1e-3)       # attack.py:275
LR_DECAY = np.float32(0.1)                  # attack.py:306
LR_FLOOR = np.float32(.1 / 256.)            # attack.py:307
LR_STOP = np.float32(1e-3)                  # attack.py:311


class CW_loss(object):
    """Holder of the criterion state (``attack.py:10-23``); evaluated by ``dp_cw_loss``.

    Kept so ``DorPatch().criterion`` exists after ``generate`` like in the reference."""

    def __init__(self, num_classes, targeted=False, confidence=0):
        self.num_classes = num_classes
        self.targeted = targeted
        self.confidence = confidence

    def __call__(self, logits, y):
        B = y.shape[0]
        S = logits.shape[0] // B
        flag = torch.full((B,), 1 if self.targeted else 0, dtype=torch.int32, device=logits.device)
        loss, _, _ = ops.cw_loss(logits.contiguous().float(), y.contiguous().long(), flag, S,
                                 self.confidence, 1.0, want_grad=False, want_pred=False)
        return loss


def _unwrap_model(model):
    """Peel ``DataParallel`` (single-process scatter/gather is replaced by one process
    per GPU) and, if present, a ``NormModel`` whose ``(x-mean)/std`` is fused into
    ``dp_apply_fwd``.  Returns ``(net, norm_or_None)``."""
    if isinstance(model, torch.nn.DataParallel):
        model = model.module
    norm = None
    inner = getattr(model, "model", None)
    nz = getattr(model, "normalize", None)
    if type(model).__name__ == "NormModel" and inner is not None and hasattr(nz, "mean") \
            and hasattr(nz, "std"):
        mean = np.asarray(torch.as_tensor(nz.mean).detach().cpu().reshape(-1).tolist(), dtype=np.float64)
        std = np.asarray(torch.as_tensor(nz.std).detach().cpu().reshape(-1).tolist(), dtype=np.float64)
        if mean.size in (1, 3) and std.size == mean.size:
            norm = (mean, std)
            model = inner
    return model, norm


class _ImageState(object):
    """Host bookkeeping of ONE image: a restatement of the reference's per-step
    control logic for B = 1 (``attack.py:249-316``) in fp32/numpy scalars."""

    def __init__(self, lr, structured, coeff_group_lasso, targeted, y):
        self.lr0 = np.float32(lr)
        self.structured = float(structured)
        self.coeff_group_lasso = float(coeff_group_lasso)
        self.flag_targeted = bool(targeted)   # the reference's local `targeted`
        self.crit_targeted = bool(targeted)   # CW_loss.targeted
        self.y = int(y)
        self.failed_idxs = []
        self.certifiable = False
        self.steps_in_stage = 0
        self.reset_stage()

    def reset_stage(self):
        """attack.py:129-132 (and 177-180 at the untargeted->targeted switch)."""
        self.lr_current = np.float32(self.lr0)
        self.loss_best = np.float32(np.inf)
        self.not_decay = 0
        self.num_failure = np.inf
        self.active = True

    def n_from_failure(self, i, sampling_size, start):
        """attack.py:193."""
        return 0 if i < start else min(len(self.failed_idxs), sampling_size // 2)

    def step(self, i, stage, loss_adv, loss_target, sampling_idxs, n_form_failure):
        """Consume this step's losses; returns (save_best, stop).  attack.py:255-316."""
        self.steps_in_stage += 1
        attack_success = loss_adv < SUCCESS_THRESHOLD            # :255
        mask_success = attack_success                           # :257 (.all(0) over one image)
        new_successes = sampling_idxs[:n_form_failure][mask_success[:n_form_failure]]
        if len(new_successes) > 0:                              # :261-262
            self.failed_idxs = np.setdiff1d(self.failed_idxs, new_successes).tolist()
        new_failures = sampling_idxs[n_form_failure:][~mask_success[n_form_failure:]]
        if len(new_failures) > 0:                               # :265-267
            self.failed_idxs = list(self.failed_idxs)
            self.failed_idxs.extend(new_failures.tolist())
            self.failed_idxs = np.unique(self.failed_idxs).tolist()
        success_all = bool(attack_success.all())                # :269
        self.certifiable = (len(self.failed_idxs) == 0)         # :270
        if len(self.failed_idxs) < self.num_failure:            # :272-273
            self.loss_best = np.float32(np.inf)
        certify_better = len(self.failed_idxs) <= self.num_failure          # :274
        loss_target = np.float32(loss_target)
        with np.errstate(invalid="ignore"):
            loss_decay = bool(certify_better and ((loss_target - self.loss_best) < IMPROVE_THRESHOLD))  # :275
        if loss_decay:                                          # :281-283
            self.num_failure = len(self.failed_idxs)
            self.loss_best = loss_target
            self.not_decay = 0                                  # :290
        else:
            self.not_decay += 1                                 # :291
        early_stop = self.not_decay > PATIENCE                  # :292
        good = success_all and self.certifiable
        if stage == 0 and i > 200:                              # :294-303
            self.coeff_group_lasso = (self.coeff_group_lasso * SCALE_UP) if good \
                else (self.coeff_group_lasso / SCALE_DOWN)
        else:
            self.structured = (self.structured * SCALE_UP) if good else (self.structured / SCALE_DOWN)
        if early_stop:                                          # :305-308
            self.lr_current = np.float32(self.lr_current * LR_DECAY)
            self.lr_current = np.float32(max(self.lr_current, LR_FLOOR))
            self.not_decay = 0
        stop = bool(self.lr_current < LR_STOP)                  # :311
        return loss_decay, stop


class _Phases(object):
    """Opt-in tracing of the step's phases (the reference has none).  ``DORPATCH_TRACE=1``: each phase is a
    roctx range (``torch.cuda.nvtx`` is roctx on ROCm) — `rocprofv3 --marker-trace --kernel-trace` then groups the
    kernels of a step under `dp:blend`, `dp:collect_failure`, `dp:sample`, `dp:regularisers`, `dp:eot_fwd_bwd`,
    `dp:allreduce`, `dp:sync+bookkeeping`, `dp:project_update`.  ``DORPATCH_TRACE=log``: the names are appended to
    ``.log`` instead (what the CPU tests read).  Off (default): every call is one attribute test."""

    def __init__(self, mode, cuda):
        self.log = []
        self.mode = None
        self._open = False
        if mode == "log":
            self.mode = "log"
        elif mode and mode != "0" and cuda:
            try:                                  # a torch build without roctx: tracing off, never an error
                torch.cuda.nvtx.range_push("dp:probe")
                torch.cuda.nvtx.range_pop()
                self.mode = "roctx"
            except Exception:                     # noqa: BLE001
                self.mode = None

    def mark(self, name):
        """End the current phase (if any) and, unless ``name`` is None, begin phase ``name``."""
        if self.mode is None:
            return
        if self.mode == "log":
            if name is not None:
                self.log.append(name)
            return
        if self._open:
            torch.cuda.nvtx.range_pop()
            self._open = False
        if name is not None:
            torch.cuda.nvtx.range_push(name)
            self._open = True


class DorPatch(object):
    """Drop-in for the reference ``attack.DorPatch`` (``attack.py:47-406``).

    Extra constructor arguments (all optional, defaults reproduce the reference):
    ``micro_batch`` — max EOT samples per backbone forward/backward (activation
    memory bound); ``process_group`` — a ``torch.distributed`` group whose ranks
    each hold a replica and process 1/world of the S sampled masks; ``verbose``;
    ``deterministic`` — run-to-run bit reproducibility of the backbone's library convolutions (the optimiser
    takes ``sign(grad)``: two runs from identical seeds otherwise drift apart).  At small batches MIOpen's
    immediate mode picks split-K implicit-GEMM kernels that accumulate with float atomics (5 of ResNetV2-50's
    23 convolution shapes at 8 samples; none at 512 — ``profiles/r02c_determinism_probe.jsonl``).
    ``True``: ``torch.backends.cudnn.deterministic = True`` while ``generate`` runs (bans MIOpen's whole NHWC
    implicit-GEMM family: 5 % at the benchmark configuration, where it changes nothing, 7.8 % at 1 image x 128 masks);
    ``False``: leave MIOpen alone; ``"auto"`` (default): decide PER CONVOLUTION PROBLEM — the first time a
    (direction, batch, shape) is seen it runs three times on its real operands, and only a problem whose results
    differ in any bit is run with deterministic kernels from then on (``dorpatch_amd/libconv.py``; with
    dorpatch_amd's own ResNetV2; for any other classifier the first micro-batch of each row count runs twice and the
    global flag is switched on only if the two gradients differ in a bit, as in round 2).
    With the table-routed 1x1 convolutions and the fixed-order reductions of every HIP kernel, identical inputs
    then give identical bits.  The reference sets ``cudnn.benchmark = True`` (``utils.py:17``) and is not run-to-run
    reproducible on a GPU.
    ``skip_satisfied`` (default False = the reference's behaviour, every sample is back-propagated, ``attack.py:247``):
    opt-in — the backward pass runs only over the EOT samples whose CW hinge is still active (a satisfied sample's
    gradient is exactly zero), see ``HotLoop._fb_taped``.  Same gradients to ~1e-6 of their scale (the compacted
    backward batches take other library kernels), up to 1.9x fewer ms per step once most samples meet their margin
    (``profiles/r02t_*``); costs one extra host sync per step and, the first time a new backward batch size occurs,
    MIOpen's one-off kernel loading for that size.
    """

    def __init__(self, micro_batch=512, process_group=None, verbose=True, deterministic="auto", skip_satisfied=False,
                 streams=None):
        self.micro_batch = int(micro_batch)
        # round 5: the step's micro-batches (and the failure sweep's forwards) are independent (disjoint images / masks);
        # enqueued round-robin on this many HIP streams the device overlaps one micro-batch's HBM-bound kernels (GroupNorm,
        # pooling) and launch tails with another's matrix-core kernels.  Results do not depend on the schedule (no atomics
        # anywhere; every micro-batch writes its own rows).  Default 2: 393 -> 369 ms per configs[1] step, 4 streams 373
        # (profiles/r05g_*); 1 = one stream, as before round 5.  Costs one more micro-batch of live activations.
        self.streams = int(os.environ.get("DORPATCH_STREAMS", "2")) if streams is None else int(streams)
        self.skip_satisfied = bool(skip_satisfied)
        if not (deterministic is True or deterministic is False or deterministic == "auto"):
            raise ValueError("deterministic must be True, False or 'auto' (got %r)" % (deterministic,))
        self.deterministic = deterministic
        self.pg = process_group
        self.verbose = verbose
        self.criterion = None
        self.last_run = None
        self._bool_universe = None        # (identity of a bool mask universe, its rectangle table) for collect_failure

    # ------------------------------------------------------------------ distributed helpers
    def _world(self):
        return dp_dist.world_rank(self.pg)

    def _log(self, *a):
        if self.verbose and self._world()[1] == 0:
            print(*a)

    # ------------------------------------------------------------------ public API
    def generate(self, model, x, patch_budget, n_classes, save_dir, batch_id, y=None, targeted=False,
                 lr=1e-2, confidence=1e-1, clip_min=0, clip_max=1, max_iterations=5000, basic_unit=7,
                 selection='topk', dropout=2, sampling_size=128, density=1e-3, structured=1e-3, eps=4.,
                 dual=False, **kwargs):
        """Same contract as the reference (``attack.py:51-361``): returns
        ``(adv_mask, adv_pattern)``, (B,1,H,W) in {0,1} and (B,3,H,W) in [0,1].

        Recognised extras in ``kwargs`` (``num_patch`` is accepted and ignored exactly
        like the reference, ``main.py:132`` / ``attack.py:53``): ``init_mask``,
        ``init_pattern`` (override the ``torch.rand`` init), ``rngs`` (one legacy
        ``np.random.RandomState`` per image), ``step_hook`` (callable receiving a dict
        of per-step internals — used by the parity tests), ``switch_iteration`` (500),
        ``failure_refresh`` (100), ``failure_sampling_start`` (1000), ``log_every`` (20), ``stem_split`` (False; True: with
        dorpatch_amd's own ResNetV2 the stem's input gradient and the S-reduction run as one kernel — bit-identical,
        measured slightly slower), ``skip_satisfied`` (default: the constructor's, False — see ``HotLoop._fb_taped``), ``skip_min_fraction`` (0.2: the
        selected-sample backward compacts a group of micro-batches only when at least this fraction of its samples is
        skippable; below it every sample is back-propagated in place),
        ``trace`` (default: the environment's ``DORPATCH_TRACE``; ``1`` = roctx ranges
        around the step's phases, ``"log"`` = phase names collected in ``last_run.phases.log`` — see ``_Phases``),
        ``retire`` (True: an image that has early-stopped in the current stage — the reference ``break``s there,
        ``attack.py:311-316`` — leaves the batch: only the images still running are occluded, forwarded, back-propagated
        and swept by ``collect_failure``, so a batch costs the SUM of its images' iterations instead of B x the slowest
        image's; False: the round-3 behaviour, finished images keep riding along with lr = 0),
        ``tape_tabs`` (micro-batches whose activations one backward may draw from, default: what fits in half of the free
        HBM, at most 8), ``backward_ladder`` (batch sizes the selected-sample backward may use), ``placement``
        (EXTENSION, not in the reference: e.g. ``dorpatch_amd.placement.RandomAffine()`` — every EOT sample sees the
        patch under its own random affine placement; ``None`` = the reference's identity placement).
        """
        run = HotLoop(self, model, x, patch_budget, n_classes, save_dir, batch_id, y, targeted, lr,
                      confidence, clip_min, clip_max, max_iterations, basic_unit, selection, dropout,
                      sampling_size, density, structured, eps, dual, kwargs)
        self.last_run = run
        try:
            return run.run()
        finally:
            run.close()

    # ------------------------------------------------------------------ attack.py:363-382
    def patch_selection(self, mask, patch_budget, basic_unit=7, selection='topk'):
        """Importance map -> binary patch mask: window-sum per ``basic_unit`` cell, keep the
        top ``floor(H*W*budget/unit^2)`` cells with positive sum, upsample back.

        Runs once per image (not on the hot path): plain torch.  Images whose side is not
        a multiple of ``basic_unit`` (e.g. 384) get the uncovered border zero-padded — the
        reference cannot handle them at all."""
        B, _, H, W = mask.shape
        ncy, ncx = (H - basic_unit) // basic_unit + 1, (W - basic_unit) // basic_unit + 1
        body = mask[:, 0, :ncy * basic_unit, :ncx * basic_unit]
        group_importance = body.reshape(B, ncy, basic_unit, ncx, basic_unit).sum(dim=(2, 4))
        num_group = int(np.floor((H * W * patch_budget) / (basic_unit ** 2)))
        if selection != 'topk':
            raise NotImplementedError("selection must be 'topk' (the reference implements nothing else)")
        flat = group_importance.reshape(B, -1)
        value_topk, idx_topk = flat.topk(num_group)
        selected = torch.zeros_like(flat)
        selected.scatter_(1, idx_topk, (value_topk > 0).to(flat.dtype))
        cells = selected.view(B, 1, ncy, ncx)
        up = cells.repeat_interleave(basic_unit, dim=2).repeat_interleave(basic_unit, dim=3)
        out = torch.zeros((B, 1, H, W), dtype=mask.dtype, device=mask.device)
        out[:, :, :ncy * basic_unit, :ncx * basic_unit] = up
        return out

    # ------------------------------------------------------------------ attack.py:384-406
    def collect_failure(self, adv_x, y, mask_set_universe, targeted, model, batch_size=128,
                        transforms=None):
        """Forward-only sweep of every mask of the universe (``attack.py:384-406``); returns the
        ascending list of mask indices on which the attack fails — for B > 1 images the union over
        the images, like the reference (``failed_idx.unique()``, ``attack.py:403``).

        ``mask_set_universe`` is a device rectangle table (``masks.universe_rects`` / ``MaskWindow(...).rects``) or the
        reference's (n,1,H,W) bool tensor (``attack.py:83-85``), which is converted once into a table with exactly the
        same occluded pixels (``masks.bool_to_rects``; cached per tensor) — masks are never materialised on the device
        path; a bool mask that is not a union of two windows is refused with a ValueError.  ``y`` may be (B,) or the
        reference's expanded (B*sampling_size,) labels (``attack.py:98, 399``).  ``transforms``
        (``attack.py:395-396``), if given, is applied to the occluded images in [0,1] right before
        ``model``, exactly where the reference applies it."""
        ops.require_gpu(adv_x, "DorPatch.collect_failure (`adv_x`)")
        table = mask_set_universe
        if isinstance(table, torch.Tensor) and table.dtype == torch.bool and table.dim() in (3, 4):
            key = (table.data_ptr(), tuple(table.shape), str(table.device), table._version)
            if self._bool_universe is None or self._bool_universe[0] != key:
                self._bool_universe = (key, ops.upload_table(masks.bool_to_rects(table, DP_MAX_RECTS),
                                                             adv_x.device))
            table = self._bool_universe[1]
        if not (isinstance(table, torch.Tensor) and table.dtype == torch.int32 and table.dim() == 3
                and table.shape[2] == 4):
            raise TypeError("collect_failure needs the mask universe as an (n, R, 4) int32 rectangle table "
                            "(dorpatch_amd.masks.universe_rects(H, dropout) / MaskWindow(...).rects uploaded with "
                            "ops.upload_table) or as the reference's (n,1,H,W) bool tensor")
        B = adv_x.shape[0]
        y_img = y.detach().to(adv_x.device).long().reshape(B, -1)[:, 0].contiguous()
        if transforms is None:
            net, norm = _unwrap_model(model)
        else:                      # the hook sees un-normalised occluded images, then the model as the caller built it
            net, norm = (lambda t: model(transforms(t))), None
        lists = _collect_failure(net, norm, adv_x.detach().contiguous().float(), y_img, table,
                                 targeted, batch_size, pg=self.pg)
        union = sorted(set().union(*[set(l) for l in lists]))
        self._log(">> %d failures collected!" % len(union))
        return union


# ======================================================================================
# implementation
# ======================================================================================

def _every_conv_under_policy(net):
    """True iff every library convolution of ``net`` goes through ``libconv`` / ``conv1x1`` (per-problem determinism):
    dorpatch_amd's own ResNetV2 with every StdConv2d folded and frozen, and no other convolution module except the
    head (which ``ResNetV2.forward`` routes through libconv itself when frozen)."""
    from . import resnetv2
    if not isinstance(net, resnetv2.ResNetV2):
        return False
    for m in net.modules():
        if isinstance(m, resnetv2.StdConv2d):
            if not m.folded or m.weight.requires_grad:
                return False
        elif isinstance(m, torch.nn.Conv2d) and m is not net.head.fc:
            return False
    return not net.head.fc.weight.requires_grad


def tape_z_channels(net):
    return net.stem.conv.out_channels


def _dp_norm(norm):
    return ops.RAW_NORM if norm is None else ops.make_norm(norm[0], norm[1], 0.5)


def sweep_plan(B, rows):
    """How the failure sweep cuts B images x n masks into forwards of ``rows`` samples: groups of g images (g a power of
    two) x rows // g masks.  The row count of a forward is then the same whatever B is — when images retire from a batch
    the sweep keeps presenting the library convolutions the shapes they have already seen (a new batch size costs MIOpen
    a kernel lookup per layer and, under deterministic="auto", three probe runs per convolution problem: measured 2.4x on
    the sweeps of a 4-image attack whose batch shrank to 3, 2, 1 images, profiles/r04a_bench_whole_attack.json).
    g is the largest power of two that fits the remaining images AND divides ``rows`` (rows = 100: groups of 4 x 25 masks,
    not 64 x 1), so the product g * (rows // g) is ``rows`` for every group."""
    plan, b = [], 0
    rows = max(1, int(rows))
    while b < B:
        g = 1
        while g * 2 <= min(B - b, rows) and rows % (g * 2) == 0:     # g divides rows: g * (rows // g) == rows exactly
            g *= 2
        plan.append((b, b + g, max(1, rows // g)))
        b += g
    return plan


def _side_streams(pool, n, dev):
    """``n`` side streams on ``dev`` from the caller's ``pool`` (a list that keeps them across calls), all waiting for the
    current stream; -> (streams, current stream)."""
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=dev))
    main = torch.cuda.current_stream(dev)
    for st in pool[:n]:
        st.wait_stream(main)
    return pool[:n], main


@torch.no_grad()
def _collect_failure(net, norm, adv_x, y_img, table, targeted_flags, batch_size, pg=None, plan=None, streams=1,
                     stream_pool=None):
    """Per-image failed-mask lists (attack.py:384-406).  ``y_img`` (B,) int64 device tensor,
    ``targeted_flags`` bool or (B,) bool array.  Ranks of ``pg`` each sweep a slice of the
    universe and exchange a (B, n_mask) failure bitmap.  ``plan``: [(first image, end image, masks per forward)]
    (default: all images together, ``batch_size`` masks per forward — the reference's loop).  With a plan the short last
    chunk of a group is padded to the full row count (its last mask repeated, the extra predictions dropped): a new
    batch size costs MIOpen a kernel lookup per layer + the determinism probes — 10 s per new size measured
    (profiles/r04b_bench_whole_attack.json: 32 s of sweeps instead of 10 when a batch shrank to 3 and 2 images)."""
    B, _, H, W = adv_x.shape
    n_mask = table.shape[0]
    dev = adv_x.device
    dn = _dp_norm(norm)
    world, rank = dp_dist.world_rank(pg)
    lo, hi = dp_dist.mask_bounds(n_mask, world, rank)
    fail = torch.zeros((B, n_mask), dtype=torch.int32, device=dev)
    tflag = torch.as_tensor(np.broadcast_to(np.asarray(targeted_flags, dtype=bool), (B,)).copy(), device=dev)
    pad_tail = plan is not None       # the reference's own loop (public collect_failure) runs its short last batch as is
    if plan is None:
        plan = [(0, B, max(1, int(batch_size)))]
    # the forwards are independent (disjoint slices of `fail`): round-robin over `streams` HIP streams, like the step's
    # micro-batches (DorPatch(streams=...)), so one forward's HBM-bound kernels overlap another's matrix-core kernels
    n_str = int(streams) if dev.type == "cuda" and streams and streams > 1 else 1
    # everything the side streams only READ is produced on the calling stream BEFORE they fork from it: the
    # labels as int32 (one conversion, the groups take views — a per-group `.to()` after the fork was ordered behind nothing
    # the side streams wait for, and its block could be re-used by the next group's while queued forwards still read it),
    # and the packed weights of the frozen convolutions (libconv builds them lazily on whatever stream asks first).
    y32 = y_img.view(B, 1).to(torch.int32)
    if dev.type == "cuda":
        from . import libconv
        libconv.prepack(net)
    if n_str > 1:
        side, main = _side_streams([] if stream_pool is None else stream_pool, n_str, dev)
    k = 0
    for b0, b1, chunk in plan:
        chunk = max(1, min(chunk, hi - lo))          # a universe (slice) smaller than one forward: no padding beyond it
        g = b1 - b0
        xg, yg, tg = adv_x[b0:b1], y32[b0:b1], tflag[b0:b1].view(g, 1)
        for j0 in range(lo, hi, chunk):
            j1 = min(hi, j0 + chunk)
            with (torch.cuda.stream(side[k % n_str]) if n_str > 1 else contextlib.nullcontext()):
                idx = torch.arange(j0, j1, dtype=torch.int32, device=dev)
                if pad_tail and j1 - j0 < chunk:      # the last, short chunk: repeat its last mask so the classifier sees
                    idx = torch.cat([idx, idx[-1:].expand(chunk - (j1 - j0))])     # the row count it has seen all along
                inp = ops.apply_fwd(xg, table, idx, None, dn)
                pred = ops.argmax(net(inp).float().contiguous()).view(g, idx.numel())[:, :j1 - j0]
                same = pred == yg
                # untargeted: still classified as y => failure; targeted: not (yet) the target => failure
                fail[b0:b1, j0:j1] = torch.where(tg, ~same, same).to(torch.int32)
                del inp, pred, same
            k += 1
    if n_str > 1:
        for st in side:
            main.wait_stream(st)
    dp_dist.allreduce_max_(fail, pg)
    fail_np = fail.cpu().numpy().astype(bool)
    return [np.nonzero(fail_np[b])[0].tolist() for b in range(B)]


def draw_indices(rng, failed_idxs, n_form_failure, sampling_size, sampling_choices):
    """Host mask sampling, attack.py:193-204: ``n_form_failure`` indices from the failure list
    first, the rest from the whole universe, each without replacement, through the legacy numpy
    generator ``rng`` (``np.random`` itself for B = 1, so the global stream advances exactly as
    in the reference)."""
    n_form_universe = sampling_size - n_form_failure
    parts = []
    if n_form_failure > 0:
        parts.append(rng.choice(failed_idxs, n_form_failure, replace=False))
    if n_form_universe > 0:
        parts.append(rng.choice(sampling_choices, n_form_universe, replace=False))
    return np.concatenate(parts)


class HotLoop(object):
    """State + one-step function of the EOT optimisation loop.  ``DorPatch.generate``
    drives it through both stages; ``bench.py`` drives ``step`` directly, so the
    benchmark times exactly the code path ``generate`` executes."""

    def __init__(self, *args):
        """Process-wide settings changed for the run (frozen parameters, TunableOp scope, libconv.MODE,
        cudnn.deterministic) are undone by close(); if construction itself raises, close() runs before the exception
        leaves, so a failed generate() leaves the caller's process as it found it."""
        from . import libconv
        self._frozen = []
        self._tuned_scope = False
        self._cudnn_det = torch.backends.cudnn.deterministic
        self._libconv_mode = libconv.MODE
        try:
            self._init(*args)
        except BaseException:
            self.close()
            raise

    def _init(self, owner, model, x, patch_budget, n_classes, save_dir, batch_id, y, targeted, lr,
              confidence, clip_min, clip_max, max_iterations, basic_unit, selection, dropout,
              sampling_size, density, structured, eps, dual, extras):
        ops.require_gpu(x, "DorPatch.generate (`x`)")
        if dropout not in (1, 2):
            raise ValueError("dropout must be 1 or 2 (reference attack.py:25-31 builds no mask set otherwise)")
        self.o = owner
        self.world, self.rank = owner._world()
        self.dev = x.device
        self.x = x.detach().contiguous().float()
        self.B, _, self.H, self.W = self.x.shape
        B, H, W, dev = self.B, self.H, self.W, self.dev
        self.patch_budget, self.n_classes = patch_budget, n_classes
        self.save_dir, self.batch_id = save_dir, batch_id
        self.lr, self.confidence = lr, confidence
        self.clip_min, self.clip_max = float(clip_min), float(clip_max)
        self.max_iterations, self.unit, self.selection = max_iterations, basic_unit, selection
        self.density, self.eps, self.dual = float(density), float(eps), bool(dual)
        self.win = int(W // 8)                                        # attack.py:77
        self.switch_iteration = extras.get("switch_iteration", 500)   # attack.py:169
        self.failure_refresh = extras.get("failure_refresh", 100)     # attack.py:187
        self.failure_start = extras.get("failure_sampling_start", 1000)  # attack.py:193
        self.log_every = extras.get("log_every", 20)                  # attack.py:318
        self.step_hook = extras.get("step_hook", None)
        self.retire = bool(extras.get("retire", True))
        # EXTENSION (dorpatch_amd/placement.py; absent from the reference, which places the patch at identity):
        # an object with .draw(rng, S, H, W) -> (S,2,3) output->source pixel maps, one draw per image per step
        self.placement = extras.get("placement", None)

        self.net, self.norm = _unwrap_model(model)
        self.dn = _dp_norm(self.norm)
        self._frozen = [(p, p.requires_grad) for p in self.net.parameters()]
        for p, _ in self._frozen:
            p.requires_grad_(False)
        # Tuned GEMM solutions for the table-routed 1x1 convolutions of dorpatch_amd's own ResNetV2: one verdict per
        # process (rank 0's numeric self-test, agreed by all ranks), in effect only between here and close().
        from . import conv1x1, resnetv2
        if any(isinstance(m, resnetv2.StdConv2d) for m in self.net.modules()):
            self._tuned_scope = conv1x1.activate(owner.pg, self.dev.type == "cuda")
        self.gemm_solutions = conv1x1.report_tuned()
        # Run-to-run bit reproducibility of the library convolutions (class docstring): True = the global flag for the
        # whole run; "auto" = per (direction, batch, shape) problem, probed at first use (dorpatch_amd/libconv.py).
        from . import libconv
        if owner.deterministic is True:
            torch.backends.cudnn.deterministic = True
        elif owner.deterministic == "auto" and not torch.backends.cudnn.deterministic:
            libconv.MODE = "auto"
        self._det_rows_merged = set()        # micro-batch row counts whose per-problem decisions the ranks have agreed on
        # The per-problem policy only reaches the library calls of dorpatch_amd's own ResNetV2.  Any other classifier is
        # checked as a whole, as in round 2: the first micro-batch of each row count runs twice, and if the two input
        # gradients differ in any bit the global flag is switched on for the rest of the run (restored by close()).
        # The policy reaches a convolution only through a FOLDED, frozen StdConv2d (resnetv2.py:49-57): an un-folded
        # ResNetV2 handed straight to generate() runs plain F.conv2d and is checked as a whole like any other net.
        self._det_whole_net = (owner.deterministic == "auto" and not torch.backends.cudnn.deterministic
                               and not _every_conv_under_policy(self.net))
        self._det_rows_checked = set()

        owner.criterion = CW_loss(n_classes, targeted, confidence)     # attack.py:57
        # attack.py:59-60 — CPU generator, mask first then pattern
        init_mask = extras.get("init_mask")
        init_pattern = extras.get("init_pattern")
        adv_mask = torch.rand([B, 1, H, W]) if init_mask is None else init_mask.detach().cpu().float()
        adv_pattern = torch.rand((B, 3, H, W)) if init_pattern is None else init_pattern.detach().cpu().float()
        self.adv_mask = adv_mask.to(dev, copy=True).contiguous()        # never alias the caller's init tensors
        self.adv_pattern = adv_pattern.to(dev, copy=True).contiguous()
        dp_dist.broadcast_(self.adv_mask, owner.pg)
        dp_dist.broadcast_(self.adv_pattern, owner.pg)
        self.best_mask = torch.zeros_like(self.adv_mask)               # attack.py:63-64
        self.best_pattern = torch.zeros_like(self.adv_pattern)

        if y is None:                                                  # attack.py:67-69
            with torch.no_grad():
                y = ops.argmax(self._forward_plain(self.x)).long()
            dp_dist.broadcast_(y, owner.pg)      # a near-tie argmax must not differ between replicas
        y = y.detach().to(dev).long().view(-1)
        assert y.numel() == B
        self.y = y.contiguous()

        # attack.py:83-85 — the mask universe, as a rectangle table
        table_np = masks.universe_rects(W, dropout)
        self.n_mask = table_np.shape[0]
        self.table = ops.upload_table(table_np, dev)
        self.S = min(int(sampling_size), self.n_mask)                  # attack.py:92-94
        self.s_lo, self.s_hi = dp_dist.shard_bounds(self.S, self.world, self.rank)
        self.S_local = self.s_hi - self.s_lo
        self.sampling_choices = np.arange(self.n_mask)                 # attack.py:95
        self.lv_x = ops.local_variance(self.x)                         # attack.py:100

        y_host = self.y.cpu().numpy()
        self.img = [_ImageState(lr, structured, 1e-5, targeted, y_host[b]) for b in range(B)]  # :87
        # Mask draws (attack.py:193-204) happen on EVERY rank from the same generator state, so the ranks
        # sample identical indices without exchanging them: the state is synchronised from rank 0 once,
        # here; afterwards the inputs of every draw (failure lists, step number) are themselves identical
        # on all ranks.  A per-step checksum rides in the step's all-reduce and is verified on the host.
        # User-supplied `rngs` must already be identical on every rank.
        rngs = extras.get("rngs")
        if rngs is None:
            if B == 1:
                if self.world > 1:
                    np.random.set_state(dp_dist.broadcast_object(np.random.get_state(), owner.pg))
                rngs = [np.random]          # the reference's global legacy stream, bit for bit
            else:
                seeds = dp_dist.broadcast_object(np.random.randint(0, 2 ** 31 - 1, size=B), owner.pg)
                rngs = [np.random.RandomState(int(s)) for s in seeds]
        self.rngs = list(rngs)
        self.idx_np = np.zeros((B, self.S), dtype=np.int64)
        self.idx2_np = np.zeros((B, self.S), dtype=np.int64) if self.dual else None
        self.n_fail = [0] * B

        # Device scratch reused every step.  ONE buffer carries everything that leaves the step:
        #   [ g_adv (B*3*H*W) | per rank: loss_adv slab (B*S_local), pred slab (B*S_local), draw checksum (1) |
        #     loss_struc (B), group_lasso (B), density (B) ]
        # With N > 1 ranks the first two regions are all-reduced (SUM) in one call: every rank fills only its
        # own slab (the others are zero, x + 0 == x exactly), so the sum IS the gather of the loss / prediction
        # columns — one collective per step, the patch-gradient all-reduce.  Everything after
        # g_adv goes to the host in one copy: the step's only device->host synchronisation.
        n_g, n_slab = B * 3 * H * W, B * self.S_local
        self._n_g, self._n_slab = n_g, n_slab
        self._n_tail = self.world * (2 * n_slab + 1)
        self._streams = []                  # side streams of the micro-batch loop (DorPatch(streams=...) > 1)
        if dev.type == "cuda":              # packed weights of the frozen convolutions: built here, on the step's own stream,
            from . import libconv           # not lazily by whichever side stream's micro-batch asks first
            libconv.prepack(self.net)
        self._comm = torch.zeros((n_g + self._n_tail + 3 * B,), dtype=torch.float32, device=dev)
        self.g_adv = self._comm[:n_g].view(B, 3, H, W)
        self._tail = self._comm[n_g:n_g + self._n_tail].view(self.world, 2 * n_slab + 1)
        own = self._tail[self.rank]
        self._own_loss, self._own_pred, self._own_chk = own[:n_slab], own[n_slab:2 * n_slab], own[2 * n_slab:]
        self._reg = self._comm[n_g + self._n_tail:]                       # loss_struc | group_lasso | density
        self.pred = torch.zeros((n_slab,), dtype=torch.int32, device=dev)
        self.pred_host = np.zeros((B, self.S), dtype=np.int64)           # argmax of the last step's logits, all ranks' columns
        self.adv_x = torch.empty_like(self.x)
        self.stage = 0
        self.samples_done = 0
        self.swept_images = 0                 # images x sweeps of collect_failure so far
        self.timing = {"sweeps_s": 0.0, "sweeps": 0}      # filled by run(): stage{0,1}_s, stage{0,1}_steps, per-image steps
        self.kernel_events = None
        self._conv_shared = False
        # Optional (extras stem_split=True), dorpatch_amd's own ResNetV2 with a frozen stem only: the backward stops at
        # the stem-convolution OUTPUT and dp_stem_dgrad_reduce turns that gradient into the S-reduced patch gradient in
        # one launch, so the per-sample (N,3,H,W) input gradient is never written.  Bit-identical to the default
        # (autograd down to the masked input: dp_stem_dgrad, then dp_apply_bwd) and NOT faster: the 2.4 GB of HBM
        # traffic it saves per step (0.4 ms) cost less than the fused kernel's lower occupancy does (measured on one
        # box: 432.3 vs 429.8 ms/step, profiles/r02h_*), so it stays off by default.
        probe = getattr(self.net, "stem_split_supported", None)
        self._stem_split = bool(probe is not None and extras.get("stem_split", False) and self.placement is None
                                and probe(self.x))
        self.theta_np = None
        # The backward pass runs only over the EOT samples that still carry gradient (dorpatch_amd/taped.py): needs
        # dorpatch_amd's own frozen ResNetV2; any other classifier goes through autograd, all samples.
        from . import taped
        self._taped = (bool(extras.get("skip_satisfied", owner.skip_satisfied)) and taped.eligible(self.net)
                       and self.placement is None)        # the placement extension keeps the autograd path (untested there)
        self._tape_tabs = extras.get("tape_tabs")            # None: sized from free memory after the first micro-batch
        self._ladder_user = extras.get("backward_ladder")
        self._skip_min_fraction = float(extras.get("skip_min_fraction", 0.2))
        self.n_forward = self.n_active = self.n_backward = 0   # samples: forwarded / carrying gradient / back-propagated (incl. padding)
        self.phases = _Phases(extras.get("trace", os.environ.get("DORPATCH_TRACE", "0")), self.dev.type == "cuda")

    # ---------------------------------------------------------------- plumbing
    @property
    def deterministic_in_effect(self):
        """What run-to-run reproducibility of the library convolutions rests on in this run (bench.py records it)."""
        from . import libconv
        if torch.backends.cudnn.deterministic:
            return "on (torch.backends.cudnn.deterministic for the whole run)"
        if libconv.MODE == "auto":
            d = libconv.summary()
            what = "per problem: %d of %d probed (direction, batch, shape) problems forced to deterministic kernels" % (
                d["forced"], d["problems"])
            if self._det_whole_net:
                what += "; foreign classifier: first micro-batch of each row count verified bit-identical twice"
            return what
        return "off"

    def close(self):
        for p, flag in self._frozen:
            p.requires_grad_(flag)
        torch.backends.cudnn.deterministic = self._cudnn_det
        from . import libconv
        libconv.MODE = self._libconv_mode
        if self._tuned_scope:                    # TunableOp back to what the caller's process had
            from . import conv1x1
            conv1x1.deactivate()
            self._tuned_scope = False

    def _forward_plain(self, imgs):
        """model(imgs) for un-occluded images in [0,1] (NormModel applied if it was peeled)."""
        idx = torch.zeros((1,), dtype=torch.int32, device=self.dev)
        empty = torch.zeros((1, 1, 4), dtype=torch.int32, device=self.dev)
        inp = ops.apply_fwd(imgs.contiguous().float(), empty, idx, None, self.dn)
        return self.net(inp).float().contiguous()

    def _dir0(self):
        parts = self.save_dir.split('/')[:-1]                          # attack.py:103
        return os.path.join(*parts) if parts else "."

    def _flags(self, attr):
        return np.array([getattr(s, attr) for s in self.img], dtype=bool)

    def _dev_f32(self, values):
        return torch.as_tensor(np.asarray(values, dtype=np.float32), device=self.dev)

    def _dev_i32(self, values):
        return torch.as_tensor(np.asarray(values, dtype=np.int32), device=self.dev)

    # ---------------------------------------------------------------- attack.py:106-122
    def _set_target(self, b, preds):
        """Per-image ``set_target``: switch image b to a targeted attack on the class most
        of its currently mis-classified masked copies fall into.  Returns True if y changed."""
        st = self.img[b]
        preds = np.asarray(preds).reshape(-1)
        wrong = preds[preds != st.y]
        if wrong.size == 0:
            return False
        st.crit_targeted = True
        if wrong.size > 1:
            vals, counts = np.unique(wrong, return_counts=True)
            target = int(vals[np.argmax(counts)])   # torch.mode: smallest of the most frequent
        else:
            target = int(wrong[0])
        changed = target != st.y
        st.y = target
        return changed

    def _sync_labels(self):
        self.y = torch.as_tensor(np.array([s.y for s in self.img], dtype=np.int64), device=self.dev)

    # ---------------------------------------------------------------- stage control
    def run(self):
        o = self.o
        dir_0 = self._dir0()
        for stage in range(2):                                         # attack.py:124
            o._log('============= Stage %d =============' % stage)
            self.stage = stage
            for s in self.img:
                s.reset_stage()
                s.steps_in_stage = 0        # per stage, not per reset: the untargeted -> targeted switch resets mid-stage
            mpath = os.path.join(dir_0, "adv_mask_%d.pt" % self.batch_id)
            # every rank follows rank 0's view of the cache (skipping stage 0 on one rank only would deadlock)
            if stage == 0 and dp_dist.broadcast_object(os.path.exists(mpath), o.pg):   # attack.py:134-141
                self.best_mask = torch.load(mpath, map_location=self.dev).float().contiguous()
                self.best_pattern = torch.load(os.path.join(dir_0, "adv_pattern_%d.pt" % self.batch_id),
                                               map_location=self.dev).float().contiguous()
                continue
            t0 = self._clock()
            if stage == 1:
                self._enter_stage1()
            last_i = -1
            for i in range(self.max_iterations):                       # attack.py:167
                last_i = i
                if not self.step(i):
                    break
            self._finish_stage(stage, last_i, dir_0)
            self.timing["stage%d_s" % stage] = self._clock() - t0       # includes this stage's sweeps
            self.timing["stage%d_steps" % stage] = last_i + 1
            self.timing["stage%d_image_steps" % stage] = [st.steps_in_stage for st in self.img]
        return self.best_mask.clone(), self.best_pattern.clone()        # attack.py:361

    def _clock(self):
        """Wall clock with the device drained (stage / sweep accounting only: a handful of calls per run)."""
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)
        import time
        return time.perf_counter()

    def _enter_stage1(self):
        """attack.py:143-165."""
        with torch.no_grad():
            adv_x, _, _ = ops.blend(self.best_mask, self.best_pattern, self.x, self.eps)   # :148-149
            if not all(self._flags("flag_targeted")):                  # :151-155 (set_target(preds_adv, y))
                preds = dp_dist.broadcast_(ops.argmax(self._forward_plain(adv_x)), self.o.pg).cpu().numpy()
                for b, st in enumerate(self.img):
                    if not st.flag_targeted:
                        st.flag_targeted = True
                        self._set_target(b, preds[b:b + 1])
                self._sync_labels()
            self.best_pattern = adv_x.clone()                          # :158-159 merged content
            self.adv_mask = self.o.patch_selection(self.best_mask, self.patch_budget, self.unit,
                                                   self.selection).contiguous()             # :161-162
            self.best_mask = self.adv_mask.clone()                     # :163
            self.adv_pattern = adv_x.clone()                           # :165

    def _finish_stage(self, stage, last_i, dir_0):
        """attack.py:344-359."""
        for b, st in enumerate(self.img):
            if np.isinf(st.loss_best):                                 # :344-346 no best saved: return last
                self.best_mask[b] = self.adv_mask[b]
                self.best_pattern[b] = self.adv_pattern[b]
        if stage == 0:
            if self.rank == 0:                                         # :351-356 stage-0 cache (pattern first:
                os.makedirs(dir_0, exist_ok=True)                      # the mask file's presence means "complete")
                for name, t in (("adv_pattern_%d.pt", self.best_pattern), ("adv_mask_%d.pt", self.best_mask)):
                    path = os.path.join(dir_0, name % self.batch_id)
                    torch.save(t, path + ".tmp")
                    os.replace(path + ".tmp", path)
            if not all(self._flags("flag_targeted")) and last_i >= 0:  # :357-359 (set_target(preds_adv, y))
                preds = self.pred_host
                for b, st in enumerate(self.img):
                    if not st.flag_targeted:
                        self._set_target(b, preds[b])
                self._sync_labels()

    def _gather_pred(self):
        """(B, S) argmax of the last step's logits over all ranks' samples.  A host array that arrived with the
        step's statistics: no collective, so callers need not be rank-symmetric."""
        return self.pred_host

    # ---------------------------------------------------------------- one optimisation step
    def _refresh_failures(self):
        # forwards of micro_batch rows (sweep_plan): same conv shapes as the hot loop (no extra MIOpen solver
        # searches), bounded activation memory, and the same shapes whatever the number of images still running
        act = self._running()
        if not act:
            return
        t0 = self._clock()
        adv_x, y, flags = self.adv_x, self.y, self._flags("flag_targeted")
        if len(act) < self.B:          # finished images are not swept (their failure lists are never read again)
            sel = torch.as_tensor(act, dtype=torch.int64, device=self.dev)
            adv_x, y, flags = adv_x.index_select(0, sel), y.index_select(0, sel), flags[act]
        lists = _collect_failure(self.net, self.norm, adv_x.contiguous(), y, self.table, flags, None, pg=self.o.pg,
                                 plan=sweep_plan(len(act), self.o.micro_batch), streams=self.o.streams,
                                 stream_pool=self._streams)
        for b, l in zip(act, lists):
            if self.img[b].active:
                self.img[b].failed_idxs = l
        self.swept_images += len(act)
        dt = self._clock() - t0
        self.timing["sweeps_s"] += dt
        self.timing["sweeps"] += 1
        key = "stage%d_sweeps_s" % self.stage
        self.timing[key] = self.timing.get(key, 0.0) + dt
        self.o._log(">> %d failures collected!" % sum(len(l) for l in lists))

    def _running(self):
        """Images the EOT pass and the failure sweep cover: all of them, or (``retire``) those still active."""
        if not self.retire:
            return list(range(self.B))
        return [b for b, st in enumerate(self.img) if st.active]

    def _draw(self, i):
        for b, st in enumerate(self.img):
            if not st.active:
                continue
            self.n_fail[b] = st.n_from_failure(i, self.S, self.failure_start)
            self.idx_np[b] = draw_indices(self.rngs[b], st.failed_idxs, self.n_fail[b], self.S,
                                          self.sampling_choices)
            if self.dual:                                              # attack.py:208-216
                self.idx2_np[b] = draw_indices(self.rngs[b], st.failed_idxs, self.n_fail[b], self.S,
                                               self.sampling_choices)
            if self.placement is not None:                             # extension: after the reference's own draws
                if self.theta_np is None:
                    self.theta_np = np.zeros((self.B, self.S, 2, 3), dtype=np.float32)
                self.theta_np[b] = self.placement.draw(self.rngs[b], self.S, self.H, self.W)

    def step(self, i):
        """One pass of attack.py:169-342 over the whole batch.  Returns False once every
        image has early-stopped in this stage."""
        o, B, S, Sl, dev, stage = self.o, self.B, self.S, self.S_local, self.dev, self.stage

        # --- attack.py:169-182: untargeted -> targeted switch
        if stage == 0 and i == self.switch_iteration and not all(self._flags("flag_targeted")):
            preds = self.pred_host
            for b, st in enumerate(self.img):
                # an image that early-stopped before the switch has LEFT stage 0, as the reference's one-image loop does at
                # attack.py:311-316 (no switch is ever reached there): it must not be revived with loss_best = inf — its
                # saved best mask / pattern would be overwritten; _finish_stage sets its target at the stage's end
                if st.flag_targeted or not st.active:
                    continue
                st.flag_targeted = True
                if self._set_target(b, preds[b]):
                    o._log(">> switch to targeted attack to category {:3d} at iteration: {:4d}".format(st.y, i))
                st.reset_stage()
            self._sync_labels()
            self._refresh_failures()        # on the previous step's adv_x, like the reference

        # --- a-2: utils.clip + adv_x (attack.py:184-185)
        ph = self.phases
        ph.mark("dp:blend")
        _, scale, l2 = ops.blend(self.adv_mask, self.adv_pattern, self.x, self.eps, out=self.adv_x)
        if i % self.failure_refresh == 0:                              # attack.py:187-190
            ph.mark("dp:collect_failure")
            self._refresh_failures()
        ph.mark("dp:sample")

        # --- a-3: mask sampling on the host (same RNG calls as the reference)
        # (every rank draws the full (B, S) index set from identical generator state and keeps its own S-slice)
        self._draw(i)
        idx = self._dev_i32(self.idx_np[:, self.s_lo:self.s_hi])
        idx2 = self._dev_i32(self.idx2_np[:, self.s_lo:self.s_hi]) if self.dual else None

        structured_pre = [st.structured for st in self.img]
        coeff_pre = [st.coeff_group_lasso for st in self.img]
        crit_flags = self._dev_i32(self._flags("crit_targeted"))

        # --- a-5 / a-6 forward terms (identical on every rank: same inputs, fixed-order reductions)
        ph.mark("dp:regularisers")
        if self.world > 1:
            self._tail.zero_()          # the other ranks' slabs must be 0 going into the SUM
            self._own_chk.fill_(self._draw_checksum())
        ops.struct_loss(self.adv_x, self.lv_x, out=self._reg[:B])
        cell = wsum = None
        if stage == 0:
            cell, wsum, _, _ = ops.mask_stats(self.adv_mask, self.unit, self.win,
                                              gl_out=self._reg[B:2 * B], dens_out=self._reg[2 * B:3 * B])

        # --- a-4, a-8, a-7: occlude -> frozen backbone fwd/bwd -> CW loss, in micro-batches
        ph.mark("dp:eot_fwd_bwd")
        self._eot_forward_backward(idx, idx2, crit_flags)
        if self.world > 1 and not self._conv_shared:    # conv1x1 "auto" only: every replica adopts rank 0's routes
            from . import conv1x1
            conv1x1.share_choices(o.pg)
            self._conv_shared = True
        # THE collective of the step: patch gradient (602 112 B per image @224) + the loss / prediction slabs
        if self.world > 1:
            ph.mark("dp:allreduce")
        dp_dist.allreduce_sum_(self._comm[:self._n_g + self._n_tail], o.pg)

        # --- the one device->host sync of the step
        ph.mark("dp:sync+bookkeeping")
        loss_adv_np, loss_struc_np, gl_np, dens_np = self._gather_stats()

        # --- a-9: per-image bookkeeping (attack.py:249-316)
        save_best = np.zeros(B, dtype=np.int32)
        lr_now = np.zeros(B, dtype=np.float32)
        for b, st in enumerate(self.img):
            if not st.active:
                continue
            loss_target = gl_np[b] if stage == 0 else loss_struc_np[b]
            save, stop = st.step(i, stage, loss_adv_np[b], loss_target, self.idx_np[b], self.n_fail[b])
            save_best[b] = 1 if save else 0
            if stop:                                                    # attack.py:311-316
                o._log("early stop at iteration: {:4d}".format(i))
                st.active = False
                if np.isinf(st.loss_best):
                    save_best[b] = 1     # best := current (pre-update) parameters, attack.py:313-315
            else:
                lr_now[b] = st.lr_current

        if i % self.log_every == 0 and o.verbose and self.rank == 0:   # attack.py:318-330 (host data only: no collective)
            preds = self.pred_host
            y_host = np.array([s.y for s in self.img]).reshape(B, 1)
            acc = float((preds == y_host).mean()) * 100
            total = loss_adv_np.mean(1) + np.asarray(structured_pre, dtype=np.float32) * loss_struc_np
            if stage == 0:
                total = total + np.float32(self.density) * dens_np + np.asarray(coeff_pre, np.float32) * gl_np
            msg = "iteration: {:4d}, accuracy: {:.2f}, loss: {:.2f}, adv: {:.2f}, l2 norm: {:.2f}, structural: {:.2f}".format(
                i, acc, float(total.mean()), float(loss_adv_np.mean()),
                float(torch.minimum(l2, torch.full_like(l2, self.eps)).mean().item()), float(loss_struc_np.mean()))
            if stage == 0:
                msg += ", group lasso: {:.2f}, density: {:.2f}".format(float(gl_np.mean()), float(dens_np.mean()))
            print(msg)

        if self.step_hook is not None:
            gp, gm = ops.project_update(
                self.x, self.adv_x, self.lv_x, self.g_adv, scale, self._dev_f32(structured_pre),
                self.adv_pattern, self.adv_mask, stage=stage, coeff_gl=self._dev_f32(coeff_pre),
                cell_sumsq=cell, win_sum=wsum, unit=self.unit, win=self.win, density=self.density,
                do_update=False, want_grads=True)
            self.step_hook(dict(i=i, stage=stage, idx=self.idx_np.copy(), adv_x=self.adv_x, scale=scale,
                                theta=None if self.theta_np is None else self.theta_np.copy(),
                                loss_adv=loss_adv_np, loss_struc=loss_struc_np, group_lasso=gl_np,
                                density=dens_np, g_adv=self.g_adv, grad_pattern=gp, grad_mask=gm,
                                mask=self.adv_mask, pattern=self.adv_pattern, lr=lr_now.copy(),
                                save_best=save_best.copy(), states=self.img))

        # --- a-2 backward + a-5/a-6 gradients + signed update (attack.py:247, 333-342)
        ph.mark("dp:project_update")
        ops.project_update(
            self.x, self.adv_x, self.lv_x, self.g_adv, scale, self._dev_f32(structured_pre),
            self.adv_pattern, self.adv_mask, stage=stage, lr=self._dev_f32(lr_now),
            coeff_gl=self._dev_f32(coeff_pre), cell_sumsq=cell, win_sum=wsum, unit=self.unit,
            win=self.win, density=self.density, clip_min=self.clip_min, clip_max=self.clip_max,
            save_best=self._dev_i32(save_best), best_pattern=self.best_pattern,
            best_mask=self.best_mask, do_update=True)
        ph.mark(None)
        self.samples_done += B * S
        return any(st.active for st in self.img)

    def _draw_checksum(self):
        """A float-exact fingerprint of this step's draws — the mask indices, the second (``dual``) index set and the
        bit patterns of the placement extension's affine maps; all ranks must agree (verified in _gather_stats)."""
        w = np.arange(1, self.S + 1, dtype=np.int64)
        h = int((self.idx_np * w).sum())
        if self.idx2_np is not None:
            h = h * 31 + int((self.idx2_np * w).sum())
        if self.theta_np is not None:
            bits = np.ascontiguousarray(self.theta_np, dtype=np.float32).view(np.uint32).astype(np.int64).reshape(-1)
            h = h * 31 + int((bits * (np.arange(bits.size, dtype=np.int64) % 8191 + 1)).sum() % 2147483629)
        return float(h % 8388593)                                   # < 2^23: exact in fp32

    def _gather_stats(self):
        """Host copies of loss_adv (B,S) over all ranks' samples, loss_struc, group lasso, density (B,);
        also refreshes ``pred_host``.  One device->host copy: THE synchronisation of the step."""
        B, Sl, world, n_slab = self.B, self.S_local, self.world, self._n_slab
        host = self._comm[self._n_g:].cpu().numpy()
        tail = host[:self._n_tail].reshape(world, 2 * n_slab + 1)
        loss_adv = tail[:, :n_slab].reshape(world, B, Sl).transpose(1, 0, 2).reshape(B, world * Sl)
        self.pred_host = tail[:, n_slab:2 * n_slab].reshape(world, B, Sl).transpose(1, 0, 2) \
            .reshape(B, world * Sl).astype(np.int64)
        if world > 1 and not (tail[:, -1] == tail[self.rank, -1]).all():
            raise RuntimeError("EOT-sample sharding: the ranks drew different mask indices (checksums %s); every rank "
                               "must enter generate() with identical RNG state / `rngs`" % tail[:, -1].tolist())
        reg = host[self._n_tail:]
        return loss_adv, reg[:B], reg[B:2 * B], reg[2 * B:3 * B]

    def _eot_forward_backward(self, idx, idx2, crit_flags):
        """Occlude (dp_apply_fwd), run the frozen backbone forward + input-gradient backward,
        CW loss (dp_cw_loss) and reduce the input gradients over the samples (dp_apply_bwd).
        Micro-batched over whole images (or over S when one image's samples exceed the
        micro-batch) so activation memory stays bounded; the reduction order is fixed.
        With ``retire`` only the images still running take part (attack.py:311-316: the reference stops its one
        image there): their rows of adv_x / idx / y are gathered into a dense batch, the results scattered back; the
        patch-gradient rows of finished images are zero (their lr is 0: nothing reads them)."""
        Sl, S, dev = self.S_local, self.S, self.dev
        act = self._running()
        B = len(act)
        compact = B < self.B
        adv_x, x, y = self.adv_x, self.x, self.y
        g_out, loss_flat, pred = self.g_adv, self._own_loss, self.pred    # this rank's slab of the all-reduce buffer
        if compact:
            sel = torch.as_tensor(act, dtype=torch.int64, device=dev)
            adv_x, x, y = adv_x.index_select(0, sel), x.index_select(0, sel), y.index_select(0, sel)
            idx = idx.index_select(0, sel)
            idx2 = None if idx2 is None else idx2.index_select(0, sel)
            crit_flags = crit_flags.index_select(0, sel)
            g_out = torch.empty((B, 3, self.H, self.W), dtype=torch.float32, device=dev)
            loss_flat = torch.empty((B * Sl,), dtype=torch.float32, device=dev)
            pred = torch.empty((B * Sl,), dtype=torch.int32, device=dev)
        self._run_y, self._run_g, self._run_pred = y, g_out, pred
        self._run_on = np.asarray([self.img[b].active for b in act], dtype=bool)
        upstream = 1.0 / float(S)                  # loss_adv.mean(1) then .sum().backward()
        mb = max(1, self.o.micro_batch)
        timer = None
        if self.kernel_events is not None:   # bench.py: HIP events stamped by the dominant kernel itself
            timer = ops.KernelTimer()
            self.kernel_events.append(timer)
        theta = theta_inv = delta = None
        if self.placement is not None:     # extension: sample (b, s) sees x + warp(delta, theta[b, s])
            from . import placement as dp_placement
            th = np.ascontiguousarray(self.theta_np[act][:, self.s_lo:self.s_hi])
            theta = torch.as_tensor(th, device=dev)
            theta_inv = torch.as_tensor(dp_placement.invert(th), device=dev)
            delta = ops.blend(self.adv_mask, self.adv_pattern, self.x, self.eps, add_x=False)[0]
            if compact:
                delta = delta.index_select(0, sel)
            inp_all = ops.apply_affine_fwd(x, delta, theta, self.table, idx, idx2, self.dn, timer=timer)
        else:
            inp_all = ops.apply_fwd(adv_x, self.table, idx, idx2, self.dn, timer=timer)   # (B*Sl,3,H,W)
        self._placement_ctx = (theta, theta_inv)
        # micro-batches: (first sample, end sample, first image, end image, first local mask, end local mask, accumulate)
        chunks = []
        if Sl <= mb:
            b0 = 0
            for k in self._image_groups(B, Sl, mb):    # whole images per micro-batch
                chunks.append((b0 * Sl, (b0 + k) * Sl, b0, b0 + k, 0, Sl, False))
                b0 += k
        else:
            for b in range(B):
                for k, s0 in enumerate(range(0, Sl, mb)):
                    s1 = min(Sl, s0 + mb)
                    chunks.append((b * Sl + s0, b * Sl + s1, b, b + 1, s0, s1, k > 0))
        self.n_forward += B * Sl
        if self._taped:
            from . import taped
            try:
                self._fb_taped(inp_all, chunks, idx, idx2, crit_flags, upstream, loss_flat)
            except taped.Unsupported as why:        # e.g. an input size the fused kernels do not take: autograd, all samples
                self.o._log(">> selected-sample backward not available here (%s): back-propagating every sample" % why)
                self._taped = False
        if not self._taped:
            # image-disjoint micro-batches (no chunk accumulates into another's rows) may run on several streams
            n_str = min(self.o.streams, len(chunks)) if dev.type == "cuda" and not any(c[6] for c in chunks) else 1
            if n_str > 1:                                 # adv_x / inp_all / idx were produced on the step's stream
                _, main = _side_streams(self._streams, n_str, dev)
            for k, c in enumerate(chunks):
                n0, n1, b0, b1, s0, s1, _ = c
                with (torch.cuda.stream(self._streams[k % n_str]) if n_str > 1 else contextlib.nullcontext()):
                    G = self._fb_chunk(inp_all[n0:n1], y[b0:b1], crit_flags[b0:b1], s1 - s0,
                                       upstream, loss_flat[n0:n1], pred[n0:n1])
                    self._reduce_chunk(G, c, idx, idx2)
                    del G
            if n_str > 1:
                for st in self._streams[:n_str]:
                    main.wait_stream(st)
            self.n_active += B * Sl
            self.n_backward += B * Sl
        if compact:                                # scatter the dense batch's results to the images' own rows
            self.g_adv.zero_()
            self.g_adv.index_copy_(0, sel, g_out)
            self._own_loss.view(self.B, Sl).index_copy_(0, sel, loss_flat.view(B, Sl))
            self.pred.view(self.B, Sl).index_copy_(0, sel, pred.view(B, Sl))    # finished images keep their last row
        self._run_y = self._run_g = self._run_pred = None
        self._own_pred.copy_(self.pred)            # int32 -> fp32 (class ids are exact), rides in the same buffer

    @staticmethod
    def _image_groups(B, Sl, mb):
        """Whole images per micro-batch for B images of Sl samples: mb // Sl each, and for what is left over (a batch not
        divisible by it, or one that shrank because images retired) the largest group whose ROW COUNT the committed
        library routes were measured for (conv1x1.TUNED: 512 / 128 / 64 / 32 rows) — 3 running images x 128 samples run as
        3 micro-batches of 128 rows, not as one of 384 (a batch size with no tuned GEMM solutions and no route column:
        measured 1.3x slower per sample, profiles/r04a_bench_whole_attack.json)."""
        from . import conv1x1
        ipm = max(1, mb // Sl)
        allowed = sorted({L // Sl for L in conv1x1.TUNED if L % Sl == 0 and 1 <= L // Sl < ipm}, reverse=True)
        groups, left = [], B
        while left > 0:
            if left >= ipm:
                k = ipm
            else:
                k = next((a for a in allowed if a <= left), left)
            groups.append(k)
            left -= k
        return groups

    def _reduce_chunk(self, G, c, idx, idx2):
        n0, n1, b0, b1, s0, s1, accumulate = c
        if s1 - s0 == self.S_local:                # whole images
            self._reduce_over_samples(G, idx[b0:b1], None if idx2 is None else idx2[b0:b1], b1 - b0,
                                      self._run_g[b0:b1], accumulate, (b0, b1, s0, s1))
        else:                                      # a slice of one image's samples
            self._reduce_over_samples(G, idx[b0:b1, s0:s1].contiguous(),
                                      None if idx2 is None else idx2[b0:b1, s0:s1].contiguous(), 1,
                                      self._run_g[b0:b1], accumulate, (b0, b1, s0, s1))

    # ---------------------------------------------------------------- backward over the samples that carry gradient
    def _fb_taped(self, inp_all, chunks, idx, idx2, crit_flags, upstream, loss_flat):
        """Forward every micro-batch on an explicit tape, then back-propagate ONLY the EOT samples whose logit gradient
        is non-zero.  The CW hinge (attack.py:16-23) gives a sample whose margin is met an exactly zero logit gradient,
        and the frozen, per-sample-normalised backbone then gives it an exactly zero input gradient — the reference
        computes those zeros (attack.py:247), here they are not computed; images that have early-stopped (their update
        is multiplied by lr = 0) are skipped as well.  The selected samples of up to ``tape_tabs`` micro-batches are
        compacted into backward batches of the sizes the library routes are tuned for."""
        from . import taped
        pos = 0
        while pos < len(chunks):
            rows = chunks[pos][1] - chunks[pos][0]
            cap = self._tape_tabs if self._tape_tabs is not None else 1      # first group of the run: measure a tape
            group = [chunks[pos]]
            while (len(group) < min(cap, taped.MAX_TABS) and pos + len(group) < len(chunks)
                   and group[-1][1] - group[-1][0] == rows):
                nxt = chunks[pos + len(group)]
                if nxt[1] - nxt[0] > rows:
                    break
                group.append(nxt)
            pos += len(group)
            self._fb_group(group, rows, inp_all, idx, idx2, crit_flags, upstream, loss_flat)

    def _fb_group(self, group, tab_rows, inp_all, idx, idx2, crit_flags, upstream, loss_flat):
        from . import taped
        net, dev = self.net, self.dev
        tape = taped.StepTape(tab_rows, len(group))
        dls = []
        for n0, n1, b0, b1, s0, s1, _ in group:
            inp = inp_all[n0:n1]
            z = None
            if self._stem_split:
                conv = net.stem.conv
                from . import libconv
                with torch.no_grad():
                    z = libconv.conv_fwd(inp, conv.weight, conv.stride, conv.padding)
            logits = taped.forward(net, inp, tape, z=z)
            _, dlogits, pred = ops.cw_loss(logits.float().contiguous(), self._run_y[b0:b1].contiguous(),
                                           crit_flags[b0:b1].contiguous(), s1 - s0, self.confidence, upstream,
                                           loss_out=loss_flat[n0:n1])
            self._run_pred[n0:n1].copy_(pred)
            dls.append(dlogits)
        if self._tape_tabs is None:               # size the tapes of the rest of the run
            self._tape_tabs = self._tabs_that_fit(tape.nbytes())
        n_group = tape.n_samples
        dl = dls[0] if len(dls) == 1 else torch.cat(dls)
        act = (dl != 0).any(dim=1).cpu().numpy()   # host sync: the forward of this group is complete
        img_on = self._run_on
        if not img_on.all():                       # early-stopped images: lr = 0, their gradient is never used
            act &= np.concatenate([np.repeat(img_on[b0:b1], s1 - s0) for _, _, b0, b1, s0, s1, _ in group])
        nz = np.flatnonzero(act)
        self.n_active += len(nz)
        through_stem = not self._stem_split
        if n_group - len(nz) < max(1.0, self._skip_min_fraction * n_group):
            # (nearly) everything carries gradient: one backward per micro-batch, in place — compaction (odd batch sizes,
            # a scatter of the result) would cost more than back-propagating the few exact zeros
            for j, c in enumerate(group):
                r = c[1] - c[0]
                sel = torch.arange(j * tab_rows, j * tab_rows + r, dtype=torch.int32, device=dev)
                G = self._taped_backward(tape, dls[j], sel, r, through_stem)
                self._reduce_chunk(G, c, idx, idx2)
            self.n_backward += n_group
            return
        plan = taped.plan_chunks(len(nz), *self._ladder(tab_rows))
        g_shape = (n_group, tape_z_channels(net)) + tuple(tape.z_hw) if self._stem_split else \
            (n_group, 3, self.H, self.W)
        G_full = torch.zeros(g_shape, dtype=torch.float32, device=dev)
        if plan:
            smap_np, keep_np, p = [], [], 0
            for real, size in plan:
                rows = nz[p:p + real]
                smap_np.append(np.concatenate([rows, np.repeat(rows[-1:], size - real)]))
                keep_np.append(np.concatenate([np.ones(real, np.int64), np.zeros(size - real, np.int64)]))
                p += real
            both = torch.as_tensor(np.stack([np.concatenate(smap_np), np.concatenate(keep_np)]), device=dev)   # one upload
            smap_l, keep = both[0], both[1].to(torch.float32)
            smap = smap_l.to(torch.int32)
            dl_sel = dl.index_select(0, smap_l) * keep[:, None]         # padding rows: zero logit gradient
            o = 0
            for real, size in plan:
                G = self._taped_backward(tape, dl_sel[o:o + size].contiguous(), smap[o:o + size].contiguous(), size,
                                         through_stem)
                G_full.index_copy_(0, smap_l[o:o + real], G[:real])
                o += size
                self.n_backward += size
        for j, c in enumerate(group):
            self._reduce_chunk(G_full[j * tab_rows:j * tab_rows + (c[1] - c[0])], c, idx, idx2)

    def _taped_backward(self, tape, dl, sel, size, through_stem):
        """taped.backward; under deterministic="auto" every library call inside decides per (batch, shape) problem
        whether it needs deterministic kernels (libconv) — rank-local decisions, no collective here."""
        from . import taped
        return taped.backward(self.net, tape, dl, sel, through_stem)

    def _ladder(self, tab_rows):
        """(batch sizes a selected-sample backward may use, cost of one pass at each size in sample-equivalents).
        Sizes: the micro-batch itself plus the smaller batches the committed library routes were measured for
        (conv1x1.TUNED); cost = size + 7 (micro-batches of 64 / 128 measured 10.9 % / 4.8 % slower per sample than 512:
        profiles/r02a_ab_small_microbatch.jsonl)."""
        from . import conv1x1
        sizes = self._ladder_user or [L for L in sorted(conv1x1.TUNED) if L < tab_rows]
        sizes = sorted(set(int(L) for L in sizes if 0 < int(L) <= tab_rows) | {int(tab_rows)})
        return sizes, {L: L + 7.0 for L in sizes}

    def _tabs_that_fit(self, tape_bytes):
        """Micro-batches whose saved activations may be alive at once: half of the free HBM, at most 8."""
        from . import taped
        if not self.dev.type == "cuda":
            return taped.MAX_TABS
        free, _ = torch.cuda.mem_get_info(self.dev)
        free += torch.cuda.memory_reserved(self.dev) - torch.cuda.memory_allocated(self.dev)
        return int(max(1, min(taped.MAX_TABS, (free // 2 + tape_bytes) // max(1, tape_bytes))))

    def _reduce_over_samples(self, G, idx, idx2, B, out, accumulate, where):
        """sum_S keep * d loss/d masked-input (/ std) -> d loss/d adv_x for B images (autograd of attack.py:206-220).
        ``where`` = (b0, b1, s0, s1): the images / local samples ``G`` covers (placement extension only)."""
        theta, theta_inv = self._placement_ctx
        if theta is not None:    # extension: d loss / d delta through the warp's exact adjoint (delta enters adv_x 1:1)
            b0, b1, s0, s1 = where
            ops.apply_affine_bwd(G, theta[b0:b1, s0:s1].contiguous(), theta_inv[b0:b1, s0:s1].contiguous(), self.table,
                                 idx, idx2, self.dn, B=B, out=out, accumulate=accumulate)
        elif self._stem_split:     # G = d loss / d stem-conv output: stem input gradient + S-reduction in one launch
            ops.stem_dgrad_reduce(G, self.net.stem.conv.weight, self.table, idx, idx2, self.dn, B=B, out=out,
                                  accumulate=accumulate)
        else:
            ops.apply_bwd(G, self.table, idx, idx2, self.dn, B=B, out=out, accumulate=accumulate)

    def _fb_chunk(self, inp, y, flags, S_chunk, upstream, loss_out, pred_out):
        G = self._fb_chunk_once(inp, y, flags, S_chunk, upstream, loss_out, pred_out)
        from . import libconv
        rows = inp.shape[0]
        if self._det_whole_net and rows not in self._det_rows_checked and not torch.backends.cudnn.deterministic:
            self._det_rows_checked.add(rows)
            again = self._fb_chunk_once(inp, y, flags, S_chunk, upstream, loss_out, pred_out)
            differ = torch.tensor([0 if torch.equal(G, again) else 1], dtype=torch.int32, device=self.dev)
            dp_dist.allreduce_max_(differ, self.o.pg)        # every replica takes the same decision
            if int(differ.item()):
                torch.backends.cudnn.deterministic = True      # restored by close()
                self.o._log(">> this classifier's library kernels are not run-to-run deterministic at batch %d: "
                            "deterministic kernels for the rest of the run" % rows)
                G = self._fb_chunk_once(inp, y, flags, S_chunk, upstream, loss_out, pred_out)
        if libconv.MODE == "auto" and self.world > 1 and rows not in self._det_rows_merged:
            # the first micro-batch of a new row count probed new convolution problems: every replica adopts the union
            # of the ranks' findings (rank-symmetric: the chunking is the same on every rank)
            libconv.merge_across(self.o.pg)
            self._det_rows_merged.add(rows)
        return G

    def _fb_chunk_once(self, inp, y, flags, S_chunk, upstream, loss_out, pred_out):
        if self._stem_split:
            conv = self.net.stem.conv
            from . import libconv
            with torch.no_grad():
                inp = libconv.conv_fwd(inp, conv.weight, conv.stride, conv.padding)
        inp = inp.detach().requires_grad_(True)
        with torch.enable_grad():
            logits = self.net.forward_after_stem_conv(inp) if self._stem_split else self.net(inp)
        lg = logits.detach().float().contiguous()
        _, dlogits, pred = ops.cw_loss(lg, y.contiguous(), flags.contiguous(), S_chunk, self.confidence,
                                       upstream, loss_out=loss_out)
        pred_out.copy_(pred)
        (G,) = torch.autograd.grad(logits, inp, dlogits.to(logits.dtype))
        return G.contiguous()
