"""ResNetV2-50x1-BiT backbone in plain PyTorch (MIOpen / rocBLAS do the convs).

The reference obtains its classifier from ``timm==0.6.7``
(``utils.py:51-63``: ``timm.create_model('resnetv2_50x1_bit_distilled')`` +
``reset_classifier`` + PatchCleanser's ``cutout2_128`` checkpoint).  timm is a
third-party dependency that is absent from ``/root/reference`` and from this
image, so this is a restatement of the published architecture with
**timm-compatible ``state_dict`` keys** so the real checkpoint loads unchanged:

    stem.conv.weight
    stages.{s}.blocks.{b}.{downsample.conv,conv1,conv2,conv3}.weight
    stages.{s}.blocks.{b}.norm{1,2,3}.{weight,bias}
    norm.{weight,bias}      head.fc.{weight,bias}

No reference test pins any logits at this boundary and timm cannot be installed offline; the
restatement is pinned against transformers' ``BitForImageClassification`` (the HF port of the same timm
model: equal logits / input gradients for equal weights, ``tests/test_backbone_vs_hf_bit.py``) and by
parameter count (25 549 352); parity against timm itself stays unpinned.

The backbone is *frozen* on the hot path, so the per-forward weight
standardisation of ``StdConv2d`` is folded once (``fold_weight_standardization``).
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F


class StdConv2d(nn.Conv2d):
    """Conv2d with per-output-channel weight standardisation (BiT), eps = 1e-8.

    ``w = (w - mean) / sqrt(var_biased + eps)`` over (in, kh, kw), recomputed every
    forward unless ``folded`` (frozen weights: standardise once, store in place)."""

    def __init__(self, in_ch, out_ch, kernel_size, stride=1, padding=0, eps=1e-8):
        super().__init__(in_ch, out_ch, kernel_size, stride=stride, padding=padding, bias=False)
        self.eps = eps
        self.folded = False

    def standardized_weight(self):
        w = self.weight
        flat = w.reshape(w.shape[0], -1)
        mean = flat.mean(dim=1, keepdim=True)
        var = flat.var(dim=1, unbiased=False, keepdim=True)
        return ((flat - mean) / torch.sqrt(var + self.eps)).reshape_as(w)

    def forward(self, x):
        if self.folded:
            from . import conv1x1
            if conv1x1.applicable(self, x):      # frozen 1x1/1 on the GPU: MIOpen or a batched GEMM, whichever measured faster
                return conv1x1.Conv1x1Function.apply(x, self.weight)
            if not self.weight.requires_grad:    # frozen: the library call goes through the per-layer determinism policy
                from . import libconv
                return libconv.FrozenConvFunction.apply(x, self.weight, self.stride, self.padding)
        w = self.weight if self.folded else self.standardized_weight()
        return F.conv2d(x, w, None, self.stride, self.padding)


class GroupNormAct(nn.GroupNorm):
    """GroupNorm(32) + ReLU, parameters named ``weight`` / ``bias`` like timm's.

    On a ROCm GPU with frozen affine parameters (the hot path: ``DorPatch.generate`` freezes the
    backbone) the pair runs as ONE hand-written HIP kernel per direction
    (``dp_gn_relu_fwd`` / ``dp_gn_relu_bwd``: 1 read + 1 write forward, 2 reads + 1 write
    backward instead of eager PyTorch's 5 and 8 passes over the activation).  On CPU tensors, or
    while the affine parameters still require gradients, it is the plain torch composition.
    ``GroupNormAct.fused = False`` disables the HIP path globally (A/B measurements)."""

    fused = True
    # round 5: on the frozen GPU hot path the norm's APPLY + ReLU are folded into the consuming convolution's operand
    # staging (ops.GnConvFunction: a statistics-only pass + dp_conv1x1_fwd / dp_conv3x3_gn_fwd on the RAW tensor) and the
    # residual add into the producing convolution's epilogue, wherever the hand-written MFMA kernels take the shape.
    # ``GroupNormAct.fold = False`` (or DORPATCH_GNFOLD=0) restores the round-4 graph (A/B measurements).
    fold = os.environ.get("DORPATCH_GNFOLD", "1") != "0"
    # ... for batches of at least this many samples.  Round 5 gated it at 256: the MFMA kernels worked in 448-pixel x 64-channel
    # tiles only, and below ~256 samples the deep layers had too few of them for 512 workgroup slots (the folded graph 2.5 %
    # slower than the library routes at 128 samples, 17 % at 64, 40 % at 32: profiles/r05e_*).  Round 6: the launchers pick
    # 256- / 128- / 64-pixel tiles where that fills the chip, and the folded graph is the faster one down to 32 samples
    # (same box, ms/step folded vs library routes: 32 samples 9.8 vs 10.4 - 11.3, 64: 14.85 vs 15.14, 128: 27.1 vs 27.4;
    # profiles/r06d_bench_*.json, r06e_bench_cfg3_*.json)
    fold_min_batch = int(os.environ.get("DORPATCH_GNFOLD_MIN_BATCH", "32"))

    def __init__(self, num_channels, num_groups=32, eps=1e-5):
        super().__init__(num_groups, num_channels, eps=eps, affine=True)

    def _use_hip(self, x):
        if GroupNormAct.fused and not (self.weight.requires_grad or self.bias.requires_grad):
            from . import ops
            return ops.gn_relu_supported(x, self.num_groups)
        return False

    def forward(self, x):
        if self._use_hip(x):
            from . import ops
            return ops.GnReluFunction.apply(x, self.weight, self.bias, self.num_groups, self.eps)
        return F.relu(F.group_norm(x, self.num_groups, self.weight, self.bias, self.eps), inplace=True)

    def add_forward(self, x, res=None):
        """(x, res) -> (s, relu(gn(s))) with s = x + res (s = x when res is None): the residual add of
        the previous bottleneck fused into this norm (one HIP kernel on the GPU hot path)."""
        if res is None:
            return x, self.forward(x)
        if self._use_hip(x):
            from . import ops
            return ops.AddGnReluFunction.apply(x, res, self.weight, self.bias, self.num_groups, self.eps)
        s = x + res
        return s, F.relu(F.group_norm(s, self.num_groups, self.weight, self.bias, self.eps), inplace=False)


class DownsampleConv(nn.Module):
    def __init__(self, in_ch, out_ch, stride):
        super().__init__()
        self.conv = StdConv2d(in_ch, out_ch, 1, stride=stride)

    def forward(self, x):
        return self.conv(x)


class PreActBottleneck(nn.Module):
    def __init__(self, in_ch, out_ch, stride, has_downsample):
        super().__init__()
        mid = out_ch // 4
        self.downsample = DownsampleConv(in_ch, out_ch, stride) if has_downsample else None
        self.norm1 = GroupNormAct(in_ch)
        self.conv1 = StdConv2d(in_ch, mid, 1)
        self.norm2 = GroupNormAct(mid)
        self.conv2 = StdConv2d(mid, mid, 3, stride=stride, padding=1)
        self.norm3 = GroupNormAct(mid)
        self.conv3 = StdConv2d(mid, out_ch, 1)

    def _dual_ok(self, pre):
        """Frozen, folded first block on fp32 NCHW GPU tensors with the fused kernels enabled."""
        from . import conv1x1, ops
        ds, c1 = self.downsample.conv, self.conv1
        return (GroupNormAct.fused and conv1x1.MODE != "miopen" and ds.folded and c1.folded
                and not (ds.weight.requires_grad or c1.weight.requires_grad)
                and ds.stride[0] in (1, 2) and ds.stride[0] == ds.stride[1] and ops.subsample2_supported(pre))

    def forward_pair(self, x, res=None):
        """Block input is ``x + res`` (``res`` = the previous block's un-added branch, or None);
        returns this block's ``(branch, shortcut)`` un-added, so the add can fuse into the next norm."""
        x, pre = self.norm1.add_forward(x, res)
        if self.downsample is not None and self._dual_ok(pre):
            from . import ops        # conv1 + strided downsample of the same tensor: one autograd node
            out, shortcut = ops.DualConv1x1Function.apply(pre, self.conv1.weight, self.downsample.conv.weight,
                                                          self.downsample.conv.stride[0])
        else:
            shortcut = self.downsample(pre) if self.downsample is not None else x
            out = self.conv1(pre)
        out = self.conv2(self.norm2(out))
        out = self.conv3(self.norm3(out))
        return out, shortcut

    def forward(self, x):
        out, shortcut = self.forward_pair(x)
        return out + shortcut

    def _sum_ok(self, x):
        """The folded form (``forward_sum``) applies: frozen + folded weights on fp32 NCHW GPU tensors, fused kernels and the
        fold enabled, and the block's three 1x1 convolutions are shapes dp_conv1x1_fwd takes."""
        from . import conv1x1, ops
        convs = [self.conv1, self.conv2, self.conv3] + ([self.downsample.conv] if self.downsample is not None else [])
        if not (GroupNormAct.fused and GroupNormAct.fold and x.shape[0] >= GroupNormAct.fold_min_batch
                and conv1x1.MODE in ("table", "mfma")
                and all(c.folded and not c.weight.requires_grad for c in convs)
                and self.norm1._use_hip(x)
                # norm2 / norm3 see other tensors: their frozen state and group shape are checked here, their planes below
                and all(GroupNormAct.fused and not (n.weight.requires_grad or n.bias.requires_grad)
                        and n.num_channels % n.num_groups == 0 for n in (self.norm2, self.norm3))):
            return False
        # what follows depends on the input's shape / layout only (the weights' shapes are fixed): decided once per shape —
        # 16 blocks ask every forward, and at 32 - 64 rows the host's time per step shows (profiles/r06y_host_path_ab.txt)
        memo = self.__dict__.setdefault("_sum_ok_shapes", {})
        key = (tuple(x.shape), x.dtype, x.is_cuda, x.is_contiguous())
        ok = memo.get(key)
        if ok is not None:
            return ok
        s = self.conv2.stride[0]
        sh, sw = (x.shape[2] + s - 1) // s, (x.shape[3] + s - 1) // s        # the plane after the block's stride
        # the BACKWARD runs the same kernel on the transposed weights: conv1 / conv3 need C % 64 == 0 there —
        # a bottleneck width that is not a multiple of 64 takes forward_pair instead of failing in the backward
        ok = bool(ops.conv1x1_supported(x, self.conv1.weight)
                  and self.conv1.weight.shape[1] % 64 == 0 and self.conv1.weight.shape[0] % 16 == 0
                  and self.conv3.weight.shape[1] % 64 == 0
                  and (self.downsample is None or self.downsample.conv.weight.shape[1] % 64 == 0)
                  and x.shape[0] * self.conv3.weight.shape[0] * sh * sw < 2 ** 31
                  and self.conv3.weight.shape[1] % 16 == 0 and self.conv3.weight.shape[0] % 64 == 0
                  and ((sh * sw) % 4 == 0 or sh * sw == 49)
                  and (self.downsample is None or (self.downsample.conv.stride[0] in (1, 2)
                                                   and self.downsample.conv.weight.shape[0] % 64 == 0
                                                   and (s == 1 or ops.subsample2_supported(x)))))
        memo[key] = ok
        return ok

    def forward_sum(self, x):
        """Folded form: ``x`` is the block's materialised input (the previous block's ``branch + shortcut``, added in that
        block's last convolution), the result is this block's materialised output.  GroupNorm-apply + ReLU run inside the
        consuming convolution wherever the MFMA kernels take the shape (``ops.GnConvFunction``, incl. the stride-2 3x3 on
        dp_conv3x3s2_fwd); elsewhere (7 x 7 planes) the norm is materialised as before."""
        from . import ops
        n1, n2, n3 = self.norm1, self.norm2, self.norm3
        G, w1, w2, w3 = n1.num_groups, self.conv1.weight, self.conv2.weight, self.conv3.weight
        fold1 = ops.gn_fold_supported(x, w1, G)
        if self.downsample is not None:
            ds = self.downsample.conv
            small_hw = (x.shape[2] // ds.stride[0]) * (x.shape[3] // ds.stride[0])
            if fold1 and small_hw % 4 == 0:
                out, shortcut = ops.GnDualConvFunction.apply(x, n1.weight, n1.bias, G, n1.eps, w1, ds.weight, ds.stride[0])
            else:       # e.g. stage 3: the subsampled plane is 7 x 7 — materialise the pre-activation, round-4 dual node
                out, shortcut = ops.DualConv1x1Function.apply(n1(x), w1, ds.weight, ds.stride[0])
        elif fold1:
            shortcut, out = ops.GnConvFunction.apply(x, n1.weight, n1.bias, G, n1.eps, w1, 1, None, True)
        else:
            from . import conv1x1
            shortcut, pre = ops.GnReluPassFunction.apply(x, n1.weight, n1.bias, G, n1.eps)
            out = conv1x1.Conv1x1Function.apply(pre, w1)
        if ops.gn_fold_supported(out, w2, n2.num_groups, self.conv2.stride, self.conv2.padding):
            out = ops.GnConvFunction.apply(out, n2.weight, n2.bias, n2.num_groups, n2.eps, w2,
                                           3 if self.conv2.stride[0] == 1 else 32, None, False)
        else:
            out = self.conv2(n2(out))
        if ops.gn_fold_supported(out, w3, n3.num_groups):
            return ops.GnConvFunction.apply(out, n3.weight, n3.bias, n3.num_groups, n3.eps, w3, 1, shortcut, False)
        return ops.Conv1x1AddFunction.apply(n3(out), w3, shortcut)


class _Stage(nn.Module):
    def __init__(self, in_ch, out_ch, stride, depth):
        super().__init__()
        blocks = []
        for i in range(depth):
            blocks.append(PreActBottleneck(in_ch if i == 0 else out_ch, out_ch,
                                           stride if i == 0 else 1, has_downsample=(i == 0)))
        self.blocks = nn.Sequential(*blocks)

    def forward(self, x):
        return self.blocks(x)


class _Stem(nn.Module):
    """'fixed' BiT stem: StdConv 7x7/2 -> zero pad 1 -> maxpool 3x3/2 (no padding).

    On the GPU the pad + pool pair is one HIP kernel per direction (``dp_pad_maxpool_fwd`` /
    ``_bwd``) and the convolution's input gradient (frozen, folded filter) is ``dp_stem_dgrad``;
    ``GroupNormAct.fused = False`` also disables both (eager A/B)."""

    def __init__(self, in_ch=3, out_ch=64):
        super().__init__()
        self.conv = StdConv2d(in_ch, out_ch, 7, stride=2, padding=3)

    def _conv(self, x):
        conv = self.conv
        if GroupNormAct.fused and x.requires_grad and conv.folded and not conv.weight.requires_grad:
            from . import ops
            if ops.stem_dgrad_supported(x, conv.weight, conv.stride, conv.padding):
                return ops.StemConvFunction.apply(x, conv.weight)    # input gradient via dp_stem_dgrad
        if GroupNormAct.fused and not x.requires_grad and conv.folded and not conv.weight.requires_grad and x.is_cuda:
            from . import libconv
            return libconv.conv_fwd(x.contiguous(), conv.weight, conv.stride, conv.padding)    # forward-only (the failure sweep)
        return conv(x)

    def pool(self, x):
        if GroupNormAct.fused:
            from . import ops
            if ops.pad_maxpool_supported(x):      # GPU fp32 NCHW only (checks x.is_cuda)
                return ops.PadMaxPoolFunction.apply(x)
        x = F.pad(x, (1, 1, 1, 1), value=0.0)
        return F.max_pool2d(x, kernel_size=3, stride=2, padding=0)

    def forward(self, x):
        return self.pool(self._conv(x))


class _Head(nn.Module):
    def __init__(self, in_ch, num_classes):
        super().__init__()
        self.fc = nn.Conv2d(in_ch, num_classes, 1, bias=True)

    def forward(self, x):
        x = F.adaptive_avg_pool2d(x, 1)
        fc = self.fc
        if not fc.weight.requires_grad and not (fc.bias is not None and fc.bias.requires_grad):
            # frozen: the 2048 -> n_classes 1x1 convolution on a (N, 2048, 1, 1) tensor is a library call like any other —
            # and at small N MIOpen runs it with a split-K kernel whose float atomics made the LOGITS differ between two
            # passes (round 3, scripts/det_trace.py) while every StdConv2d was already under the determinism policy
            from . import libconv
            y = libconv.FrozenConvFunction.apply(x, fc.weight, (1, 1), (0, 0))
            if fc.bias is not None:
                y = y + fc.bias.view(1, -1, 1, 1)
            return y.flatten(1)
        return fc(x).flatten(1)


class ResNetV2(nn.Module):
    def __init__(self, layers=(3, 4, 6, 3), channels=(256, 512, 1024, 2048), num_classes=1000,
                 stem_ch=64):
        super().__init__()
        self.stem = _Stem(3, stem_ch)
        stages, prev = [], stem_ch
        for i, (depth, ch) in enumerate(zip(layers, channels)):
            stages.append(_Stage(prev, ch, 1 if i == 0 else 2, depth))
            prev = ch
        self.stages = nn.Sequential(*stages)
        self.norm = GroupNormAct(prev)
        self.head = _Head(prev, num_classes)
        self.num_classes = num_classes

    def forward(self, x):
        return self.forward_after_stem_conv(self.stem._conv(x))

    def stem_split_supported(self, x):
        """True when the caller may run the stem convolution itself (``F.conv2d(x, stem.conv.weight, None, 2, 3)``,
        no autograd), continue with ``forward_after_stem_conv`` and turn the gradient w.r.t. the stem-conv output
        into the S-reduced patch gradient with ``ops.stem_dgrad_reduce`` (frozen folded stem, fused kernels on)."""
        conv = self.stem.conv
        if not (GroupNormAct.fused and conv.folded and not conv.weight.requires_grad):
            return False
        from . import ops
        return ops.stem_dgrad_supported(x, conv.weight, conv.stride, conv.padding)

    def forward_after_stem_conv(self, z):
        """Everything after ``stem.conv``: pad + max-pool, the four stages, final norm, head."""
        x, res = self.stem.pool(z), None
        for stage in self.stages:
            for block in stage.blocks:
                if res is None and block._sum_ok(x):
                    x = block.forward_sum(x)            # round 5: adds in the conv epilogue, norms in the conv staging
                else:
                    x, res = block.forward_pair(x, res)     # residual adds ride along into the next norm
        _, y = self.norm.add_forward(x, res)
        return self.head(y)

    def reset_classifier(self, num_classes):
        """timm API used by the reference (utils.py:58)."""
        in_ch = self.head.fc.in_channels
        self.head = _Head(in_ch, num_classes)
        self.num_classes = num_classes

    @torch.no_grad()
    def fold_weight_standardization(self):
        """Frozen backbone: standardise every StdConv2d weight once, in place."""
        for m in self.modules():
            if isinstance(m, StdConv2d) and not m.folded:
                m.weight.copy_(m.standardized_weight())
                m.folded = True
        return self

    def freeze(self):
        for p in self.parameters():
            p.requires_grad_(False)
        return self.eval()


def resnetv2_50x1_bit(num_classes=1000):
    """Architecture of timm's ``resnetv2_50x1_bit_distilled`` (25 549 352 params @1000 classes)."""
    return ResNetV2((3, 4, 6, 3), (256, 512, 1024, 2048), num_classes)


@torch.no_grad()
def seeded_init_(model, seed=1234, gn_bias=0.0):
    """Deterministic stand-in weights (no checkpoint is available offline):
    conv ~ N(0, 2/fan_in) before standardisation, GN gamma=1 beta=``gn_bias``, head ~ N(0, 0.01).
    Drawn on the CPU generator so CPU oracle and GPU product share the weights.

    ``gn_bias = 0`` (default; the benchmark's weights) gives a network that is chaotic in fp32: half of all
    pre-activations sit within a rounding error's reach of a ReLU gate somewhere along 50 layers, and its
    input gradient differs between two correct fp32 evaluations (or fp32 vs fp64) by ~1e-2 relative.
    ``gn_bias = 3.5`` is the WELL-CONDITIONED set used by the tight whole-network parity tests: every
    GroupNorm output is shifted 3.5 sigma into the linear side of its ReLU (0.02 % of the gates stay closed,
    so gating is still exercised), gate flips under re-ordered fp32 sums become rare, and fp32 agrees with
    fp64 to ~2e-6 of the gradient scale — tight enough to see a 1e-4 error anywhere in the
    GroupNorm / stem / conv1x1 / pooling path (tests/test_backbone_parity_gpu.py, smoke())."""
    gen = torch.Generator(device="cpu").manual_seed(seed)
    for name, p in model.named_parameters():
        if p.dim() == 4 and not name.startswith("head."):
            fan_in = p.shape[1] * p.shape[2] * p.shape[3]
            v = torch.randn(p.shape, generator=gen) * math.sqrt(2.0 / fan_in)
        elif name.startswith("head.fc.weight"):
            v = torch.randn(p.shape, generator=gen) * 0.01
        elif name.endswith("norm.weight") or ".norm" in name and name.endswith("weight"):
            v = torch.ones(p.shape)
        elif "norm" in name and name.endswith("bias"):
            v = torch.full(p.shape, float(gn_bias))
        else:
            v = torch.zeros(p.shape)
        p.copy_(v.to(p.device))
    return model


WELL_CONDITIONED_GN_BIAS = 3.5
