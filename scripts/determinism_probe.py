#!/usr/bin/env python
"""Which kernels of the frozen ResNetV2-50x1-BiT forward + input-gradient backward are NOT run-to-run
deterministic on this box?  (tests/test_backbone_parity_gpu.py::test_two_fresh_runs_are_bit_identical.)

For every convolution (each distinct (in, out, kernel, stride, input shape)) and for the hand-written
kernels: run the forward 3x and the backward-data 3x on the same inputs and compare bits.  Then the whole
network, 4 runs.  Prints one JSON line per finding.  Diagnostic only.
    python scripts/determinism_probe.py [--n 8 512] [--deterministic 0 1]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from dorpatch_amd import conv1x1  # noqa: E402
from dorpatch_amd.resnetv2 import StdConv2d, resnetv2_50x1_bit, seeded_init_  # noqa: E402


def same(ts):
    return all(torch.equal(ts[0], t) for t in ts[1:])


def probe_convs(net, n, reps=3):
    shapes = {}

    def hook(mod, inp, out):
        key = (mod.in_channels, mod.out_channels, mod.kernel_size[0], mod.stride[0], tuple(inp[0].shape[1:]))
        shapes.setdefault(key, (mod, tuple(inp[0].shape), tuple(out.shape)))
    hs = [m.register_forward_hook(hook) for m in net.modules() if isinstance(m, StdConv2d)]
    from dorpatch_amd.resnetv2 import GroupNormAct
    GroupNormAct.fused_saved = GroupNormAct.fused
    with torch.no_grad():
        mode = conv1x1.MODE
        conv1x1.MODE = "miopen"          # plain modules so that every conv's forward hook fires
        net(torch.randn(n, 3, 224, 224, device="cuda"))
        conv1x1.MODE = mode
    for h in hs:
        h.remove()
    bad = []
    for key, (mod, ishape, oshape) in sorted(shapes.items()):
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(ishape, device="cuda", generator=g)
        dy = torch.randn(oshape, device="cuda", generator=g)
        for route in (("miopen", "gemm") if key[2] == 1 and key[3] == 1 else ("miopen",)):
            if key[2] == 1 and key[3] == 1:
                f = lambda: conv1x1._IMPL[("fwd", route)](x, mod.weight, None)
                b = lambda: conv1x1._IMPL[("bwd", route)](dy, mod.weight, x)
            else:
                f = lambda: torch.nn.functional.conv2d(x, mod.weight, None, mod.stride, mod.padding)
                b = lambda: torch.ops.aten.convolution_backward(dy, x, mod.weight, None, mod.stride, mod.padding, (1, 1),
                                                                False, (0, 0), 1, (True, False, False))[0]
            with torch.no_grad():
                fo = [f().clone() for _ in range(reps)]
                bo = [b().clone() for _ in range(reps)]
            rec = dict(n=n, cin=key[0], cout=key[1], k=key[2], stride=key[3], hw=list(key[4][1:]), route=route,
                       fwd_deterministic=same(fo), bwd_deterministic=same(bo))
            if not (rec["fwd_deterministic"] and rec["bwd_deterministic"]):
                rec["fwd_maxdiff"] = max(float((fo[0] - t).abs().max()) for t in fo[1:])
                rec["bwd_maxdiff"] = max(float((bo[0] - t).abs().max()) for t in bo[1:])
                bad.append(rec)
                print(json.dumps(rec), flush=True)
    print(json.dumps(dict(n=n, conv_shapes=len(shapes), nondeterministic=len(bad))), flush=True)


def probe_net(net, n, runs=4):
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(n, 3, 224, 224, generator=g) * 2 - 1).cuda()
    dl = torch.randn(n, 1000, generator=g).cuda()
    outs = []
    for _ in range(runs):
        xi = x.clone().requires_grad_(True)
        lg = net(xi)
        (gx,) = torch.autograd.grad(lg, xi, dl)
        outs.append((lg.detach().clone(), gx.clone()))
    print(json.dumps(dict(n=n, whole_net_logits_equal=[torch.equal(outs[0][0], o[0]) for o in outs[1:]],
                          whole_net_input_grad_equal=[torch.equal(outs[0][1], o[1]) for o in outs[1:]],
                          grad_rel_diff=[float((outs[0][1] - o[1]).norm() / outs[0][1].norm()) for o in outs[1:]])), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="+", default=[8, 512])
    ap.add_argument("--deterministic", type=int, nargs="+", default=[0, 1])
    args = ap.parse_args()
    net = seeded_init_(resnetv2_50x1_bit(1000)).fold_weight_standardization().freeze().cuda()
    for det in args.deterministic:
        torch.backends.cudnn.deterministic = bool(det)
        print(json.dumps(dict(cudnn_deterministic=bool(det), conv1x1_mode=conv1x1.MODE)), flush=True)
        for n in args.n:
            probe_convs(net, n)
            probe_net(net, n)


if __name__ == "__main__":
    main()
