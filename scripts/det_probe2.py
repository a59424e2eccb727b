#!/usr/bin/env python
"""How should a convolution problem be probed for run-to-run reproducibility?  (libconv.guard; round-3 GPU call 1: the
back-to-back 3-run probe judged problems reproducible whose results then differed between two steps of a test.)

For every convolution problem of ResNetV2-50 at batch n (forward and backward-data, MIOpen route): the result
 (a) three times back to back;
 (b) six times with DIFFERENT work in between (a large elementwise kernel, another convolution, an idle gap);
 (c) three times while a bandwidth hog runs on a second stream;
and which of the three protocols sees a difference.  One JSON line per problem that any protocol flags, then totals.
    python scripts/det_probe2.py --n 8 16 128"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from dorpatch_amd import conv1x1  # noqa: E402
from dorpatch_amd.resnetv2 import StdConv2d, resnetv2_50x1_bit, seeded_init_  # noqa: E402


def collect(net, n):
    shapes = {}

    def hook(mod, inp, out):
        key = (mod.in_channels, mod.out_channels, mod.kernel_size[0], mod.stride[0], tuple(inp[0].shape[1:]))
        shapes.setdefault(key, (mod, tuple(inp[0].shape), tuple(out.shape)))
    hs = [m.register_forward_hook(hook) for m in net.modules() if isinstance(m, StdConv2d)]
    with torch.no_grad():
        mode, conv1x1.MODE = conv1x1.MODE, "miopen"
        net(torch.randn(n, 3, 224, 224, device="cuda"))
        conv1x1.MODE = mode
    for h in hs:
        h.remove()
    return shapes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="+", default=[8, 16, 128])
    args = ap.parse_args()
    net = seeded_init_(resnetv2_50x1_bit(1000)).fold_weight_standardization().freeze().cuda()
    hog_src = torch.randn(64 * 1024 * 1024, device="cuda")
    hog_dst = torch.empty_like(hog_src)
    side = torch.cuda.Stream()
    other_x = torch.randn(32, 64, 56, 56, device="cuda")
    other_w = torch.randn(64, 64, 3, 3, device="cuda")
    for n in args.n:
        shapes = collect(net, n)
        totals = dict(n=n, problems=0, back_to_back=0, varied_context=0, concurrent_hog=0)
        for key, (mod, ishape, oshape) in sorted(shapes.items()):
            g = torch.Generator(device="cuda").manual_seed(1)
            x = torch.randn(ishape, device="cuda", generator=g)
            dy = torch.randn(oshape, device="cuda", generator=g)
            fns = {"fwd": lambda: F.conv2d(x, mod.weight, None, mod.stride, mod.padding),
                   "bwd": lambda: torch.ops.aten.convolution_backward(dy, x, mod.weight, None, mod.stride, mod.padding,
                                                                      (1, 1), False, (0, 0), 1, (True, False, False))[0]}
            for direction, fn in fns.items():
                with torch.no_grad():
                    fn()
                    a = [fn().clone() for _ in range(3)]
                    b = []
                    for k in range(6):
                        if k % 3 == 0:
                            hog_dst.copy_(hog_src)
                        elif k % 3 == 1:
                            F.conv2d(other_x, other_w, None, 1, 1)
                        else:
                            torch.cuda.synchronize()
                            time.sleep(0.002)
                        b.append(fn().clone())
                    c = []
                    for k in range(3):
                        side.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(side):
                            for _ in range(4):
                                hog_dst.copy_(hog_src)
                        c.append(fn().clone())
                        torch.cuda.current_stream().wait_stream(side)
                    torch.cuda.synchronize()
                ref = a[0]
                flags = dict(back_to_back=not all(torch.equal(ref, t) for t in a[1:]),
                             varied_context=not all(torch.equal(ref, t) for t in b),
                             concurrent_hog=not all(torch.equal(ref, t) for t in c))
                totals["problems"] += 1
                for k, v in flags.items():
                    totals[k] += int(v)
                if any(flags.values()):
                    print(json.dumps(dict(n=n, direction=direction, cin=key[0], cout=key[1], k=key[2], stride=key[3],
                                          hw=list(key[4][1:]), **flags)), flush=True)
        print(json.dumps(totals), flush=True)


if __name__ == "__main__":
    main()
