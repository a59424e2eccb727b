#!/usr/bin/env python
"""Where does the frozen ResNetV2-50x1-BiT forward + input-gradient backward spend its time on
MI355X (fp32)?  Prints samples/s for a few (micro-batch, memory-format, MIOpen-find) settings and a
per-kernel breakdown from torch.profiler.  Diagnostic only (not part of the product or the tests).

    python scripts/backbone_probe.py [--mb 64 256] [--profile]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from dorpatch_amd.resnetv2 import resnetv2_50x1_bit, seeded_init_


def run(net, mb, channels_last, steps=4, warm=2):
    x = torch.rand(mb, 3, 224, 224, device="cuda") * 2 - 1
    if channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
    dl = torch.randn(mb, 1000, device="cuda")

    def one():
        inp = x.detach().requires_grad_(True)
        out = net(inp)
        (g,) = torch.autograd.grad(out, inp, dl)
        return g

    t0 = time.perf_counter()
    for _ in range(warm):
        one()
    torch.cuda.synchronize()
    t_warm = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return mb / dt, dt * 1e3, t_warm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, nargs="+", default=[64, 256])
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--find", type=int, nargs="+", default=[0, 1])
    ap.add_argument("--formats", nargs="+", default=["nchw", "nhwc"])
    ap.add_argument("--no-fused", action="store_true",
                    help="eager GroupNorm/ReLU/max-pool and plain F.conv2d everywhere, so that the memory format is "
                         "the only difference between the runs (set PYTORCH_MIOPEN_SUGGEST_NHWC=1 for nhwc)")
    args = ap.parse_args()
    if args.no_fused:
        from dorpatch_amd import conv1x1
        from dorpatch_amd.resnetv2 import GroupNormAct
        GroupNormAct.fused = False
        conv1x1.MODE = "miopen"
    net = seeded_init_(resnetv2_50x1_bit(1000)).fold_weight_standardization().freeze().cuda()
    for find in args.find:
        torch.backends.cudnn.benchmark = bool(find)
        for fmt in args.formats:
            m = net.to(memory_format=torch.channels_last) if fmt == "nhwc" else net.to(memory_format=torch.contiguous_format)
            for mb in args.mb:
                sps, ms, tw = run(m, mb, fmt == "nhwc")
                print("find=%d fmt=%s mb=%4d: %8.1f samples/s  %8.2f ms/iter  (warm-up %.1f s)  mem %.1f GB" % (
                    find, fmt, mb, sps, ms, tw, torch.cuda.max_memory_allocated() / 1e9), flush=True)
    if args.profile:
        from torch.profiler import ProfilerActivity, profile
        torch.backends.cudnn.benchmark = bool(args.find[-1])
        m = net.to(memory_format=torch.contiguous_format)
        mb = args.mb[-1]
        run(m, mb, False, steps=1, warm=1)
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            run(m, mb, False, steps=2, warm=0)
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=90))


if __name__ == "__main__":
    main()
