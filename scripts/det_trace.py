#!/usr/bin/env python
"""Which library call makes two passes of the frozen backbone differ under the per-problem determinism policy?
Every libconv.guard call (and every routed GEMM) records an exact fingerprint of its result; several passes of the same
forward + input-gradient backward are compared call by call.  Diagnostic (round 3, GPU call 3).
    python scripts/det_trace.py --n 8 32 128"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from dorpatch_amd import conv1x1, libconv  # noqa: E402
from dorpatch_amd.resnetv2 import resnetv2_50x1_bit, seeded_init_  # noqa: E402

trace = []


def fp(t):
    return int(t.contiguous().view(torch.int32).to(torch.int64).sum().item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="+", default=[8, 32, 128])
    ap.add_argument("--passes", type=int, default=5)
    args = ap.parse_args()
    net = seeded_init_(resnetv2_50x1_bit(1000)).fold_weight_standardization().freeze().cuda()
    conv1x1.activate(None, True)
    libconv.MODE = "auto"
    orig_guard = libconv.guard

    def guard(key, fn, forced_fn=None):
        out = orig_guard(key, fn, forced_fn)
        trace.append((key, bool(libconv.POLICY.get(key)), fp(out)))
        return out
    libconv.guard = guard
    orig_impl = dict(conv1x1._IMPL)
    for n in args.n:
        g = torch.Generator().manual_seed(3)
        x = (torch.rand(n, 3, 224, 224, generator=g) * 2 - 1).cuda()
        dl = torch.randn(n, 1000, generator=g).cuda()
        runs = []
        for p in range(args.passes + 1):
            trace.clear()
            xi = x.clone().requires_grad_(True)
            lg = net(xi)
            (gx,) = torch.autograd.grad(lg, xi, dl)
            runs.append((list(trace), fp(lg), fp(gx)))
        runs = runs[1:]                      # pass 0 probed
        base = runs[0]
        out = dict(n=n, calls=len(base[0]), forced=sum(1 for k, f, _ in base[0] if f),
                   logits_equal=[r[1] == base[1] for r in runs[1:]], grad_equal=[r[2] == base[2] for r in runs[1:]])
        bad = {}
        for r in runs[1:]:
            for (k, f, a), (k2, f2, b) in zip(base[0], r[0]):
                assert k == k2
                if a != b:
                    bad.setdefault(str(k), dict(forced=f, passes_differing=0))["passes_differing"] += 1
        out["calls_whose_result_differs_between_passes"] = bad
        print(json.dumps(out), flush=True)
        # the first differing call in order (its inputs were still identical)
        for r in runs[1:]:
            for i, ((k, f, a), (_, _, b)) in enumerate(zip(base[0], r[0])):
                if a != b:
                    print(json.dumps(dict(n=n, first_differing_call=i, key=list(k), forced=f)), flush=True)
                    break


if __name__ == "__main__":
    main()
