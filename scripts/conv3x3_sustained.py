"""Does dp_conv3x3_fwd keep its burst rate under sustained load?  In the step's kernel trace (profiles/r04g_kernel_stats_
timed_headline.txt) the 64@56^2 launches average 0.971 ms, in a 10-launch burst 0.880 (profiles/r04e_).  This runs each
route back to back for ~2 s and prints the average of consecutive blocks of 100 launches, plus an interleaved pattern
(conv, a 1x1 GEMM-sized matmul, a streaming add) that looks more like the step."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

from dorpatch_amd import ops


def blocks(fn, n_blocks=16, per=100):
    out = []
    for _ in range(n_blocks):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(per):
            fn()
        b.record()
        b.synchronize()
        out.append(round(a.elapsed_time(b) / per, 4))
    return out


def main():
    torch.backends.cudnn.benchmark = False
    g = torch.Generator().manual_seed(0)
    res = {}
    for C, S in ((64, 56), (128, 28)):
        x = torch.randn(512, C, S, S, generator=g).cuda()
        w = (torch.randn(C, C, 3, 3, generator=g) / (3.0 * C ** 0.5)).cuda()
        wt = ops.pack_conv3x3_weights(w)
        res["mfma %d@%d" % (C, S)] = blocks(lambda: ops.conv3x3_fwd(x, wt))
        res["miopen %d@%d" % (C, S)] = blocks(lambda: F.conv2d(x, w, padding=1))
        other = torch.randn(512, 256, S, S, generator=g).cuda() if S == 56 else None
        if other is not None:      # conv between unrelated memory-bound work, each launch timed by its own events
            evs = []
            for i in range(300):
                other.add_(1.0)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); ops.conv3x3_fwd(x, wt); b.record()
                evs.append((a, b))
            torch.cuda.synchronize()
            t = [a.elapsed_time(b) for a, b in evs]
            res["mfma 64@56 between streaming adds (per-launch events)"] = [round(sum(t[i:i + 100]) / 100, 4) for i in (0, 100, 200)]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
