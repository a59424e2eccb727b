#!/usr/bin/env python
"""Throughput of the PatchCleanser certification phase (reference main.py:143-153 runs
PatchCleanser.robust_predict(img, True) for the 4 mask ratios on every adversarial image: 4 x (36 + 630) = 2664 masked
forwards per image, 2.66 M for BASELINE configs[4]'s 1000 images).  ResNetV2-50x1-BiT, seeded weights, 224 x 224,
`--images` random images per batched call.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from dorpatch_amd.patchcleanser import MaskWindow, PatchCleanser  # noqa: E402
from dorpatch_amd.resnetv2 import resnetv2_50x1_bit, seeded_init_  # noqa: E402
from dorpatch_amd.utils import NormModel, get_normalize  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=16)
    ap.add_argument("--size", type=int, default=224)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    net = seeded_init_(resnetv2_50x1_bit(1000), seed=1234).fold_weight_standardization().freeze()
    model = NormModel(net, get_normalize("imagenet", "resnetv2")).to(dev).eval()
    x = torch.rand(args.images, 3, args.size, args.size, generator=torch.Generator().manual_seed(5)).to(dev)
    ratios = (0.015, 0.03, 0.06, 0.12)                       # main.py:61
    pcs = [PatchCleanser(MaskWindow(args.size, r, 1), model) for r in ratios]
    pcs[0].robust_predict_batch(x[:2], True)                 # warm-up: library kernels for these batch shapes
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    certified = 0
    for pc in pcs:
        recs = pc.robust_predict_batch(x, True)
        certified += sum(int(r.certification) for r in recs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    forwards = args.images * len(ratios) * (36 + 630)
    print(json.dumps({"images": args.images, "ratios": list(ratios), "masked_forwards": forwards, "seconds": round(dt, 3),
                      "forwards_per_s": round(forwards / dt, 1), "seconds_per_image_all_4_ratios": round(dt / args.images, 4),
                      "certified_records": certified,
                      "config4_1000_images_minutes_on_1_gpu": round(1000 * dt / args.images / 60, 2)}))


if __name__ == "__main__":
    main()
