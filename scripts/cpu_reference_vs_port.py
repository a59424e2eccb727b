#!/usr/bin/env python
"""BUILD-CONTAINER ONLY (needs /root/reference): is bench.py's CPU baseline — the oracle's restatement of the
reference step ("kind": "port") — as fast as the UNMODIFIED reference step?  

Times, on this host's cores, B = 1 x S sampled masks at 224x224 through the seeded ResNetV2-50x1-BiT:
  * reference: /root/reference/attack.py DorPatch.generate executed as is through oracle/ref_shim.py; the
    classifier handed to it is wrapped in a module that time-stamps every hot-loop forward (batch == S) and
    aborts the run after `warm + timed + 1` stage-0 steps — step time = interval between consecutive hot-loop
    forwards (forward, losses, backward, bookkeeping, update, next clip / sampling / occlusion), the
    collect_failure sweep of iteration 0 (attack.py:187-190) is timed apart;
  * port: oracle/restatement.eot_step, exactly what bench.py's cpu_baseline leg runs.
Both "as-is" (backbone weights keep requires_grad=True, as the reference leaves them) and "frozen".
Prints one JSON object; committed as profiles/r02_cpu_reference_vs_port.json.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


class Stop(Exception):
    pass


class Stamp(torch.nn.Module):
    def __init__(self, model, S, n_steps):
        super().__init__()
        self.inner, self.S, self.n_steps = model, S, n_steps
        self.hot, self.t0 = [], time.perf_counter()

    def forward(self, inp):
        if inp.requires_grad and inp.shape[0] == self.S:          # the hot loop's forward (the sweep runs under no_grad)
            self.hot.append(time.perf_counter())
            if len(self.hot) > self.n_steps:
                raise Stop()
        return self.inner(inp)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=32)
    ap.add_argument("--warm", type=int, default=3)
    ap.add_argument("--timed", type=int, default=5)
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    import bench
    from oracle import ref_shim
    ref = ref_shim.load_reference()
    threads = args.threads or min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    S, n_steps = args.samples, args.warm + args.timed
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(1, 3, 224, 224, generator=g)
    y = torch.randint(0, 1000, (1,), generator=g)
    out = {"host_cores": os.cpu_count(), "threads": threads, "S": S, "warm": args.warm, "timed": args.timed,
           "torch": torch.__version__, "what": "B=1 x %d masks @224x224 fp32, seeded ResNetV2-50x1-BiT" % S}
    for variant, trainable in (("as_is", True), ("frozen", False)):
        model = bench.build_model("cpu")
        for p in model.parameters():
            p.requires_grad_(trainable)
        stamp = Stamp(model, S, n_steps)
        cwd, tmp = os.getcwd(), tempfile.mkdtemp(prefix="dp_refbase_")
        os.makedirs(os.path.join(tmp, "res", "cfg", "sub"))
        os.chdir(tmp)
        torch.manual_seed(1234)
        np.random.seed(1234)
        t_begin = time.perf_counter()
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                ref.attack.DorPatch().generate(stamp, x, 0.0204, 1000, "res/cfg/sub", 0, y=y, targeted=True,
                                               sampling_size=S, max_iterations=n_steps + 5)
        except Stop:
            pass
        finally:
            os.chdir(cwd)
        steps = np.diff(stamp.hot)[args.warm:]
        out["reference_" + variant] = dict(samples_per_s=round(S / float(np.median(steps)), 3),
                                           median_step_s=round(float(np.median(steps)), 3), timed_steps=len(steps),
                                           setup_plus_sweep_s=round(stamp.hot[0] - t_begin, 1))
    port = bench.cpu_baseline(224, n_masks=S, warm=args.warm, timed=args.timed, budget_s=600, threads=threads)
    out["port_as_is"], out["port_frozen"] = port["detail"]["as_is"], port["detail"]["frozen"]
    out["port_over_reference"] = {v: round(out["port_" + v]["samples_per_s"] / out["reference_" + v]["samples_per_s"], 3)
                                  for v in ("as_is", "frozen")}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
