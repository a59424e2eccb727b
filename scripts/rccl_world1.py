#!/usr/bin/env python
"""Execute every collective of the multi-rank path ONCE on RCCL, on the one GPU a gpurun box has:
a world-size-1 `nccl` process group bound to the device — init_process_group("nccl", device_id=...),
broadcast, broadcast_object_list, the SUM all-reduce of a view of a HotLoop-shaped _comm buffer, the int32 MAX-reduce of
a failure bitmap, all_gather_object (libconv.merge_across / conv1x1 agreement), barrier — through dorpatch_amd.dist, i.e.
the product's own call sites.  Not a scaling measurement: it proves the calls, dtypes and buffer views are accepted by
RCCL and return what the protocol expects.  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from dorpatch_amd import dist as dp_dist


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    t0 = time.perf_counter()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    pg = dist.group.WORLD
    out = {"backend": dist.get_backend(pg), "init_s": round(time.perf_counter() - t0, 3),
           "nccl_version": list(torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None}
    B, S, H = 64, 32, 224
    n_g, n_slab = B * 3 * H * H, B * S
    comm = torch.randn(n_g + (2 * n_slab + 1) + 3 * B, device=dev)
    before = comm.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dp_dist.allreduce_sum_(comm[:n_g + 2 * n_slab + 1], pg)           # THE data-path collective, on the real view
    torch.cuda.synchronize()
    out["allreduce_first_call_ms"] = round(1e3 * (time.perf_counter() - t0), 3)
    out["allreduce_sum_identity_world1"] = bool(torch.equal(comm, before))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20):
        dp_dist.allreduce_sum_(comm[:n_g + 2 * n_slab + 1], pg)
    ev1.record()
    ev1.synchronize()
    out["allreduce_38MB_ms"] = round(ev0.elapsed_time(ev1) / 20, 4)
    fail = torch.randint(0, 2, (B, 2520), dtype=torch.int32, device=dev)
    fb = fail.clone()
    dp_dist.allreduce_max_(fail, pg)                                   # collect_failure's bitmap
    out["allreduce_max_int32_identity"] = bool(torch.equal(fail, fb))
    t = torch.arange(8, device=dev, dtype=torch.float32)
    out["broadcast_ok"] = bool(torch.equal(dp_dist.broadcast_(t.clone(), pg), t))
    y = torch.arange(5, device=dev, dtype=torch.int64)
    out["broadcast_int64_ok"] = bool(torch.equal(dp_dist.broadcast_(y.clone(), pg), y))
    state = np.random.RandomState(3).get_state()
    got = dp_dist.broadcast_object(state, pg)                           # RNG state at HotLoop construction
    out["broadcast_object_rng_state_ok"] = bool(got[0] == state[0] and np.array_equal(got[1], state[1]))
    out["all_true"] = [dp_dist.all_true(True, pg), dp_dist.all_true(False, pg)]
    from dorpatch_amd import libconv
    libconv.POLICY[("fwd", 8, 1, 1, 3, 1, 4, 4)] = True
    libconv.merge_across(pg)
    out["merge_across_ok"] = libconv.POLICY == {("fwd", 8, 1, 1, 3, 1, 4, 4): True}
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    out["ok"] = all(v is True for k, v in out.items() if k.endswith("_ok") or k.endswith("identity") or k.endswith("world1")) \
        and out["all_true"] == [True, False]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
