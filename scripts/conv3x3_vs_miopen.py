"""Go / no-go measurement: dp_conv3x3_fwd (direct implicit GEMM on v_mfma_f32_32x32x2_f32) against
MIOpen's route for the same convolution, on the GPU box, same process, same tensors.

    python scripts/conv3x3_vs_miopen.py [N]        # default 512 = the headline micro-batch

Prints one JSON line: ms and effective TFLOP/s of both, the max abs difference relative to the output scale, and the same
for the input gradient (dgrad = the same kernel with flipped / transposed weights)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

from dorpatch_amd import ops


def timed(fn, iters=10):
    fn(); fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / iters


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    torch.backends.cudnn.benchmark = False          # the product's setting: MIOpen immediate mode
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, 64, 56, 56, generator=g).cuda()
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24).cuda()
    wt = ops.pack_conv3x3_weights(w)
    wt_bwd = ops.pack_conv3x3_weights(w.flip(2, 3).transpose(0, 1).contiguous())     # dgrad as a forward convolution
    flop = 2.0 * N * 3136 * 64 * 576
    want = F.conv2d(x, w, padding=1)
    got = ops.conv3x3_fwd(x, wt)
    err = float((got - want).abs().max() / want.abs().max())
    dy = torch.randn(N, 64, 56, 56, generator=g).cuda()
    bwd_lib = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                          (True, False, False))[0]
    want_b, got_b = bwd_lib(), ops.conv3x3_fwd(dy, wt_bwd)
    err_b = float((got_b - want_b).abs().max() / want_b.abs().max())
    ms = dict(miopen_fwd=timed(lambda: F.conv2d(x, w, padding=1)), mfma_fwd=timed(lambda: ops.conv3x3_fwd(x, wt)),
              miopen_bwd_data=timed(bwd_lib), mfma_bwd_data=timed(lambda: ops.conv3x3_fwd(dy, wt_bwd)))
    out = dict(shape="N=%d 64->64 3x3/1 @56x56 fp32" % N, gflop=flop / 1e9, ms={k: round(v, 4) for k, v in ms.items()},
               tflops_effective={k: round(flop / (v * 1e-3) / 1e12, 1) for k, v in ms.items()},
               max_rel_diff_fwd=err, max_rel_diff_bwd_data=err_b,
               go_threshold="mfma_fwd >= 130 TFLOP/s effective and logits within 2e-6 (VERDICT r3 item 7)")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
