"""TEST INFRASTRUCTURE — tests/bench_emu_hook.py (kernels through the host emulation on CPU tensors) plus: 32 x 32 images and
a toy classifier in place of ResNetV2-50, so that the EXACT command line a driver runs for an 8-GPU configuration
(`python bench.py --gpus 8 --config 3 --scaling strong ...`: 512 EOT samples of one image, 64 per rank) finishes on 8
emulated ranks in about a minute.  Everything else — the launcher, the preset arithmetic, the rank code, the process group,
the step's all-reduce — is bench.py's own.  exec'd by bench.py in every rank process (DORPATCH_BENCH_RANK_HOOK)."""
import os

_here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(_here, "bench_emu_hook.py")) as _f:
    exec(compile(_f.read(), os.path.join(_here, "bench_emu_hook.py"), "exec"),
         {"__name__": "bench_rank_hook", "__file__": os.path.join(_here, "bench_emu_hook.py"), "bench": bench})  # noqa: F821


def _tiny_model(device):
    from dorpatch_amd.utils import NormModel, get_normalize
    from oracle.toy_models import make_toy
    return NormModel(make_toy(n_classes=1000, width=8), get_normalize("imagenet", "resnetv2")).to(device).eval()


bench.SIZE_OVERRIDE = 32            # noqa: F821
bench.build_model = _tiny_model     # noqa: F821
