"""TEST INFRASTRUCTURE (build container only) — execute the UNMODIFIED reference driver, /root/reference/main.py, against
this repository's drop-in modules.

    python tests/run_reference_main.py <workdir> [main.py flags ...]

main.py's own imports (``from utils import *``, ``from attack import DorPatch``, ``from defenses.PatchCleanser import
PatchCleanser, MaskWindow`` — main.py:1-4) resolve to the repo-root shims because the repo root is first on sys.path; the
file itself is loaded from /root/reference byte for byte and its ``main(args)`` is called with its own parser's arguments.
What the harness supplies, all outside main.py:
 * no GPU here: the dp_* kernels run through the host emulation (tests/hipemu) and ``Tensor.cuda`` / ``Module.cuda`` are
   identity (main.py:54, 87-88, 122 hard-code them);
 * ``utils.get_model`` / ``utils.get_dataset`` (checkpoint + ImageNet: unavailable offline) are replaced BEFORE main.py's
   star import by a seeded toy classifier with 1000 classes and a synthetic loader (DORPATCH_REFMAIN_BATCHES batches, default 1) of 224 x 224 images labelled
   with the classifier's own prediction (so main.py:91-100 keeps them);
 * main.py does not pass ``max_iterations`` / ``sampling_size`` to ``generate`` (5000 / 128: hours under emulation), so
   ``DorPatch.generate``'s DEFAULTS are lowered to 3 iterations and 4 masks by a subclass installed in the ``attack`` shim;
   every argument main.py does pass goes through untouched and is recorded for the test to check.
Prints one JSON line: the recorded generate() keyword arguments, the files written, and main.py's captured stdout."""
import contextlib
import importlib.util
import io
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_MAIN = "/root/reference/main.py"


def main():
    work = sys.argv[1]
    flags = sys.argv[2:]
    sys.path.insert(0, ROOT)
    os.chdir(work)
    import torch
    spec = importlib.util.spec_from_file_location("tests_hipemu", os.path.join(HERE, "hipemu", "__init__.py"),
                                                  submodule_search_locations=[os.path.join(HERE, "hipemu")])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["tests_hipemu"] = mod
    spec.loader.exec_module(mod)
    from tests_hipemu import patch as emu_patch
    from oracle import toy_models

    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self

    import attack as attack_shim          # the repo-root drop-ins main.py will import
    import utils as utils_shim
    import defenses.PatchCleanser as pc_shim
    assert os.path.dirname(os.path.abspath(attack_shim.__file__)) == ROOT
    assert os.path.dirname(os.path.abspath(utils_shim.__file__)) == ROOT

    net = toy_models.make_toy(n_classes=1000, gain=2.0)

    def get_model(dataset_name, model_name, model_dir='pretrained_models'):
        return net

    def get_dataset(dataset_name, data_dir='/home/data', train=False, batch_size=128, shuffle=True):
        model = utils_shim.NormModel(net, utils_shim.get_normalize(dataset_name, "resnetv2")).eval()
        out = []
        for i in range(int(os.environ.get("DORPATCH_REFMAIN_BATCHES", "1"))):
            x = torch.rand(batch_size, 3, 224, 224, generator=torch.Generator().manual_seed(77 + i))
            with torch.no_grad():
                out.append((x, model(x).argmax(-1)))
        return out

    utils_shim.get_model, utils_shim.get_dataset = get_model, get_dataset
    calls = []

    class ShortDorPatch(attack_shim.DorPatch):
        def generate(self, *a, **k):
            calls.append({key: (val if isinstance(val, (int, float, str, bool, type(None))) else
                                ("tensor%s" % (tuple(val.shape),) if torch.is_tensor(val) else type(val).__name__))
                          for key, val in k.items()})
            calls[-1]["n_positional"] = len(a)
            k.setdefault("max_iterations", 3)
            k.setdefault("sampling_size", 4)
            return super().generate(*a, **k)

    attack_shim.DorPatch = ShortDorPatch

    buf = io.StringIO()
    with emu_patch.emulated_ops():
        spec = importlib.util.spec_from_file_location("reference_main", REF_MAIN)
        ref_main = importlib.util.module_from_spec(spec)
        sys.dont_write_bytecode = True
        with contextlib.redirect_stdout(buf):
            spec.loader.exec_module(ref_main)                     # main.py:1-44: imports + parser
            args = ref_main.parser.parse_args(flags)
            ref_main.main(args)                                   # main.py:47-187, unmodified
    files = sorted(os.path.relpath(os.path.join(d, f), work) for d, _, fs in os.walk(work) for f in fs)
    bound = dict(DorPatch=ref_main.DorPatch.__mro__[1].__module__, PatchCleanser=ref_main.PatchCleanser.__module__,
                 MaskWindow=ref_main.MaskWindow.__module__, clip=ref_main.clip.__module__,
                 NormModel=ref_main.NormModel.__module__)
    print(json.dumps(dict(calls=calls, files=files, stdout=buf.getvalue(), bound=bound,
                          record_type="%s.%s" % (pc_shim.PatchCleanserRecord.__module__, pc_shim.PatchCleanserRecord.__name__))))


if __name__ == "__main__":
    main()
