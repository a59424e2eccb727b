"""Drop-in for the reference's ``defenses/PatchCleanser.py`` module path
(``from defenses.PatchCleanser import PatchCleanser, MaskWindow`` — reference ``main.py:4``;
``adv_PC_{i}.pt`` pickles name ``defenses.PatchCleanser.PatchCleanserRecord``).
Implementation: ``dorpatch_amd.patchcleanser`` (rectangle tables + ``dp_apply_fwd``).
"""
from dorpatch_amd.patchcleanser import (MaskWindow, PatchCleanser, PatchCleanserRecord,  # noqa: F401
                                        PatchCleanserResult)
