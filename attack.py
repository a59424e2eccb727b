"""Drop-in for the reference's ``attack.py`` module name: ``from attack import DorPatch``
(reference ``main.py:3``) resolves here to the MI355X-native optimiser.

Everything lives in ``dorpatch_amd.attack``; this file only keeps the import path the
reference driver uses.  See INTEGRATION.md.
"""
from dorpatch_amd.attack import CW_loss, DorPatch  # noqa: F401
from dorpatch_amd.patchcleanser import MaskWindow  # noqa: F401  (reference attack.py:5)
from dorpatch_amd.utils import clip  # noqa: F401            (reference attack.py:8)


def get_mask_set(img_size, dropout_size, dropout):
    """reference attack.py:25-31 — bool masks (True = keep).  The optimiser itself uses the
    rectangle tables of ``dorpatch_amd.masks`` and never materialises these."""
    mask_window = MaskWindow(img_size, dropout_size)
    if dropout == 1:
        return mask_window.mask_set
    elif dropout == 2:
        return mask_window.double_mask_set
