"""Drop-in for the reference's ``utils.py`` module name: ``from utils import *``
(reference ``main.py:1``) must provide ``set_device, set_random_seed, generate_saving_path,
get_model, NormModel, get_normalize, get_dataset, clip, NUM_CLASSES_DICT,
convert_float_list_to_str`` and the re-exported ``os, torch, np``.  Implementation:
``dorpatch_amd.utils`` (``clip`` runs the HIP ``dp_sumsq_partials`` + ``dp_blend`` kernels).
"""
import os  # noqa: F401
import random  # noqa: F401

import numpy as np  # noqa: F401
import torch  # noqa: F401

from dorpatch_amd.utils import (NUM_CLASSES_DICT, NormModel, Normalize, clip,  # noqa: F401
                                convert_float_list_to_str, generate_saving_path, get_dataset,
                                get_model, get_normalize, set_device, set_random_seed)
