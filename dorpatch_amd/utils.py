"""Drop-in twin of the reference ``utils.py`` for the symbols ``main.py`` star-imports
(``main.py:1``): ``set_device, set_random_seed, generate_saving_path, get_model,
NormModel, get_normalize, get_dataset, clip, NUM_CLASSES_DICT,
convert_float_list_to_str`` (+ the re-exported ``os, torch, np``).

Only ``clip`` and ``NormModel`` are on the hot path; the
rest is kept so the reference driver runs unchanged against this package.
"""
import json
import os
import random

import numpy as np
import torch

from . import ops
from .resnetv2 import resnetv2_50x1_bit

NUM_CLASSES_DICT = {'imagenet': 1000, 'cifar10': 10, 'cifar100': 100}


def set_device(device):
    """reference utils.py:12-13 (ROCm honours CUDA_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES)."""
    os.environ['CUDA_VISIBLE_DEVICES'] = device


def set_random_seed(seed=1234):
    """reference utils.py:16-21 — the three RNG streams the attack consumes."""
    torch.backends.cudnn.benchmark = True
    random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    np.random.seed(seed)


def generate_saving_path(configs):
    """reference utils.py:24-44: ``results/<remaining k=v joined by _>/<num_patch=.._patch_budget=..>``.

    Mutates ``configs`` exactly like the reference (pops the keys it consumes)."""
    json.dumps(configs, indent=4)
    for key in ("device", "model_dir", "data_dir", "batch_size", "lr", "epsilon"):
        configs.pop(key)
    subdir = None
    if configs["attack"] == 'DorPatch':
        subdir = '_'.join("%s=%s" % (k, configs.pop(k)) for k in ("num_patch", "patch_budget"))
    print(subdir)
    top = "_".join("%s=%s" % (k, v) for k, v in configs.items())
    save_path = os.path.join("results", top, subdir)
    os.makedirs(save_path, exist_ok=True)
    return save_path


class Normalize(torch.nn.Module):
    """Channel-wise ``(x - mean) / std`` (stand-in for torchvision.transforms.Normalize,
    which is absent from this image).  Exposes ``.mean`` / ``.std`` like torchvision so
    ``DorPatch.generate`` can fuse it into the occlusion kernel."""

    def __init__(self, mean, std):
        super().__init__()
        self.mean = [float(m) for m in mean]
        self.std = [float(s) for s in std]

    def forward(self, x):
        mean = torch.as_tensor(self.mean, dtype=x.dtype, device=x.device).view(1, -1, 1, 1)
        std = torch.as_tensor(self.std, dtype=x.dtype, device=x.device).view(1, -1, 1, 1)
        return (x - mean) / std


def get_normalize(dataset_name, model_name):
    """reference utils.py:66-68: mean = std = 0.5 for every dataset/model."""
    return Normalize(mean=[0.5, 0.5, 0.5], std=[0.5, 0.5, 0.5])


class NormModel(torch.nn.Module):
    """reference utils.py:71-78: ``model(normalize(x))``."""

    def __init__(self, model, normalize):
        super().__init__()
        self.model = model
        self.normalize = normalize

    def forward(self, x):
        return self.model(self.normalize(x))


def get_model(dataset_name, model_name, model_dir='pretrained_models'):
    """reference utils.py:47-63.  timm is replaced by the in-repo restatement of
    ``resnetv2_50x1_bit_distilled`` (timm-compatible state_dict keys); the
    PatchCleanser ``cutout2_128`` checkpoint is loaded from the same path."""
    archs = {'resnetv2_50x1_bit_distilled': resnetv2_50x1_bit}
    model = None
    for full_name, ctor in archs.items():
        if model_name in full_name:
            model = ctor(num_classes=NUM_CLASSES_DICT[dataset_name])
            ckpt_path = os.path.join(model_dir, dataset_name,
                                     full_name + '_cutout2_128_{}.pth'.format(dataset_name))
            checkpoint = torch.load(ckpt_path, map_location='cpu')
            model.load_state_dict(checkpoint['state_dict'])
    return model


def get_dataset(dataset_name, data_dir='/home/data', train=False, batch_size=128, shuffle=True):
    """reference ``utils.py:81-102``: a shuffled ``DataLoader`` over the named torchvision dataset (``cifar10`` /
    ``cifar100`` / ``imagenet`` under ``<data_dir>/<name>``), every image resized to 256 on its short side, centre-cropped to
    224 x 224 and converted to a float tensor in [0,1].  Dataset acquisition is outside the accelerated path
    but the driver's default (non ``--synthetic``) route needs it, so the loader is built here on torchvision, imported
    lazily; without torchvision the ImportError says what to do."""
    try:
        from torchvision import datasets, transforms
    except ImportError as e:
        raise ImportError("get_dataset(%r) needs torchvision (not installed here): install it, pass your own DataLoader "
                          "to dorpatch_amd.driver.run(args, dataloader=...), or use `main.py --synthetic`" % dataset_name) from e
    size = 224
    pipeline = transforms.Compose([transforms.Resize(int(size / 0.875)), transforms.CenterCrop((size, size)),
                                   transforms.ToTensor()])
    root = os.path.join(data_dir, dataset_name)
    if dataset_name == 'imagenet':
        dataset = datasets.ImageNet(root=root, split='train' if train else 'val', transform=pipeline)
    elif dataset_name in ('cifar10', 'cifar100'):
        cls = datasets.CIFAR10 if dataset_name == 'cifar10' else datasets.CIFAR100
        dataset = cls(root=root, train=train, download=True, transform=pipeline)
    else:
        raise KeyError(dataset_name)
    print('Dataset has {} instances'.format(len(dataset)))
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, num_workers=1, pin_memory=True)


def clip(mask, pattern, x, eps):
    """reference utils.py:105-110: ``delta = mask*(pattern-x)``, rescaled so that its
    per-image L2 norm is at most ``eps`` (scale detached).  Runs the HIP
    ``dp_sumsq_partials`` + ``dp_blend`` kernels; forward only (the attack applies the
    analytic backward inside ``dp_project_update``)."""
    B = x.shape[0]
    mask = mask.detach().expand(B, 1, x.shape[2], x.shape[3]).contiguous().float()
    delta, _, _ = ops.blend(mask, pattern.detach().contiguous().float(),
                            x.detach().contiguous().float(), eps, add_x=False)
    return delta


def convert_float_list_to_str(l):
    return ', '.join(["%.2f" % i for i in l])
