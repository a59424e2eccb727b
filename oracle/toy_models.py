"""TEST INFRASTRUCTURE — tiny deterministic backbones for fixtures and fast tests.

The hot path treats the classifier as an opaque ``nn.Module``; parity of the
DorPatch-specific arithmetic does not need the 25 M-parameter ResNetV2, so the
golden fixtures use these (CPU reference run in seconds, fixtures stay small).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ToyNet(nn.Module):
    """conv3x3/2 -> ReLU -> conv3x3/2 -> ReLU -> global avg pool -> linear."""

    def __init__(self, n_classes=10, width=8):
        super().__init__()
        self.c1 = nn.Conv2d(3, width, 3, stride=2, padding=1)
        self.c2 = nn.Conv2d(width, 2 * width, 3, stride=2, padding=1)
        self.fc = nn.Linear(2 * width, n_classes)

    def forward(self, x):
        x = F.relu(self.c1(x))
        x = F.relu(self.c2(x))
        return self.fc(x.mean((2, 3)))


@torch.no_grad()
def make_toy(n_classes=10, width=8, seed=7, gain=4.0):
    """Seeded weights drawn on the CPU generator (identical on every box)."""
    net = ToyNet(n_classes, width)
    gen = torch.Generator(device="cpu").manual_seed(seed)
    for p in net.parameters():
        fan_in = p[0].numel() if p.dim() > 1 else p.numel()
        p.copy_(torch.randn(p.shape, generator=gen) * (gain / fan_in ** 0.5 if p.dim() > 1 else 0.1))
    return net.eval()


class Normalize(nn.Module):
    """(x - mean) / std with list attributes like torchvision's Normalize."""

    def __init__(self, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
        super().__init__()
        self.mean, self.std = list(mean), list(std)

    def forward(self, x):
        m = torch.tensor(self.mean, dtype=x.dtype, device=x.device).view(1, -1, 1, 1)
        s = torch.tensor(self.std, dtype=x.dtype, device=x.device).view(1, -1, 1, 1)
        return (x - m) / s


class NormModel(nn.Module):
    """Same shape as the reference's NormModel (utils.py:71-78); named identically so the
    product's fusion of the normalisation into dp_apply_fwd is exercised."""

    def __init__(self, model, normalize):
        super().__init__()
        self.model, self.normalize = model, normalize

    def forward(self, x):
        return self.model(self.normalize(x))


class PeakNet(nn.Module):
    """conv5x5/2 -> ReLU -> conv3x3/2 -> global MAX pool -> linear: a classifier whose decision
    hangs on localised features, so occluding them flips the label (exercises PatchCleanser's
    disagreement / second-round branches, which the average-pooling ToyNet never reaches)."""

    def __init__(self, n_classes=10, width=8):
        super().__init__()
        self.c1 = nn.Conv2d(3, width, 5, stride=2, padding=2)
        self.c2 = nn.Conv2d(width, 2 * width, 3, stride=2, padding=1)
        self.fc = nn.Linear(2 * width, n_classes)

    def forward(self, x):
        x = F.relu(self.c1(x))
        x = self.c2(x)
        return self.fc(x.amax((2, 3)))


@torch.no_grad()
def make_peaky(n_classes=10, width=8, seed=11, gain=6.0):
    net = PeakNet(n_classes, width)
    gen = torch.Generator(device="cpu").manual_seed(seed)
    for p in net.parameters():
        fan_in = p[0].numel() if p.dim() > 1 else p.numel()
        p.copy_(torch.randn(p.shape, generator=gen) * (gain / fan_in ** 0.5 if p.dim() > 1 else 0.1))
    return net.eval()


def blob_image(H, seed, n_blobs=2, size=12):
    """Flat grey image with a few saturated random blobs (seeded): masking a blob changes what
    PeakNet sees."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(3, H, H, generator=g) * 0.2 + 0.4
    for _ in range(n_blobs):
        cy, cx = torch.randint(0, H - size, (2,), generator=g).tolist()
        img[:, cy:cy + size, cx:cx + size] = torch.rand(3, size, size, generator=g).round()
    return img
