"""CPU oracle for the build-defined ResNet50 extractor: torch-CPU functional restatement of torchvision's resnet50 (v1.5)
cut after `block` of its top-level children [conv1, bn1, relu, maxpool, layer1, layer2, layer3] (block = 5 / 6 / 7).

TEST INFRASTRUCTURE ONLY (same rules as ncc_oracle.py).  PARITY UNPINNED by the reference: it has no ResNet branch at all
(network.py:121-182) - BASELINE.json config 3 names one - and torchvision is not importable here; the graph below is
torchvision's published resnet50: stem 7x7/2 + BN + ReLU + maxpool 3x3/2, bottlenecks (1x1, 3x3 with the stride, 1x1 x4,
BatchNorm after each, downsample 1x1 + BN on the first block of a layer, ReLU after the sum), [3, 4, 6] blocks.
Pre-processing as the reference's default transforms (network.py:51-71): ToTensor, repeat(3), Normalize(ImageNet mean/std).
"""

from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)  # network.py:52-53
BLOCKS = (3, 4, 6)


def conv_specs(block: int):
    """(cin, cout, ksize, stride, role) of every convolution in module order; role 0 stem, 1/2/3 bottleneck, 4 downsample."""
    specs = [(3, 64, 7, 2, 0)]
    cin = 64
    for layer in range(block - 4):
        mid = 64 << layer
        for b in range(BLOCKS[layer]):
            stride = 2 if (b == 0 and layer > 0) else 1
            specs += [(cin, mid, 1, 1, 1), (mid, mid, 3, stride, 2), (mid, 4 * mid, 1, 1, 3)]
            if b == 0:
                specs.append((cin, 4 * mid, 1, stride, 4))
            cin = 4 * mid
    return specs


def _cbn(x, p, stride, pad):
    w, b, gamma, beta, mu, var = (torch.from_numpy(np.asarray(t, dtype=np.float32)) for t in p)
    x = F.conv2d(x, w, b, stride=stride, padding=pad)
    return F.batch_norm(x, mu, var, gamma, beta, training=False, eps=1e-5)


def get_feature_maps(img: np.ndarray, block: int, parameters) -> np.ndarray:
    """uint8 [H,W] (already CLAHE'd) -> float32 [C,h,w]; parameters[i] = (w, b, gamma, beta, running_mean, running_var)
    of convolution i of conv_specs(block) and its BatchNorm."""
    x = torch.from_numpy(img.astype(np.float32) / np.float32(255.0))[None].repeat(3, 1, 1)
    mean = torch.tensor(MEAN, dtype=torch.float32)[:, None, None]
    std = torch.tensor(STD, dtype=torch.float32)[:, None, None]
    x = ((x - mean) / std)[None]
    specs = conv_specs(block)
    with torch.no_grad():
        x = F.relu(_cbn(x, parameters[0], 2, 3))
        x = F.max_pool2d(x, 3, 2, 1)
        i = 1
        while i < len(specs):
            down = i + 3 < len(specs) and specs[i + 3][4] == 4
            y = F.relu(_cbn(x, parameters[i], 1, 0))
            y = F.relu(_cbn(y, parameters[i + 1], specs[i + 1][3], 1))
            y = _cbn(y, parameters[i + 2], 1, 0)
            idn = _cbn(x, parameters[i + 3], specs[i + 3][3], 0) if down else x
            x = F.relu(y + idn)
            i += 4 if down else 3
    return x.numpy().squeeze(0)
