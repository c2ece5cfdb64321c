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


def _round(x: torch.Tensor, compute: str | None) -> torch.Tensor:
    if not compute:
        return x
    return x.to({"float16": torch.float16, "bfloat16": torch.bfloat16}[compute]).to(torch.float32)


def _cbn(x, p, stride, pad, compute=None):
    w, b, gamma, beta, mu, var = (np.asarray(t, dtype=np.float32) for t in p)
    if compute:
        # 16-bit plans: the library folds the BatchNorm into the convolution in float32, THEN rounds the weights; the operand is
        # a stored (rounded) activation; sums, bias in float32
        scale = gamma / np.sqrt(var + np.float32(1e-5))
        w = _round(torch.from_numpy(np.ascontiguousarray(w * scale[:, None, None, None])), compute)
        return F.conv2d(_round(x, compute), w, torch.from_numpy(np.ascontiguousarray((b - mu) * scale + beta)), stride=stride, padding=pad)
    x = F.conv2d(x, torch.from_numpy(w), torch.from_numpy(b), stride=stride, padding=pad)
    return F.batch_norm(x, torch.from_numpy(mu), torch.from_numpy(var), torch.from_numpy(gamma), torch.from_numpy(beta),
                        training=False, eps=1e-5)


def get_feature_maps(img: np.ndarray, block: int, parameters, compute: str | None = None) -> np.ndarray:
    """uint8 [H,W] (already CLAHE'd) -> float32 [C,h,w]; parameters[i] = (w, b, gamma, beta, running_mean, running_var)
    of convolution i of conv_specs(block) and its BatchNorm.  ``compute`` = "float16" | "bfloat16": the 16-bit compute type of
    spr_resnet_plan_create_ex - every convolution, the stem included (its operand is the normalised image), takes rounded
    weights and a rounded operand, the residual operand is a rounded stored activation, everything else float32."""
    x = torch.from_numpy(img.astype(np.float32) / np.float32(255.0))[None].repeat(3, 1, 1)
    mean = torch.tensor(MEAN, dtype=torch.float32)[:, None, None]
    std = torch.tensor(STD, dtype=torch.float32)[:, None, None]
    x = ((x - mean) / std)[None]
    specs = conv_specs(block)
    with torch.no_grad():
        x = F.relu(_cbn(x, parameters[0], 2, 3, compute))
        x = F.max_pool2d(x, 3, 2, 1)
        i = 1
        while i < len(specs):
            down = i + 3 < len(specs) and specs[i + 3][4] == 4
            y = F.relu(_cbn(x, parameters[i], 1, 0, compute))
            y = F.relu(_cbn(y, parameters[i + 1], specs[i + 1][3], 1, compute))
            y = _cbn(y, parameters[i + 2], 1, 0, compute)
            idn = _cbn(x, parameters[i + 3], specs[i + 3][3], 0, compute) if down else x
            x = F.relu(y + _round(idn, compute))
            i += 4 if down else 3
    return x.numpy().squeeze(0)
