"""CPU oracle for the feature extractor: torch-CPU restatement of the reference's forward path.

TEST INFRASTRUCTURE ONLY (same rules as ncc_oracle.py).  Follows network.py:60-71 (ToTensor, repeat(3),
Normalize), :125-134 (VGG16 mean / std), :185-186 (features[:block]) and :228-244 (batch of one, squeeze)
with torch.nn.functional ops in float32.  PARITY UNPINNED by the reference: network.py needs cv2,
torchvision and downloaded weights, none available offline, and the reference holds no fixtures for it;
the layer table is torchvision's published vgg16 "D" configuration.
"""

from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

VGG16_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")
VGG19_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M")
MEAN = (0.48235, 0.45882, 0.40784)          # network.py:128 (VGG16)
STD = (0.00392156862745098,) * 3
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)  # network.py:52-53 (VGG19, VGG19_BN)
ARCHS = {"VGG16": (VGG16_CFG, False, MEAN, STD), "VGG19": (VGG19_CFG, False, IMAGENET_MEAN, IMAGENET_STD),
         "VGG19_BN": (VGG19_CFG, True, IMAGENET_MEAN, IMAGENET_STD)}


def feature_ops(block: int, arch: str = "VGG16"):
    cfg, bn, _, _ = ARCHS[arch]
    ops, cin = [], 3
    for v in cfg:
        if v == "M":
            ops.append(("pool",))
        else:
            ops.append(("conv", cin, v))
            if bn:
                ops.append(("bn", v))
            ops.append(("relu",))
            cin = v
    return ops[:block]


def conv_shapes(block: int, arch: str = "VGG16"):
    return [(op[1], op[2]) for op in feature_ops(block, arch) if op[0] == "conv"]


def get_feature_maps(img: np.ndarray, block: int, parameters, arch: str = "VGG16") -> np.ndarray:
    """uint8 [H,W] or RGB [H,W,3] (already CLAHE'd) -> float32 [C,h,w].  ``parameters[i]`` = (w, b) or, for a convolution whose
    BatchNorm2d is part of the truncation, (w, b, gamma, beta, running_mean, running_var)."""
    if img.ndim == 3:  # RGB [H,W,3]: transform_rgb = ToTensor + Normalize (network.py:74-87)
        x = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)).astype(np.float32) / np.float32(255.0))
    else:
        x = torch.from_numpy(img.astype(np.float32) / np.float32(255.0))[None].repeat(3, 1, 1)  # ToTensor + repeat
    mean = torch.tensor(ARCHS[arch][2], dtype=torch.float32)[:, None, None]
    std = torch.tensor(ARCHS[arch][3], dtype=torch.float32)[:, None, None]
    x = ((x - mean) / std)[None]
    k = 0
    with torch.no_grad():
        for op in feature_ops(block, arch):
            if op[0] == "conv":
                p = parameters[k]
                k += 1
                x = F.conv2d(x, torch.from_numpy(p[0]), torch.from_numpy(p[1]), stride=1, padding=1)
            elif op[0] == "bn":
                gamma, beta, mu, var = (torch.from_numpy(t) for t in parameters[k - 1][2:])
                x = F.batch_norm(x, mu, var, gamma, beta, training=False, eps=1e-5)
            elif op[0] == "relu":
                x = F.relu(x)
            else:
                x = F.max_pool2d(x, 2, 2)
    return x.numpy().squeeze(0)
