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
MEAN = (0.48235, 0.45882, 0.40784)
STD = (0.00392156862745098,) * 3


def feature_ops(block: int):
    ops, cin = [], 3
    for v in VGG16_CFG:
        if v == "M":
            ops.append(("pool",))
        else:
            ops.append(("conv", cin, v))
            ops.append(("relu",))
            cin = v
    return ops[:block]


def conv_shapes(block: int):
    return [(op[1], op[2]) for op in feature_ops(block) if op[0] == "conv"]


def get_feature_maps(img: np.ndarray, block: int, parameters) -> np.ndarray:
    """uint8 [H,W] (already CLAHE'd) -> float32 [C,h,w]."""
    x = torch.from_numpy(img.astype(np.float32) / np.float32(255.0))[None].repeat(3, 1, 1)  # ToTensor + repeat
    mean = torch.tensor(MEAN, dtype=torch.float32)[:, None, None]
    std = torch.tensor(STD, dtype=torch.float32)[:, None, None]
    x = ((x - mean) / std)[None]
    k = 0
    with torch.no_grad():
        for op in feature_ops(block):
            if op[0] == "conv":
                w, b = parameters[k]
                k += 1
                x = F.conv2d(x, torch.from_numpy(w), torch.from_numpy(b), stride=1, padding=1)
            elif op[0] == "relu":
                x = F.relu(x)
            else:
                x = F.max_pool2d(x, 2, 2)
    return x.numpy().squeeze(0)
