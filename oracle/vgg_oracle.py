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


def ops_after(block: int, arch: str, conv_index: int):
    """The (at most one) operation that directly follows convolution `conv_index` inside features[:block]."""
    ops = feature_ops(block, arch)
    at = [i for i, op in enumerate(ops) if op[0] == "conv"][conv_index]
    return ops[at + 1:at + 2]


def conv_shapes(block: int, arch: str = "VGG16"):
    return [(op[1], op[2]) for op in feature_ops(block, arch) if op[0] == "conv"]


def round_to(x: torch.Tensor, compute: str) -> torch.Tensor:
    """float32 values rounded (nearest even) to float16 / bfloat16 and back: what a 16-bit operand of the matrix cores holds."""
    return x.to({"float16": torch.float16, "bfloat16": torch.bfloat16}[compute]).to(torch.float32)


def get_feature_maps(img: np.ndarray, block: int, parameters, arch: str = "VGG16", compute: str | None = None) -> np.ndarray:
    """uint8 [H,W] or RGB [H,W,3] (already CLAHE'd) -> float32 [C,h,w].  ``parameters[i]`` = (w, b) or, for a convolution whose
    BatchNorm2d is part of the truncation, (w, b, gamma, beta, running_mean, running_var).

    ``compute`` = "float16" | "bfloat16" restates the 16-bit compute type of spr_vgg_plan_create_ex (BUILD-DEFINED: the
    reference runs float32, network.py:235): every convolution - the first one's input is the normalised image - takes its input
    and its (BatchNorm-folded) weights ROUNDED to that type, products and sums in float32 (exact products, so only the order of the f32 additions
    differs from the matrix cores), bias / ReLU / pool and the last output unrounded."""
    if img.ndim == 3:  # RGB [H,W,3]: transform_rgb = ToTensor + Normalize (network.py:74-87)
        x = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)).astype(np.float32) / np.float32(255.0))
    else:
        x = torch.from_numpy(img.astype(np.float32) / np.float32(255.0))[None].repeat(3, 1, 1)  # ToTensor + repeat
    mean = torch.tensor(ARCHS[arch][2], dtype=torch.float32)[:, None, None]
    std = torch.tensor(ARCHS[arch][3], dtype=torch.float32)[:, None, None]
    x = ((x - mean) / std)[None]
    k = 0
    n_convs = len(conv_shapes(block, arch))
    with torch.no_grad():
        for op in feature_ops(block, arch):
            if op[0] == "conv":
                p = parameters[k]
                k += 1
                w, b = torch.from_numpy(np.asarray(p[0], np.float32)), torch.from_numpy(np.asarray(p[1], np.float32))
                if compute and len(p) == 6 and ("bn", op[2]) in ops_after(block, arch, k - 1):
                    # the library folds an eval-mode BatchNorm into the convolution BEFORE the weights are rounded
                    gamma, beta, mu, var = (np.asarray(t, np.float32) for t in p[2:])
                    scale = gamma / np.sqrt(var + np.float32(1e-5))
                    w = torch.from_numpy(np.ascontiguousarray(p[0] * scale[:, None, None, None]))
                    b = torch.from_numpy(np.ascontiguousarray((p[1] - mu) * scale + beta))
                if compute and n_convs > 1:  # (a plan that is its first convolution alone stays float32)
                    x, w = round_to(x, compute), round_to(w, compute)
                x = F.conv2d(x, w, b, stride=1, padding=1)
            elif op[0] == "bn":
                if compute:
                    continue  # folded above
                gamma, beta, mu, var = (torch.from_numpy(t) for t in parameters[k - 1][2:])
                x = F.batch_norm(x, mu, var, gamma, beta, training=False, eps=1e-5)
            elif op[0] == "relu":
                x = F.relu(x)
            else:
                x = F.max_pool2d(x, 2, 2)
    return x.numpy().squeeze(0)
