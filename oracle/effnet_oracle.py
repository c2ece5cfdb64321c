"""CPU oracle for the EfficientNet (B-series and V2) extractors: torch-CPU restatement of the reference's forward path.

TEST INFRASTRUCTURE ONLY (same rules as ncc_oracle.py).  Follows network.py:163-175 (model choice, mean / std), :60-71 /
:74-87 (ToTensor, repeat(3), Normalize), :185-186 (features[:block]) and :228-244 with torch.nn.functional ops in float32,
on torchvision's efficientnet_v2 graph restated from its published definition: stem 3x3/2 + BatchNorm + SiLU; FusedMBConv =
3x3 expansion (BN, SiLU) + 1x1 projection (BN), or one 3x3 (BN, SiLU) when the expansion is 1; MBConv = 1x1 expansion (BN,
SiLU), depthwise 3x3 (BN, SiLU), squeeze-excitation (mean, 1x1, SiLU, 1x1, sigmoid, scale), 1x1 projection (BN); residual sum
where stride 1 and equal widths (stochastic depth is the identity in eval mode); BatchNorm eps 1e-3 (V2, B5, B7) or 1e-5.  The
B-series is MBConv throughout (kernel 3 or 5), without the expansion convolution where the ratio is 1.
PARITY UNPINNED by the reference: network.py needs cv2, torchvision and downloaded weights, none available offline.
"""

from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3


def get_feature_maps(img: np.ndarray, ops, parameters, mean, std, bn_eps: float = BN_EPS) -> np.ndarray:
    """uint8 [H,W] or RGB [H,W,3] (already CLAHE'd) -> float32 [C,h,w].  ``ops``: the layer list of Model.effnet_ops (kind,
    widths, kernel, stride, activation, residual flag, block_end); ``parameters[i]``: (w, b, gamma, beta, running mean,
    running variance) or, for a squeeze-excitation, (fc1 w, fc1 b, fc2 w, fc2 b)."""
    if img.ndim == 3:
        x = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)).astype(np.float32) / np.float32(255.0))
    else:
        x = torch.from_numpy(img.astype(np.float32) / np.float32(255.0))[None].repeat(3, 1, 1)
    m = torch.tensor(mean, dtype=torch.float32)[:, None, None]
    s = torch.tensor(std, dtype=torch.float32)[:, None, None]
    x = ((x - m) / s)[None]
    block_in, scale = x, None
    with torch.no_grad():
        for op, p in zip(ops, parameters):
            t = [torch.from_numpy(np.asarray(a, dtype=np.float32)) for a in p]
            if op["kind"] == 2:
                z = x.mean(dim=(2, 3), keepdim=True)
                z = F.silu(F.conv2d(z, t[0].reshape(op["sq"], op["cin"], 1, 1), t[1]))
                scale = torch.sigmoid(F.conv2d(z, t[2].reshape(op["cin"], op["sq"], 1, 1), t[3]))
                continue
            if op["kind"] == 0 and scale is not None:
                x = x * scale
                scale = None
            groups = op["cin"] if op["kind"] == 1 else 1
            y = F.conv2d(x, t[0], t[1], stride=op["stride"], padding=op["ks"] // 2, groups=groups)
            y = F.batch_norm(y, t[4], t[5], t[2], t[3], training=False, eps=bn_eps)
            if op["act"] == 2:
                y = F.silu(y)
            if op["kind"] == 0 and op["res"]:
                y = y + block_in
            x = y
            if op["block_end"]:
                block_in = x
    return x.numpy().squeeze(0)
