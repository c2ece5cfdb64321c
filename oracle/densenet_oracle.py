"""CPU oracle for the DenseNet_201 extractor: torch-CPU restatement of the reference's forward path.

TEST INFRASTRUCTURE ONLY (same rules as ncc_oracle.py).  Follows network.py:176-179 (model choice, ImageNet mean / std),
:60-71 / :74-87 (transforms), :185-186 (features[:block]) with torch.nn.functional ops in float32, on torchvision's
densenet201 `features` restated from its published definition: conv0 7x7/2 (3 -> 64), norm0, relu0, pool0 (3x3/2 max pool),
four dense blocks of 6 / 12 / 48 / 32 layers (BatchNorm, ReLU, 1x1 -> 128, BatchNorm, ReLU, 3x3 -> 32, concatenated behind
the input) with transitions between them (BatchNorm, ReLU, 1x1 halving the width, 2x2 average pool), norm5 (no ReLU inside
`features`).  PARITY UNPINNED by the reference: network.py needs cv2, torchvision and downloaded weights, none available offline.
"""

from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5

# ---- the architecture, restated here independently of the library under test: torchvision densenet201 = growth rate 32,
# block_config (6, 12, 48, 32), 64 initial features, bottleneck width 4 x 32 ------------------------------------------------
GROWTH, BLOCKS, INIT, BOTTLENECK = 32, (6, 12, 48, 32), 64, 128
FEATURE_NAMES = ("conv0", "norm0", "relu0", "pool0", "denseblock1", "transition1", "denseblock2", "transition2", "denseblock3",
                 "transition3", "denseblock4", "norm5")


def arch_ops(block: int) -> list[dict]:
    """features[:block] (network.py:185-186) as the layer list the tests compare Model.densenet_ops with: kind 0 stem (flags:
    1 norm0, 2 relu0, 4 pool0 inside the cut), 1 dense 1x1 (with both BatchNorms), 2 dense 3x3, 3 transition, 4 norm5;
    `ctot` = the width of the tensor the layer reads from / appends to once its block is complete; module names attached."""
    ops = []
    if block < 1:
        return ops
    ops.append(dict(kind=0, cin=3, cout=INIT, c_off=0, ctot=INIT, flags=(block >= 2) | 2 * (block >= 3) | 4 * (block >= 4),
                    feature=0, names=("features.conv0", "features.norm0")))
    width = INIT
    for b, layers in enumerate(BLOCKS):
        f = 4 + 2 * b
        if block <= f:
            break
        total = width + layers * GROWTH
        for l in range(layers):
            pre = f"features.denseblock{b + 1}.denselayer{l + 1}"
            ops.append(dict(kind=1, cin=width + l * GROWTH, cout=BOTTLENECK, c_off=0, ctot=total, flags=0, feature=f,
                            names=(f"{pre}.norm1", f"{pre}.conv1", f"{pre}.norm2")))
            ops.append(dict(kind=2, cin=BOTTLENECK, cout=GROWTH, c_off=width + l * GROWTH, ctot=total, flags=0, feature=f,
                            names=(f"{pre}.conv2",)))
        width = total
        if b == len(BLOCKS) - 1:
            if block > f + 1:
                ops.append(dict(kind=4, cin=width, cout=width, c_off=0, ctot=width, flags=0, feature=f + 1, names=("features.norm5",)))
            break
        if block <= f + 1:
            break
        ops.append(dict(kind=3, cin=width, cout=width // 2, c_off=0, ctot=width, flags=0, feature=f + 1,
                        names=(f"features.transition{b + 1}.norm", f"features.transition{b + 1}.conv")))
        width //= 2
    return ops


ARCH_KEYS = ("kind", "cin", "cout", "flags", "feature")


def _bn(x, p):
    g, b, m, v = (torch.from_numpy(np.asarray(t, dtype=np.float32)) for t in p)
    return F.batch_norm(x, m, v, g, b, training=False, eps=BN_EPS)


def get_feature_maps(img: np.ndarray, ops, parameters, mean, std) -> np.ndarray:
    """uint8 [H,W] or RGB [H,W,3] (already CLAHE'd) -> float32 [C,h,w]; ``ops`` / ``parameters`` as Model.densenet_ops and
    synth.densenet_parameters give them."""
    if img.ndim == 3:
        x = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)).astype(np.float32) / np.float32(255.0))
    else:
        x = torch.from_numpy(img.astype(np.float32) / np.float32(255.0))[None].repeat(3, 1, 1)
    m = torch.tensor(mean, dtype=torch.float32)[:, None, None]
    s = torch.tensor(std, dtype=torch.float32)[:, None, None]
    x = ((x - m) / s)[None]
    w = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    with torch.no_grad():
        for op, p in zip(ops, parameters):
            if op["kind"] == 0:
                x = F.conv2d(x, w(p[0]), None, stride=2, padding=3)
                if op["flags"] & 1:
                    x = _bn(x, p[1:5])
                if op["flags"] & 2:
                    x = F.relu(x)
                if op["flags"] & 4:
                    x = F.max_pool2d(x, 3, 2, 1)
            elif op["kind"] == 1:
                y = F.conv2d(F.relu(_bn(x, p[0:4])), w(p[4]))
                y = F.relu(_bn(y, p[5:9]))
            elif op["kind"] == 2:
                x = torch.cat([x, F.conv2d(y, w(p[0]), padding=1)], dim=1)
            elif op["kind"] == 3:
                x = F.avg_pool2d(F.conv2d(F.relu(_bn(x, p[0:4])), w(p[4])), 2, 2)
            else:
                x = _bn(x, p[0:4])
    return x.numpy().squeeze(0)
