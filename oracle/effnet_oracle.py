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

# ---- the architecture, restated here independently of the library under test -------------------------------------------------
# torchvision.models.efficientnet: (fused, expand ratio, kernel, stride, input width, output width, layers) per stage.
_V2 = {
    "EfficientNetV2_S": [(1, 1, 3, 1, 24, 24, 2), (1, 4, 3, 2, 24, 48, 4), (1, 4, 3, 2, 48, 64, 4), (0, 4, 3, 2, 64, 128, 6),
                         (0, 6, 3, 1, 128, 160, 9), (0, 6, 3, 2, 160, 256, 15)],
    "EfficientNetV2_M": [(1, 1, 3, 1, 24, 24, 3), (1, 4, 3, 2, 24, 48, 5), (1, 4, 3, 2, 48, 80, 5), (0, 4, 3, 2, 80, 160, 7),
                         (0, 6, 3, 1, 160, 176, 14), (0, 6, 3, 2, 176, 304, 18), (0, 6, 3, 1, 304, 512, 5)],
    "EfficientNetV2_L": [(1, 1, 3, 1, 32, 32, 4), (1, 4, 3, 2, 32, 64, 7), (1, 4, 3, 2, 64, 96, 7), (0, 4, 3, 2, 96, 192, 10),
                         (0, 6, 3, 1, 192, 224, 19), (0, 6, 3, 2, 224, 384, 25), (0, 6, 3, 1, 384, 640, 7)],
}
_B0 = [(0, 1, 3, 1, 32, 16, 1), (0, 6, 3, 2, 16, 24, 2), (0, 6, 5, 2, 24, 40, 2), (0, 6, 3, 2, 40, 80, 3),
       (0, 6, 5, 1, 80, 112, 3), (0, 6, 5, 2, 112, 192, 4), (0, 6, 3, 1, 192, 320, 1)]
_B_MULT = {"EfficientNet_B1": (1.0, 1.1), "EfficientNet_B2": (1.1, 1.2), "EfficientNet_B3": (1.2, 1.4),
           "EfficientNet_B4": (1.4, 1.8), "EfficientNet_B5": (1.6, 2.2), "EfficientNet_B6": (1.8, 2.6),
           "EfficientNet_B7": (2.0, 3.1)}
# BatchNorm eps as torchvision builds the models: 1e-3 for efficientnet_v2_* and b5 / b6 / b7, the default 1e-5 otherwise
EPS = {**{k: 1e-3 for k in _V2}, **{k: (1e-3 if k[-1] in "567" else 1e-5) for k in _B_MULT}}


def make_divisible(v: float, divisor: int = 8) -> int:
    """torchvision.models._utils._make_divisible with min_value = divisor."""
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def stages(model_str: str):
    if model_str in _V2:
        return list(_V2[model_str])
    wm, dm = _B_MULT[model_str]
    import math
    return [(f, e, k, s, make_divisible(ci * wm), make_divisible(co * wm), int(math.ceil(n * dm))) for f, e, k, s, ci, co, n in _B0]


def arch_ops(model_str: str, block: int) -> list[dict]:
    """features[:block] (network.py:185-186) flattened into convolutions (kind 0), depthwise convolutions (1) and
    squeeze-excitations (2), with the module name of every layer in a torchvision state dict.  act: 2 = SiLU, 0 = none."""
    st = stages(model_str)
    ops = [dict(kind=0, cin=3, cout=st[0][4], ks=3, stride=2, act=2, res=0, sq=0, feature=0, block_end=1,
                names=("features.0.0", "features.0.1"))]
    for si, (fused, expand, ks, stride0, cin0, cout, layers) in enumerate(st[: max(0, block - 1)]):
        for l in range(layers):
            cin, stride = (cin0, stride0) if l == 0 else (cout, 1)
            exp = make_divisible(cin * expand)
            res = int(stride == 1 and cin == cout)
            pre = f"features.{si + 1}.{l}.block"
            k = 0

            def conv(ci, co, kk, s, act, r, end):
                nonlocal k
                ops.append(dict(kind=0, cin=ci, cout=co, ks=kk, stride=s, act=act, res=r, sq=0, feature=si + 1, block_end=end,
                                names=(f"{pre}.{k}.0", f"{pre}.{k}.1")))
                k += 1

            if fused:
                if exp != cin:
                    conv(cin, exp, ks, stride, 2, 0, 0)
                    conv(exp, cout, 1, 1, 0, res, 1)
                else:
                    conv(cin, cout, ks, stride, 2, res, 1)
                continue
            if exp != cin:
                conv(cin, exp, 1, 1, 2, 0, 0)
            ops.append(dict(kind=1, cin=exp, cout=exp, ks=ks, stride=stride, act=2, res=0, sq=0, feature=si + 1, block_end=0,
                            names=(f"{pre}.{k}.0", f"{pre}.{k}.1")))
            k += 1
            ops.append(dict(kind=2, cin=exp, cout=exp, ks=0, stride=0, act=0, res=0, sq=max(1, cin // 4), feature=si + 1,
                            block_end=0, names=(f"{pre}.{k}.fc1", f"{pre}.{k}.fc2")))
            k += 1
            conv(exp, cout, 1, 1, 0, res, 1)
    if block >= len(st) + 2:  # the whole of `features`: the closing 1x1 convolution + BatchNorm + SiLU
        cin, f = st[-1][5], len(st) + 1
        ops.append(dict(kind=0, cin=cin, cout=1280 if model_str in _V2 else 4 * cin, ks=1, stride=1, act=2, res=0, sq=0, feature=f,
                        block_end=1, names=(f"features.{f}.0", f"features.{f}.1")))
    return ops


ARCH_KEYS = ("kind", "cin", "cout", "ks", "stride", "act", "res", "sq", "feature", "block_end")


def _round(x: torch.Tensor, compute) -> torch.Tensor:
    if not compute:
        return x
    return x.to({"float16": torch.float16, "bfloat16": torch.bfloat16}[compute]).to(torch.float32)


def get_feature_maps(img: np.ndarray, ops, parameters, mean, std, bn_eps: float = BN_EPS, compute: str | None = None) -> np.ndarray:
    """uint8 [H,W] or RGB [H,W,3] (already CLAHE'd) -> float32 [C,h,w].  ``ops``: the layer list of Model.effnet_ops (kind,
    widths, kernel, stride, activation, residual flag, block_end); ``parameters[i]``: (w, b, gamma, beta, running mean,
    running variance) or, for a squeeze-excitation, (fc1 w, fc1 b, fc2 w, fc2 b).

    ``compute`` = "float16" | "bfloat16" restates the 16-bit plans of spr_effnet_plan_create_ex (BUILD-DEFINED): every tensor
    between layers is stored rounded to that type (the stem reads the rounded normalised image); a convolution takes its
    BatchNorm-folded weights rounded and, where a squeeze-excitation scales its input, the rounded product x * factor; sums,
    bias, SiLU, the residual sum, the squeeze-excitation mean / factors and the depthwise weights are float32; the last
    layer's output is not rounded."""
    if img.ndim == 3:
        x = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)).astype(np.float32) / np.float32(255.0))
    else:
        x = torch.from_numpy(img.astype(np.float32) / np.float32(255.0))[None].repeat(3, 1, 1)
    m = torch.tensor(mean, dtype=torch.float32)[:, None, None]
    s = torch.tensor(std, dtype=torch.float32)[:, None, None]
    x = ((x - m) / s)[None]
    block_in, scale = x, None
    if compute:
        return _forward16(x, ops, parameters, bn_eps, compute)
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


def _forward16(x, ops, parameters, bn_eps, compute):
    x = _round(x, compute)  # the stored tensor: what the next layer reads
    block_in, scale = x, None
    with torch.no_grad():
        for i, (op, p) in enumerate(zip(ops, parameters)):
            last = i + 1 == len(ops)
            t = [np.asarray(a, dtype=np.float32) for a in p]
            if op["kind"] == 2:
                z = x.mean(dim=(2, 3), keepdim=True)
                z = F.silu(F.conv2d(z, torch.from_numpy(t[0].reshape(op["sq"], op["cin"], 1, 1)), torch.from_numpy(t[1])))
                scale = torch.sigmoid(F.conv2d(z, torch.from_numpy(t[2].reshape(op["cin"], op["sq"], 1, 1)), torch.from_numpy(t[3])))
                continue
            w, b, gamma, beta, mu, var = t
            s = gamma / np.sqrt(var + np.float32(bn_eps))  # eval-mode BatchNorm folded as the library folds it (float32)
            wf = torch.from_numpy(np.ascontiguousarray(w * s[:, None, None, None]))
            bf = torch.from_numpy(np.ascontiguousarray((b - mu) * s + beta))
            a = x
            if op["kind"] == 0:
                if scale is not None:
                    a = _round(a * scale, compute)
                    scale = None
                wf = _round(wf, compute)
            groups = op["cin"] if op["kind"] == 1 else 1
            y = F.conv2d(a, wf, bf, stride=op["stride"], padding=op["ks"] // 2, groups=groups)
            if op["act"] == 2:
                y = F.silu(y)
            if op["kind"] == 0 and op["res"]:
                y = y + block_in
            x = y if last else _round(y, compute)
            if op["block_end"]:
                block_in = x
    return x.numpy().squeeze(0)
