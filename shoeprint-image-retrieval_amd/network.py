"""Feature extractor on MI355X — host mirror of the reference's ``network.py``.

Drop-in for ``src/shoeprint_image_retrieval/network.py``:

* ``Model(config, block)``                                        (network.py:93-97)
* ``Model.get_feature_maps(img) -> float32 [C, h, w]``            (network.py:210-244)
* ``Model.get_multiple_feature_maps(images, *, progress=True)``   (network.py:246-269)

``block`` is the slice end into ``vgg16().features`` (network.py:185).  The forward pass is the HIP
library's ``spr_vgg16_forward`` (implicit-GEMM 3x3 convolutions on the fp32 matrix cores with bias /
ReLU / max-pool fused, pre-processing fused into the first layer) and — unlike the reference's one
image per launch (network.py:228) — runs whole batches; ``extract_device`` keeps the features in HBM
for the scorer.  Unknown ``model.type`` raises ``LookupError("Model string not found")`` as the
reference does (network.py:180-182).  ``EfficientNetV2_S / _M / _L`` (network.py:163-175; run.toml's default) and
``EfficientNet_B1 .. B5, B7`` (network.py:139-162) run on ``spr_effnet_forward`` (stem, FusedMBConv and MBConv stages;
BatchNorm folded and parameters packed here) and ``DenseNet_201`` (network.py:176-179) on ``spr_densenet_forward``: every
backbone of the reference's list is built.  ``model.type = "ResNet50"`` is BUILD-DEFINED (BASELINE.json config 3 names a ResNet50
layer3 extractor, the reference has none): torchvision's resnet50 cut after ``block`` of its top-level children
[conv1, bn1, relu, maxpool, layer1, layer2, layer3], block = 5 / 6 / 7, ImageNet mean / std.

Weights: torchvision downloads ``IMAGENET1K_FEATURES`` by name (network.py:126), impossible offline.
``config["mi355x"]["weights"]`` may name a local state dict (``features.N.weight/bias``, loaded with
``torch.load(weights_only=True)``); otherwise seeded He-normal weights from ``synth.vgg16_parameters``
are used (and a warning is printed once).
"""

from __future__ import annotations

import ctypes as C
import sys
from typing import Any

import numpy as np

from . import _lib, synth

VGG16_MEAN = (0.48235, 0.45882, 0.40784)                     # network.py:128
VGG16_STD = (0.00392156862745098,) * 3                       # network.py:129
IMAGENET_MEAN = (0.485, 0.456, 0.406)                        # network.py:52 (the default transforms)
IMAGENET_STD = (0.229, 0.224, 0.225)                         # network.py:53
BN_EPS = 1e-5                                                # torch.nn.BatchNorm2d default
# model.type -> (spr_vgg_arch, mean, std): the plain-VGG branches of network.py:121-139
_VGG_MODELS = {"VGG16": (0, VGG16_MEAN, VGG16_STD), "VGG19": (1, IMAGENET_MEAN, IMAGENET_STD),
               "VGG19_BN": (2, IMAGENET_MEAN, IMAGENET_STD)}
# BUILD-DEFINED (BASELINE.json config 3; the reference has no ResNet branch): torchvision's resnet50 cut after `block` of its
# top-level children [conv1, bn1, relu, maxpool, layer1, layer2, layer3] - block 5 / 6 / 7 - with the default transforms
_RESNET_MODELS = {"ResNet50": (IMAGENET_MEAN, IMAGENET_STD)}
# network.py:163-175: arch id of spr_effnet_plan_create, mean, std (EfficientNetV2_L was trained on 0.5 / 0.5)
# network.py:139-175: arch id of spr_effnet_plan_create, mean, std (EfficientNetV2_L was trained on 0.5 / 0.5), BatchNorm eps
# (torchvision builds efficientnet_v2_* and efficientnet_b5 / b6 / b7 with eps = 1e-3, the others with the default 1e-5)
_EFFNET_MODELS = {"EfficientNetV2_S": (0, IMAGENET_MEAN, IMAGENET_STD, 1e-3), "EfficientNetV2_M": (1, IMAGENET_MEAN, IMAGENET_STD, 1e-3),
                  "EfficientNetV2_L": (2, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5), 1e-3),
                  "EfficientNet_B1": (3, IMAGENET_MEAN, IMAGENET_STD, 1e-5), "EfficientNet_B2": (4, IMAGENET_MEAN, IMAGENET_STD, 1e-5),
                  "EfficientNet_B3": (5, IMAGENET_MEAN, IMAGENET_STD, 1e-5), "EfficientNet_B4": (6, IMAGENET_MEAN, IMAGENET_STD, 1e-5),
                  "EfficientNet_B5": (7, IMAGENET_MEAN, IMAGENET_STD, 1e-3), "EfficientNet_B7": (8, IMAGENET_MEAN, IMAGENET_STD, 1e-3)}
_REFERENCE_MODELS = {"EfficientNet_B1", "EfficientNet_B2", "EfficientNet_B3", "EfficientNet_B4",
                     "EfficientNet_B5", "EfficientNet_B7", "EfficientNetV2_S", "EfficientNetV2_M", "EfficientNetV2_L",
                     "DenseNet_201"}
_COMPUTE = {"float32": _lib.F32, "float16": _lib.F16, "bfloat16": _lib.BF16}
_warned = False


def effnet_state_names(ops) -> list[tuple[str, str]]:
    """Module names of the layers in a torchvision efficientnet_v2 state dict, in ``Model.effnet_ops`` order: (convolution,
    BatchNorm) or, for a squeeze-excitation, (fc1, fc2).  Written from torchvision's module layout - features.0 = stem,
    features.s = stage s, .l = block l of the stage, .block.k = its k-th sub-module - and not checked against a checkpoint:
    none is available offline."""
    names, layer, slot = [], {}, 0
    for i, op in enumerate(ops):
        f = op["feature"]
        if f == 0:
            names.append(("features.0.0", "features.0.1"))
            continue
        # the closing 1x1 convolution of `features` is a Conv2dNormActivation of its own, like the stem: a feature that
        # consists of one plain 1x1 convolution with SiLU (inside a stage a 1x1 + SiLU is an expansion, never a block's end)
        if (op["kind"] == 0 and op["ks"] == 1 and op["act"] == 2 and op["block_end"] and i == len(ops) - 1
                and (i == 0 or ops[i - 1]["feature"] != f)):
            names.append((f"features.{f}.0", f"features.{f}.1"))
            continue
        pre = f"features.{f}.{layer.setdefault(f, 0)}.block.{slot}"
        names.append((f"{pre}.fc1", f"{pre}.fc2") if op["kind"] == 2 else (f"{pre}.0", f"{pre}.1"))
        slot += 1
        if op["block_end"]:
            layer[f] += 1
            slot = 0
    return names


def effnet_plan_ops(lib, handle) -> list[dict]:
    """The op list of an spr_effnet_plan (see Model.effnet_ops)."""
    keys = ("kind", "cin", "cout", "cin_p", "cout_p", "ks", "stride", "act", "res", "sq", "feature", "w_off", "b_off",
            "w2_off", "b2_off", "block_end")
    out = []
    for i in range(lib.spr_effnet_num_ops(handle)):
        info = (C.c_int32 * 16)()
        lib.check(lib.spr_effnet_op_info(handle, i, info))
        out.append(dict(zip(keys, list(info))))
    return out


def densenet_plan_ops(lib, handle) -> list[dict]:
    """The op list of an spr_densenet_plan (see Model.densenet_ops)."""
    keys = ("kind", "cin", "cout", "c_off", "ctot", "flags", "feature", "w_off", "b_off", "s_off", "t_off")
    out = []
    for i in range(lib.spr_densenet_num_ops(handle)):
        info = (C.c_int32 * 12)()
        lib.check(lib.spr_densenet_op_info(handle, i, info))
        out.append(dict(zip(keys, list(info))))
    return out


def densenet_state_names(ops) -> list[tuple]:
    """Module names of the layers in a torchvision densenet201 state dict, in ``Model.densenet_ops`` order (written from
    torchvision's module layout; not checked against a checkpoint: none is available offline)."""
    names, layer = [], {}
    for op in ops:
        f = op["feature"]
        if op["kind"] == 0:
            names.append(("features.conv0", "features.norm0"))
        elif op["kind"] == 1:
            b = (f - 4) // 2 + 1
            pre = f"features.denseblock{b}.denselayer{layer.setdefault(b, 0) + 1}"
            names.append((f"{pre}.norm1", f"{pre}.conv1", f"{pre}.norm2"))
        elif op["kind"] == 2:
            b = (f - 4) // 2 + 1
            names.append((f"features.denseblock{b}.denselayer{layer[b] + 1}.conv2",))
            layer[b] += 1
        elif op["kind"] == 3:
            b = (f - 5) // 2 + 1
            names.append((f"features.transition{b}.norm", f"features.transition{b}.conv"))
        else:
            names.append(("features.norm5",))
    return names


class Model:
    """Operate on the truncated VGG16 and its pre-processing (reference network.py:90-269)."""

    def __init__(self, config: dict, block: int, *, device=None, library: _lib.Library | None = None,
                 parameters: list[tuple[np.ndarray, np.ndarray]] | None = None, batch_size: int = 16):
        self.config = config
        model_cfg = config["model"]
        self.clahe_clip_limit = float(model_cfg.get("clahe_clip_limit", 2.0))
        self.clahe_tile_grid_size = tuple(model_cfg.get("clahe_tile_grid_size", (8, 8)))
        model_str = model_cfg["type"]
        self.resnet = model_str in _RESNET_MODELS
        self.effnet = model_str in _EFFNET_MODELS
        self.densenet = model_str == "DenseNet_201"
        if model_str not in _VGG_MODELS and not self.resnet and not self.effnet and not self.densenet:
            if model_str in _REFERENCE_MODELS:
                raise NotImplementedError(f"backbone {model_str} is not built on MI355X yet (SURVEY §8 f4); "
                                          f"use one of {sorted(_VGG_MODELS) + sorted(_EFFNET_MODELS)}")
            raise LookupError("Model string not found")  # network.py:180-182
        self.model_str = model_str
        if self.resnet:
            self.arch, (self.mean, self.std) = -1, _RESNET_MODELS[model_str]
        elif self.effnet:
            self.arch, self.mean, self.std, self.bn_eps = _EFFNET_MODELS[model_str]
        elif self.densenet:
            self.arch, self.mean, self.std, self.bn_eps = -2, IMAGENET_MEAN, IMAGENET_STD, 1e-5
        else:
            self.arch, self.mean, self.std = _VGG_MODELS[model_str]
        self.block = int(block)
        self.batch_size = int(batch_size)
        self.lib = library or _lib.load_library()
        if device is None:
            from .device import TorchDevice

            device = TorchDevice()
        self.dev = device
        # [mi355x].extractor_dtype: compute type of the convolutions behind the first layer ("float32": the f32 matrix cores,
        # exact, the reference's arithmetic; "bfloat16" / "float16": 16-bit operands, f32 accumulation - BASELINE configs 3 / 5)
        self.compute = str((config.get("mi355x") or {}).get("extractor_dtype", "float32") or "float32")
        if self.compute not in _COMPUTE:
            raise ValueError(f"[mi355x].extractor_dtype = {self.compute!r}: expected one of {sorted(_COMPUTE)}")
        if self.compute != "float32" and self.densenet:
            raise NotImplementedError(f"{model_str}: the 16-bit matrix-core path is built for the VGG, ResNet50 and EfficientNet "
                                      "extractors; use [mi355x].extractor_dtype = \"float32\"")
        handle = C.c_void_p()
        if self.densenet:
            self.lib.check(self.lib.spr_densenet_plan_create(self.block, C.byref(handle)))
            self.handle = handle
            self.n_convs = self.lib.spr_densenet_num_ops(handle)
            if parameters is None:
                parameters = self._load_densenet_parameters(config)
            self._set_densenet_parameters(parameters)
            return
        if self.effnet:
            self.lib.check(self.lib.spr_effnet_plan_create_ex(self.arch, self.block, _COMPUTE[self.compute], C.byref(handle)))
            self.handle = handle
            self.n_convs = self.lib.spr_effnet_num_ops(handle)
            if parameters is None:
                parameters = self._load_effnet_parameters(config)
            self._set_effnet_parameters(parameters)
            return
        if self.resnet:
            self.lib.check(self.lib.spr_resnet_plan_create_ex(self.block, _COMPUTE[self.compute], C.byref(handle)))
            self.handle = handle
            self.n_convs = self.lib.spr_resnet_num_convs(handle)
            if parameters is None:
                parameters = self._load_resnet_parameters(config)
            self._set_resnet_parameters(parameters)
            return
        self.lib.check(self.lib.spr_vgg_plan_create_ex(self.arch, self.block, _COMPUTE[self.compute], C.byref(handle)))
        self.handle = handle
        self.n_convs = self.lib.spr_vgg16_num_convs(handle)
        if parameters is None:
            parameters = self._load_parameters(config)
        self._set_parameters(parameters)

    # ------------------------------------------------------------------ EfficientNet B1 .. B7 and V2 (network.py:139-175)
    def effnet_ops(self) -> list[dict]:
        """The flattened layers of features[:block]: kind (0 convolution, 1 depthwise 3x3, 2 squeeze-excitation), real and
        padded widths, kernel size, stride, activation, residual flag, hidden width, index into ``features`` and the offsets
        (floats) of the layer's parameters in the packed buffer."""
        return effnet_plan_ops(self.lib, self.handle)

    def _load_effnet_parameters(self, config):
        global _warned
        path = config.get("mi355x", {}).get("weights", "")
        ops = self.effnet_ops()
        if path:
            import torch

            state = torch.load(path, map_location="cpu", weights_only=True)
            params = []
            for op, names in zip(ops, effnet_state_names(ops)):
                if op["kind"] == 2:
                    fc1, fc2 = names
                    params.append(tuple(state[f"{m}.{n}"].float().numpy() for m in (fc1, fc2) for n in ("weight", "bias")))
                else:
                    conv, bn = names
                    w = state[f"{conv}.weight"].float().numpy()
                    params.append((w, np.zeros(w.shape[0], np.float32)) + tuple(
                        state[f"{bn}.{n}"].float().numpy() for n in ("weight", "bias", "running_mean", "running_var")))
            return params
        if not _warned:
            print(f"shoeprint_image_retrieval_amd: no [mi355x].weights given — using seeded synthetic {self.model_str} "
                  "weights (pretrained ImageNet weights cannot be downloaded offline)", file=sys.stderr)
            _warned = True
        return synth.effnet_parameters(1234, ops)

    def _set_effnet_parameters(self, parameters):
        ops = self.effnet_ops()
        if len(parameters) < len(ops):
            raise ValueError(f"{len(ops)} layers need parameters, got {len(parameters)}")
        packed = np.zeros(self.lib.spr_effnet_packed_bytes(self.handle) // 4, np.float32)
        packed16 = packed.view(np.uint16)  # 16-bit plans: the convolutions' weights as float16 / bfloat16 bit patterns
        half = self.compute != "float32"
        if half and len(ops) < 2:
            raise NotImplementedError("a 16-bit EfficientNet plan needs layers behind the stem (block >= 2)")

        def bits16(a):
            a = np.ascontiguousarray(a, dtype=np.float32)
            return synth.bfloat16_bits(a) if self.compute == "bfloat16" else a.astype(np.float16).view(np.uint16)

        for k_op, (op, p) in enumerate(zip(ops, parameters)):
            p = [np.asarray(t, dtype=np.float32) for t in p]
            if op["kind"] == 2:
                w1, b1, w2, b2 = p
                c, cp, sq = op["cin"], op["cin_p"], op["sq"]
                if w1.reshape(-1).size != sq * c or w2.reshape(-1).size != c * sq:
                    raise ValueError(f"squeeze-excitation of width {c}: parameters do not match (hidden width {sq})")
                a = np.zeros((sq, cp), np.float32); a[:, :c] = w1.reshape(sq, c)
                b = np.zeros((cp, sq), np.float32); b[:c] = w2.reshape(c, sq)
                bb = np.zeros(cp, np.float32); bb[:c] = b2
                packed[op["w_off"]:op["w_off"] + a.size] = a.ravel()
                packed[op["b_off"]:op["b_off"] + sq] = b1
                packed[op["w2_off"]:op["w2_off"] + b.size] = b.ravel()
                packed[op["b2_off"]:op["b2_off"] + cp] = bb
                continue
            w, b, gamma, beta, mu, var = p
            scale = gamma / np.sqrt(var + np.float32(self.bn_eps))  # eval-mode BatchNorm folded into the convolution
            w = w * scale[:, None, None, None]
            b = (b - mu) * scale + beta
            ks, cin, cout, cin_p, cout_p = op["ks"], op["cin"], op["cout"], op["cin_p"], op["cout_p"]
            if op["kind"] == 1:
                if w.shape != (cin, 1, ks, ks):
                    raise ValueError(f"depthwise parameter shape {w.shape} does not match width {cin}, kernel {ks}")
                a = np.zeros((ks * ks, cin_p), np.float32); a[:, :cin] = w.reshape(cin, ks * ks).T
                bb = np.zeros(cin_p, np.float32); bb[:cin] = b
                packed[op["w_off"]:op["w_off"] + a.size] = a.ravel()
                packed[op["b_off"]:op["b_off"] + cin_p] = bb
                continue
            if w.shape != (cout, cin, ks, ks):
                raise ValueError(f"parameter shape {w.shape} does not match conv {cin}->{cout} {ks}x{ks}")
            bb = np.zeros(cout_p, np.float32); bb[:cout] = b
            if half and k_op == 0:
                # the stem of a 16-bit plan: [k / 8][64][8], k = tap * 3 + plane, 27 real values of 32 (stem16_kernel)
                ws = np.zeros((32, 64), np.float32)
                ws[:27, :cout] = w.transpose(2, 3, 1, 0).reshape(27, cout)  # [ky][kx][c][n] -> k = (ky * 3 + kx) * 3 + c
                ws = ws.reshape(4, 8, 64).transpose(0, 2, 1)               # [k / 8][n][k % 8]
                packed16[2 * op["w_off"]:2 * op["w_off"] + ws.size] = bits16(ws).ravel()
                packed[op["b_off"]:op["b_off"] + cout_p] = bb
                continue
            if half:
                # [cout_p / 64][K / 64][64][64] with K index = tap * cin_p + c (conv_gemm16_kernel)
                wk = np.zeros((cout_p, ks * ks, cin_p), np.float32)
                wk[:cout, :, :cin] = w.reshape(cout, cin, ks * ks).transpose(0, 2, 1)
                kk = ks * ks * cin_p
                wk = wk.reshape(cout_p // 64, 64, kk // 64, 64).transpose(0, 2, 1, 3)
                packed16[2 * op["w_off"]:2 * op["w_off"] + wk.size] = bits16(wk).ravel()
                packed[op["b_off"]:op["b_off"] + cout_p] = bb
                continue
            wp = np.zeros((cout_p, ks * ks, cin_p), np.float32)
            wp[:cout, :, :cin] = w.reshape(cout, cin, ks * ks).transpose(0, 2, 1)  # K index = tap * cin_p + c
            k = ks * ks * cin_p
            wp = wp.reshape(cout_p // 64, 64, k // 16, 16).transpose(0, 2, 1, 3)  # [cout/64][K/16][64][16]
            bb = np.zeros(cout_p, np.float32); bb[:cout] = b
            packed[op["w_off"]:op["w_off"] + wp.size] = np.ascontiguousarray(wp).ravel()
            packed[op["b_off"]:op["b_off"] + cout_p] = bb
        self.packed = self.dev.to_device(packed)

    # ------------------------------------------------------------------ DenseNet_201 (network.py:176-179)
    def densenet_ops(self) -> list[dict]:
        """The layers of features[:block]: kind (0 stem, 1 dense 1x1, 2 dense 3x3, 3 transition, 4 closing BatchNorm), widths,
        channel offset / width of the block tensor, stem flags, index into ``features`` and packed offsets (floats)."""
        return densenet_plan_ops(self.lib, self.handle)

    def _load_densenet_parameters(self, config):
        global _warned
        path = config.get("mi355x", {}).get("weights", "")
        ops = self.densenet_ops()
        if path:
            import torch

            state = torch.load(path, map_location="cpu", weights_only=True)
            bn = lambda m: tuple(state[f"{m}.{n}"].float().numpy() for n in ("weight", "bias", "running_mean", "running_var"))
            wt = lambda m: state[f"{m}.weight"].float().numpy()
            params = []
            for op, names in zip(ops, densenet_state_names(ops)):
                if op["kind"] == 0:
                    params.append((wt(names[0]),) + bn(names[1]))
                elif op["kind"] == 1:
                    params.append(bn(names[0]) + (wt(names[1]),) + bn(names[2]))
                elif op["kind"] == 2:
                    params.append((wt(names[0]),))
                elif op["kind"] == 3:
                    params.append(bn(names[0]) + (wt(names[1]),))
                else:
                    params.append(bn(names[0]))
            return params
        if not _warned:
            print(f"shoeprint_image_retrieval_amd: no [mi355x].weights given — using seeded synthetic {self.model_str} "
                  "weights (pretrained ImageNet weights cannot be downloaded offline)", file=sys.stderr)
            _warned = True
        return synth.densenet_parameters(1234, ops)

    def _set_densenet_parameters(self, parameters):
        ops = self.densenet_ops()
        if len(parameters) < len(ops):
            raise ValueError(f"{len(ops)} layers need parameters, got {len(parameters)}")
        eps = np.float32(self.bn_eps)
        packed = np.zeros(self.lib.spr_densenet_packed_bytes(self.handle) // 4, np.float32)

        def affine(gamma, beta, mu, var):  # eval-mode BatchNorm as x * s + t
            s = gamma / np.sqrt(var + eps)
            return s, beta - mu * s

        def gemm_pack(w, cout_p):  # [cout][cin][k][k] -> [cout_p/64][K/16][64][16], K index = tap * cin + c
            cout, cin, k, _ = w.shape
            wp = np.zeros((cout_p, k * k, cin), np.float32)
            wp[:cout] = w.reshape(cout, cin, k * k).transpose(0, 2, 1)
            return np.ascontiguousarray(wp.reshape(cout_p // 64, 64, k * k * cin // 16, 16).transpose(0, 2, 1, 3)).ravel()

        def put(off, a):
            a = np.asarray(a, np.float32).ravel()
            packed[off:off + a.size] = a

        for op, p in zip(ops, parameters):
            p = [np.asarray(t, dtype=np.float32) for t in p]
            if op["kind"] == 0:
                w, b = p[0], np.zeros(64, np.float32)
                if w.shape != (64, 3, 7, 7):
                    raise ValueError(f"conv0 parameter shape {w.shape}")
                if op["flags"] & 1:
                    s, t = affine(*p[1:5])
                    w, b = w * s[:, None, None, None], t
                put(op["w_off"], w.reshape(64, 3, 49).transpose(2, 1, 0))  # [tap][c][n]
                put(op["b_off"], b)
            elif op["kind"] == 1:
                s1, t1 = affine(*p[0:4])
                s2, t2 = affine(*p[5:9])
                w = p[4]
                if w.shape != (128, op["cin"], 1, 1):
                    raise ValueError(f"dense 1x1 parameter shape {w.shape} for {op['cin']} input channels")
                put(op["s_off"], s1); put(op["t_off"], t1)
                put(op["w_off"], gemm_pack(w * s2[:, None, None, None], 128)); put(op["b_off"], t2)
            elif op["kind"] == 2:
                if p[0].shape != (32, 128, 3, 3):
                    raise ValueError(f"dense 3x3 parameter shape {p[0].shape}")
                put(op["w_off"], gemm_pack(p[0], 64))
            elif op["kind"] == 3:
                s, t = affine(*p[0:4])
                if p[4].shape != (op["cout"], op["cin"], 1, 1):
                    raise ValueError(f"transition parameter shape {p[4].shape}")
                put(op["s_off"], s); put(op["t_off"], t)
                put(op["w_off"], gemm_pack(p[4], op["cout"]))
            else:
                s, t = affine(*p[0:4])
                put(op["s_off"], s); put(op["t_off"], t)
        self.packed = self.dev.to_device(packed)

    # ------------------------------------------------------------------ ResNet50 (build-defined)
    def conv_specs(self) -> list[tuple[int, int, int, int, int]]:
        """(cin, cout, ksize, stride, role) of every convolution in torchvision's module order."""
        out = []
        for i in range(self.n_convs):
            v = [C.c_int32() for _ in range(5)]
            self.lib.check(self.lib.spr_resnet_conv_shape(self.handle, i, *[C.byref(t) for t in v]))
            out.append(tuple(t.value for t in v))
        return out

    def _resnet_state_names(self) -> list[tuple[str, str]]:
        """(convolution, BatchNorm) module names in a torchvision resnet50 state dict, in conv_specs order."""
        names = [("conv1", "bn1")]
        for layer in range(self.block - 4):
            for b in range((3, 4, 6)[layer]):
                pre = f"layer{layer + 1}.{b}"
                names += [(f"{pre}.conv1", f"{pre}.bn1"), (f"{pre}.conv2", f"{pre}.bn2"), (f"{pre}.conv3", f"{pre}.bn3")]
                if b == 0:
                    names.append((f"{pre}.downsample.0", f"{pre}.downsample.1"))
        return names

    def _load_resnet_parameters(self, config):
        global _warned
        path = config.get("mi355x", {}).get("weights", "")
        if path:
            import torch

            state = torch.load(path, map_location="cpu", weights_only=True)
            params = []
            for conv, bn in self._resnet_state_names():
                w = state[f"{conv}.weight"].float().numpy()
                b = state[f"{conv}.bias"].float().numpy() if f"{conv}.bias" in state else np.zeros(w.shape[0], np.float32)
                params.append((w, b) + tuple(state[f"{bn}.{n}"].float().numpy()
                                             for n in ("weight", "bias", "running_mean", "running_var")))
            return params
        if not _warned:
            print(f"shoeprint_image_retrieval_amd: no [mi355x].weights given — using seeded synthetic {self.model_str} "
                  "weights (pretrained ImageNet weights cannot be downloaded offline)", file=sys.stderr)
            _warned = True
        return synth.resnet_parameters(1234, self.conv_specs())

    def _set_resnet_parameters(self, parameters):
        specs = self.conv_specs()
        if len(parameters) < len(specs):
            raise ValueError(f"{len(specs)} convolutions need parameters, got {len(parameters)}")
        dev = self.dev
        self._w_dev, self._b_dev = [], []
        for (cin, cout, ks, _stride, _role), p in zip(specs, parameters):
            if len(p) != 6:
                raise ValueError("every ResNet convolution needs (w, b, gamma, beta, running_mean, running_var)")
            w, b, gamma, beta, mu, var = (np.asarray(t, dtype=np.float32) for t in p)
            if w.shape != (cout, cin, ks, ks) or b.shape != (cout,):
                raise ValueError(f"parameter shape {w.shape}/{b.shape} does not match conv {cin}->{cout} {ks}x{ks}")
            scale = gamma / np.sqrt(var + np.float32(BN_EPS))  # eval-mode BatchNorm folded into the convolution
            self._w_dev.append(dev.to_device(np.ascontiguousarray(w * scale[:, None, None, None])))
            self._b_dev.append(dev.to_device(np.ascontiguousarray((b - mu) * scale + beta)))
        n = len(specs)
        wp = (C.c_void_p * n)(*[dev.ptr(t) for t in self._w_dev])
        bp = (C.c_void_p * n)(*[dev.ptr(t) for t in self._b_dev])
        self.packed = dev.empty_bytes(max(16, self.lib.spr_resnet_packed_bytes(self.handle)))
        self.lib.check(self.lib.spr_resnet_pack_weights(self.handle, wp, bp, dev.ptr(self.packed), dev.stream()))
        dev.synchronize()

    # ------------------------------------------------------------------ parameters
    def conv_shapes(self) -> list[tuple[int, int]]:
        out = []
        for i in range(self.n_convs):
            cin, cout = C.c_int32(), C.c_int32()
            self.lib.check(self.lib.spr_vgg16_conv_shape(self.handle, i, C.byref(cin), C.byref(cout)))
            out.append((cin.value, cout.value))
        return out

    def conv_info(self) -> list[tuple[int, bool]]:
        """(index in model.features, BatchNorm2d inside the truncation) of every convolution."""
        out = []
        for i in range(self.n_convs):
            k, bn = C.c_int32(), C.c_int32()
            self.lib.check(self.lib.spr_vgg_conv_info(self.handle, i, C.byref(k), C.byref(bn)))
            out.append((k.value, bool(bn.value)))
        return out

    def _load_parameters(self, config):
        global _warned
        path = config.get("mi355x", {}).get("weights", "")
        if path:
            import torch

            state = torch.load(path, map_location="cpu", weights_only=True)
            params = []
            for k, bn in self.conv_info():
                p = [state[f"features.{k}.weight"].float().numpy(), state[f"features.{k}.bias"].float().numpy()]
                if bn:  # BatchNorm2d at features.<k+1>: gamma, beta, running mean, running variance
                    p += [state[f"features.{k + 1}.{n}"].float().numpy()
                          for n in ("weight", "bias", "running_mean", "running_var")]
                params.append(tuple(p))
            return params
        if not _warned:
            print(f"shoeprint_image_retrieval_amd: no [mi355x].weights given — using seeded synthetic {self.model_str} "
                  "weights (pretrained ImageNet weights cannot be downloaded offline)", file=sys.stderr)
            _warned = True
        return synth.vgg_parameters(1234, self.conv_shapes(), [bn for _, bn in self.conv_info()])

    def _set_parameters(self, parameters):
        shapes = self.conv_shapes()
        if len(parameters) < len(shapes):
            raise ValueError(f"{len(shapes)} convolutions need parameters, got {len(parameters)}")
        dev = self.dev
        self._w_dev, self._b_dev = [], []
        for (cin, cout), (_, bn), p in zip(shapes, self.conv_info(), parameters):
            w = np.ascontiguousarray(p[0], dtype=np.float32)
            b = np.ascontiguousarray(p[1], dtype=np.float32)
            if bn:
                # eval-mode BatchNorm2d after the convolution is y = (x - mean) * gamma / sqrt(var + eps) + beta:
                # folded into the convolution (float32, as the layer itself computes)
                if len(p) != 6:
                    raise ValueError("a convolution followed by BatchNorm2d needs (w, b, gamma, beta, mean, var)")
                gamma, beta, mu, var = (np.asarray(t, dtype=np.float32) for t in p[2:])
                scale = gamma / np.sqrt(var + np.float32(BN_EPS))
                w = np.ascontiguousarray(w * scale[:, None, None, None])
                b = np.ascontiguousarray((b - mu) * scale + beta)
            if w.shape != (cout, cin, 3, 3) or b.shape != (cout,):
                raise ValueError(f"parameter shape {w.shape}/{b.shape} does not match conv {cin}->{cout}")
            self._w_dev.append(dev.to_device(w))
            self._b_dev.append(dev.to_device(b))
        n = len(shapes)
        wp = (C.c_void_p * n)(*[dev.ptr(t) for t in self._w_dev])
        bp = (C.c_void_p * n)(*[dev.ptr(t) for t in self._b_dev])
        self.packed = dev.empty_bytes(max(16, self.lib.spr_vgg16_packed_bytes(self.handle)))
        self.lib.check(self.lib.spr_vgg16_pack_weights(self.handle, wp, bp, dev.ptr(self.packed), dev.stream()))
        dev.synchronize()

    # ------------------------------------------------------------------ shapes
    def output_shape(self, in_h: int, in_w: int) -> tuple[int, int, int]:
        c, h, w = C.c_int32(), C.c_int32(), C.c_int32()
        fn = (self.lib.spr_densenet_output_shape if self.densenet else self.lib.spr_effnet_output_shape if self.effnet else
              self.lib.spr_resnet_output_shape if self.resnet else self.lib.spr_vgg16_output_shape)
        self.lib.check(fn(self.handle, in_h, in_w, C.byref(c), C.byref(h), C.byref(w)))
        return c.value, h.value, w.value

    # ------------------------------------------------------------------ forward
    def extract_device(self, images_dev, in_channels: int = 1):
        """uint8 device batch [N,H,W] (or [N,H,W,3]) -> float32 device features [N,C,h,w] (stays in HBM)."""
        dev = self.dev
        shape = dev.shape(images_dev)
        n, h, w = shape[0], shape[1], shape[2]
        if self.arch >= 0 and not self.effnet and dev.name == "hip" and self.lib is _lib.load_library():
            from . import _torch_ops

            if _torch_ops.enabled() and images_dev.is_contiguous() and len(shape) == (4 if in_channels == 3 else 3):
                # the registered PyTorch-ROCm custom op (csrc/torch_ops.cpp): same plan, same kernels, current stream
                return _torch_ops.load().extract(images_dev, self.packed, self.arch, self.block, [float(m) for m in self.mean],
                                                 [float(s) for s in self.std], _COMPUTE[self.compute])
        c, oh, ow = self.output_shape(h, w)
        out = dev.empty((n, c, oh, ow), np.float32)
        ws_fn = (self.lib.spr_densenet_workspace_bytes if self.densenet else self.lib.spr_effnet_workspace_bytes if self.effnet else
                 self.lib.spr_resnet_workspace_bytes if self.resnet else self.lib.spr_vgg16_workspace_bytes)
        fwd = (self.lib.spr_densenet_forward if self.densenet else self.lib.spr_effnet_forward if self.effnet else
               self.lib.spr_resnet_forward if self.resnet else self.lib.spr_vgg16_forward)
        ws = dev.empty_bytes(max(16, ws_fn(self.handle, n, h, w)))
        mean = (C.c_float * 3)(*self.mean)
        inv_std = (C.c_float * 3)(*[np.float32(1.0) / np.float32(s) for s in self.std])
        self.lib.check(fwd(self.handle, dev.ptr(images_dev), n, h, w, in_channels, mean, inv_std,
                                                  dev.ptr(self.packed), dev.ptr(ws), dev.ptr(out), dev.stream()))
        return out

    def extract_taps_device(self, images_dev, tap_features, in_channels: int = 1):
        """One pass of the (plain VGG) extractor that also returns the activations at ``tap_features`` - slice ends into
        ``model.features`` like ``block``, each just behind a ReLU (16 = conv3_3, 23 = conv4_3, 30 = conv5_3 for VGG16) -
        as float32 device arrays [N, C_l, h_l, w_l], in that order; a tap equal to ``block`` is the network's output.
        Multi-layer scoring (BASELINE config 5) feeds on this: the reference would run the extractor once per block."""
        if self.resnet or self.effnet or self.densenet:
            raise NotImplementedError("feature taps are built for the plain VGG backbones")
        dev = self.dev
        n, h, w = dev.shape(images_dev)[:3]
        info = self.conv_info()          # (index in model.features, BatchNorm inside) per convolution
        shapes = self.conv_shapes()
        c, oh, ow = self.output_shape(h, w)
        out = dev.empty((n, c, oh, ow), np.float32)
        taps, tap_convs, tap_bufs = [], [], []
        for t in tap_features:
            if t == self.block:
                taps.append(out)
                continue
            # the convolution whose ReLU ends the slice features[:t]
            ordinal = [i for i, (k, bn) in enumerate(info) if k + (2 if bn else 1) == t - 1]
            if not ordinal or ordinal[0] == 0:
                raise ValueError(f"tap {t}: not the ReLU behind one of the convolutions 1.. of features[:{self.block}]")
            i = ordinal[0]
            # spatial size of convolution i's output = image size halved once per max-pool in front of it
            pools = sum(1 for j in range(i) if self._stage_pool(j))
            hh, ww = h >> pools, w >> pools
            buf = dev.empty((n, shapes[i][1], hh, ww), np.float32)
            taps.append(buf)
            tap_convs.append(i)
            tap_bufs.append(buf)
        ws = dev.empty_bytes(max(16, self.lib.spr_vgg16_workspace_bytes(self.handle, n, h, w)))
        mean = (C.c_float * 3)(*self.mean)
        inv_std = (C.c_float * 3)(*[np.float32(1.0) / np.float32(s) for s in self.std])
        nt = len(tap_convs)
        self.lib.check(self.lib.spr_vgg16_forward_taps(
            self.handle, dev.ptr(images_dev), n, h, w, in_channels, mean, inv_std, dev.ptr(self.packed), dev.ptr(ws),
            dev.ptr(out), nt, (C.c_int32 * max(1, nt))(*tap_convs), (C.c_void_p * max(1, nt))(*[dev.ptr(b) for b in tap_bufs]),
            dev.stream()))
        return taps

    def _stage_pool(self, i: int) -> bool:
        """Does a max-pool follow convolution i (and its BatchNorm / ReLU) inside features[:block]?"""
        info = self.conv_info()
        k, bn = info[i]
        pool_at = k + (3 if bn else 2)  # where a pool would sit: behind the ReLU
        if i + 1 < len(info):
            return info[i + 1][0] == pool_at + 1  # the next convolution starts right behind a pool, not behind the ReLU
        return pool_at < self.block

    def clahe_device(self, images_dev):
        """CLAHE of a uint8 device batch (network.py:108-111, 197-208) - HIP kernels, stays in HBM.  [N,H,W]: on the
        image; [N,H,W,3] (RGB): on the L channel of its 8-bit L*a*b* form and back (network.py:199-204)."""
        dev = self.dev
        shape = dev.shape(images_dev)
        if len(shape) == 4:
            if shape[3] != 3:
                raise ValueError("colour images must be [N, H, W, 3] (RGB)")
            n, h, w = shape[:3]
            if getattr(self, "_color_tables", None) is None:
                from . import color

                self._color_tables = dev.to_device(color.tables())
            lab = dev.empty((n, h, w, 3), np.uint8)
            self.lib.check(self.lib.spr_rgb_to_lab_u8(dev.ptr(images_dev), dev.ptr(lab), n * h * w, dev.ptr(self._color_tables),
                                                      dev.stream()))
            dev.set_channel(lab, 0, self.clahe_device(dev.channel(lab, 0)))
            rgb = dev.empty((n, h, w, 3), np.uint8)
            self.lib.check(self.lib.spr_lab_to_rgb_u8(dev.ptr(lab), dev.ptr(rgb), n * h * w, dev.ptr(self._color_tables),
                                                      dev.stream()))
            return rgb
        n, h, w = shape
        tx, ty = int(self.clahe_tile_grid_size[0]), int(self.clahe_tile_grid_size[1])
        out = dev.empty((n, h, w), np.uint8)
        ws = dev.empty_bytes(max(16, self.lib.spr_clahe_workspace_bytes(n, tx, ty)))
        self.lib.check(self.lib.spr_clahe_u8(dev.ptr(images_dev), dev.ptr(out), n, h, w, self.clahe_clip_limit, tx, ty,
                                             dev.ptr(ws), dev.stream()))
        return out

    def _clahe(self, img: np.ndarray) -> np.ndarray:
        """CLAHE before the network (network.py:197-208): grey [H,W] or RGB [H,W,3]."""
        batch = self.dev.to_device(np.ascontiguousarray(img, dtype=np.uint8)[None])
        return self.dev.to_host(self.clahe_device(batch))[0]

    def get_feature_maps(self, img: np.ndarray) -> np.ndarray:
        """One image (uint8 [H,W]) -> float32 [C,h,w], a fresh C-contiguous array (network.py:210-244)."""
        return self.get_multiple_feature_maps([img], progress=False)[0]

    def get_multiple_feature_maps(self, images: list[np.ndarray], *, progress: bool = True) -> list[np.ndarray]:
        """List of images -> list of feature stacks (network.py:246-269).  Images of equal size are
        batched through one launch sequence; sizes may differ between images."""
        results: list[Any] = [None] * len(images)
        groups: dict[tuple, list[int]] = {}
        for i, im in enumerate(images):
            groups.setdefault(tuple(im.shape), []).append(i)
        done = 0
        for shape, idx in groups.items():
            for start in range(0, len(idx), self.batch_size):
                part = idx[start:start + self.batch_size]
                if len(shape) not in (2, 3) or (len(shape) == 3 and shape[2] != 3):
                    raise ValueError("images must be grey [H,W] or RGB [H,W,3] uint8 arrays")
                # grey: transform (repeat to three planes); RGB: transform_rgb (network.py:236-241, 60-87)
                batch = self.dev.to_device(np.stack([np.ascontiguousarray(images[i], dtype=np.uint8) for i in part]))
                feats = self.dev.to_host(self.extract_device(self.clahe_device(batch), in_channels=3 if len(shape) == 3 else 1))
                for k, i in enumerate(part):
                    results[i] = np.ascontiguousarray(feats[k])
                done += len(part)
                if progress:
                    print(f"\rfeatures {done}/{len(images)}", end="" if done < len(images) else "\n", file=sys.stderr)
        return results

    def close(self):
        if getattr(self, "handle", None):
            (self.lib.spr_densenet_plan_destroy if getattr(self, "densenet", False) else
             self.lib.spr_effnet_plan_destroy if getattr(self, "effnet", False) else
             self.lib.spr_resnet_plan_destroy if self.resnet else self.lib.spr_vgg16_plan_destroy)(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass
