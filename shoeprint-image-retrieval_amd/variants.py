"""Rotation / scale variants of query feature maps on the device (reference similarity.py:230-284, 321-353).

The reference converts every channel to a Pillow mode-"F" image and calls ``Image.rotate(angle)`` or
``Image.resize((int(w*s), int(h*s)))``.  Here the host restates the few scalars Pillow derives
(affine matrix rounded to 15 decimals and converted to 16.16 fixed point; per-output-pixel window bounds
and double-precision bicubic weights) and two HIP kernels apply them to whole [N,C,h,w] batches in HBM,
bit-identically to Pillow (tests/golden/variants.npz was produced by the real reference).
"""

from __future__ import annotations

import ctypes as C
import math

import numpy as np


# ------------------------------------------------------------------ Pillow Image.rotate (NEAREST, no expand)
def rotate_plan(angle: float, w: int, h: int):
    """(mode, fixed6) for spr_rotate_nearest, following PIL.Image.Image.rotate and Geometry.c affine_fixed."""
    angle = angle % 360.0
    if angle == 0:
        return 0, None
    if angle == 180:
        return 1, None
    if angle in (90, 270) and w == h:
        return (2 if angle == 90 else 3), None
    cx, cy = w / 2, h / 2
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2] = m[0] * -cx + m[1] * -cy + m[2]
    m[5] = m[3] * -cx + m[4] * -cy + m[5]
    m[2] += cx
    m[5] += cy

    def fix(v: float) -> int:
        return int(math.floor(v * 65536.0 + 0.5))

    fixed = [fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5),
             fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5)]
    return 4, fixed


# ------------------------------------------------------------------ Pillow Resample.c, BICUBIC (a = -0.5)
def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_tables(in_size: int, out_size: int):
    """bounds int32 [out][2], coeffs float64 [out][ksize], ksize — precompute_coeffs of Resample.c."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coeffs = np.zeros((out_size, ksize), dtype=np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ww = 0.0
        k = []
        for x in range(xmax):
            wv = _bicubic((x + xmin - center + 0.5) * ss)
            k.append(wv)
            ww += wv
        for x in range(xmax):
            coeffs[xx, x] = k[x] / ww if ww != 0.0 else k[x]
        bounds[xx] = (xmin, xmax)
    return bounds, coeffs, ksize


class VariantBuilder:
    """Applies rotations / scales to device batches [N,C,h,w] through the C ABI."""

    def __init__(self, lib, dev):
        self.lib, self.dev = lib, dev
        self._tables: dict[tuple[int, int], tuple] = {}  # (in, out) -> device bounds / coefficients, kept alive here

    def rotate(self, maps_dev, angle: float):
        n, c, h, w = self.dev.shape(maps_dev)
        mode, fixed = rotate_plan(angle, w, h)
        out = self.dev.empty((n, c, h, w), np.float32)
        arr = (C.c_int64 * 6)(*(fixed or [0] * 6))
        self.lib.check(self.lib.spr_rotate_nearest(self.dev.ptr(maps_dev), self.dev.ptr(out), n * c, h, w, mode, arr,
                                                   self.dev.stream()))
        return out

    def scale(self, maps_dev, factor: float):
        n, c, h, w = self.dev.shape(maps_dev)
        ow, oh = int(w * factor), int(h * factor)  # similarity.py:270-273 (truncation)
        if ow < 1 or oh < 1:
            raise ValueError("height and width must be > 0")  # Pillow's message for an empty target size
        cur, cur_w = maps_dev, w
        if ow != w:  # horizontal pass first, as ImagingResample does
            cur = self._axis(cur, n * c, h, w, 1, ow)
            cur_w = ow
        if oh != h:
            cur = self._axis(cur, n * c, h, cur_w, 0, oh)
        if cur is maps_dev:  # nothing to do: Pillow returns a copy
            cur = self.rotate(maps_dev, 0.0)
        return self._reshape(cur, (n, c, oh, ow))

    def _axis(self, src, n_maps, h, w, axis, out_size):
        key = (w if axis == 1 else h, out_size)
        if key not in self._tables:  # uploaded once per size pair and owned by the builder: no sync after the launch
            bounds, coeffs, ksize = resample_tables(*key)
            self._tables[key] = (self.dev.to_device(bounds), self.dev.to_device(coeffs), ksize)
        b_dev, c_dev, ksize = self._tables[key]
        shape = (n_maps, h, out_size) if axis == 1 else (n_maps, out_size, w)
        out = self.dev.empty(shape, np.float32)
        self.lib.check(self.lib.spr_resample_axis(self.dev.ptr(src), self.dev.ptr(out), n_maps, h, w, axis, out_size,
                                                  self.dev.ptr(b_dev), self.dev.ptr(c_dev), ksize, self.dev.stream()))
        return out

    def _reshape(self, buf, shape):
        return buf.reshape(shape)

    def variants(self, maps_dev, rotations, scales) -> list:
        """Variant batches in the reference's order (similarity.py:321-353, 282): [orig] + rotations,
        [orig] + scales, or — both set — [orig] + every scale of [orig, rot...]; the rotation-only
        variants are dropped by the reference in that mode and therefore here too."""
        if rotations is None and scales is None:
            return [maps_dev]
        if scales is None:
            return [maps_dev] + [self.rotate(maps_dev, r) for r in rotations]
        if rotations is None:
            return [maps_dev] + [self.scale(maps_dev, s) for s in scales]
        rotated = [maps_dev] + [self.rotate(maps_dev, r) for r in rotations]
        return [maps_dev] + [self.scale(v, s) for v in rotated for s in scales]
