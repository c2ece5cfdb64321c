"""CPU oracle of the RGB route of CLAHE (network.py:199-204): 8-bit RGB -> L*a*b*, CLAHE on L, L*a*b* -> RGB.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED by the reference: cv2 is not importable offline and the reference holds no
fixtures; this restates the published 8-bit convention (L * 255/100, a + 128, b + 128; sRGB primaries, D65) with the
fixed-point, table-driven forward transform OpenCV's 8-bit path is built on and a float32 inverse (one rounding per
operation), i.e. the arithmetic csrc/color.hip is held to bit for bit."""

from __future__ import annotations

import numpy as np

M = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
WHITE = np.array([0.950456, 1.0, 1.088754])


def _tables():
    x = np.arange(256, dtype=np.float64) / 255.0
    gamma = np.rint(255.0 * 8.0 * np.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4)).astype(np.int64)
    t = np.arange(3072, dtype=np.float64) / (255.0 * 8.0)
    cbrt = np.minimum(np.rint(32768.0 * np.where(t < 0.008856, t * 7.787 + 0.13793103448275862, np.cbrt(t))), 65535).astype(np.int64)
    coeff = np.rint(4096.0 * M / WHITE[:, None]).astype(np.int64)
    y = np.arange(4096, dtype=np.float64) / 4095.0
    inv = np.clip(np.rint(255.0 * np.where(y <= 0.0031308, 12.92 * y, 1.055 * y ** (1.0 / 2.4) - 0.055)), 0, 255).astype(np.int64)
    return gamma, cbrt, coeff, inv


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def rgb_to_lab(rgb: np.ndarray) -> np.ndarray:
    gamma, cbrt, c, _ = _tables()
    R, G, B = (gamma[rgb[..., k].astype(np.int64)] for k in range(3))
    fx = cbrt[_descale(R * c[0, 0] + G * c[0, 1] + B * c[0, 2], 12)]
    fy = cbrt[_descale(R * c[1, 0] + G * c[1, 1] + B * c[1, 2], 12)]
    fz = cbrt[_descale(R * c[2, 0] + G * c[2, 1] + B * c[2, 2], 12)]
    lscale, lshift = (116 * 255 + 50) // 100, -((16 * 255 * (1 << 15) + 50) // 100)
    L = _descale(lscale * fy + lshift, 15)
    a = _descale(500 * (fx - fy) + 128 * (1 << 15), 15)
    b = _descale(200 * (fy - fz) + 128 * (1 << 15), 15)
    return np.clip(np.stack([L, a, b], axis=-1), 0, 255).astype(np.uint8)


def lab_to_rgb(lab: np.ndarray) -> np.ndarray:
    f32 = np.float32
    _, _, _, inv = _tables()
    L = lab[..., 0].astype(f32) * f32(100.0) / f32(255.0)
    a = (lab[..., 1].astype(np.int32) - 128).astype(f32)
    b = (lab[..., 2].astype(np.int32) - 128).astype(f32)
    fy = (L + f32(16.0)) / f32(116.0)
    fx = fy + a / f32(500.0)
    fz = fy + (-(b / f32(200.0)))

    def finv(t):
        return np.where(t > f32(0.20689655), (t * t) * t, (t + f32(-0.13793103)) / f32(7.787)).astype(f32)

    X, Y, Z = f32(0.950456) * finv(fx), finv(fy), f32(1.088754) * finv(fz)
    lin = [(f32(3.240479) * X + f32(-1.53715) * Y) + f32(-0.498535) * Z,
           (f32(-0.969256) * X + f32(1.875991) * Y) + f32(0.041556) * Z,
           (f32(0.055648) * X + f32(-0.204043) * Y) + f32(1.057311) * Z]
    out = []
    for v in lin:
        v = np.clip(v, f32(0.0), f32(1.0)).astype(f32)
        idx = np.minimum((v * f32(4095.0) + f32(0.5)).astype(np.int64), 4095)
        out.append(inv[idx])
    return np.stack(out, axis=-1).astype(np.uint8)


def clahe_rgb(img: np.ndarray, clip_limit: float, grid) -> np.ndarray:
    """network.py:199-204 on an RGB image."""
    from . import clahe_oracle

    lab = rgb_to_lab(img)
    lab[..., 0] = clahe_oracle.clahe(np.ascontiguousarray(lab[..., 0]), clip_limit, grid)
    return lab_to_rgb(lab)
