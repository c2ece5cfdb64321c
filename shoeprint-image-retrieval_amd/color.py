"""Tables of the 8-bit RGB <-> L*a*b* kernels (csrc/color.hip): sRGB transfer function, f(t) of CIE L*a*b*, the sRGB -> XYZ
(D65) matrix normalised by the white point, and the inverse transfer function.  int32, in the order the kernels read them."""

from __future__ import annotations

import numpy as np

_M = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
_WHITE = np.array([0.950456, 1.0, 1.088754])


def tables() -> np.ndarray:
    x = np.arange(256, dtype=np.float64) / 255.0
    lin = np.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4)
    gamma = np.rint(255.0 * 8.0 * lin).astype(np.int64)
    t = np.arange(3072, dtype=np.float64) / (255.0 * 8.0)
    f = np.where(t < 0.008856, t * 7.787 + 0.13793103448275862, np.cbrt(t))
    cbrt = np.minimum(np.rint(32768.0 * f), 65535).astype(np.int64)
    coeff = np.rint(4096.0 * _M / _WHITE[:, None]).astype(np.int64).ravel()
    y = np.arange(4096, dtype=np.float64) / 4095.0
    srgb = np.where(y <= 0.0031308, 12.92 * y, 1.055 * y ** (1.0 / 2.4) - 0.055)
    inv_gamma = np.clip(np.rint(255.0 * srgb), 0, 255).astype(np.int64)
    return np.concatenate([gamma, cbrt, coeff, inv_gamma]).astype(np.int32)
