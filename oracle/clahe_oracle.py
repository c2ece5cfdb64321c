"""CPU oracle for CLAHE pre-processing (reference network.py:108-111, 197-208: cv2.createCLAHE(...).apply).

TEST INFRASTRUCTURE ONLY (same rules as ncc_oracle.py): the product runs csrc/clahe.hip.
The reference runs OpenCV's CLAHE on the CPU, once per image, before the network; OpenCV is not
installable offline, so this is a numpy restatement of OpenCV's published 8-bit algorithm
(modules/imgproc/src/clahe.cpp, 4.x): reflect-101 padding to a multiple of the tile grid, per-tile
256-bin histogram, clip at max(1, int(clipLimit * tileArea / 256)) with uniform redistribution of the
excess plus the strided residual, cumulative LUT scaled by 255 / tileArea, bilinear interpolation of
the four neighbouring tile LUTs in float32.  PARITY UNPINNED: no cv2 here and the reference holds no
CLAHE fixtures; the HIP kernels are held bit-for-bit to this restatement.
"""

from __future__ import annotations

import numpy as np


def _round_half_even_u8(x: np.ndarray) -> np.ndarray:
    return np.clip(np.rint(x), 0, 255).astype(np.uint8)  # cvRound + saturate_cast<uchar>


def clahe(img: np.ndarray, clip_limit: float = 2.0, tile_grid_size=(8, 8)) -> np.ndarray:
    if img.dtype != np.uint8 or img.ndim != 2:
        raise ValueError("CLAHE expects a 2-D uint8 image")
    tiles_x, tiles_y = int(tile_grid_size[0]), int(tile_grid_size[1])
    h, w = img.shape
    if h % tiles_y or w % tiles_x:
        ext = np.pad(img, ((0, tiles_y - h % tiles_y), (0, tiles_x - w % tiles_x)), mode="reflect")
    else:
        ext = img
    th, tw = ext.shape[0] // tiles_y, ext.shape[1] // tiles_x
    area = th * tw
    clip = 0
    if clip_limit > 0.0:
        clip = max(int(clip_limit * area / 256), 1)
    lut_scale = np.float32(255.0) / np.float32(area)
    luts = np.empty((tiles_y, tiles_x, 256), dtype=np.uint8)
    for ty in range(tiles_y):
        for tx in range(tiles_x):
            tile = ext[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
            hist = np.bincount(tile.ravel(), minlength=256).astype(np.int64)
            if clip > 0:
                excess = int(np.maximum(hist - clip, 0).sum())
                hist = np.minimum(hist, clip)
                batch, residual = divmod(excess, 256)
                hist += batch
                if residual:
                    step = max(256 // residual, 1)
                    idx = np.arange(0, 256, step)[:residual]
                    hist[idx] += 1
            luts[ty, tx] = _round_half_even_u8(np.cumsum(hist).astype(np.float32) * lut_scale)
    ys = np.arange(h, dtype=np.float32) * np.float32(1.0 / th) - np.float32(0.5)
    xs = np.arange(w, dtype=np.float32) * np.float32(1.0 / tw) - np.float32(0.5)
    ty1 = np.floor(ys).astype(np.int64)
    tx1 = np.floor(xs).astype(np.int64)
    ya = (ys - ty1).astype(np.float32)[:, None]
    xa = (xs - tx1).astype(np.float32)[None, :]
    ty2 = np.minimum(ty1 + 1, tiles_y - 1)
    tx2 = np.minimum(tx1 + 1, tiles_x - 1)
    ty1 = np.maximum(ty1, 0)
    tx1 = np.maximum(tx1, 0)
    v = img.astype(np.int64)
    l11 = luts[ty1[:, None], tx1[None, :], v].astype(np.float32)
    l12 = luts[ty1[:, None], tx2[None, :], v].astype(np.float32)
    l21 = luts[ty2[:, None], tx1[None, :], v].astype(np.float32)
    l22 = luts[ty2[:, None], tx2[None, :], v].astype(np.float32)
    one = np.float32(1.0)
    res = (l11 * (one - xa) + l12 * xa) * (one - ya) + (l21 * (one - xa) + l22 * xa) * ya
    return _round_half_even_u8(res)
