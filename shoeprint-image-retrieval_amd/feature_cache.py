"""Persisted gallery features: one flat little-endian file of float32 / float16 maps + a JSON index.

The reference re-extracts every gallery print for every size cluster of every run (run.py:20-26); gallery
features depend only on (image file, crop, scale, block, weights), so a run can keep them.
The file is memory-mapped on load — items are views, nothing is read until the scorer uploads them — and
holds the FEATURES only: the scorer's prepared form (spectra + 1/sigma maps) is specific to one
(query shape, gallery shape) plan and is rebuilt on the device in ~40 us per item and channel-stack.

Layout: ``<path>``      raw item data, each item C-contiguous [C,h,w], 256-byte aligned
        ``<path>.json`` {"version": 1, "key": {...caller's identity of the extraction...},
                         "dtype": "float32", "items": [{"offset": o, "shape": [C,h,w]}, ...]}
"""

from __future__ import annotations

import json
import os
from typing import Any, Sequence

import numpy as np

VERSION = 1
_ALIGN = 256


def save_features(path: str, maps: Sequence[np.ndarray], key: dict[str, Any] | None = None, dtype=np.float32) -> None:
    """Write ``maps`` (list of [C,h,w] arrays, shapes may differ) to ``path`` + ``path.json`` atomically."""
    dtype = np.dtype(dtype)
    if dtype not in (np.dtype(np.float32), np.dtype(np.float16)):
        raise ValueError("feature cache stores float32 or float16")
    items, offset = [], 0
    tmp = f"{path}.tmp.{os.getpid()}"
    with open(tmp, "wb") as fh:
        for m in maps:
            a = np.ascontiguousarray(m, dtype=dtype)
            if a.ndim != 3:
                raise ValueError(f"feature maps are [C,h,w]; got shape {a.shape}")
            pad = (-offset) % _ALIGN
            fh.write(b"\0" * pad)
            offset += pad
            items.append({"offset": offset, "shape": list(a.shape)})
            fh.write(a.tobytes())
            offset += a.nbytes
    with open(f"{tmp}.json", "w") as fh:
        json.dump({"version": VERSION, "key": key or {}, "dtype": dtype.name, "bytes": offset, "items": items}, fh)
    os.replace(tmp, path)
    os.replace(f"{tmp}.json", f"{path}.json")


def load_features(path: str, key: dict[str, Any] | None = None) -> list[np.ndarray] | None:
    """Memory-mapped items of a cache written by ``save_features``; ``None`` if the cache is absent, was
    written for a different ``key``, or does not match its index (a stale or truncated file is never trusted)."""
    try:
        with open(f"{path}.json") as fh:
            index = json.load(fh)
        size = os.path.getsize(path)
    except (OSError, ValueError):
        return None
    if index.get("version") != VERSION or (key is not None and index.get("key") != _jsonable(key)):
        return None
    if size != index.get("bytes"):
        return None
    dtype = np.dtype(index["dtype"])
    if size == 0:
        return []
    flat = np.memmap(path, dtype=np.uint8, mode="r")
    out = []
    for it in index["items"]:
        n = int(np.prod(it["shape"])) * dtype.itemsize
        if it["offset"] + n > size:
            return None
        out.append(flat[it["offset"]: it["offset"] + n].view(dtype).reshape(it["shape"]))
    return out


def _jsonable(obj):
    return json.loads(json.dumps(obj))
