"""run.toml loader with the reference's schema (config.py:11-64) plus an optional ``[mi355x]`` table.

The reference reads the file with the ``toml`` package; ``tomli`` (same TOML 1.0 grammar) is what
this image ships.  As in the reference, ``rotations = ""`` / ``scales = ""`` mean "none"
(config.py:60-63).  Reference files stay valid: keys under ``[mi355x]`` are build-only extras
(``dtype``, ``ncc_method``, ``max_prepared_gib``, ``weights``) and all have defaults.
"""

from __future__ import annotations

from pathlib import Path
from typing import Any, TypedDict

import tomli


class DatasetConfig(TypedDict, total=True):
    dir: str
    type: str
    crop: list[float]
    n_processes: int
    n_clusters: int
    cluster_minimise_tolerance: float


class ModelConfig(TypedDict, total=True):
    type: str
    clahe_clip_limit: float
    clahe_tile_grid_size: list[int]
    start_block: int
    end_block: int
    skip_blocks: list[int]
    minimum_dim: int
    maximum_dim: int


class ComparisonConfig(TypedDict, total=True):
    n_processes: int
    rotations: list[int] | None
    scales: list[float] | None


class Mi355xConfig(TypedDict, total=False):
    dtype: str
    ncc_method: str
    max_prepared_gib: float
    weights: str
    gallery_cache: str  # directory for persisted gallery features (feature_cache.py); "" = off
    extractor_dtype: str  # "float32" (the reference's arithmetic) | "bfloat16" | "float16": compute type of the extractor


class Config(TypedDict, total=False):
    dataset: DatasetConfig
    model: ModelConfig
    comparison: ComparisonConfig
    mi355x: Mi355xConfig


MI355X_DEFAULTS: dict[str, Any] = {"dtype": "float32", "ncc_method": "auto", "max_prepared_gib": 0.0, "weights": "",
                                   "gallery_cache": "", "extractor_dtype": "float32"}


def normalise(raw: dict) -> Config:
    comp = raw.setdefault("comparison", {})
    for key in ("rotations", "scales"):
        if comp.get(key, "") == "":
            comp[key] = None
    extra = dict(MI355X_DEFAULTS)
    extra.update(raw.get("mi355x", {}))
    raw["mi355x"] = extra
    return raw  # type: ignore[return-value]


def load_config(config_file: Path | str) -> Config:
    with Path(config_file).open("rb") as fh:
        return normalise(tomli.load(fh))
