"""Dataset loading: size clustering, scale / block selection, crop, resize, id parsing.

Host-side mirror of the reference's ``dataloader.py`` (``Dataloader(config)`` is an iterator over size
clusters of the query images; each step yields ``(query images, ALL gallery images at the cluster's scale,
gallery index of every query's true match, network block)`` — dataloader.py:26-113).  This is CPU I/O that
runs once per image; it uses Pillow like the reference does.

Behaviour kept from the reference because it decides the scale and block (and therefore the ranks):
  * "Algorithm 1" scale / block search incl. its asymmetric skip-block loops       (dataloader.py:366-419)
  * cluster merging by |scale difference| <= tolerance and equal block               (:314-362)
  * ``_image_extremes``: Pillow's ``size`` is (width, height) but is unpacked as (height, width), so the two
    crop ratios are applied to the opposite axes, and the ``elif`` lets one image update only one extreme (:446-464)
  * crop box from floor(size*ratio), resize to int(size*scale) with LANCZOS            (:217-237)
  * ids: WVU2019 = first 3 characters, Impress = before the first "_" / ".", FID-300 = stem + label_table.csv (:245-250, 101-107)
Deliberate differences:
  * K-means is seeded (``random_state=0``; the reference's is unseeded, dataloader.py:284); clusters are
    reported in order of first appearance in the directory listing, as the reference's dict does.
  * the directory listing is sorted before clustering (``os.listdir`` order is file-system dependent).
  * images are loaded by a thread pool over correct chunks: the reference's chunking drops or mis-sizes items
    unless ``len(files) % n_processes == 0`` and then papers over it (:143, :178-181).
"""

from __future__ import annotations

import csv
import os
from concurrent.futures import ThreadPoolExecutor
from math import floor
from pathlib import Path
from typing import Any

import numpy as np
from PIL import Image


class Dataloader:
    """Initialise, pre-process and provide access to datasets (reference dataloader.py:26-113)."""

    def __init__(self, config: dict) -> None:
        self.config = config
        self.dataset_dir = Path(config["dataset"]["dir"])
        self.shoeprint_dir = self.dataset_dir / "Gallery"
        self.shoemark_dir = self.dataset_dir / "Query"
        self.shoeprint_files = sorted(os.listdir(self.shoeprint_dir))
        self.shoemark_files = sorted(os.listdir(self.shoemark_dir))
        print("The dataset contains: \n", f"    {len(self.shoeprint_files)} reference shoeprints\n",
              f"    {len(self.shoemark_files)} shoemarks")
        clustered = self._cluster_images_by_size(self.shoemark_dir, config["dataset"]["n_clusters"])
        self.scales, self.blocks, self.clusters = self._minimise_clusters(clustered)
        self.num_clusters = len(self.clusters)
        self._current_cluster = 0

    def __iter__(self):
        return self

    def __next__(self) -> tuple[list[np.ndarray], list[np.ndarray], list[int], int]:
        if self._current_cluster >= self.num_clusters:
            raise StopIteration
        scale = self.scales[self._current_cluster]
        shoemark_images, shoemark_ids = self._load_images(self.clusters[self._current_cluster], self.shoemark_dir, scale)
        shoeprint_images, shoeprint_ids = self._load_images(self.shoeprint_files, self.shoeprint_dir, scale)
        if self.config["dataset"]["type"] != "FID-300":
            # ValueError if a query's id has no gallery item, as list.index does in the reference (:99)
            matching_pairs = [shoeprint_ids.index(i) for i in shoemark_ids]
        else:
            table: dict[int, int] = {}
            with (self.dataset_dir / "label_table.csv").open() as fh:
                for row in csv.reader(fh):
                    table[int(row[0])] = int(row[1])
            matching_pairs = [table[i] - 1 for i in shoemark_ids]
        block = self.blocks[self._current_cluster]
        self._current_cluster += 1
        return shoemark_images, shoeprint_images, matching_pairs, block

    # ------------------------------------------------------------------ loading
    def _load_images(self, image_files: list[str], image_directory: Path, scale: float):
        files = sorted(image_files)
        crop = self.config["dataset"]["crop"]
        kind = self.config["dataset"]["type"]
        workers = max(1, min(int(self.config["dataset"].get("n_processes", 1)), len(files) or 1))
        with ThreadPoolExecutor(workers) as pool:
            loaded = list(pool.map(lambda f: _load_one(image_directory / f, f, scale, crop, kind), files))
        return [im for im, _ in loaded], [i for _, i in loaded]

    # ------------------------------------------------------------------ clustering
    def _cluster_images_by_size(self, image_dir: Path, n_clusters: int) -> dict[int, list[str]]:
        from sklearn.cluster import KMeans

        sizes, names = [], []
        for name in sorted(os.listdir(image_dir)):
            with Image.open(image_dir / name) as im:
                width, height = im.size
            sizes.append([min(width, height)])  # group by the smallest image dimension (:275-279)
            names.append(name)
        labels = KMeans(n_clusters=n_clusters, random_state=0, n_init=10).fit(sizes).labels_
        clusters: dict[int, list[str]] = {}
        for name, label in zip(names, labels):
            clusters.setdefault(int(label), []).append(name)
        return clusters

    def _minimise_clusters(self, clusters: dict[int, list[str]]):
        tolerance = self.config["dataset"]["cluster_minimise_tolerance"]
        scales: list[float] = []
        blocks: list[int] = []
        groups: list[list[str]] = []
        largest_print, smallest_print = self._image_extremes(self.shoeprint_files, self.shoeprint_dir)
        for files in clusters.values():
            largest_mark, smallest_mark = self._image_extremes(files, self.shoemark_dir)
            smallest_dim = min(smallest_mark[1], smallest_print[1])
            largest_dim = max(largest_mark[1], largest_print[1])
            scale, block = self._find_best_scale(smallest_dim, largest_dim, minimum_dim=self.config["model"]["minimum_dim"],
                                                 block=self.config["model"]["start_block"])
            for index, other in enumerate(scales):  # first selected scale within the tolerance (:314-319)
                if abs(scale - other) <= tolerance:
                    if blocks[index] == block:
                        groups[index] += files
                    else:
                        scales.append(scale); blocks.append(block); groups.append(list(files))
                    break
            else:
                scales.append(scale); blocks.append(block); groups.append(list(files))
        return scales, blocks, groups

    def _find_best_scale(self, smallest_dim: int, largest_dim: int, minimum_dim: int, block: int) -> tuple[float, int]:
        """Ideal input scale and network block ("Algorithm 1 of the paper", dataloader.py:366-419)."""
        maximum_dim = self.config["model"]["maximum_dim"]
        end_block = self.config["model"]["end_block"]
        skip_blocks = self.config["model"]["skip_blocks"]
        scale: float = 1
        if smallest_dim < minimum_dim:
            if block > end_block:
                block -= 1
                while block in skip_blocks:
                    block -= 1
                return self._find_best_scale(smallest_dim, largest_dim, int(minimum_dim / 2), block)
            return 1, block
        if largest_dim > maximum_dim:
            scale = maximum_dim / largest_dim
            if smallest_dim * scale < minimum_dim:
                if block > end_block:
                    block -= 1
                    while block in skip_blocks and block != end_block:
                        block -= 1
                else:
                    scale = minimum_dim / smallest_dim
        return scale, block

    def _image_extremes(self, image_files: list[str], image_directory: Path):
        """((largest name, largest dim), (smallest name, smallest dim)) after cropping — with the reference's
        axis swap and one-update-per-image rule (see the module docstring)."""
        crop = self.config["dataset"]["crop"]
        largest_dim, largest_name = 0, ""
        smallest_dim, smallest_name = 2**31 - 1, ""
        for name in image_files:
            with Image.open(image_directory / name) as im:
                height, width = im.size  # sic: Pillow returns (width, height)
            height -= floor(height * crop[0] * 2)
            width -= floor(width * crop[1] * 2)
            big, small = max(width, height), min(width, height)
            if big > largest_dim:
                largest_name, largest_dim = name, big
            elif small < smallest_dim:
                smallest_name, smallest_dim = name, small
        return (largest_name, largest_dim), (smallest_name, smallest_dim)


def _load_one(path: Path, name: str, scale: float, crop, dataset_type: str) -> tuple[np.ndarray, int]:
    with Image.open(path) as image:
        crop_height = floor(image.height * crop[0])
        crop_width = floor(image.width * crop[1])
        image = image.crop((crop_width, crop_height, image.width - crop_width, image.height - crop_height))
        resized = image.resize((int(image.width * scale), int(image.height * scale)), Image.Resampling.LANCZOS)
        array = np.array(resized)
    if dataset_type == "Impress":
        ident = int(name.split("_")[0].split(".")[0])
    elif dataset_type == "WVU2019":
        ident = int(name[:3])
    elif dataset_type == "FID-300":
        ident = int(name[:-4])
    else:
        raise ValueError(f"unknown dataset type {dataset_type!r} (FID-300, Impress or WVU2019)")
    return array, ident
