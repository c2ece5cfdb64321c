"""Dataloader restatement against outputs of the reference's own Dataloader on the same synthetic directories
(tests/golden/dataloader.json, made by oracle/make_golden_dataloader.py in the build container)."""
import contextlib
import hashlib
import io
import json
import os

import numpy as np
import pytest

import dataset_util
from shoeprint_image_retrieval_amd.dataloader import Dataloader

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dataloader.json")))


def _sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("case", dataset_util.CASES, ids=lambda c: c["name"])
def test_selection_and_images_match_reference(tmp_path, case):
    want = next(c for c in GOLDEN["cases"] if c["name"] == case["name"])
    config = dataset_util.write_dataset(str(tmp_path), case)
    with contextlib.redirect_stdout(io.StringIO()):
        loader = Dataloader(config)
    assert loader.scales == pytest.approx(want["scales"], abs=0)
    assert loader.blocks == want["blocks"]
    assert [sorted(c) for c in loader.clusters] == want["clusters"]
    assert loader.num_clusters == len(want["steps"])
    for step, (queries, gallery, matches, block) in zip(want["steps"], loader):
        assert block == step["block"] and matches == step["matches"]
        assert [list(a.shape) for a in queries] == step["q_shapes"]
        assert [list(a.shape) for a in gallery] == step["g_shapes"]
        assert [_sha(a) for a in queries] == step["q_sha"]  # crop + LANCZOS resize, byte for byte
        assert [_sha(a) for a in gallery] == step["g_sha"]
    with pytest.raises(StopIteration):
        next(loader)


def test_find_best_scale_grid():
    loader = Dataloader.__new__(Dataloader)
    loader.config = {"model": GOLDEN["find_best_scale"]["config"]}
    model = loader.config["model"]
    for small, large, scale, block in GOLDEN["find_best_scale"]["grid"]:
        got = loader._find_best_scale(small, large, model["minimum_dim"], model["start_block"])
        assert (float(got[0]), got[1]) == (scale, block), (small, large)


def test_uneven_worker_counts_keep_every_image(tmp_path):
    # the reference's chunking loses or mis-sizes items unless the count divides by n_processes; here it may not
    case = dataset_util.CASES[1]
    config = dataset_util.write_dataset(str(tmp_path), case)
    for workers in (1, 2, 4, 7):
        config["dataset"]["n_processes"] = workers
        with contextlib.redirect_stdout(io.StringIO()):
            steps = list(Dataloader(config))
        assert sum(len(q) for q, *_ in steps) == len(case["query"])
        assert all(len(g) == len(case["gallery"]) for _, g, *_ in steps)


def test_unknown_dataset_type_and_missing_match(tmp_path):
    case = dict(dataset_util.CASES[2])
    config = dataset_util.write_dataset(str(tmp_path), case)
    config["dataset"]["type"] = "Nope"
    with contextlib.redirect_stdout(io.StringIO()):
        loader = Dataloader(config)
    with pytest.raises(ValueError):
        next(loader)
    os.remove(os.path.join(tmp_path, "Gallery", "3.png"))  # query 3_y.png now has no gallery item
    config["dataset"]["type"] = "Impress"
    with contextlib.redirect_stdout(io.StringIO()):
        loader = Dataloader(config)
    with pytest.raises(ValueError):
        next(loader)


def test_fid300_label_table(tmp_path):
    case = {"name": "fid", "type": "Impress", "crop": [0.0, 0.0], "n_clusters": 1, "tolerance": 0.05,
            "gallery": {"1.png": (128, 64), "2.png": (128, 64), "3.png": (128, 64)}, "query": {"1.png": (128, 64), "2.png": (128, 64)}}
    config = dataset_util.write_dataset(str(tmp_path), case)
    config["dataset"]["type"] = "FID-300"
    with open(os.path.join(tmp_path, "label_table.csv"), "w") as fh:
        fh.write("1,3\n2,1\n")
    with contextlib.redirect_stdout(io.StringIO()):
        _, _, matches, _ = next(Dataloader(config))
    assert matches == [2, 0]  # 1-based gallery numbers in the table (dataloader.py:101-107)
