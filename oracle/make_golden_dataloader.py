#!/usr/bin/env python3
"""Golden vectors for the dataloader restatement: runs the REAL reference Dataloader (build container only)
on a synthetic image directory written by tests/dataset_util.py and records what it selects and yields.
``python oracle/make_golden_dataloader.py`` -> tests/golden/dataloader.json"""
import contextlib, hashlib, io, json, os, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, "/root/reference")
import numpy as np
from src.shoeprint_image_retrieval.dataloader import Dataloader as RefLoader  # noqa: E402
import dataset_util  # noqa: E402

out = {"cases": []}
for case in dataset_util.CASES:
    with tempfile.TemporaryDirectory() as tmp:
        cfg = dataset_util.write_dataset(tmp, case)
        cfg["dataset"]["n_processes"] = 1  # the reference's chunking is only correct when the count divides evenly
        with contextlib.redirect_stdout(io.StringIO()):
            dl = RefLoader(cfg)
            steps = []
            for q, g, m, block in dl:
                steps.append({"block": int(block), "matches": [int(v) for v in m],
                              "q_shapes": [list(a.shape) for a in q], "g_shapes": [list(a.shape) for a in g],
                              "q_sha": [hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest() for a in q],
                              "g_sha": [hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest() for a in g]})
        out["cases"].append({"name": case["name"], "scales": [float(s) for s in dl.scales], "blocks": [int(b) for b in dl.blocks],
                             "clusters": [sorted(c) for c in dl.clusters], "steps": steps})
        print(case["name"], dl.scales, dl.blocks, [len(c) for c in dl.clusters])
# Algorithm 1 on its own: a grid of (smallest, largest) pairs
dl = RefLoader.__new__(RefLoader)
dl.config = {"model": {"maximum_dim": 800, "end_block": 4, "skip_blocks": [5], "minimum_dim": 300, "start_block": 6}}
grid = []
for small in (40, 80, 140, 160, 299, 300, 420, 640, 900):
    for large in (small, small * 2, 700, 801, 1600, 3200):
        if large >= small:
            s, b = dl._find_best_scale(small, large, 300, 6)
            grid.append([small, large, float(s), int(b)])
out["find_best_scale"] = {"config": dl.config["model"], "grid": grid}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "dataloader.json"), "w"), indent=1)
