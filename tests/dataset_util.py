"""Synthetic image directories (Gallery/ + Query/) for the dataloader tests and the golden generator."""
import os

import numpy as np
from PIL import Image

from shoeprint_image_retrieval_amd import synth

MODEL = {"type": "VGG16", "clahe_clip_limit": 2.0, "clahe_tile_grid_size": [8, 8], "start_block": 16, "end_block": 9,
         "skip_blocks": [], "minimum_dim": 120, "maximum_dim": 200}
CASES = [
    {"name": "wvu_two_sizes", "type": "WVU2019", "crop": [0.1, 0.2], "n_clusters": 2, "tolerance": 0.05,
     "gallery": {"001.png": (260, 120), "002.png": (300, 150), "003.png": (220, 100), "004.png": (280, 140)},
     "query": {"001_a.png": (250, 118), "002_b.png": (120, 60), "003_c.png": (244, 110), "004_d.png": (124, 64)}},
    {"name": "wvu_split", "type": "WVU2019", "crop": [0.05, 0.1], "n_clusters": 2, "tolerance": 0.05,
     "gallery": {"001.png": (240, 130), "002.png": (250, 125), "003.png": (236, 128)},
     "query": {"001_a.png": (200, 90), "002_b.png": (190, 96), "003_c.png": (640, 330), "001_d.png": (600, 300),
               "002_e.png": (180, 92)}},
    {"name": "impress_uniform", "type": "Impress", "crop": [0.0, 0.0], "n_clusters": 2, "tolerance": 0.05,
     "gallery": {"1.png": (256, 128), "2.png": (256, 128), "3.png": (256, 128)},
     "query": {"1_x.png": (256, 128), "3_y.png": (256, 128), "2_z.png": (256, 128)}},
]


def write_dataset(root: str, case: dict) -> dict:
    for sub, items in (("Gallery", case["gallery"]), ("Query", case["query"])):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
        for k, (name, (h, w)) in enumerate(sorted(items.items())):
            ident = int(name[:3]) if case["type"] == "WVU2019" else int(name.split("_")[0].split(".")[0])
            img = synth.shoeprint_image(77, ident, 320, 160)  # one print per identity ...
            pil = Image.fromarray(img).resize((w, h), Image.Resampling.BILINEAR)  # ... at this file's size
            if sub == "Query":
                arr = np.array(pil).astype(np.int32) + np.random.default_rng(k).integers(-20, 20, (h, w))
                pil = Image.fromarray(np.clip(arr, 0, 255).astype(np.uint8))
            pil.save(os.path.join(root, sub, name))
    return {"dataset": {"dir": root, "type": case["type"], "crop": case["crop"], "n_processes": 3,
                        "n_clusters": case["n_clusters"], "cluster_minimise_tolerance": case["tolerance"]},
            "model": dict(MODEL), "comparison": {"n_processes": 2, "rotations": None, "scales": None}}
