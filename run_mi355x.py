#!/usr/bin/env python3
"""Run shoeprint image retrieval on MI355X — the reference's ``run.py`` (run.py:1-34) on this package.

    python run_mi355x.py [run.toml]

Same flow: load the config, iterate the size clusters of the query set, extract features of the queries and
of the whole gallery with the truncated network chosen for the cluster, rank every query's true match, print
the S-scores of that cluster against the whole-data-set totals.  Additionally prints rank-1 and mAP over all
clusters at the end.
"""

import os
import sys

from shoeprint_image_retrieval_amd.config import load_config
from shoeprint_image_retrieval_amd import feature_cache
from shoeprint_image_retrieval_amd.dataloader import Dataloader
from shoeprint_image_retrieval_amd.network import Model
from shoeprint_image_retrieval_amd.parse_results import cmp_all, mean_average_precision, rank1
from shoeprint_image_retrieval_amd.similarity import compare_maps


def main(config_file: str = "run.toml") -> list[int]:
    config = load_config(config_file)
    dataloader = Dataloader(config)
    print(f"{dataloader.num_clusters} clusters of image sizes found.")
    all_ranks: list[int] = []
    cache_dir = config["mi355x"].get("gallery_cache", "")
    for cluster, (shoemark_images, shoeprint_images, matching_shoeprint_ids, block) in enumerate(dataloader):
        print(f"Cluster has {len(shoemark_images)} items.")
        model = Model(config, block)
        shoemark_features = model.get_multiple_feature_maps(shoemark_images)
        # gallery features depend only on what is in `key`: keep them across runs when [mi355x].gallery_cache is set
        key = {"files": dataloader.shoeprint_files, "scale": dataloader.scales[cluster], "block": block,
               "crop": list(config["dataset"]["crop"]), "model": dict(config["model"]),
               "weights": config["mi355x"].get("weights", ""),
               "extractor_dtype": config["mi355x"].get("extractor_dtype", "float32")}
        path = os.path.join(cache_dir, f"gallery_block{block}_cluster{cluster}.f32") if cache_dir else ""
        shoeprint_features = feature_cache.load_features(path, key) if path else None
        if shoeprint_features is None:
            shoeprint_features = model.get_multiple_feature_maps(shoeprint_images)
            if path:
                os.makedirs(cache_dir, exist_ok=True)
                feature_cache.save_features(path, shoeprint_features, key)
        else:
            print(f"Gallery features from {path}")
        print("Calculating ranks:")
        ranks = compare_maps(shoemark_features, shoeprint_features, matching_shoeprint_ids, config, progress=True)
        cmp_all(list(ranks), total_shoeprints=len(dataloader.shoeprint_files),
                total_shoemarks=len(dataloader.shoemark_files))
        all_ranks += [int(r) for r in ranks]
    if all_ranks:
        print(f"rank-1: {rank1(all_ranks):.4f}  mAP: {mean_average_precision(all_ranks):.4f}  ({len(all_ranks)} queries)")
    return all_ranks


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "run.toml")
