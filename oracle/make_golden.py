#!/usr/bin/env python3
"""Generate tests/golden/* by running the REAL reference scorer.

Build-container only: imports ``/root/reference/src/shoeprint_image_retrieval``
(similarity.py, parse_results.py — SURVEY §8c says both import here; network.py
needs cv2/torchvision and does not).  The reference never travels to the GPU
box; what is committed are the small input/output vectors this script writes,
plus this script.  Large inputs are not stored: they are regenerated from the
seeded generator in ``shoeprint_image_retrieval_amd.synth`` (the seed and shape
are stored instead).

Run:  python oracle/make_golden.py            (takes ~1 minute)
"""

from __future__ import annotations

import contextlib
import io
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("SPR_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from src.shoeprint_image_retrieval import parse_results as ref_parse  # noqa: E402
from src.shoeprint_image_retrieval import similarity as ref_sim  # noqa: E402

from shoeprint_image_retrieval_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED = 1234


def half_normal(rng, shape):
    return np.maximum(rng.standard_normal(shape), 0).astype(np.float32)


def golden_normxcorr():
    """(1) full NCC maps for odd / even / ragged shapes and degenerate channels."""
    rng = np.random.default_rng(SEED)
    cases = {}
    shapes = [((28, 12), (28, 12)), ((27, 13), (30, 11)), ((12, 28), (13, 27)), ((60, 28), (60, 28)),
              ((9, 7), (16, 12)), ((16, 12), (9, 7))]
    for k, (ts, is_) in enumerate(shapes):
        t, i = half_normal(rng, ts), half_normal(rng, is_)
        cases[f"rand{k}_t"], cases[f"rand{k}_i"] = t, i
        cases[f"rand{k}_out"] = ref_sim.normxcorr(t, i, "same")
    t, i = half_normal(rng, (28, 12)), half_normal(rng, (28, 12))
    for name, tt, ii in (
        ("zero_template", np.zeros_like(t), i),
        ("zero_image", t, np.zeros_like(i)),
        ("const_image", t, np.full_like(i, 0.75)),
        ("const_template", np.full_like(t, 2.0), i),
        ("sparse_image", t, np.where(rng.random(i.shape) < 0.03, i + 1, 0).astype(np.float32)),
        ("big_values", t * 300, i * 1000),
    ):
        cases[f"{name}_t"], cases[f"{name}_i"] = tt, ii
        cases[f"{name}_out"] = ref_sim.normxcorr(tt, ii, "same")
    np.savez_compressed(os.path.join(OUT, "normxcorr_maps.npz"), **cases)
    print("normxcorr_maps:", len(cases) // 3, "cases")


def golden_get_similarity():
    """(2) get_similarity scalars on seeded synthetic stacks (inputs regenerated from the seed)."""
    rows = []
    for c, h, w, dead in [(4, 16, 12, []), (8, 20, 14, [1, 5]), (64, 32, 16, [0, 63]), (512, 32, 16, [7]),
                          (512, 64, 32, []), (256, 128, 64, [3, 100])]:
        q = synth.query_features(SEED, 0, 0, c, h, w)
        g0 = synth.gallery_features(SEED, 0, c, h, w)
        g1 = synth.gallery_features(SEED, 1, c, h, w)
        for d in dead:
            g0[d] = 0
            g1[d] = 0
        q_dead = q.copy()
        if dead:
            q_dead[dead[0]] = 0
        rows.append({
            "c": c, "h": h, "w": w, "dead": dead, "seed": SEED,
            "sim_q0_g0": float(ref_sim.get_similarity(q, g0)),
            "sim_q0_g1": float(ref_sim.get_similarity(q, g1)),
            "sim_qdead_g0": float(ref_sim.get_similarity(q_dead, g0)),
        })
        print("get_similarity", rows[-1])
    with open(os.path.join(OUT, "get_similarity.json"), "w") as f:
        json.dump(rows, f, indent=1)


def ref_matrix(queries, gallery):
    m = np.zeros((len(queries), len(gallery)), dtype=np.float32)
    for qi, q in enumerate(queries):
        for gi, g in enumerate(gallery):
            s = ref_sim.get_similarity(q, g)
            if s > m[qi, gi]:  # similarity.py:366-367
                m[qi, gi] = s
    return m


def run_compare_maps(queries, gallery, matches, n_proc, rotations=None, scales=None):
    cfg = {"comparison": {"n_processes": n_proc, "rotations": rotations, "scales": scales}}
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        ranks = ref_sim.compare_maps(queries, gallery, matches, cfg)
    return np.array(ranks, dtype=np.int32).copy()


def golden_compare_maps():
    """(3) ranks from the real multi-process compare_maps + the float32 [Q,G] matrix."""
    out = {}
    q, g, m = synth.dataset(SEED, 10, 100, 4, 16, 12)
    out["tiny_matrix"] = ref_matrix(q, g)
    out["tiny_ranks"] = run_compare_maps(q, g, m, n_proc=3)
    out["tiny_matches"] = np.array(m, dtype=np.int32)
    out["tiny_shape"] = np.array([10, 100, 4, 16, 12, SEED])
    # hard set: weak signal (1) under strong noise (6) => true matches spread over the ranking
    q, g, m = synth.dataset(SEED, 10, 100, 4, 16, 12, signal=1, noise=6)
    out["hard_matrix"] = ref_matrix(q, g)
    out["hard_ranks"] = run_compare_maps(q, g, m, n_proc=4)
    out["hard_matches"] = np.array(m, dtype=np.int32)
    out["hard_shape"] = np.array([10, 100, 4, 16, 12, SEED, 1, 6])
    q, g, m = synth.dataset(SEED + 1, 2, 8, 256, 128, 64)
    out["conv3_matrix"] = ref_matrix(q, g)
    out["conv3_ranks"] = run_compare_maps(q, g, m, n_proc=2)
    out["conv3_matches"] = np.array(m, dtype=np.int32)
    out["conv3_shape"] = np.array([2, 8, 256, 128, 64, SEED + 1])
    # ragged: queries and gallery items of different spatial sizes (dataloader.py:231-237)
    rq = [synth.query_features(SEED + 2, i, i, 6, h, w) for i, (h, w) in enumerate([(18, 12), (16, 14), (18, 12)])]
    rg = [synth.gallery_features(SEED + 2, i, 6, h, w) for i, (h, w) in
          enumerate([(18, 12), (20, 12), (16, 14), (18, 12), (17, 15)])]
    out["ragged_matrix"] = ref_matrix(rq, rg)
    out["ragged_ranks"] = run_compare_maps(rq, rg, [0, 2, 3], n_proc=2)
    np.savez_compressed(os.path.join(OUT, "compare_maps.npz"), **out)
    print("compare_maps: tiny ranks", out["tiny_ranks"], "hard ranks", out["hard_ranks"], "conv3 ranks", out["conv3_ranks"],
          "ragged ranks", out["ragged_ranks"])


def golden_variants():
    """(6) rotated / scaled query variants and the ranks they lead to (f1)."""
    out = {}
    q, g, m = synth.dataset(SEED + 3, 3, 6, 3, 20, 14)
    for r in (-15, 3, 180):
        v = ref_sim._apply_transformations([list(q)], [r], list(q), "rotate")
        out[f"rot_{r}"] = np.stack(v[1])
    for s in (1.02, 1.08, 0.9):
        v = ref_sim._apply_transformations([list(q)], [s], list(q), "scale")
        out[f"scale_{s}"] = np.stack(v[1])
    out["ranks_rot"] = run_compare_maps(q, g, m, 2, rotations=[-15, 3, 180])
    out["ranks_scale"] = run_compare_maps(q, g, m, 2, scales=[1.02, 1.08])
    out["shape"] = np.array([3, 6, 3, 20, 14, SEED + 3])
    # matrix for rotation-only variants through the reference's own building blocks
    variants = ref_sim._apply_transformations([list(q)], [-15, 3, 180], list(q), "rotate")
    mat = np.zeros((3, 6), dtype=np.float32)
    for var in variants:
        mat = np.maximum(mat, ref_matrix(var, g))
    out["matrix_rot"] = mat
    np.savez_compressed(os.path.join(OUT, "variants.npz"), **out)
    print("variants: ranks_rot", out["ranks_rot"], "ranks_scale", out["ranks_scale"])


def golden_rank_and_scores():
    """(4) _get_rank on tie-free vectors, (5) cmp / cmp_all output."""
    rng = np.random.default_rng(SEED + 4)
    rank_cases = []
    for n in (5, 17, 100, 1500):
        s = rng.permutation(n).astype(np.float32) / n  # distinct values: no ties
        match = [int(rng.integers(0, n)) for _ in range(4)]
        rank_cases.append({"sims": s.tolist(), "matches": match,
                           "ranks": [int(ref_sim._get_rank(s, match, i)) for i in range(4)]})
    score_cases = []
    for ranks, g, q in [([1, 1, 3, 7, 50], 100, 5), ([1, 2, 3, 4, 5, 6, 7, 8, 9, 10], 20, 10),
                        ([15, 16, 75, 76, 300], 1500, 100), ([1], 1, 1)]:
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ref_parse.cmp_all(ranks, g, q)
        score_cases.append({"ranks": ranks, "total_shoeprints": g, "total_shoemarks": q,
                            "line": buf.getvalue().strip(),
                            "cmp": [ref_parse.cmp(ranks, p, g, q) for p in (1, 5, 10, 15, 20)]})
    with open(os.path.join(OUT, "rank_and_scores.json"), "w") as f:
        json.dump({"rank": rank_cases, "scores": score_cases}, f)
    print("rank_and_scores:", [c["line"] for c in score_cases])


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    golden_normxcorr()
    golden_get_similarity()
    golden_rank_and_scores()
    golden_variants()
    golden_compare_maps()
