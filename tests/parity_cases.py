"""Parity checks shared by the CPU-emulation tests (not gpu) and the MI355X tests (gpu).

Every check drives the product's host code (similarity.py) over a C-ABI library — the
emulation twin or the real gfx950 build — and compares with the oracle and the golden vectors.
Tolerances: scores within 1e-4 of the reference (north_star); in practice the float32 FFT path is
within ~1e-6 and that tighter bound is what is asserted for small maps.  Ranks: identical.
"""

import json
import os

import numpy as np
import pytest

from oracle import ncc_oracle as oracle
from shoeprint_image_retrieval_amd import similarity, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4      # the contract (BASELINE.json: "NCC scores agree within 1e-4 fp32")
TIGHT = 5e-6    # what the kernels are expected to reach on these sizes


def _cfg(rot=None, sc=None):
    return {"comparison": {"n_processes": 4, "rotations": rot, "scales": sc}}


def check_golden_compare_maps(scorer, name):
    z = np.load(os.path.join(GOLDEN, "compare_maps.npz"))
    shape = [int(v) for v in z[f"{name}_shape"]]
    nq, ng, c, h, w, seed = shape[:6]
    kw = {"signal": shape[6], "noise": shape[7]} if len(shape) > 6 else {}
    q, g, m = synth.dataset(seed, nq, ng, c, h, w, **kw)
    mat = scorer.score_matrix(q, g)
    assert mat.dtype == np.float32 and mat.shape == (nq, ng)
    np.testing.assert_allclose(mat, z[f"{name}_matrix"], atol=TIGHT, rtol=0)
    ranks = similarity.compare_maps(q, g, m, _cfg(), scorer=scorer)
    assert ranks.dtype == np.int32
    np.testing.assert_array_equal(ranks, z[f"{name}_ranks"])
    return mat


def check_golden_ragged(scorer):
    z = np.load(os.path.join(GOLDEN, "compare_maps.npz"))
    seed = 1236
    rq = [synth.query_features(seed, i, i, 6, h, w) for i, (h, w) in enumerate([(18, 12), (16, 14), (18, 12)])]
    rg = [synth.gallery_features(seed, i, 6, h, w) for i, (h, w) in
          enumerate([(18, 12), (20, 12), (16, 14), (18, 12), (17, 15)])]
    mat = scorer.score_matrix(rq, rg)
    np.testing.assert_allclose(mat, z["ragged_matrix"], atol=TIGHT, rtol=0)
    np.testing.assert_array_equal(similarity.compare_maps(rq, rg, [0, 2, 3], _cfg(), scorer=scorer), z["ragged_ranks"])


def check_golden_normxcorr(scorer):
    z = np.load(os.path.join(GOLDEN, "normxcorr_maps.npz"))
    names = sorted({k.rsplit("_", 1)[0] for k in z.files if k.endswith("_out")})
    for name in names:
        t, i, ref = z[f"{name}_t"], z[f"{name}_i"], z[f"{name}_out"]
        got = similarity.normxcorr(t, i, "same", scorer=scorer)
        assert got.shape == ref.shape
        if name == "const_image":
            continue  # rounding noise normalised to O(1) in the reference itself (see test_oracle_golden)
        if name in ("zero_template", "zero_image", "const_template"):
            assert not got.any(), name  # degenerate maps give exact zeros (similarity.py:68-70)
            continue
        np.testing.assert_allclose(got, ref, atol=2e-5, rtol=0, err_msg=name)


def check_golden_get_similarity(scorer, max_elems):
    rows = json.load(open(os.path.join(GOLDEN, "get_similarity.json")))
    done = 0
    for r in rows:
        c, h, w, seed = r["c"], r["h"], r["w"], r["seed"]
        if c * h * w > max_elems:
            continue
        q = synth.query_features(seed, 0, 0, c, h, w)
        g0 = synth.gallery_features(seed, 0, c, h, w)
        g1 = synth.gallery_features(seed, 1, c, h, w)
        for d in r["dead"]:
            g0[d] = 0
            g1[d] = 0
        qd = q.copy()
        if r["dead"]:
            qd[r["dead"][0]] = 0
        assert abs(similarity.get_similarity(q, g0, scorer=scorer) - r["sim_q0_g0"]) < TIGHT
        assert abs(similarity.get_similarity(q, g1, scorer=scorer) - r["sim_q0_g1"]) < TIGHT
        assert abs(similarity.get_similarity(qd, g0, scorer=scorer) - r["sim_qdead_g0"]) < TIGHT
        done += 1
    assert done > 0


SHAPE_CASES = [  # (C, qh, qw, gh, gw): tight / general variants, odd sizes, template != image size
    (3, 32, 16, 32, 16), (2, 40, 40, 40, 40), (2, 30, 17, 33, 15), (2, 16, 31, 17, 30),
    (2, 20, 20, 50, 40), (2, 50, 40, 20, 20), (1, 13, 9, 20, 16), (2, 64, 32, 64, 32),
    (3, 60, 30, 64, 32), (2, 32, 16, 30, 18), (2, 44, 22, 44, 22),
]
BIG_SHAPE_CASES = [(3, 128, 64, 128, 64), (2, 100, 70, 100, 70), (2, 90, 50, 120, 70), (2, 126, 62, 128, 64),
                   (2, 128, 64, 130, 60), (2, 97, 41, 110, 52)]


def check_narrow_maps(scorer):
    """Full correlation maps of narrow and short images (stretches of the blocked table scans that are empty or ragged:
    a width of 6 leaves the fourth column stretch empty) against the oracle - every pixel, not only the maximum."""
    rng = np.random.default_rng(11)
    for (th, tw), (ih, iw) in [((8, 6), (8, 6)), ((5, 3), (5, 3)), ((4, 6), (9, 7)), ((7, 2), (11, 5)), ((3, 3), (6, 17))]:
        t = rng.random((th, tw), dtype=np.float32)
        i = rng.random((ih, iw), dtype=np.float32)
        got = similarity.normxcorr(t, i, "same", scorer=scorer)
        want = oracle.normxcorr(t, i, precise=True)
        np.testing.assert_allclose(got, want, atol=2e-5, rtol=0, err_msg=f"{th}x{tw} on {ih}x{iw}")


def check_shape_case(scorer, case, tol=TIGHT):
    c, qh, qw, gh, gw = case
    same = (qh, qw) == (gh, gw)
    q = [synth.query_features(7, i, i, c, qh, qw) if same else synth.gallery_features(8, 10 + i, c, qh, qw)
         for i in range(2)]
    g = [synth.gallery_features(7, i, c, gh, gw) for i in range(3)]
    g[1][0] = 0  # dead gallery channel: contributes 0 but still counts in the divisor (similarity.py:96,108)
    q[1][c - 1] = 0  # dead query channel
    mat = scorer.score_matrix(q, g)
    ref = oracle.similarity_matrix(q, g, precise=True)
    np.testing.assert_allclose(mat, ref, atol=tol, rtol=0)


def check_team_mode(scorer, monkeypatch):
    """The pair kernel's persistent "team" schedule (opt-in: SPR_NCC_TEAM=1)
    scores exactly what the one-pair-per-workgroup schedule does: ragged epochs, several query strips."""
    for nq, ng, case in ((5, 7, (2, 32, 16, 32, 16)), (19, 3, (1, 20, 12, 24, 14)), (1, 9, (2, 32, 16, 32, 16))):
        c, qh, qw, gh, gw = case
        q = [synth.gallery_features(31, 100 + i, c, qh, qw) for i in range(nq)]
        g = [synth.gallery_features(31, i, c, gh, gw) for i in range(ng)]
        monkeypatch.setenv("SPR_NCC_TEAM", "0")
        tiles = scorer.score_matrix(q, g)
        monkeypatch.setenv("SPR_NCC_TEAM", "1")
        team = scorer.score_matrix(q, g)
        monkeypatch.setenv("SPR_NCC_TEAM_EVERY", "1")  # a soft barrier every channel as well
        team_every = scorer.score_matrix(q, g)
        monkeypatch.delenv("SPR_NCC_TEAM_EVERY")
        np.testing.assert_array_equal(team, tiles)
        np.testing.assert_array_equal(team_every, tiles)
        np.testing.assert_allclose(team, oracle.similarity_matrix(q, g, precise=True), atol=TIGHT, rtol=0)


def check_config3_bf16_resnet_layer3(scorer, channels):
    """BASELINE config 3 in miniature: ResNet50-layer3-shaped maps [C,32,16] stored as bfloat16 (extractor
    bypassed, half-normal synthetic features as SURVEY §8d says).  Parity = the oracle on the same rounded
    features upcast to float32; the arithmetic stays float32/float64."""
    nq, ng = 3, 5
    g = [np.maximum(synth.gallery_features(41, i, channels, 32, 16), 0) for i in range(ng)]
    q = [np.maximum(synth.query_features(41, i, i, channels, 32, 16), 0) for i in range(nq)]
    qb, gb = synth.bfloat16_bits(np.stack(q)), synth.bfloat16_bits(np.stack(g))
    dev = scorer.dev
    got = dev.to_host(scorer.scores_device(dev.to_device(qb), dev.to_device(gb)))
    ref = oracle.similarity_matrix(list(synth.from_bfloat16_bits(qb)), list(synth.from_bfloat16_bits(gb)), precise=True)
    np.testing.assert_allclose(got, ref, atol=TIGHT, rtol=0)
    assert (got.argmax(axis=1) == np.arange(nq)).all()  # each query still finds its own print


def _bf16_sets(seed, channels, nq, ng):
    g = [np.maximum(synth.gallery_features(seed, i, channels, 32, 16), 0) for i in range(ng)]
    q = [np.maximum(synth.query_features(seed, i % ng, i, channels, 32, 16), 0) for i in range(nq)]
    qb, gb = synth.bfloat16_bits(np.stack(q)), synth.bfloat16_bits(np.stack(g))
    return qb, gb, list(synth.from_bfloat16_bits(qb)), list(synth.from_bfloat16_bits(gb))


def check_mfma_method(make_scorer, channels, nq, ng, tol=TIGHT):
    """The matrix-core direct form (ncc_mfma.hip) on ResNet50-layer3-shaped bfloat16 maps: chosen by "auto", scores equal to
    the oracle on the rounded features, query counts that leave waves and lanes of the last 64-query block without work,
    the running maximum over variants, and the per-channel maps of spr_ncc_maps."""
    from shoeprint_image_retrieval_amd import _lib

    qb, gb, qf, gf = _bf16_sets(43, channels, nq, ng)
    sc = make_scorer("auto")
    dev = sc.dev
    plan = sc.plan(channels, (32, 16), (32, 16), dtype="bfloat16")
    assert plan.method == _lib.NCC_MFMA
    got = dev.to_host(sc.scores_device(dev.to_device(qb), dev.to_device(gb)))
    ref = oracle.similarity_matrix(qf, gf, precise=True)
    np.testing.assert_allclose(got, ref, atol=tol, rtol=0)
    # running maximum (similarity.py:364-367): a second pass over other queries keeps the larger score
    scores = dev.to_device(np.full((nq, ng), 0.05, np.float32))
    sc.scores_device(dev.to_device(qb), dev.to_device(gb), scores=scores, accumulate_max=True)
    np.testing.assert_allclose(dev.to_host(scores), np.maximum(ref, 0.05), atol=tol, rtol=0)
    # per-channel maps of one pair
    pq = sc.prepare_queries(plan, dev.to_device(qb[:1]))
    pg = sc.prepare_gallery(plan, dev.to_device(gb[1:2]))
    maps = dev.to_host(sc.ncc_maps_device(plan, pq, pg))
    want = np.stack([oracle.normxcorr(qf[0][c, 2:-2, 2:-2], gf[1][c, 2:-2, 2:-2], precise=True) for c in range(channels)])
    np.testing.assert_allclose(maps, want, atol=20 * tol, rtol=0)
    # larger maps and float32 storage stay with the other methods; asking for the method outright is refused there
    assert sc.plan(channels, (32, 16), (33, 16), dtype="bfloat16").method != _lib.NCC_MFMA   # map beyond the 28 x 12 frame
    assert sc.plan(channels, (36, 16), (32, 16), dtype="bfloat16").method != _lib.NCC_MFMA   # template beyond 30 x 16
    assert sc.plan(channels, (30, 16), (32, 16), dtype="bfloat16").method == _lib.NCC_MFMA   # the general instance
    assert sc.plan(channels, (32, 16), (32, 16), dtype=np.float32).method != _lib.NCC_MFMA
    with pytest.raises(Exception, match="matrix-core"):
        make_scorer("mfma").plan(channels, (32, 16), (32, 16), dtype=np.float32)


# (raw query map, raw gallery map) sizes of the GENERAL matrix-core instance (crop 2 per edge: templates up to 30 x 16 on maps
# up to 28 x 12): the scaled variants of a 32 x 16 query at the reference's run.toml scales (1.04 -> 33 x 16, 1.08 -> 34 x 17;
# similarity.py:264-278 truncates int(w * s)), a shrunk one, the largest template, an odd-sized template on a smaller map
# (ragged sets, dataloader.py:231-237), a small template (8 x 5 taps; windows of a handful of post-ReLU pixels are often
# constant, and their variance is rounding noise in the reference itself), a map narrower than the frame
MFMA_GENERAL_SHAPES = [((33, 16), (32, 16)), ((34, 17), (32, 16)), ((28, 14), (32, 16)), ((34, 20), (32, 16)),
                       ((31, 15), (30, 14)), ((12, 9), (32, 16)), ((32, 16), (21, 9))]


def check_mfma_general_shapes(make_scorer, channels=5, nq=3, ng=4, tol=TIGHT, shapes=None):
    """Unequal template / map sizes on the matrix cores (MCfg<28, 12, 30, 8>): "auto" picks the method, scores equal the
    oracle on the rounded features - both storage types, both forms of the method - and the per-channel maps of one pair."""
    from shoeprint_image_retrieval_amd import _lib

    for k, (qs, gs) in enumerate(shapes or MFMA_GENERAL_SHAPES):
        g = [np.maximum(synth.gallery_features(61 + k, i, channels, *gs), 0) for i in range(ng)]
        q = [np.maximum(synth.gallery_features(67 + k, 10 + i, channels, *qs), 0) for i in range(nq)]
        q[1][channels - 1] = 0.0  # a dead query channel
        g[ng - 1][0] = 0.0        # a dead gallery channel
        qb, gb = synth.bfloat16_bits(np.stack(q)), synth.bfloat16_bits(np.stack(g))
        qf, gf = list(synth.from_bfloat16_bits(qb)), list(synth.from_bfloat16_bits(gb))
        ref = oracle.similarity_matrix(qf, gf, precise=True)
        sc = make_scorer("auto")
        dev = sc.dev
        plan = sc.plan(channels, qs, gs, dtype="bfloat16")
        assert plan.method == _lib.NCC_MFMA, (qs, gs)
        got = dev.to_host(sc.scores_device(dev.to_device(qb), dev.to_device(gb)))
        np.testing.assert_allclose(got, ref, atol=tol, rtol=0, err_msg=f"bfloat16 {qs} on {gs}")
        pq = sc.prepare_queries(plan, dev.to_device(qb[:1]))
        pg = sc.prepare_gallery(plan, dev.to_device(gb[1:2]))
        maps = dev.to_host(sc.ncc_maps_device(plan, pq, pg))
        want = np.stack([oracle.normxcorr(qf[0][c, 2:-2, 2:-2], gf[1][c, 2:-2, 2:-2], precise=True) for c in range(channels)])
        assert maps.shape == want.shape
        np.testing.assert_allclose(maps, want, atol=20 * tol, rtol=0, err_msg=f"maps {qs} on {gs}")
        q16, g16 = np.stack(q).astype(np.float16), np.stack(g).astype(np.float16)
        ref16 = oracle.similarity_matrix(list(q16.astype(np.float32)), list(g16.astype(np.float32)), precise=True)
        got16 = dev.to_host(sc.scores_device(dev.to_device(q16), dev.to_device(g16)))
        np.testing.assert_allclose(got16, ref16, atol=tol, rtol=0, err_msg=f"float16 {qs} on {gs}")


def check_mfma_method_fp16(make_scorer, channels, nq, ng, tol=TIGHT):
    """The same kernel on float16-stored maps (VGG16 conv5_3 of BASELINE config 5): v_mfma_f32_16x16x32_f16, exact form."""
    from shoeprint_image_retrieval_amd import _lib

    g = [np.maximum(synth.gallery_features(53, i, channels, 32, 16), 0).astype(np.float16) for i in range(ng)]
    q = [np.maximum(synth.query_features(53, i % ng, i, channels, 32, 16), 0).astype(np.float16) for i in range(nq)]
    sc = make_scorer("auto")
    dev = sc.dev
    assert sc.plan(channels, (32, 16), (32, 16), dtype=np.float16).method == _lib.NCC_MFMA
    got = dev.to_host(sc.scores_device(dev.to_device(np.stack(q)), dev.to_device(np.stack(g))))
    ref = oracle.similarity_matrix([a.astype(np.float32) for a in q], [a.astype(np.float32) for a in g], precise=True)
    np.testing.assert_allclose(got, ref, atol=tol, rtol=0)


def check_mfma_conditioning(make_scorer, channels=6, nq=3, ng=3, tol=TIGHT):
    """Maps riding on a large offset (mean / sigma up to ~1000): with the raw operands on the matrix cores the numerator
    would be a difference of numbers (mean / sigma)^2 times its size (4e-4 on the scores at an offset of 100, 5e-2 at 1000,
    before the prep kernels shifted such channels exactly in their storage type); channels spread over several binades (offset
    0) take the other branch.  Both forms of the method and both storage types."""
    for offset in (0.0, 3.0, 100.0, 1000.0):
        g = [np.maximum(synth.gallery_features(41, i, channels, 32, 16), 0) + np.float32(offset) for i in range(ng)]
        q = [np.maximum(synth.query_features(41, i, i, channels, 32, 16), 0) + np.float32(offset) for i in range(nq)]
        qb, gb = synth.bfloat16_bits(np.stack(q)), synth.bfloat16_bits(np.stack(g))
        ref = oracle.similarity_matrix(list(synth.from_bfloat16_bits(qb)), list(synth.from_bfloat16_bits(gb)), precise=True)
        sc = make_scorer("mfma")
        dev = sc.dev
        got = dev.to_host(sc.scores_device(dev.to_device(qb), dev.to_device(gb)))
        np.testing.assert_allclose(got, ref, atol=tol, rtol=0, err_msg=f"bfloat16, offset {offset}")
        if offset <= 100.0:  # (float16 keeps 11 bits: an offset of 1000 leaves steps of 0.5 - still a valid map, same rule)
            q16, g16 = np.stack(q).astype(np.float16), np.stack(g).astype(np.float16)
            ref16 = oracle.similarity_matrix(list(q16.astype(np.float32)), list(g16.astype(np.float32)), precise=True)
            got16 = dev.to_host(sc.scores_device(dev.to_device(q16), dev.to_device(g16)))
            np.testing.assert_allclose(got16, ref16, atol=tol, rtol=0, err_msg=f"float16, offset {offset}")


def check_mfma_degenerate_channels(make_scorer, tol=TIGHT):
    """All-zero, constant and single-spike channels on either side (similarity.py:65-70: they contribute 0, or what little
    the spike correlates) through the matrix-core method: finite, equal to the oracle."""
    c, nq, ng = 6, 2, 2
    g = [np.maximum(synth.gallery_features(41, i, c, 32, 16), 0) for i in range(ng)]
    q = [np.maximum(synth.query_features(41, i, i, c, 32, 16), 0) for i in range(nq)]
    g[0][1] = 0.0
    q[1][2] = 0.0
    g[1][3] = 7.0
    q[0][4] = 0.5
    g[1][5] = 0.0
    g[1][5][9, 7] = 3.0
    qb, gb = synth.bfloat16_bits(np.stack(q)), synth.bfloat16_bits(np.stack(g))
    ref = oracle.similarity_matrix(list(synth.from_bfloat16_bits(qb)), list(synth.from_bfloat16_bits(gb)), precise=True)
    sc = make_scorer("mfma")
    got = sc.dev.to_host(sc.scores_device(sc.dev.to_device(qb), sc.dev.to_device(gb)))
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, ref, atol=tol, rtol=0)


def check_mfma_large_gallery(make_scorer, monkeypatch, channels, nq, ng, oracle_pairs=12):
    qb, gb, qf, gf = _bf16_sets(47, channels, nq, ng)
    mf, ff = make_scorer("mfma"), make_scorer("fft")
    dev = mf.dev
    qd, gd = dev.to_device(qb), dev.to_device(gb)
    got = dev.to_host(mf.scores_device(qd, gd))
    other = dev.to_host(ff.scores_device(qd, gd))
    np.testing.assert_allclose(got, other, atol=TIGHT, rtol=0)
    rng = np.random.default_rng(5)
    for qi, gi in zip(rng.integers(0, nq, oracle_pairs), rng.integers(0, ng, oracle_pairs)):
        want = float(oracle.get_similarity(qf[qi], gf[gi], precise=True))
        assert abs(got[qi, gi] - max(want, 0.0)) <= TIGHT, (qi, gi, got[qi, gi], want)
    match = dev.to_device((np.arange(nq) % ng).astype(np.int32))
    r_m = dev.to_host(mf.ranks_device(dev.to_device(got), match))
    r_f = dev.to_host(ff.ranks_device(dev.to_device(other), match))
    np.testing.assert_array_equal(r_m, r_f)
    monkeypatch.setenv("SPR_NCC_MAX_TILES", str(max(1, ng // 3)))  # three launches over the gallery
    np.testing.assert_array_equal(dev.to_host(mf.scores_device(qd, gd)), got)
    monkeypatch.delenv("SPR_NCC_MAX_TILES")


def check_config5_multi_layer_fp16(scorer, channels):
    """BASELINE config 5 in miniature: conv3_3 + conv4_3 + conv5_3 shaped maps in float16, score = mean of the
    three get_similarity values (build-defined)."""
    nq, ng = 2, 3
    dev = scorer.dev
    layers, refs = [], []
    for k, (c, h, w) in enumerate(zip(channels, (128, 64, 32), (64, 32, 16))):
        g = np.stack([synth.gallery_features(50 + k, i, c, h, w) for i in range(ng)]).astype(np.float16)
        q = np.stack([synth.query_features(50 + k, i, i, c, h, w) for i in range(nq)]).astype(np.float16)
        layers.append((dev.to_device(q), dev.to_device(g)))
        refs.append(oracle.similarity_matrix(list(q.astype(np.float32)), list(g.astype(np.float32)), precise=True))
    got = scorer.multi_layer_score_matrix(layers)
    np.testing.assert_allclose(got, np.mean(refs, axis=0), atol=TIGHT, rtol=0)


def check_big_mode(make_scorer, monkeypatch, full):
    """Maps beyond LDS: the prep / pair kernels keep their working set in the plan's global workspace.
    (1) forced on shapes that also fit LDS (SPR_NCC_FORCE_BIG=1) the scores are bit-identical to the LDS mode;
    (2) a conv3_3-of-an-800x400-print sized pair (200x100 maps -> 384x192 grid), only possible in this mode,
    against the oracle."""
    for case in ((2, 32, 16, 32, 16), (2, 40, 40, 40, 40), (2, 30, 17, 33, 15), (1, 64, 32, 64, 32)):
        c, qh, qw, gh, gw = case
        q = [synth.gallery_features(61, 100 + i, c, qh, qw) for i in range(3)]
        g = [synth.gallery_features(61, i, c, gh, gw) for i in range(4)]
        monkeypatch.delenv("SPR_NCC_FORCE_BIG", raising=False)
        lds = make_scorer().score_matrix(q, g)
        monkeypatch.setenv("SPR_NCC_FORCE_BIG", "1")
        big = make_scorer().score_matrix(q, g)  # a fresh scorer: the mode is fixed when the plan is created
        monkeypatch.delenv("SPR_NCC_FORCE_BIG")
        np.testing.assert_array_equal(big, lds)
    c, h, w = (4, 200, 100) if full else (1, 200, 100)
    sc = make_scorer()
    assert sc.plan(c, (h, w), (h, w)).fft_size == (384, 192)
    q = [synth.query_features(62, i, i, c, h, w) for i in range(2)]
    g = [synth.gallery_features(62, i, c, h, w) for i in range(2 if not full else 3)]
    got = sc.score_matrix(q, g)
    ref = oracle.similarity_matrix(q, g, precise=True)
    np.testing.assert_allclose(got, ref, atol=TIGHT, rtol=0)
    if not full:
        return
    # the instance's other two accumulator layouts: maps wider than 108 columns (11 kept outputs per row transform) and a
    # search map taller than 256 rows under a smaller template (five row rounds: the general variant)
    for (c, qh, qw, gh, gw) in ((2, 200, 124, 200, 124), (1, 150, 60, 280, 100)):
        sc = make_scorer()
        assert sc.plan(c, (qh, qw), (gh, gw)).fft_size == (384, 192)
        q = [synth.gallery_features(63, 50 + i, c, qh, qw) for i in range(2)]
        g = [synth.gallery_features(63, i, c, gh, gw) for i in range(2)]
        np.testing.assert_allclose(sc.score_matrix(q, g), oracle.similarity_matrix(q, g, precise=True), atol=TIGHT, rtol=0)


def check_sparse_channels(scorer, c=5):
    """Post-ReLU maps of real prints have channels that are entirely zero (or constant) in one image: they contribute
    exactly 0 and still count in the divisor (similarity.py:68-70, 96, 108).  The prep kernels flag them and the six-wave
    pair kernel walks only the channels live on both sides: dead patterns that differ between the two queries of a
    workgroup and between gallery items, a constant non-zero channel, and a pair with NO live channel (score 0)."""
    qh, qw = 128, 64
    q = [synth.gallery_features(95, 100 + i, c, qh, qw) for i in range(3)]
    g = [synth.gallery_features(95, i, c, qh, qw) for i in range(3)]
    q[0][[0, c - 2]] = 0
    q[1][1] = 0.75      # constant, non-zero: zero after centring
    g[0][[1, min(2, c - 1)]] = 0
    g[1][:] = 0         # no live channel against anybody
    g[2][c - 1] = 0
    q[2][:c - 1] = 0    # live only in the channel g[2] lacks: nothing in common with g[2]
    mat = scorer.score_matrix(q, g)
    ref = oracle.similarity_matrix(q, g, precise=True)
    np.testing.assert_allclose(mat, ref, atol=TIGHT, rtol=0)
    assert np.all(mat[:, 1] == 0.0) and mat[2, 2] == 0.0


def check_launch_slicing(make_scorer, monkeypatch, big_c=2):
    """HIP refuses grids of 2^32 work-items and more, so the pair launches are cut into slices of pair tiles
    (gallery items for the direct kernel).  With the slice size forced down to ONE tile (SPR_NCC_MAX_TILES=1) a
    multi-tile score matrix must come out bit-identical to the single launch - for the one-wave / four-wave
    kernels, the six-wave kernel (conv3_3-sized maps: two queries per workgroup, odd query count) and the direct one."""
    cases = [("fft", 17, 35, (2, 20, 12, 20, 12)), ("fft", 3, 17, (big_c, 128, 64, 128, 64)), ("direct", 5, 4, (2, 20, 12, 20, 12))]
    for method, nq, ng, (c, qh, qw, gh, gw) in cases:
        q = [synth.gallery_features(91, 100 + i, c, qh, qw) for i in range(nq)]
        g = [synth.gallery_features(91, i, c, gh, gw) for i in range(ng)]
        monkeypatch.delenv("SPR_NCC_MAX_TILES", raising=False)
        whole = make_scorer(method).score_matrix(q, g)
        monkeypatch.setenv("SPR_NCC_MAX_TILES", "1")
        sliced = make_scorer(method).score_matrix(q, g)
        monkeypatch.delenv("SPR_NCC_MAX_TILES")
        np.testing.assert_array_equal(sliced, whole)
        np.testing.assert_allclose(whole[:2, :3], oracle.similarity_matrix(q[:2], g[:3], precise=True), atol=TIGHT, rtol=0)


def check_config_selects_the_scorer(make_from_config):
    """[mi355x] of run.toml is what builds the scorer behind compare_maps: ncc_method picks the kernel, dtype the HBM
    storage type of uploaded maps (scores then equal the oracle on the ROUNDED maps), max_prepared_gib the chunk budget."""
    from shoeprint_image_retrieval_amd import _lib
    from shoeprint_image_retrieval_amd.config import normalise

    q = [synth.query_features(93, i, i, 3, 22, 14) for i in range(3)]
    g = [synth.gallery_features(93, i, 3, 22, 14) for i in range(5)]
    ref = oracle.similarity_matrix(q, g, precise=True)
    for method, code in (("direct", _lib.NCC_DIRECT), ("fft", _lib.NCC_FFT)):
        cfg = normalise({"comparison": {"n_processes": 1, "rotations": "", "scales": ""}, "mi355x": {"ncc_method": method}})
        sc = make_from_config(cfg)
        assert sc.plan(3, (22, 14), (22, 14)).method == code
        np.testing.assert_array_equal(similarity.compare_maps(q, g, [0, 1, 2], cfg, scorer=sc),
                                      oracle.compare_maps(q, g, [0, 1, 2], cfg))
    cfg = normalise({"comparison": {"n_processes": 1, "rotations": "", "scales": ""},
                     "mi355x": {"dtype": "float16", "max_prepared_gib": 0.001}})
    sc = make_from_config(cfg)
    assert sc.storage == "float16" and sc._budget() == int(0.001 * (1 << 30))
    q16 = [a.astype(np.float16).astype(np.float32) for a in q]
    g16 = [a.astype(np.float16).astype(np.float32) for a in g]
    np.testing.assert_allclose(sc.score_matrix(q, g), oracle.similarity_matrix(q16, g16, precise=True), atol=TIGHT, rtol=0)
    assert np.abs(sc.score_matrix(q, g) - ref).max() < 5e-3
    with np.testing.assert_raises(ValueError):
        make_from_config(normalise({"comparison": {}, "mi355x": {"ncc_method": "fastest"}}))


def check_empty_and_degenerate_sets(scorer):
    """Edge cases as the reference behaves (run there): no queries -> empty int32 ranks; one query works;
    queries against an empty gallery -> IndexError from the rank lookup (similarity.py:386); a match index
    outside the gallery -> IndexError as well."""
    g = [synth.gallery_features(71, i, 2, 12, 10) for i in range(3)]
    q = [synth.query_features(71, i, i, 2, 12, 10) for i in range(2)]
    r = similarity.compare_maps([], g, [], _cfg(), scorer=scorer)
    assert r.dtype == np.int32 and r.shape == (0,)
    one = similarity.compare_maps(q[:1], g, [0], _cfg(), scorer=scorer)
    np.testing.assert_array_equal(one, oracle.compare_maps(q[:1], g, [0], _cfg()))
    for bad in (lambda: similarity.compare_maps(q, [], [0, 1], _cfg(), scorer=scorer),
                lambda: similarity.compare_maps(q, g, [0, 7], _cfg(), scorer=scorer)):
        try:
            bad()
        except IndexError:
            pass
        else:
            raise AssertionError("expected IndexError")
    assert scorer.score_matrix([], g).shape == (0, 3) and scorer.score_matrix(q, []).shape == (2, 0)


def check_rank_kernel(scorer):
    rng = np.random.default_rng(5)
    for nq, ng in [(1, 1), (3, 7), (5, 300), (2, 1500)]:
        s = rng.random((nq, ng)).astype(np.float32)
        s[:, ::3] = s[:, :1]  # plenty of exact ties, some involving the true match
        m = rng.integers(0, ng, nq)
        got = scorer.ranks(s, m)
        np.testing.assert_array_equal(got, oracle.ranks_from_matrix(s, m))
    d = json.load(open(os.path.join(GOLDEN, "rank_and_scores.json")))
    for case in d["rank"]:  # the reference's own _get_rank outputs (tie-free)
        s = np.array(case["sims"], dtype=np.float32)[None].repeat(4, 0)
        np.testing.assert_array_equal(scorer.ranks(s, case["matches"]), case["ranks"])
    try:
        scorer.ranks(np.zeros((2, 4), np.float32), [0, 4])
    except IndexError:
        pass
    else:
        raise AssertionError("a match index outside the gallery must raise IndexError (similarity.py:386)")


def check_synth_twin(scorer, lib):
    """The device generator is bit-identical to the numpy generator."""
    dev = scorer.dev
    c, h, w, seed = 3, 10, 7, 99
    out = dev.zeros((4, c, h, w), np.float32)
    lib.check(lib.spr_synth_gallery(dev.ptr(out), 2, 4, c, h, w, seed, dev.stream()))
    got = dev.to_host(out)
    for i in range(4):
        np.testing.assert_array_equal(got[i], synth.gallery_features(seed, 2 + i, c, h, w))
    match = np.array([5, 0, 3], dtype=np.int32)
    m_dev = dev.to_device(match)
    outq = dev.zeros((3, c, h, w), np.float32)
    lib.check(lib.spr_synth_queries(dev.ptr(outq), 1, 3, dev.ptr(m_dev), c, h, w, seed, 3, 1, 6, dev.stream()))
    gotq = dev.to_host(outq)
    for i in range(3):
        np.testing.assert_array_equal(gotq[i], synth.query_features(seed, 1 + i, int(match[i]), c, h, w, signal=1, noise=6))


def check_variant_accumulate(scorer):
    """accumulate_max keeps the running maximum over calls (similarity.py:364-367) and the floor at 0."""
    dev = scorer.dev
    q, g, _ = synth.dataset(3, 2, 3, 2, 16, 12, signal=1, noise=6)
    qb, gb = dev.to_device(np.stack(q)), dev.to_device(np.stack(g))
    scores = scorer.scores_device(qb, gb)
    first = dev.to_host(scores).copy()
    assert (first >= 0).all()
    big = dev.to_device(np.full((2, 3), 0.9, np.float32))
    scorer.scores_device(qb, gb, scores=big, accumulate_max=True)
    np.testing.assert_array_equal(dev.to_host(big), np.maximum(first, np.float32(0.9)))
    # anti-correlated pair: the raw similarity is negative, the stored score is the floor 0
    a = np.zeros((1, 12, 10), np.float32)
    a[0, 4:8, 3:7] = 1.0
    mat = scorer.score_matrix([a], [1.0 - a])
    ref = oracle.similarity_matrix([a], [1.0 - a], precise=True)
    np.testing.assert_allclose(mat, ref, atol=TIGHT)


def check_variant_chunk_budget(make_scorer):
    """Scaled variants give one plan (and one prepared form of every gallery chunk) per distinct query shape; all of them
    are alive while a chunk is scored, so together they must fit the HBM budget of 'one prepared chunk' - and chunking must
    not change a score."""
    q, g, _ = synth.dataset(11, 3, 7, 2, 20, 14, signal=1, noise=6)
    scales = [1.1, 1.25]  # 16 x 10 cropped maps -> three distinct variant shapes
    whole = make_scorer()
    ref = whole.score_matrix(q, g, scales=scales)
    plans = [p for p in whole._plans.values()]
    assert len({p.q_hw for p in plans}) == 3
    per_item_all = sum(p.gallery_item_bytes for p in plans)
    budget = 3 * max(p.gallery_item_bytes for p in plans) * 2 + 17  # room for two items of every form, not for three
    small = make_scorer(max_prepared_bytes=budget)
    got = small.score_matrix(q, g, scales=scales)
    assert small.last_chunk_items == 2 and small.last_chunk_items * per_item_all <= budget
    np.testing.assert_array_equal(got, ref)


def check_variants(scorer):
    """f1: rotated / scaled query variants are bit-identical to Pillow's (golden from the real reference),
    and the running-max matrix / ranks over variants match the reference."""
    from shoeprint_image_retrieval_amd.variants import VariantBuilder

    z = np.load(os.path.join(GOLDEN, "variants.npz"))
    nq, ng, c, h, w, seed = (int(v) for v in z["shape"])
    q, g, m = synth.dataset(seed, nq, ng, c, h, w)
    dev = scorer.dev
    vb = VariantBuilder(scorer.lib, dev)
    qd = dev.to_device(np.stack(q))
    for r in (-15, 3, 180):
        np.testing.assert_array_equal(dev.to_host(vb.rotate(qd, r)), z[f"rot_{r}"])
    for sc in (1.02, 1.08, 0.9):
        np.testing.assert_array_equal(dev.to_host(vb.scale(qd, sc)), z[f"scale_{sc}"])
    mat = scorer.score_matrix(q, g, rotations=[-15, 3, 180])
    np.testing.assert_allclose(mat, z["matrix_rot"], atol=TIGHT, rtol=0)
    np.testing.assert_array_equal(similarity.compare_maps(q, g, m, _cfg(rot=[-15, 3, 180]), scorer=scorer), z["ranks_rot"])
    np.testing.assert_array_equal(similarity.compare_maps(q, g, m, _cfg(sc=[1.02, 1.08]), scorer=scorer), z["ranks_scale"])
    # both set: 1 + (R+1)*S variant lists, against the oracle (which drives real Pillow)
    rot, sca = [9, 180], [1.08, 0.9]
    assert len(vb.variants(qd, rot, sca)) == 1 + 3 * 2
    both = scorer.score_matrix(q, g, rotations=rot, scales=sca)
    ref = oracle.similarity_matrix(q, g, rotations=rot, scales=sca, precise=True)
    np.testing.assert_allclose(both, ref, atol=TIGHT, rtol=0)
    # square maps: 90 / 270 are exact transposes in Pillow
    from PIL import Image
    sq = np.stack([synth.gallery_features(3, i, 2, 12, 12) for i in range(2)])
    sd = dev.to_device(sq)
    for r in (90, 270, 45, -90):
        want = np.stack([[np.array(Image.fromarray(ch).rotate(r)) for ch in item] for item in sq])
        np.testing.assert_array_equal(dev.to_host(vb.rotate(sd, r)), want)
