"""`gpu`: the parity tests proper — the real gfx950 library on an MI355X, through the C ABI,
against the oracle, the reference-generated golden vectors, and size-independent properties at
BASELINE.json's full feature sizes."""

import os

import numpy as np
import pytest

import parity_cases as pc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from shoeprint_image_retrieval_amd import _lib

    return _lib.load_library()  # raises if the in-tree .so is missing: no fallback


@pytest.fixture(scope="module", params=["fft", "fft_pow2", "direct"])
def scorer(request, lib):
    from shoeprint_image_retrieval_amd.similarity import NccScorer

    return NccScorer(method=request.param, library=lib)


@pytest.fixture(scope="module")
def fft_scorer(lib):
    from shoeprint_image_retrieval_amd.similarity import NccScorer

    return NccScorer(method="fft", library=lib)


def test_native_library_is_loaded(lib):
    assert os.path.basename(lib.path) == "libshoeprint_mi355x.so"
    maps = open("/proc/self/maps").read()
    assert "libshoeprint_mi355x.so" in maps


@pytest.mark.parametrize("name", ["tiny", "hard"])
def test_golden_compare_maps(scorer, name):
    pc.check_golden_compare_maps(scorer, name)


def test_golden_ragged(scorer):
    pc.check_golden_ragged(scorer)


def test_golden_normxcorr(scorer):
    pc.check_golden_normxcorr(scorer)


def test_narrow_maps(scorer):
    pc.check_narrow_maps(scorer)


def test_golden_get_similarity_all_sizes(scorer):
    # includes the 512x64x32 (conv4_3) and 256x128x64 (conv3_3) stacks
    pc.check_golden_get_similarity(scorer, max_elems=1 << 40)


@pytest.mark.parametrize("case", pc.SHAPE_CASES + pc.BIG_SHAPE_CASES)
def test_shape_classes(scorer, case):
    pc.check_shape_case(scorer, case, tol=2e-5)


def test_golden_conv3_shaped_compare_maps(scorer):
    """Q=2 x G=8 at the VGG16 conv3_3 shape: matrix and ranks from the real reference."""
    pc.check_golden_compare_maps(scorer, "conv3")


def test_team_schedule(fft_scorer, monkeypatch):
    pc.check_team_mode(fft_scorer, monkeypatch)


def test_config3_bf16_resnet_layer3_maps(fft_scorer):
    pc.check_config3_bf16_resnet_layer3(fft_scorer, channels=1024)


def _make_scorer(lib):
    from shoeprint_image_retrieval_amd.similarity import NccScorer

    return lambda method: NccScorer(method=method, library=lib)


@pytest.mark.parametrize("channels,nq,ng", [(1024, 5, 3), (64, 70, 4), (16, 129, 2)])
def test_matrix_core_method(lib, channels, nq, ng):
    pc.check_mfma_method(_make_scorer(lib), channels, nq, ng)


def test_matrix_core_method_fp16(lib):
    pc.check_mfma_method_fp16(_make_scorer(lib), 512, 70, 4)


@pytest.mark.parametrize("exact", ["1", "0"])
def test_matrix_core_method_conditioning(lib, monkeypatch, exact):
    monkeypatch.setenv("SPR_NCC_MFMA_EXACT", exact)
    pc.check_mfma_conditioning(_make_scorer(lib), channels=64)
    pc.check_mfma_degenerate_channels(_make_scorer(lib))


@pytest.mark.parametrize("exact", ["1", "0"])
def test_matrix_core_general_shapes(lib, monkeypatch, exact, capsys):
    """Templates up to 30 x 16 on maps up to 28 x 12 (the scaled / rotated query variants of 32 x 16 maps at the reference's
    run.toml scales, ragged sets) on the matrix cores at EfficientNetV2_M's block-6 width (176 channels), both forms; and the
    rate of one such plan against the FFT form it replaces, printed."""
    import time

    monkeypatch.setenv("SPR_NCC_MFMA_EXACT", exact)
    pc.check_mfma_general_shapes(_make_scorer(lib), channels=176, nq=5, ng=6)
    if exact == "0":
        return
    import torch
    from shoeprint_image_retrieval_amd import synth

    nq, ng, c = 64, 2048, 176
    rates = {}
    for method in ("auto", "fft"):
        sc = _make_scorer(lib)(method)
        dev = sc.dev
        g = dev.zeros((ng, c, 32, 16), np.float32)
        lib.check(lib.spr_synth_gallery(dev.ptr(g), 0, ng, c, 32, 16, 5, dev.stream()))
        q = dev.zeros((nq, c, 33, 16), np.float32)   # a query batch scaled by 1.04 (similarity.py:264-278)
        lib.check(lib.spr_synth_gallery(dev.ptr(q), 5000, nq, c, 33, 16, 5, dev.stream()))
        q, g = q.to(torch.bfloat16), g.to(torch.bfloat16)
        plan = sc.plan(c, (33, 16), (32, 16), dtype="bfloat16")
        pq, pg = sc.prepare_queries(plan, q), sc.prepare_gallery(plan, g)
        out = dev.zeros((nq, ng), np.float32)
        sc.score_prepared(plan, pq, nq, pg, ng, out, ng, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            sc.score_prepared(plan, pq, nq, pg, ng, out, ng, 0)
        torch.cuda.synchronize()
        rates[method] = (3 * nq * ng / (time.perf_counter() - t0), dev.to_host(out))
    np.testing.assert_allclose(rates["auto"][1], rates["fft"][1], atol=pc.TIGHT, rtol=0)
    with capsys.disabled():
        print(f"\n[33x16 on 32x16 bf16, 176 ch] matrix cores {rates['auto'][0] / 1e6:.2f} M pairs/s, FFT form {rates['fft'][0] / 1e6:.2f} M pairs/s")
    assert rates["auto"][0] > rates["fft"][0]


def test_matrix_core_method_split_form(lib, monkeypatch):
    monkeypatch.setenv("SPR_NCC_MFMA_EXACT", "0")
    pc.check_mfma_method(_make_scorer(lib), 256, 70, 3)


def test_matrix_core_method_large_gallery(lib, monkeypatch):
    """G = 1 100 (beyond one slice of the correction matrix) at the full ResNet50-layer3 shape [1024,32,16] bfloat16 (BASELINE config 3 in shape): the matrix-core form
    against the FFT form on every pair, against the oracle on sampled pairs, identical ranks, launches sliced."""
    pc.check_mfma_large_gallery(_make_scorer(lib), monkeypatch, channels=1024, nq=8, ng=1100)


def test_config5_multi_layer_fp16(fft_scorer):
    pc.check_config5_multi_layer_fp16(fft_scorer, channels=(256, 512, 512))


def test_config_selects_the_scorer(lib):
    from shoeprint_image_retrieval_amd.similarity import scorer_from_config

    pc.check_config_selects_the_scorer(lambda cfg: scorer_from_config(cfg, library=lib))


def test_sparse_channels(fft_scorer):
    pc.check_sparse_channels(fft_scorer, c=7)


def test_launch_slicing(lib, monkeypatch):
    from shoeprint_image_retrieval_amd.similarity import NccScorer

    pc.check_launch_slicing(lambda m: NccScorer(method=m, library=lib), monkeypatch)


def test_big_mode(lib, monkeypatch):
    from shoeprint_image_retrieval_amd.similarity import NccScorer

    pc.check_big_mode(lambda: NccScorer(method="fft", library=lib), monkeypatch, full=True)


def test_empty_and_degenerate_sets(fft_scorer):
    pc.check_empty_and_degenerate_sets(fft_scorer)


def test_rank_kernel(scorer):
    pc.check_rank_kernel(scorer)


def test_synth_twin(scorer, lib):
    pc.check_synth_twin(scorer, lib)


def test_query_variants(scorer):
    pc.check_variants(scorer)


def test_accumulate_and_floor(scorer):
    pc.check_variant_accumulate(scorer)


def test_variant_chunk_budget(lib):
    from shoeprint_image_retrieval_amd.similarity import NccScorer

    pc.check_variant_chunk_budget(lambda **kw: NccScorer(library=lib, method="fft", **kw))


def test_fft_and_direct_agree_on_device_generated_conv3_features(lib):
    """Two independent HIP formulations on device-generated features (config-2 shape, fewer items)."""
    from shoeprint_image_retrieval_amd.similarity import NccScorer
    from shoeprint_image_retrieval_amd import synth

    fft, direct = NccScorer(method="fft", library=lib), NccScorer(method="direct", library=lib)
    dev = fft.dev
    nq, ng, c, h, w, seed = 3, 6, 256, 128, 64, 77
    g = dev.zeros((ng, c, h, w), np.float32)
    q = dev.zeros((nq, c, h, w), np.float32)
    m = dev.to_device(synth.default_matches(nq, ng))
    lib.check(lib.spr_synth_gallery(dev.ptr(g), 0, ng, c, h, w, seed, dev.stream()))
    lib.check(lib.spr_synth_queries(dev.ptr(q), 0, nq, dev.ptr(m), c, h, w, seed, 3, 1, 6, dev.stream()))
    a = dev.to_host(fft.scores_device(q, g))
    b = dev.to_host(direct.scores_device(q, g))
    np.testing.assert_allclose(a, b, atol=2e-5, rtol=0)
    np.testing.assert_array_equal(dev.to_host(fft.ranks_device(dev.to_device(a), m)),
                                  dev.to_host(direct.ranks_device(dev.to_device(b), m)))


def test_full_size_properties(fft_scorer, lib):
    """Size-independent properties at the full conv3_3 size with a few hundred pairs:
    (1) a gallery item scored against itself gives exactly the per-channel autocorrelation peak
        ~1 (all channels alive), (2) scores are invariant to a positive rescaling of either side,
    (3) permuting the gallery permutes the columns, (4) chunked == single pass, (5) scores in [0, 1+eps]."""
    from shoeprint_image_retrieval_amd import synth
    from shoeprint_image_retrieval_amd.similarity import NccScorer

    sc = fft_scorer
    dev = sc.dev
    nq, ng, c, h, w, seed = 8, 40, 256, 128, 64, 4242
    g = dev.zeros((ng, c, h, w), np.float32)
    lib.check(lib.spr_synth_gallery(dev.ptr(g), 0, ng, c, h, w, seed, dev.stream()))
    q = g[:nq].clone()
    base = dev.to_host(sc.scores_device(q, g))
    assert np.all(base >= 0) and np.all(base <= 1.0 + 1e-4)
    np.testing.assert_allclose(np.diag(base[:, :nq]), 1.0, atol=2e-5)       # (1)
    assert np.all(base[~np.eye(nq, ng, dtype=bool)] < 0.5)
    scaled = dev.to_host(sc.scores_device(q * 4.0, g * 0.5))                # (2) power-of-two scales: exact inputs
    np.testing.assert_allclose(scaled, base, atol=2e-6)
    perm = np.random.default_rng(0).permutation(ng)
    import torch
    gp = g[torch.as_tensor(perm, device=g.device)]
    np.testing.assert_array_equal(dev.to_host(sc.scores_device(q, gp)), base[:, perm])   # (3)
    plan = sc.plan(c, (h, w), (h, w))
    small = NccScorer(method="fft", library=lib, max_prepared_bytes=7 * plan.gallery_item_bytes)
    np.testing.assert_array_equal(dev.to_host(small.scores_device(q, g)), base)          # (4)
    ranks = dev.to_host(sc.ranks_device(dev.to_device(base), dev.to_device(np.arange(nq, dtype=np.int32))))
    np.testing.assert_array_equal(ranks, np.ones(nq, np.int32))


@pytest.mark.parametrize("config", [4, 5])
def test_bench_step_on_chunked_fp16_gallery(config):
    """BASELINE configs 4 / 5 through bench.py's OWN step on one GPU: conv3_3 maps stored as float16 (config 5: three layers
    on three streams, fused mean), the prepared gallery forced into >= 2 chunks by a small HBM budget, and the sampled block
    of scores + the full rank vectors of the sampled queries against the oracle on the float16-rounded maps."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    gib = 0.45 if config == 4 else 1.35  # 28.7 MB per prepared conv3_3 item (config 5: a third of the budget per layer)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--config", str(config), "--steps", "1", "--warmup", "0",
           "--queries", "6", "--gallery-per-gpu", "40", "--max-prepared-gib", str(gib), "--no-extractor",
           "--cpu-sample-queries", "4", "--cpu-sample-gallery", "10"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["config"]["workload"].startswith(f"config {config}:") and out["config"]["storage"] == "float16"
    assert out["config"]["gallery_chunks_per_step"] >= 2, out["config"]
    assert out["config"]["streams"] == (3 if config == 5 else 1)
    assert out["roofline"]["kernel"] == "pair6_kernel" and out["roofline"]["launches"] >= 2
    ps = out["parity_sample"]
    assert ps["pairs"] == 40 and ps["max_abs_err_vs_oracle"] < 1e-4 and ps["layers"] == (3 if config == 5 else 1)
    assert ps["true_match_ranks_equal"], ps
    if config == 4:
        assert ps["full_rank_vectors_equal"], ps


def test_oracle_sample_at_full_size(fft_scorer, lib):
    """A sample of config-2 pairs (device-generated features) against the float64 oracle."""
    from oracle import ncc_oracle as oracle
    from shoeprint_image_retrieval_amd import synth

    sc = fft_scorer
    dev = sc.dev
    nq, ng, c, h, w, seed = 2, 3, 256, 128, 64, 1234
    m = synth.default_matches(nq, ng)
    g = dev.zeros((ng, c, h, w), np.float32)
    q = dev.zeros((nq, c, h, w), np.float32)
    lib.check(lib.spr_synth_gallery(dev.ptr(g), 0, ng, c, h, w, seed, dev.stream()))
    lib.check(lib.spr_synth_queries(dev.ptr(q), 0, nq, dev.ptr(dev.to_device(m)), c, h, w, seed, 3, 3, 2, dev.stream()))
    got = dev.to_host(sc.scores_device(q, g))
    ref = oracle.similarity_matrix(list(dev.to_host(q)), list(dev.to_host(g)), precise=True)
    np.testing.assert_allclose(got, ref, atol=1e-5, rtol=0)


# ------------------------------------------------------------------------------ extractor (VGG16 on MFMA)
@pytest.fixture(scope="module")
def torch_dev():
    from shoeprint_image_retrieval_amd.device import TorchDevice

    return TorchDevice()


@pytest.mark.parametrize("block,hw", [(2, (20, 24)), (5, (36, 40)), (10, (40, 36)), (16, (64, 48)), (17, (64, 48)),
                                      (23, (64, 64)), (30, (64, 64))])
def test_vgg16_truncations(block, hw, torch_dev, lib):
    import extractor_cases as ec

    ec.check_block(block, hw, torch_dev, lib)


def test_vgg16_conv3_3_full_size_image(torch_dev, lib):
    """512x256 print through features[:16] (the headline extractor shape) vs torch-CPU."""
    import extractor_cases as ec

    ec.check_block(16, (512, 256), torch_dev, lib, n_images=1)


@pytest.mark.parametrize("arch,block,hw", [("VGG19", 19, (64, 48)), ("VGG19", 37, (64, 64)), ("VGG19_BN", 2, (40, 36)),
                                           ("VGG19_BN", 27, (64, 48)), ("VGG19_BN", 53, (64, 64))])
def test_other_vgg_backbones(arch, block, hw, torch_dev, lib):
    import extractor_cases as ec

    ec.check_other_vgg(arch, block, hw, torch_dev, lib)


@pytest.mark.parametrize("block,hw", [(5, (64, 48)), (6, (100, 72)), (7, (512, 256))])
def test_resnet50_extractor(torch_dev, lib, block, hw):
    """Build-defined ResNet50 (BASELINE config 3) through layer1 / layer2 / layer3 - the last at the full 512x256 print,
    [1024, 32, 16] out - vs torch-CPU with the same seeded parameters."""
    import extractor_cases as ec

    ec.check_resnet50(block, hw, torch_dev, lib, n_images=1 if block == 7 else 2)


@pytest.mark.parametrize("model,block,hw", [("EfficientNetV2_M", 4, (128, 96)), ("EfficientNetV2_M", 6, (512, 256)),
                                            ("EfficientNetV2_S", 5, (96, 64)), ("EfficientNetV2_L", 5, (64, 64)),
                                            ("EfficientNet_B1", 5, (96, 64)), ("EfficientNet_B4", 6, (256, 128)),
                                            ("EfficientNet_B7", 4, (64, 64))])
def test_efficientnet_v2_extractor(torch_dev, lib, model, block, hw):
    """EfficientNetV2 truncations (the reference's run.toml default: EfficientNetV2_M, blocks 4 .. 6) - block 6 at the full
    512x256 print, [176, 32, 16] out - vs torch-CPU with the same seeded parameters."""
    import extractor_cases as ec

    ec.check_effnet(model, block, hw, torch_dev, lib, n_images=1 if hw[0] >= 512 else 2, tol=1e-4)


@pytest.mark.parametrize("block,hw", [(4, (64, 48)), (7, (96, 64)), (9, (128, 96)), (12, (512, 256))])
def test_densenet201_extractor(torch_dev, lib, block, hw):
    """DenseNet_201 truncations - block 12 (all of `features`, [1920, 16, 8] out) at the full 512x256 print - vs torch-CPU
    with the same seeded parameters."""
    import extractor_cases as ec

    ec.check_densenet(block, hw, torch_dev, lib, n_images=1 if hw[0] >= 512 else 2, tol=1e-4)


def test_multi_layer_pipeline(torch_dev, lib, fft_scorer):
    """Config 5: conv3_3 + conv4_3 + conv5_3 taps of one VGG16 pass, scored on separate streams, fused on the device."""
    import extractor_cases as ec

    ec.check_multi_layer_pipeline(torch_dev, lib, fft_scorer, hw=(128, 96), taps=(16, 23, 30), n_gallery=9, n_queries=4, batch=4)
    for compute in ("bfloat16", "float16"):
        ec.check_multi_layer_pipeline(torch_dev, lib, fft_scorer, hw=(128, 96), taps=(16, 23, 30), n_gallery=9, n_queries=4,
                                      batch=4, compute=compute)


def test_rgb_route(torch_dev, lib):
    import extractor_cases as ec

    ec.check_rgb_route(torch_dev, lib, hw=(96, 64), block=10)


def test_extractor_reference_surface(torch_dev, lib):
    import extractor_cases as ec

    ec.check_reference_surface(torch_dev, lib)


def test_end_to_end_pipeline(torch_dev, lib, fft_scorer):
    import extractor_cases as ec

    ec.check_end_to_end(torch_dev, lib, fft_scorer, block=16, hw=(128, 96), n_gallery=10, n_queries=4)


@pytest.mark.parametrize("hw,grid,clip", [((512, 256), (8, 8), 2.0), ((67, 53), (8, 8), 2.0), ((40, 90), (4, 2), 3.5)])
def test_clahe_kernels_match_the_restatement(hw, grid, clip, torch_dev, lib):
    """HIP CLAHE == numpy restatement of OpenCV's algorithm, bit for bit (f2; unpinned by the reference)."""
    import extractor_cases as ec
    from oracle import clahe_oracle
    from shoeprint_image_retrieval_amd import synth

    cfg = {"model": {"type": "VGG16", "clahe_clip_limit": clip, "clahe_tile_grid_size": list(grid)}}
    m = ec.network.Model(cfg, 2, device=torch_dev, library=lib)
    imgs = np.stack([synth.shoeprint_image(3, i, *hw) for i in range(3)])
    got = torch_dev.to_host(m.clahe_device(torch_dev.to_device(imgs)))
    for i in range(len(imgs)):
        np.testing.assert_array_equal(got[i], clahe_oracle.clahe(imgs[i], clip, grid))


# ------------------------------------------------------------------------------ run.toml-driven end to end (f3)
def test_run_driver_on_an_image_directory(tmp_path, capsys):
    """run_mi355x.main(run.toml) on a two-cluster Gallery/Query directory == dataloader -> oracle chain."""
    import dataset_util
    import run_mi355x
    from oracle import clahe_oracle, ncc_oracle, vgg_oracle
    from shoeprint_image_retrieval_amd import synth
    from shoeprint_image_retrieval_amd.dataloader import Dataloader

    case = next(c for c in dataset_util.CASES if c["name"] == "wvu_split")
    cfg = dataset_util.write_dataset(str(tmp_path), case)
    toml = tmp_path / "run.toml"
    toml.write_text(
        f'[dataset]\ndir = "{tmp_path}"\ntype = "WVU2019"\ncrop = {case["crop"]}\nn_processes = 3\nn_clusters = 2\n'
        f'cluster_minimise_tolerance = 0.05\n[model]\ntype = "VGG16"\nclahe_clip_limit = 2.0\nclahe_tile_grid_size = [8, 8]\n'
        f'start_block = 16\nend_block = 9\nskip_blocks = []\nminimum_dim = 120\nmaximum_dim = 200\n'
        f'[comparison]\nn_processes = 2\n')
    got = run_mi355x.main(str(toml))
    out = capsys.readouterr().out
    assert "2 clusters of image sizes found." in out and "rank-1:" in out
    # second run with a gallery feature cache: first fills it, then reads it; ranks unchanged
    toml.write_text(toml.read_text() + f'[mi355x]\ngallery_cache = "{tmp_path}/cache"\n')
    assert run_mi355x.main(str(toml)) == got
    assert run_mi355x.main(str(toml)) == got
    assert "Gallery features from" in capsys.readouterr().out
    want = []
    for queries, gallery, matches, block in Dataloader(cfg):
        params = synth.vgg16_parameters(1234, vgg_oracle.conv_shapes(block))
        feats = lambda ims: [vgg_oracle.get_feature_maps(clahe_oracle.clahe(im, 2.0, (8, 8)), block, params) for im in ims]
        want += [int(r) for r in ncc_oracle.compare_maps(feats(queries), feats(gallery), matches, cfg)]
    assert got == want


def test_run_driver_with_a_16bit_extractor_and_16bit_maps(tmp_path, capsys):
    """run_mi355x.main with [mi355x] extractor_dtype = "bfloat16" and dtype = "bfloat16" (features extracted on the 16-bit matrix
    cores, stored as bfloat16 for the scorer): runs end to end, uses a gallery cache of its own, and on this easy two-cluster
    directory every query still ranks where the float32 run ranks it, give or take one place."""
    import dataset_util
    import run_mi355x

    case = next(c for c in dataset_util.CASES if c["name"] == "wvu_split")
    dataset_util.write_dataset(str(tmp_path), case)
    toml = tmp_path / "run.toml"
    base = (f'[dataset]\ndir = "{tmp_path}"\ntype = "WVU2019"\ncrop = {case["crop"]}\nn_processes = 3\nn_clusters = 2\n'
            f'cluster_minimise_tolerance = 0.05\n[model]\ntype = "VGG16"\nclahe_clip_limit = 2.0\nclahe_tile_grid_size = [8, 8]\n'
            f'start_block = 16\nend_block = 9\nskip_blocks = []\nminimum_dim = 120\nmaximum_dim = 200\n'
            f'[comparison]\nn_processes = 2\n[mi355x]\ngallery_cache = "{tmp_path}/cache"\n')
    toml.write_text(base)
    exact = run_mi355x.main(str(toml))
    toml.write_text(base + 'extractor_dtype = "bfloat16"\ndtype = "bfloat16"\n')
    got = run_mi355x.main(str(toml))
    out = capsys.readouterr().out
    assert "rank-1:" in out and out.count("Gallery features from") == 0  # (the float32 cache is not reused: other key)
    assert len(got) == len(exact) and max(abs(a - b) for a, b in zip(got, exact)) <= 1, (got, exact)
    assert run_mi355x.main(str(toml)) == got and "Gallery features from" in capsys.readouterr().out


def test_run_driver_with_the_reference_default_model(tmp_path, capsys):
    """The [model] and [comparison] sections of the reference's own run.toml - EfficientNetV2_M, start_block 6, end_block 4,
    block 5 skipped, rotated and scaled query variants - on a two-cluster image directory == dataloader -> oracle chain
    (EfficientNetV2 oracle with the same seeded parameters, NCC oracle with the same variants)."""
    import dataset_util
    import run_mi355x
    from oracle import clahe_oracle, effnet_oracle, ncc_oracle
    from shoeprint_image_retrieval_amd import network, synth
    from shoeprint_image_retrieval_amd.dataloader import Dataloader

    case = next(c for c in dataset_util.CASES if c["name"] == "wvu_split")
    cfg = dataset_util.write_dataset(str(tmp_path), case)
    toml = tmp_path / "run.toml"
    toml.write_text(
        f'[dataset]\ndir = "{tmp_path}"\ntype = "WVU2019"\ncrop = {case["crop"]}\nn_processes = 3\nn_clusters = 2\n'
        f'cluster_minimise_tolerance = 0.05\n[model]\ntype = "EfficientNetV2_M"\nclahe_clip_limit = 2.0\n'
        f'clahe_tile_grid_size = [8, 8]\nstart_block = 6\nend_block = 4\nskip_blocks = [5]\nminimum_dim = 120\nmaximum_dim = 200\n'
        f'[comparison]\nn_processes = 2\nrotations = [-9, 3, 180]\nscales = [1.04]\n')
    got = run_mi355x.main(str(toml))
    assert "rank-1:" in capsys.readouterr().out
    from shoeprint_image_retrieval_amd.config import load_config

    config = load_config(str(toml))
    want, blocks = [], set()
    for queries, gallery, matches, block in Dataloader(config):
        blocks.add(block)
        m = network.Model(config, block)
        ops = m.effnet_ops()
        params = synth.effnet_parameters(1234, ops)
        feats = lambda ims: [effnet_oracle.get_feature_maps(clahe_oracle.clahe(im, 2.0, (8, 8)), ops, params, m.mean, m.std, m.bn_eps)
                             for im in ims]
        want += [int(r) for r in ncc_oracle.compare_maps(feats(queries), feats(gallery), matches, config)]
        m.close()
    assert blocks <= {4, 6} and got == want


def test_torch_ops_equal_the_ctypes_route_bit_for_bit(lib, monkeypatch):
    """torch.ops.shoeprint_mi355x.{ncc_scores, ranks, extract} against the ctypes binding of the same C ABI: identical bits,
    on float32 conv4_3-sized maps (FFT form), bf16 28 x 12 maps (matrix-core form), a chunked gallery, and VGG16 features;
    and the host mirror (NccScorer / Model / compare_maps) routes through the ops by default."""
    import torch
    from shoeprint_image_retrieval_amd import _torch_ops, network, similarity, synth
    from shoeprint_image_retrieval_amd.similarity import NccScorer

    ops = _torch_ops.load()
    monkeypatch.setenv("SPR_TORCH_OPS", "0")
    plain = NccScorer(library=lib, method="auto")
    assert plain._torch_ops() is None
    monkeypatch.setenv("SPR_TORCH_OPS", "1")
    routed = NccScorer(library=lib, method="auto")
    assert routed._torch_ops() is not None
    dev = plain.dev
    for (c, h, w), dtype, nq, ng in (((64, 64, 32), torch.float32, 5, 11), ((32, 32, 16), torch.bfloat16, 7, 19),
                                     ((8, 36, 20), torch.float16, 3, 6)):
        m = synth.default_matches(nq, ng)
        g = dev.zeros((ng, c, h, w), np.float32)
        lib.check(lib.spr_synth_gallery(dev.ptr(g), 0, ng, c, h, w, 77, dev.stream()))
        q = dev.zeros((nq, c, h, w), np.float32)
        md = dev.to_device(m)
        lib.check(lib.spr_synth_queries(dev.ptr(q), 0, nq, dev.ptr(md), c, h, w, 77, 2, 1, 20, dev.stream()))
        q, g = q.to(dtype), g.to(dtype)
        want = plain.scores_device(q, g)
        got = ops.ncc_scores(q, g)
        assert torch.equal(got, want) and got.dtype == torch.float32 and tuple(got.shape) == (nq, ng)
        assert torch.equal(routed.scores_device(q, g), want)
        item = plain.plan(c, (h, w), (h, w), dtype=dtype).gallery_item_bytes
        assert torch.equal(ops.ncc_scores(q, g, 2, "auto", 4 * item), want)   # three gallery chunks
        assert torch.equal(ops.ranks(got, md), plain.ranks_device(want, md))
        assert torch.equal(routed.ranks_device(want, md), plain.ranks_device(want, md))
    # on another stream: the ops take PyTorch's current stream
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        again = ops.ncc_scores(q, g)
    side.synchronize()
    assert torch.equal(again, want)
    # extractor
    cfg = {"model": {"type": "VGG16", "clahe_clip_limit": 2.0, "clahe_tile_grid_size": [8, 8]}}
    model = network.Model(cfg, 10, library=lib)
    imgs = torch.randint(0, 256, (3, 64, 48), dtype=torch.uint8, device="cuda")
    monkeypatch.setenv("SPR_TORCH_OPS", "0")
    want = model.extract_device(imgs)
    monkeypatch.setenv("SPR_TORCH_OPS", "1")
    got = ops.extract(imgs, model.packed, 0, 10, list(model.mean), list(model.std))
    assert torch.equal(got, want) and torch.equal(model.extract_device(imgs), want)
    rgb = torch.randint(0, 256, (2, 40, 32, 3), dtype=torch.uint8, device="cuda")
    monkeypatch.setenv("SPR_TORCH_OPS", "0")
    want = model.extract_device(rgb, in_channels=3)
    monkeypatch.setenv("SPR_TORCH_OPS", "1")
    assert torch.equal(model.extract_device(rgb, in_channels=3), want)
    with pytest.raises(RuntimeError, match="channel mismatch"):
        ops.ncc_scores(q, g[:, :4].contiguous())
    model.close()


@pytest.mark.parametrize("compute,arch,block,hw", [("bfloat16", "VGG16", 16, (512, 256)), ("float16", "VGG16", 16, (512, 256)),
                                                   ("bfloat16", "VGG16", 30, (128, 96)), ("float16", "VGG19_BN", 27, (96, 64)),
                                                   ("bfloat16", "VGG16", 7, (50, 38))])
def test_vgg_on_the_16bit_matrix_cores(torch_dev, lib, compute, arch, block, hw):
    """spr_vgg_plan_create_ex(SPR_BF16 / SPR_F16): features[:block] with 16-bit operands and f32 accumulation against the
    torch-CPU oracle on the SAME rounded weights and activations (tolerance stated in extractor_cases.TOL16)."""
    import extractor_cases as ec

    ec.check_block16(block, hw, torch_dev, lib, compute, arch=arch, n_images=1 if hw[0] >= 512 else 2)


@pytest.mark.parametrize("compute,block,hw", [("bfloat16", 7, (512, 256)), ("float16", 7, (160, 96)), ("bfloat16", 6, (100, 70))])
def test_resnet50_on_the_16bit_matrix_cores(torch_dev, lib, compute, block, hw):
    import extractor_cases as ec

    ec.check_resnet50_16(block, hw, torch_dev, lib, compute)


@pytest.mark.parametrize("model,block,hw,compute", [("EfficientNetV2_M", 6, (512, 256), "bfloat16"), ("EfficientNetV2_S", 7, (192, 128), "float16"),
                                                    ("EfficientNet_B3", 6, (160, 96), "bfloat16"), ("EfficientNetV2_L", 9, (96, 64), "float16")])
def test_efficientnet_on_the_16bit_matrix_cores(torch_dev, lib, model, block, hw, compute):
    import extractor_cases as ec

    ec.check_effnet16(model, block, hw, torch_dev, lib, compute)
