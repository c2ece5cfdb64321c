"""Extractor parity checks shared by the emulated (not gpu) and MI355X (gpu) tests.
Oracle = torch-CPU float32 ops with the same seeded weights (oracle/vgg_oracle.py; unpinned by the
reference, see its header).  Tolerance: 2e-5 of the largest activation (float32 summation order)."""

import numpy as np

from oracle import clahe_oracle, color_oracle, ncc_oracle, resnet_oracle, vgg_oracle
from shoeprint_image_retrieval_amd import network, similarity, synth

CFG = {"model": {"type": "VGG16", "clahe_clip_limit": 2.0, "clahe_tile_grid_size": [8, 8]},
       "comparison": {"n_processes": 2, "rotations": None, "scales": None}}


def make_model(block, device, lib, **kw):
    return network.Model(CFG, block, device=device, library=lib, **kw)


def check_block(block, hw, device, lib, n_images=2):
    m = make_model(block, device, lib)
    params = synth.vgg16_parameters(1234, m.conv_shapes())
    assert m.conv_shapes() == vgg_oracle.conv_shapes(block)
    imgs = np.stack([synth.shoeprint_image(5, i, *hw) for i in range(n_images)])
    got = device.to_host(m.extract_device(device.to_device(imgs)))
    assert got.dtype == np.float32 and got.shape[1:] == m.output_shape(*hw)
    for i in range(n_images):
        ref = vgg_oracle.get_feature_maps(imgs[i], block, params)
        assert got[i].shape == ref.shape
        np.testing.assert_allclose(got[i], ref, atol=2e-5 * np.abs(ref).max(), rtol=0)
    m.close()


# 16-bit compute types (spr_vgg_plan_create_ex; build-defined, BASELINE configs 3 / 5): the oracle takes the SAME rounded
# weights and rounds the activations at the same places, so what is left is the order of the f32 additions - plus, rarely,
# an activation whose f32 value sits on a rounding boundary and lands one 16-bit step apart (2^-8 relative for bfloat16,
# 2^-11 for float16), which the layers behind it spread (measured on VGG16[:16] at 512 x 256, float16: one element of 2 M off
# by 5.3e-4 of the largest activation).  Tolerance = two such steps, relative to the largest activation:
TOL16 = {"bfloat16": 8e-3, "float16": 1e-3}
TOL16_DEEP = {"bfloat16": 3e-2, "float16": 4e-3}  # the ResNet's 40 layers: see check_resnet50_16


def check_block16(block, hw, device, lib, compute, arch="VGG16", n_images=2):
    cfg = {"model": dict(CFG["model"], type=arch), "comparison": CFG["comparison"], "mi355x": {"extractor_dtype": compute}}
    m = network.Model(cfg, block, device=device, library=lib)
    assert m.compute == compute and lib.spr_vgg_plan_compute(m.handle) == {"float16": 1, "bfloat16": 2}[compute]
    params = synth.vgg_parameters(1234, m.conv_shapes(), [bn for _, bn in m.conv_info()])
    imgs = np.stack([synth.shoeprint_image(5, i, *hw) for i in range(n_images)])
    got = device.to_host(m.extract_device(device.to_device(imgs)))
    assert got.dtype == np.float32 and got.shape[1:] == m.output_shape(*hw)
    exact = []
    for i in range(n_images):
        ref = vgg_oracle.get_feature_maps(imgs[i], block, params, arch, compute=compute)
        assert got[i].shape == ref.shape
        np.testing.assert_allclose(got[i], ref, atol=TOL16[compute] * max(1.0, np.abs(ref).max()), rtol=0)
        exact.append(vgg_oracle.get_feature_maps(imgs[i], block, params, arch))
    # and the 16-bit features stay close to the float32 network's (a sanity bound, not a parity claim)
    e = np.stack(exact)
    assert np.abs(got - e).max() <= (0.05 if compute == "bfloat16" else 0.01) * max(1.0, np.abs(e).max())
    m.close()


def check_other_vgg(arch, block, hw, device, lib):
    """VGG19 / VGG19_BN truncations (reference network.py:121-139) against the torch-CPU oracle, including cuts
    between a convolution and its BatchNorm and between the BatchNorm and its ReLU."""
    cfg = {"model": dict(CFG["model"], type=arch), "comparison": CFG["comparison"]}
    m = network.Model(cfg, block, device=device, library=lib)
    assert m.conv_shapes() == vgg_oracle.conv_shapes(block, arch)
    params = synth.vgg_parameters(1234, m.conv_shapes(), [bn for _, bn in m.conv_info()])
    imgs = np.stack([synth.shoeprint_image(6, i, *hw) for i in range(2)])
    got = device.to_host(m.extract_device(device.to_device(imgs)))
    for i in range(2):
        ref = vgg_oracle.get_feature_maps(imgs[i], block, params, arch)
        assert got[i].shape == ref.shape
        np.testing.assert_allclose(got[i], ref, atol=2e-5 * max(1.0, np.abs(ref).max()), rtol=0)
    m.close()


def check_resnet50(block, hw, device, lib, n_images=2, tol=5e-5):
    """The build-defined ResNet50 extractor (BASELINE config 3) against the torch-CPU oracle with the same seeded
    parameters; parity unpinned by the reference (it has no ResNet).  Tolerance: relative to the largest activation."""
    cfg = {"model": dict(CFG["model"], type="ResNet50"), "comparison": CFG["comparison"]}
    m = network.Model(cfg, block, device=device, library=lib)
    assert m.conv_specs() == resnet_oracle.conv_specs(block)
    params = synth.resnet_parameters(1234, m.conv_specs())
    imgs = np.stack([synth.shoeprint_image(8, i, *hw) for i in range(n_images)])
    got = device.to_host(m.extract_device(device.to_device(imgs)))
    assert got.dtype == np.float32 and got.shape[1:] == m.output_shape(*hw)
    for i in range(n_images):
        ref = resnet_oracle.get_feature_maps(imgs[i], block, params)
        assert got[i].shape == ref.shape
        np.testing.assert_allclose(got[i], ref, atol=tol * max(1.0, np.abs(ref).max()), rtol=0)
    fm = m.get_feature_maps(imgs[0])
    ref = resnet_oracle.get_feature_maps(clahe_oracle.clahe(imgs[0], 2.0, (8, 8)), block, params)
    np.testing.assert_allclose(fm, ref, atol=tol * max(1.0, np.abs(ref).max()), rtol=0)
    m.close()


def check_resnet50_16(block, hw, device, lib, compute, n_images=1):
    """The ResNet50 extractor on the 16-bit matrix cores (spr_resnet_plan_create_ex) against the oracle on the same rounded
    weights, operands and residuals; tolerance TOL16, relative to the largest activation."""
    cfg = {"model": dict(CFG["model"], type="ResNet50"), "comparison": CFG["comparison"], "mi355x": {"extractor_dtype": compute}}
    m = network.Model(cfg, block, device=device, library=lib)
    assert lib.spr_resnet_plan_compute(m.handle) == {"float16": 1, "bfloat16": 2}[compute]
    params = synth.resnet_parameters(1234, m.conv_specs())
    imgs = np.stack([synth.shoeprint_image(8, i, *hw) for i in range(n_images)])
    got = device.to_host(m.extract_device(device.to_device(imgs)))
    assert got.dtype == np.float32 and got.shape[1:] == m.output_shape(*hw)
    for i in range(n_images):
        ref = resnet_oracle.get_feature_maps(imgs[i], block, params, compute=compute)
        # up to 40 layers with residual sums: a stored activation that lands one 16-bit step apart (see TOL16) is carried and
        # amplified by everything behind it (measured through layer3 at 512 x 256, bfloat16: 52 of 524 288 elements beyond
        # two steps, the largest at 1.3 % of the largest activation).  Hence: 99.9 % of the elements within TOL16, all of them
        # within TOL16_DEEP.
        scale = max(1.0, np.abs(ref).max())
        err = np.abs(got[i] - ref)
        assert np.mean(err > TOL16[compute] * scale) < 1e-3, float(np.mean(err > TOL16[compute] * scale))
        assert err.max() <= TOL16_DEEP[compute] * scale, float(err.max() / scale)
        exact = resnet_oracle.get_feature_maps(imgs[i], block, params)
        assert np.abs(got[i] - exact).max() <= (0.08 if compute == "bfloat16" else 0.02) * max(1.0, np.abs(exact).max())
    m.close()


def check_effnet_tables(model_str, block, ops):
    """The library's flattened layer list and the state-dict names derived from it against the oracle's OWN restatement of
    torchvision's architecture tables (stage settings, width / depth scaling, _make_divisible, squeeze widths, residual
    flags, module names) - the oracle graph must not be taken from the library it checks."""
    from oracle import effnet_oracle

    want = effnet_oracle.arch_ops(model_str, block)
    assert len(ops) == len(want), (model_str, block, len(ops), len(want))
    for i, (a, b) in enumerate(zip(ops, want)):
        keys = [k for k in effnet_oracle.ARCH_KEYS if not (b["kind"] == 2 and k in ("ks", "stride", "act"))]
        assert {k: a[k] for k in keys} == {k: b[k] for k in keys}, (model_str, block, i, a, b)
    assert network.effnet_state_names(ops) == [o["names"] for o in want]
    assert network._EFFNET_MODELS[model_str][3] == effnet_oracle.EPS[model_str]


def check_densenet_tables(block, ops):
    from oracle import densenet_oracle

    want = densenet_oracle.arch_ops(block)
    assert len(ops) == len(want), (block, len(ops), len(want))
    for i, (a, b) in enumerate(zip(ops, want)):
        keys = densenet_oracle.ARCH_KEYS + (("c_off", "ctot") if b["kind"] in (1, 2) else ())
        assert {k: a[k] for k in keys} == {k: b[k] for k in keys}, (block, i, a, b)
    assert [tuple(n) for n in network.densenet_state_names(ops)] == [o["names"] for o in want]


def check_effnet(model_str, block, hw, device, lib, n_images=2, tol=5e-5, rgb=False):
    """An EfficientNetV2 truncation features[:block] (the reference's run.toml default is EfficientNetV2_M, blocks 4 .. 6)
    against the torch-CPU oracle with the same seeded parameters; parity unpinned (no torchvision offline).  Tolerance:
    relative to the largest activation."""
    from oracle import effnet_oracle

    cfg = {"model": dict(CFG["model"], type=model_str), "comparison": CFG["comparison"]}
    m = network.Model(cfg, block, device=device, library=lib)
    ops = m.effnet_ops()
    check_effnet_tables(model_str, block, ops)
    params = synth.effnet_parameters(1234, ops)
    if rgb:
        imgs = np.stack([np.stack([synth.shoeprint_image(8 + c, i, *hw) for c in range(3)], axis=-1) for i in range(n_images)])
    else:
        imgs = np.stack([synth.shoeprint_image(8, i, *hw) for i in range(n_images)])
    got = device.to_host(m.extract_device(device.to_device(imgs), in_channels=3 if rgb else 1))
    assert got.dtype == np.float32 and got.shape[1:] == m.output_shape(*hw)
    for i in range(n_images):
        ref = effnet_oracle.get_feature_maps(imgs[i], ops, params, m.mean, m.std, m.bn_eps)
        assert got[i].shape == ref.shape
        np.testing.assert_allclose(got[i], ref, atol=tol * max(1.0, np.abs(ref).max()), rtol=0)
    m.close()


def check_effnet16(model_str, block, hw, device, lib, compute, n_images=1):
    """An EfficientNet truncation on the 16-bit matrix cores (spr_effnet_plan_create_ex) against the oracle with the same
    rounded weights and stored activations.

    Through up to ~110 layers with SiLU and squeeze-excitation two correct 16-bit evaluations that only ORDER their float32
    additions differently drift apart: a stored activation on a rounding boundary lands one step apart, and everything behind
    it carries and amplifies that.  Measured under the bit-exact emulation, EfficientNetV2_M[:6]: rms(kernel - oracle16) =
    0.77 x rms(oracle16 - float32 network) in bfloat16 and 0.79 x in float16 - the distance scales with the step of the type
    (so it is rounding, not a wrong operation) and is as large as the 16-bit oracle's own distance from the float32 network.
    The check is therefore statistical, and weaker than the VGG one: the kernel must be (1) no farther from the 16-bit oracle
    than 1.25 x the oracle is from the float32 network, and (2) no farther from the float32 network than 1.5 x the oracle is
    (rms over the output; maxima within 2 x); shallow truncations (few rounding points) additionally hold TOL16_DEEP."""
    from oracle import effnet_oracle

    cfg = {"model": dict(CFG["model"], type=model_str), "comparison": CFG["comparison"], "mi355x": {"extractor_dtype": compute}}
    m = network.Model(cfg, block, device=device, library=lib)
    assert lib.spr_effnet_plan_compute(m.handle) == {"float16": 1, "bfloat16": 2}[compute]
    ops = m.effnet_ops()
    params = synth.effnet_parameters(1234, ops)
    imgs = np.stack([synth.shoeprint_image(8, i, *hw) for i in range(n_images)])
    got = device.to_host(m.extract_device(device.to_device(imgs)))
    assert got.dtype == np.float32 and got.shape[1:] == m.output_shape(*hw)
    rms = lambda a: float(np.sqrt(np.mean(np.square(a, dtype=np.float64))))
    for i in range(n_images):
        ref = effnet_oracle.get_feature_maps(imgs[i], ops, params, m.mean, m.std, m.bn_eps, compute=compute)
        exact = effnet_oracle.get_feature_maps(imgs[i], ops, params, m.mean, m.std, m.bn_eps)
        assert got[i].shape == ref.shape
        scale = max(1.0, np.abs(exact).max())
        step = {"bfloat16": 2.0 ** -8, "float16": 2.0 ** -11}[compute] * scale  # one step of the type at the largest activation
        noise_rms, noise_max = max(rms(ref - exact), 0.25 * step), max(np.abs(ref - exact).max(), step)
        assert rms(got[i] - ref) <= 1.25 * noise_rms, (rms(got[i] - ref), noise_rms)
        assert rms(got[i] - exact) <= 1.5 * noise_rms, (rms(got[i] - exact), noise_rms)
        assert np.abs(got[i] - ref).max() <= 2.0 * noise_max and np.abs(got[i] - exact).max() <= 2.0 * noise_max
        if len(ops) <= 30:
            assert np.abs(got[i] - ref).max() <= TOL16_DEEP[compute] * scale
    m.close()


def check_densenet(block, hw, device, lib, n_images=2, tol=5e-5):
    """DenseNet_201 features[:block] against the torch-CPU oracle with the same seeded parameters; parity unpinned."""
    from oracle import densenet_oracle

    cfg = {"model": dict(CFG["model"], type="DenseNet_201"), "comparison": CFG["comparison"]}
    m = network.Model(cfg, block, device=device, library=lib)
    ops = m.densenet_ops()
    check_densenet_tables(block, ops)
    params = synth.densenet_parameters(1234, ops)
    imgs = np.stack([synth.shoeprint_image(8, i, *hw) for i in range(n_images)])
    got = device.to_host(m.extract_device(device.to_device(imgs)))
    assert got.dtype == np.float32 and got.shape[1:] == m.output_shape(*hw)
    for i in range(n_images):
        ref = densenet_oracle.get_feature_maps(imgs[i], ops, params, m.mean, m.std)
        assert got[i].shape == ref.shape
        np.testing.assert_allclose(got[i], ref, atol=tol * max(1.0, np.abs(ref).max()), rtol=0)
    m.close()


def check_multi_layer_pipeline(device, lib, scorer, hw=(64, 48), taps=(9, 14, 16), n_gallery=7, n_queries=3, batch=3,
                               compute=None):
    """BASELINE config 5 in miniature: one extractor pass with feature taps, per-layer scoring chains on their own
    streams (gallery in batches, the last one ragged), fusion on the device - against the oracle chain: torch-CPU features
    of every tapped layer, the NCC oracle per layer, the mean.  ``compute`` = a 16-bit extractor plan (the taps are the
    float32 activations BEFORE the rounding the next layer reads them through, i.e. the oracle's truncation at that tap)."""
    from shoeprint_image_retrieval_amd import pipeline

    if compute:
        cfg = dict(CFG)
        cfg["mi355x"] = dict(cfg.get("mi355x", {}), extractor_dtype=compute)
        m = network.Model(cfg, max(taps), device=device, library=lib)
    else:
        m = make_model(max(taps), device, lib)
    params = synth.vgg16_parameters(1234, m.conv_shapes())
    gallery = np.stack([synth.shoeprint_image(23, g, *hw) for g in range(n_gallery)])
    rng = np.random.default_rng(5)
    queries = np.stack([np.clip(np.roll(gallery[q].astype(np.int32), (1, -2), axis=(0, 1)) + rng.normal(0, 20, hw), 0, 255)
                        .astype(np.uint8) for q in range(n_queries)])
    pipe = pipeline.MultiLayerPipeline(m, scorer, taps=taps, batch_size=batch)
    got = device.to_host(pipe.scores_device(device.to_device(queries), device.to_device(gallery)))
    ref = np.zeros((n_queries, n_gallery), dtype=np.float64)
    for t in taps:
        gf = [vgg_oracle.get_feature_maps(clahe_oracle.clahe(im, 2.0, (8, 8)), t, params, compute=compute) for im in gallery]
        qf = [vgg_oracle.get_feature_maps(clahe_oracle.clahe(im, 2.0, (8, 8)), t, params, compute=compute) for im in queries]
        ref += ncc_oracle.similarity_matrix(qf, gf, precise=True).astype(np.float32)
    ref /= len(taps)
    # 16-bit plans: the features agree with the oracle's to a few 16-bit steps (TOL16); the NCC score of such features moves
    # by the same relative amount at most
    np.testing.assert_allclose(got, ref, atol=TOL16[compute] if compute else 1e-4, rtol=0)
    ranks = pipe.ranks(device.to_device(queries), device.to_device(gallery), list(range(n_queries)))
    if not compute:
        np.testing.assert_array_equal(ranks, ncc_oracle.ranks_from_matrix(ref.astype(np.float32), list(range(n_queries))))
    m.close()


def check_rgb_route(device, lib, hw=(40, 32), block=5):
    """RGB images (network.py:199-204, 74-87): RGB -> L*a*b* -> CLAHE(L) -> RGB bit for bit against the oracle's
    restatement, round-trip sanity of the colour transform, and the features of an RGB image (transform_rgb: three
    distinct input planes) against the torch-CPU oracle."""
    rng = np.random.default_rng(11)
    m = make_model(block, device, lib)
    params = synth.vgg16_parameters(1234, m.conv_shapes())
    base = np.stack([synth.shoeprint_image(31, k, *hw) for k in range(3)], axis=-1)      # three decorrelated planes
    imgs = [base, rng.integers(0, 256, size=hw + (3,), dtype=np.uint8), np.full(hw + (3,), 200, np.uint8)]
    for im in imgs:
        got = m._clahe(im)
        want = color_oracle.clahe_rgb(im, 2.0, (8, 8))
        np.testing.assert_array_equal(got, want)
    # the colour transform alone: every grey level maps to a = b = 128 and back to itself within one level
    grey = np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, axis=2)
    lab = color_oracle.rgb_to_lab(grey)
    assert np.all(np.abs(lab[..., 1:].astype(int) - 128) <= 1) and np.all(np.diff(lab[0, :, 0].astype(int)) >= 0)
    assert np.abs(color_oracle.lab_to_rgb(lab).astype(int) - grey).max() <= 2
    fm = m.get_multiple_feature_maps([imgs[0], imgs[1][:, :, 0].copy()], progress=False)
    ref_rgb = vgg_oracle.get_feature_maps(color_oracle.clahe_rgb(imgs[0], 2.0, (8, 8)), block, params)
    ref_grey = vgg_oracle.get_feature_maps(clahe_oracle.clahe(imgs[1][:, :, 0].copy(), 2.0, (8, 8)), block, params)
    np.testing.assert_allclose(fm[0], ref_rgb, atol=2e-5 * np.abs(ref_rgb).max(), rtol=0)
    np.testing.assert_allclose(fm[1], ref_grey, atol=2e-5 * np.abs(ref_grey).max(), rtol=0)
    m.close()


def check_reference_surface(device, lib):
    """Model(config, block), get_feature_maps, get_multiple_feature_maps keep the reference's behaviour."""
    m = make_model(5, device, lib, batch_size=2)
    params = synth.vgg16_parameters(1234, m.conv_shapes())
    imgs = [synth.shoeprint_image(9, i, 40, 32) for i in range(3)] + [synth.shoeprint_image(9, 7, 24, 48)]
    maps = m.get_multiple_feature_maps(imgs, progress=False)
    assert isinstance(maps, list) and len(maps) == 4
    for im, fm in zip(imgs, maps):
        assert fm.dtype == np.float32 and fm.flags["C_CONTIGUOUS"] and fm.ndim == 3
        ref = vgg_oracle.get_feature_maps(clahe_oracle.clahe(im, 2.0, (8, 8)), 5, params)
        np.testing.assert_allclose(fm, ref, atol=2e-5 * np.abs(ref).max(), rtol=0)
    single = m.get_feature_maps(imgs[1])
    np.testing.assert_array_equal(single, maps[1])
    for bad, exc in (("NoSuchNet", LookupError),):
        cfg = {"model": dict(CFG["model"], type=bad)}
        try:
            network.Model(cfg, 5, device=device, library=lib)
        except exc as e:
            if exc is LookupError:
                assert str(e) == "Model string not found"  # network.py:181
        else:
            raise AssertionError(bad)


def check_end_to_end(device, lib, scorer, block=10, hw=(64, 48), n_gallery=6, n_queries=3):
    """SURVEY §8d config 1 in miniature: images -> Model -> compare_maps, against the oracle chain."""
    m = make_model(block, device, lib)
    params = synth.vgg16_parameters(1234, m.conv_shapes())
    gallery = [synth.shoeprint_image(21, g, *hw) for g in range(n_gallery)]
    rng = np.random.default_rng(3)
    queries = []
    for q in range(n_queries):  # degraded copies of gallery[q]: noise + a small shift
        im = np.roll(gallery[q].astype(np.int32), (2, -3), axis=(0, 1)) + rng.normal(0, 25, hw)
        queries.append(np.clip(im, 0, 255).astype(np.uint8))
    matches = list(range(n_queries))
    gf = m.get_multiple_feature_maps(gallery, progress=False)
    qf = m.get_multiple_feature_maps(queries, progress=False)
    ranks = similarity.compare_maps(qf, gf, matches, CFG, scorer=scorer)
    ref_gf = [vgg_oracle.get_feature_maps(clahe_oracle.clahe(im, 2.0, (8, 8)), block, params) for im in gallery]
    ref_qf = [vgg_oracle.get_feature_maps(clahe_oracle.clahe(im, 2.0, (8, 8)), block, params) for im in queries]
    ref_ranks, ref_mat = ncc_oracle.compare_maps(ref_qf, ref_gf, matches, CFG, return_matrix=True)
    mat = scorer.score_matrix(qf, gf)
    np.testing.assert_allclose(mat, ref_mat, atol=1e-4, rtol=0)
    np.testing.assert_array_equal(ranks, ref_ranks)
