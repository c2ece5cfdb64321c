"""`not gpu`: the VGG16 extractor kernels under the CPU emulation vs the torch-CPU oracle, the CLAHE
restatement's invariants, and the end-to-end images -> features -> ranks pipeline."""

import numpy as np
import pytest

import extractor_cases as ec
from emu_util import emu_library, emu_scorer
from host_device import HostDevice
from oracle import clahe_oracle as clahe
from shoeprint_image_retrieval_amd import config as cfgmod


@pytest.mark.parametrize("block,hw", [(1, (18, 20)), (2, (20, 24)), (3, (17, 33)), (5, (36, 40)), (7, (32, 32)),
                                      (10, (40, 36)), (12, (24, 40))])
def test_emu_vgg16_truncations(block, hw):
    ec.check_block(block, hw, HostDevice(), emu_library())


def test_emu_vgg16_conv3_3_block16():
    ec.check_block(16, (32, 48), HostDevice(), emu_library(), n_images=1)


@pytest.mark.parametrize("arch,block,hw", [("VGG19", 12, (24, 32)), ("VGG19", 19, (32, 32)), ("VGG19_BN", 1, (18, 20)),
                                           ("VGG19_BN", 2, (18, 20)), ("VGG19_BN", 3, (20, 18)), ("VGG19_BN", 14, (24, 24))])
def test_emu_other_vgg_backbones(arch, block, hw):
    ec.check_other_vgg(arch, block, hw, HostDevice(), emu_library())


@pytest.mark.parametrize("compute,arch,block,hw", [("bfloat16", "VGG16", 5, (36, 40)), ("float16", "VGG16", 10, (40, 36)),
                                                   ("bfloat16", "VGG16", 3, (17, 33)), ("float16", "VGG19_BN", 14, (24, 24)),
                                                   ("bfloat16", "VGG16", 2, (18, 20))])
def test_emu_vgg_16bit_matrix_cores(compute, arch, block, hw):
    """The extractor on the 16-bit matrix cores (v_mfma_f32_16x16x32_{bf16,f16}; spr_vgg_plan_create_ex) under emulation:
    truncations ending behind a pool, a ReLU and a bare convolution; a BatchNorm folded before the weights are rounded; a
    cut (block 2) that leaves only the f32 first layer."""
    ec.check_block16(block, hw, HostDevice(), emu_library(), compute, arch=arch, n_images=1)


def test_emu_resnet50_layer1():
    """The build-defined ResNet50 extractor under emulation: stem, max pool, the three bottlenecks of layer1 (1x1 and 3x3
    implicit GEMMs, downsample branch, residual sums) on a 40x36 image - odd-sized maps after the strided layers."""
    ec.check_resnet50(5, (40, 36), HostDevice(), emu_library(), n_images=1)


@pytest.mark.parametrize("compute,hw", [("bfloat16", (40, 36)), ("float16", (34, 47))])
def test_emu_resnet50_layer1_16bit(compute, hw):
    """layer1 on the 16-bit GEMM kernel under emulation: 1x1 and 3x3 implicit GEMMs with 128-pixel tiles (ragged last tile),
    the downsample branch, rounded residual operands, float32 NCHW out."""
    ec.check_resnet50_16(5, hw, HostDevice(), emu_library(), compute)


def test_emu_resnet50_16bit_wide_tiles():
    """The same with the 128-channel tiles forced (SPR_GEMM16_BN=128; read once per process, hence the child process): the
    launcher picks them only for grids of 512 workgroups and more, which no emulated shape reaches."""
    import os
    import subprocess
    import sys

    env = dict(os.environ, SPR_GEMM16_BN="128")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.abspath(__file__), "-k",
                        "resnet50_layer1_16bit and bfloat16"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("model,block,hw,rgb", [("EfficientNetV2_M", 2, (40, 36), False), ("EfficientNetV2_M", 5, (40, 32), False),
                                                ("EfficientNetV2_S", 3, (34, 32), True), ("EfficientNet_B1", 4, (40, 32), False)])
def test_emu_efficientnet_v2(model, block, hw, rgb):
    """EfficientNetV2 truncations under emulation: stem, FusedMBConv stages (3x3 expansion + 1x1 projection on the GEMM
    kernel, channel counts padded to 64), and with block 5 an MBConv stage (1x1 expansion, depthwise 3x3 / stride 2,
    squeeze-excitation, scaled 1x1 projection)."""
    ec.check_effnet(model, block, hw, HostDevice(), emu_library(), n_images=1, rgb=rgb)


@pytest.mark.parametrize("model,block,hw,compute", [("EfficientNetV2_M", 3, (40, 36), "bfloat16"), ("EfficientNetV2_S", 5, (40, 32), "float16"),
                                                    ("EfficientNet_B1", 4, (40, 32), "bfloat16")])
def test_emu_efficientnet_16bit(model, block, hw, compute):
    """EfficientNet truncations on the 16-bit kernels under emulation: matrix-core stem (3x3 / stride 2, SiLU), FusedMBConv
    and MBConv stages on the 16-bit GEMM kernel (SiLU in front of the residual sum, squeeze-excitation factors on the
    operand), 16-bit depthwise convolutions (3x3 and, in the B-series, 5x5) and squeeze-excitation means."""
    ec.check_effnet16(model, block, hw, HostDevice(), emu_library(), compute)


@pytest.mark.parametrize("block,hw", [(1, (34, 32)), (3, (34, 32)), (5, (40, 36)), (6, (36, 40))])
def test_emu_densenet201(block, hw):
    """DenseNet_201 truncations under emulation: the stem with and without norm0 / relu0 / pool0, the first dense block
    (BatchNorm + ReLU on the operand load of the 1x1 convolutions, 3x3 outputs stored into their channel range of the block
    tensor), a transition (average pool into the next block's tensor)."""
    ec.check_densenet(block, hw, HostDevice(), emu_library(), n_images=1)


def test_architecture_tables_against_the_oracles_own():
    """Every EfficientNet / DenseNet truncation the library builds, against the architecture tables the ORACLES hold on their
    own (torchvision's published settings), plus constants pinned by hand from torchvision's definitions."""
    from oracle import densenet_oracle, effnet_oracle
    from shoeprint_image_retrieval_amd import network

    import ctypes

    lib = emu_library()  # (plans only: no parameters are generated, nothing is launched)
    for model_str, (arch, *_rest) in network._EFFNET_MODELS.items():
        n_stages = len(effnet_oracle.stages(model_str))
        for block in range(1, n_stages + 3):  # up to the whole of `features`, the closing 1x1 convolution included
            handle = ctypes.c_void_p()
            lib.check(lib.spr_effnet_plan_create(arch, block, ctypes.byref(handle)))
            ec.check_effnet_tables(model_str, block, network.effnet_plan_ops(lib, handle))
            lib.spr_effnet_plan_destroy(handle)
    for block in range(1, 13):
        handle = ctypes.c_void_p()
        lib.check(lib.spr_densenet_plan_create(block, ctypes.byref(handle)))
        ec.check_densenet_tables(block, network.densenet_plan_ops(lib, handle))
        lib.spr_densenet_plan_destroy(handle)
    # pinned by hand
    out = lambda model, block: [o for o in effnet_oracle.arch_ops(model, block) if o["kind"] == 0][-1]["cout"]
    assert out("EfficientNetV2_M", 6) == 176 and out("EfficientNetV2_S", 7) == 256 and out("EfficientNetV2_L", 8) == 640
    assert out("EfficientNet_B7", 1) == 64 and out("EfficientNet_B1", 1) == 32 and out("EfficientNet_B4", 1) == 48
    assert out("EfficientNet_B7", 8) == 640 and out("EfficientNet_B3", 8) == 384 and out("EfficientNet_B5", 8) == 512
    assert out("EfficientNetV2_M", 9) == 1280 and out("EfficientNetV2_S", 8) == 1280 and out("EfficientNet_B7", 9) == 2560
    assert out("EfficientNet_B2", 9) == 1408 and effnet_oracle.arch_ops("EfficientNetV2_S", 8)[-1]["names"] == ("features.7.0", "features.7.1")
    depth = lambda model: [s[6] for s in effnet_oracle.stages(model)]
    assert depth("EfficientNet_B7") == [4, 7, 7, 10, 10, 13, 4] and depth("EfficientNet_B1") == [2, 3, 3, 4, 4, 5, 2]
    assert depth("EfficientNet_B4") == [2, 4, 4, 6, 6, 8, 2] and depth("EfficientNetV2_M") == [3, 5, 5, 7, 14, 18, 5]
    assert sum(1 for o in effnet_oracle.arch_ops("EfficientNetV2_M", 6) if o["kind"] != 2) + \
        sum(1 for o in effnet_oracle.arch_ops("EfficientNetV2_M", 6) if o["kind"] == 2) == len(effnet_oracle.arch_ops("EfficientNetV2_M", 6))
    se = [o for o in effnet_oracle.arch_ops("EfficientNetV2_S", 5) if o["kind"] == 2]
    assert [o["sq"] for o in se[:2]] == [16, 32] and se[0]["names"] == ("features.4.0.block.2.fc1", "features.4.0.block.2.fc2")
    b1 = effnet_oracle.arch_ops("EfficientNet_B1", 2)  # stage 1 has expansion 1: no expansion convolution, SE hidden width 8
    assert [o["kind"] for o in b1[:4]] == [0, 1, 2, 0] and b1[2]["sq"] == 8 and b1[1]["names"][0] == "features.1.0.block.0.0"
    assert effnet_oracle.make_divisible(40 * 1.1) == 48 and effnet_oracle.make_divisible(16 * 1.1) == 16
    d = densenet_oracle.arch_ops(12)
    assert d[-1]["kind"] == 4 and d[-1]["cin"] == 1920 and len(d) == 1 + 2 * (6 + 12 + 48 + 32) + 3 + 1
    assert [o["cout"] for o in d if o["kind"] == 3] == [128, 256, 896]
    assert d[1]["names"] == ("features.denseblock1.denselayer1.norm1", "features.denseblock1.denselayer1.conv1",
                             "features.denseblock1.denselayer1.norm2")
    assert densenet_oracle.arch_ops(11)[-1]["names"] == ("features.denseblock4.denselayer32.conv2",)


def test_emu_multi_layer_pipeline():
    ec.check_multi_layer_pipeline(HostDevice(), emu_library(), emu_scorer("fft"), hw=(40, 32), n_gallery=5, n_queries=2, batch=2)


def test_emu_multi_layer_pipeline_16bit_extractor():
    ec.check_multi_layer_pipeline(HostDevice(), emu_library(), emu_scorer("fft"), hw=(40, 32), n_gallery=4, n_queries=2, batch=2,
                                  compute="bfloat16")


def test_emu_rgb_route():
    ec.check_rgb_route(HostDevice(), emu_library())


def test_emu_reference_surface():
    ec.check_reference_surface(HostDevice(), emu_library())


def test_emu_end_to_end_pipeline():
    ec.check_end_to_end(HostDevice(), emu_library(), emu_scorer("auto"))


def test_bad_block_is_rejected():
    from shoeprint_image_retrieval_amd import _lib
    for block in (0, 32):
        with pytest.raises(_lib.SprError):
            ec.make_model(block, HostDevice(), emu_library())


def test_clahe_invariants():
    rng = np.random.default_rng(0)
    flat = np.full((64, 48), 77, np.uint8)
    out = clahe.clahe(flat, 2.0, (8, 8))
    assert out.shape == flat.shape and out.dtype == np.uint8 and len(np.unique(out)) == 1
    img = (rng.random((67, 53)) * 60 + 90).astype(np.uint8)  # low contrast, size not divisible by the grid
    out = clahe.clahe(img, 2.0, (8, 8))
    assert out.shape == img.shape
    assert int(out.max()) - int(out.min()) > int(img.max()) - int(img.min())  # contrast is stretched
    # monotone within a tile-constant neighbourhood: order of grey levels is preserved on a ramp
    ramp = np.tile(np.arange(0, 256, 4, dtype=np.uint8), (64, 1))
    r = clahe.clahe(ramp, 0.0, (1, 1))  # no clipping, one tile = plain histogram equalisation
    assert (np.diff(r[0].astype(int)) >= 0).all()
    with pytest.raises(ValueError):
        clahe.clahe(img.astype(np.float32))


@pytest.mark.parametrize("hw,grid,clip", [((64, 48), (8, 8), 2.0), ((67, 53), (8, 8), 2.0), ((40, 90), (4, 2), 3.5),
                                          ((33, 31), (8, 8), 0.0), ((128, 64), (8, 8), 40.0)])
def test_emu_clahe_kernels_match_the_restatement(hw, grid, clip):
    """HIP CLAHE == numpy restatement of OpenCV's algorithm, bit for bit (incl. reflect-101 extension)."""
    from shoeprint_image_retrieval_amd import synth

    dev, lib = HostDevice(), emu_library()
    cfg = {"model": {"type": "VGG16", "clahe_clip_limit": clip, "clahe_tile_grid_size": list(grid)}}
    m = ec.network.Model(cfg, 2, device=dev, library=lib)
    rng = np.random.default_rng(1)
    imgs = np.stack([synth.shoeprint_image(3, 0, *hw), (rng.random(hw) * 60 + 90).astype(np.uint8),
                     np.full(hw, 200, np.uint8)])
    got = dev.to_host(m.clahe_device(dev.to_device(imgs)))
    for i in range(len(imgs)):
        np.testing.assert_array_equal(got[i], clahe.clahe(imgs[i], clip, grid))


def test_load_config_matches_reference_schema(tmp_path):
    p = tmp_path / "run.toml"
    p.write_text('[dataset]\ndir="x"\ntype="WVU2019"\ncrop=[0,0]\nn_processes=2\nn_clusters=1\n'
                 'cluster_minimise_tolerance=0.05\n[model]\ntype="VGG16"\nclahe_clip_limit=2.0\n'
                 'clahe_tile_grid_size=[8,8]\nstart_block=16\nend_block=9\nskip_blocks=[]\nminimum_dim=300\n'
                 'maximum_dim=800\n[comparison]\nn_processes=30\nrotations=""\nscales=[1.02, 1.04]\n')
    cfg = cfgmod.load_config(p)
    assert cfg["comparison"]["rotations"] is None and cfg["comparison"]["scales"] == [1.02, 1.04]  # config.py:60-63
    assert cfg["model"]["type"] == "VGG16" and cfg["mi355x"]["ncc_method"] == "auto"
