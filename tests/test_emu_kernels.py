"""`not gpu`: run the HIP kernels under the CPU emulation (tests/emu) against oracle + golden vectors,
and check the C ABI of the real gfx950 build (symbols only — no compute without a GPU)."""

import ctypes
import os
import re

import numpy as np
import pytest

import parity_cases as pc
from emu_util import emu_library, emu_scorer
from shoeprint_image_retrieval_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", params=["fft", "direct"])
def scorer(request):
    return emu_scorer(request.param)


@pytest.mark.parametrize("name", ["tiny", "hard"])
def test_emu_golden_compare_maps(scorer, name):
    pc.check_golden_compare_maps(scorer, name)


def test_emu_golden_ragged(scorer):
    pc.check_golden_ragged(scorer)


def test_emu_golden_normxcorr(scorer):
    pc.check_golden_normxcorr(scorer)


def test_emu_golden_get_similarity(scorer):
    pc.check_golden_get_similarity(scorer, max_elems=64 * 32 * 16)


def test_emu_narrow_maps(scorer):
    pc.check_narrow_maps(scorer)


def test_emu_reverse_work_item_order():
    """The emulation runs the work-items of a workgroup one after another, in ascending order, so a data race between
    work-items always resolves the same way - possibly the right way, where the hardware does not (round 2: two work-items
    of one wave storing to the same table entry).  SPR_EMU_ORDER=reverse runs them in descending order: the checks that
    compare whole maps and matrices must not notice."""
    import subprocess
    import sys

    env = dict(os.environ, SPR_EMU_ORDER="reverse")
    sel = "narrow_maps or golden_normxcorr or golden_ragged or accumulate or matrix_core or sparse_channels"
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.abspath(__file__), "-k", sel],
                       env=env, capture_output=True, text=True, cwd=ROOT, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("case", pc.SHAPE_CASES)
def test_emu_shape_classes(scorer, case):
    pc.check_shape_case(scorer, case)


@pytest.mark.parametrize("method", ["fft", "fft_pow2"])
@pytest.mark.parametrize("case", pc.BIG_SHAPE_CASES)
def test_emu_conv3_sized_maps_fft(case, method):
    pc.check_shape_case(emu_scorer(method), case)


def test_emu_team_schedule(monkeypatch):
    pc.check_team_mode(emu_scorer("fft"), monkeypatch)


def test_emu_big_mode(monkeypatch):
    pc.check_big_mode(lambda: emu_scorer("fft"), monkeypatch, full=False)


def test_emu_sparse_channels():
    pc.check_sparse_channels(emu_scorer("fft"), c=3)


def test_emu_launch_slicing(monkeypatch):
    pc.check_launch_slicing(lambda m: emu_scorer(m), monkeypatch, big_c=1)


def test_emu_config_selects_the_scorer():
    from host_device import HostDevice
    from shoeprint_image_retrieval_amd.similarity import scorer_from_config

    dev = HostDevice()
    pc.check_config_selects_the_scorer(lambda cfg: scorer_from_config(cfg, device=dev, library=emu_library()))


def test_emu_empty_and_degenerate_sets():
    pc.check_empty_and_degenerate_sets(emu_scorer("auto"))


def test_emu_rank_kernel(scorer):
    pc.check_rank_kernel(scorer)


def test_emu_synth_twin(scorer):
    pc.check_synth_twin(scorer, emu_library())


def test_emu_query_variants(scorer):
    pc.check_variants(scorer)


def test_emu_accumulate_and_floor(scorer):
    pc.check_variant_accumulate(scorer)


def test_emu_variant_chunk_budget():
    pc.check_variant_chunk_budget(lambda **kw: emu_scorer("fft", **kw))


def test_emu_gallery_chunking_matches_single_pass():
    from shoeprint_image_retrieval_amd import synth

    q, g, _ = synth.dataset(11, 3, 9, 2, 16, 12, signal=1, noise=6)
    whole = emu_scorer("fft").score_matrix(q, g)
    one = emu_scorer("fft")
    plan = one.plan(2, (16, 12), (16, 12))
    chunked = emu_scorer("fft", max_prepared_bytes=2 * plan.gallery_item_bytes)
    dev = chunked.dev
    got = dev.to_host(chunked.scores_device(dev.to_device(np.stack(q)), dev.to_device(np.stack(g))))
    np.testing.assert_array_equal(got, whole)


def test_emu_auto_method_and_errors():
    lib = emu_library()
    sc = emu_scorer("auto")
    assert sc.plan(4, (16, 12), (16, 12)).method == _lib.NCC_FFT
    assert sc.plan(4, (16, 12), (16, 12)).fft_size == (32, 16)
    assert sc.plan(256, (128, 64), (128, 64)).fft_size == (192, 96)   # VGG16 conv3_3 of a 512x256 print: 3*2^k grid
    assert emu_scorer("fft_pow2").plan(256, (128, 64), (128, 64)).fft_size == (256, 128)
    assert sc.plan(512, (64, 32), (64, 32)).fft_size == (96, 48)      # conv4_3
    assert emu_scorer("fft_pow2").plan(512, (64, 32), (64, 32)).fft_size == (128, 64)
    assert sc.plan(512, (32, 16), (32, 16)).fft_size == (48, 24)      # conv5_3: 3*2^k grid, one wave per pair
    assert emu_scorer("fft_pow2").plan(512, (32, 16), (32, 16)).fft_size == (64, 32)
    with pytest.raises(_lib.SprError) as e:
        sc.plan(4, (4, 4), (16, 12))  # vanishes under the 2-pixel crop
    assert e.value.code == -2
    with pytest.raises(_lib.SprError) as e:
        sc.plan(4, (400, 300), (400, 300))  # nothing LDS-resident fits
    assert e.value.code == -3
    shape = _lib.NccShape(4, 16, 12, 16, 12, 2, 7, 0)
    handle = ctypes.c_void_p()
    assert lib.spr_ncc_plan_create(ctypes.byref(shape), ctypes.byref(handle)) == -1
    assert b"dtype" in lib.spr_last_error()
    assert lib.spr_rank_true_match(None, 4, 2, 4, None, None, None) == -1


def test_float16_feature_storage():
    """fp16 feature storage: parity = oracle on the same rounded features upcast to fp32."""
    from oracle import ncc_oracle as oracle
    from shoeprint_image_retrieval_amd import synth

    sc = emu_scorer("fft")
    dev = sc.dev
    q, g, _ = synth.dataset(21, 2, 3, 3, 16, 12, signal=1, noise=6)
    q16, g16 = np.stack(q).astype(np.float16), np.stack(g).astype(np.float16)
    plan = sc.plan(3, (16, 12), (16, 12), dtype=np.float16)
    pq = sc.prepare_queries(plan, dev.to_device(q16))
    pg = sc.prepare_gallery(plan, dev.to_device(g16))
    scores = dev.zeros((2, 3), np.float32)
    sc.score_prepared(plan, pq, 2, pg, 3, scores, 3, 0)
    ref = oracle.similarity_matrix(list(q16.astype(np.float32)), list(g16.astype(np.float32)), precise=True)
    np.testing.assert_allclose(dev.to_host(scores), ref, atol=pc.TIGHT)


def test_emu_reduced_precision_configs():
    sc = emu_scorer("fft")
    pc.check_config3_bf16_resnet_layer3(sc, channels=16)
    pc.check_config5_multi_layer_fp16(sc, channels=(2, 4, 4))


@pytest.mark.parametrize("channels,nq,ng", [(3, 5, 2), (2, 70, 2), (2, 17, 3), (17, 3, 2)])
def test_emu_matrix_core_method(channels, nq, ng):
    pc.check_mfma_method(emu_scorer, channels, nq, ng)


def test_emu_matrix_core_method_fp16():
    pc.check_mfma_method_fp16(emu_scorer, 3, 18, 2)


def test_emu_matrix_core_general_shapes():
    """The general instance of the matrix-core scorer (templates up to 30 x 16 on maps up to 28 x 12) under emulation."""
    pc.check_mfma_general_shapes(emu_scorer, channels=3, nq=2, ng=3)


def test_emu_matrix_core_general_shapes_split_form(monkeypatch):
    monkeypatch.setenv("SPR_NCC_MFMA_EXACT", "0")
    pc.check_mfma_general_shapes(emu_scorer, channels=2, nq=2, ng=2, shapes=pc.MFMA_GENERAL_SHAPES[:3],
                                 )


@pytest.mark.parametrize("exact", ["1", "0"])
def test_emu_matrix_core_method_conditioning(monkeypatch, exact):
    monkeypatch.setenv("SPR_NCC_MFMA_EXACT", exact)
    pc.check_mfma_conditioning(emu_scorer)
    pc.check_mfma_degenerate_channels(emu_scorer)


def test_emu_matrix_core_method_split_form(monkeypatch):
    """SPR_NCC_MFMA_EXACT=0: the centred search map as hi + lo on the matrix cores instead of the raw map + correction matrix."""
    monkeypatch.setenv("SPR_NCC_MFMA_EXACT", "0")
    pc.check_mfma_method(emu_scorer, 3, 20, 2)


# ------------------------------------------------------------------------- real library: ABI only
def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "shoeprint_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spr_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_lib.SIGNATURES)


def test_real_library_exports_every_declared_symbol():
    path = _lib.DEFAULT_PATH
    if not os.path.exists(path):
        import __graft_entry__

        __graft_entry__.build()
    cdll = ctypes.CDLL(path)
    for name in _declared_symbols():
        assert hasattr(cdll, name), name
    cdll.spr_abi_version.restype = ctypes.c_int
    assert cdll.spr_abi_version() == 1


def test_product_refuses_to_run_without_gpu_or_library(tmp_path):
    import torch

    from shoeprint_image_retrieval_amd.device import TorchDevice

    with pytest.raises(RuntimeError, match="no fallback"):
        _lib.Library(str(tmp_path / "missing.so"))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            TorchDevice()


def test_emu_kernels_under_ubsan():
    """Sanitizer pass on the CPU build (GPU sanitizers are not available on the pool): the UBSan-instrumented
    (trap mode) emulation library runs a scorer + ranker + variants + extractor slice; any undefined
    behaviour (misaligned or out-of-bounds-typed access, signed overflow, bad shift ...) kills the process."""
    import subprocess
    import sys

    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import build_emu
from host_device import HostDevice
from shoeprint_image_retrieval_amd import _lib, synth, network
from shoeprint_image_retrieval_amd.similarity import NccScorer, compare_maps
lib = _lib.load_library(build_emu.build(sanitize=True))
cfg = {"comparison": {"n_processes": 1, "rotations": [7], "scales": [1.1]},
       "model": {"type": "VGG16", "clahe_clip_limit": 2.0, "clahe_tile_grid_size": [8, 8]}}
for method in ("fft", "direct"):
    sc = NccScorer(device=HostDevice(), library=lib, method=method)
    q, g, m = synth.dataset(3, 2, 5, 3, 20, 14, signal=1, noise=6)
    print(method, compare_maps(q, g, m, cfg, scorer=sc))
sc = NccScorer(device=HostDevice(), library=lib, method="fft")
q, g, m = synth.dataset(4, 1, 2, 2, 128, 64)
print("conv3 grid", sc.plan(2, (128, 64), (128, 64)).fft_size, sc.score_matrix(q, g))
mdl = network.Model(cfg, 5, device=HostDevice(), library=lib)
print(mdl.get_feature_maps(synth.shoeprint_image(1, 0, 24, 20)).shape)
print("UBSAN-CLEAN")
""" % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emu"))
    env = dict(os.environ, SPR_EMU_THREADS="2")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "UBSAN-CLEAN" in out.stdout


def test_torch_ops_are_registered_and_refuse_cpu_tensors():
    """torch.ops.shoeprint_mi355x (csrc/torch_ops.cpp): the TORCH_LIBRARY registration north_star names.  Without a GPU:
    the extension loads, the three operators carry the schemas SURVEY 8(b) lists, and - there being no CPU path - a CPU
    tensor is refused loudly."""
    import torch
    from shoeprint_image_retrieval_amd import _torch_ops

    ops = _torch_ops.load()
    schemas = {n: str(getattr(ops, n).default._schema) for n in ("ncc_scores", "ranks", "extract")}
    assert schemas["ncc_scores"] == ('shoeprint_mi355x::ncc_scores(Tensor q, Tensor g, int crop=2, str method="auto", '
                                     'int max_prepared_bytes=0) -> Tensor')
    assert schemas["ranks"] == "shoeprint_mi355x::ranks(Tensor scores, Tensor match) -> Tensor"
    assert schemas["extract"].startswith("shoeprint_mi355x::extract(Tensor images, Tensor packed, int arch, int block, float[] mean")
    with pytest.raises(RuntimeError, match="must live in HBM"):
        ops.ncc_scores(torch.zeros(1, 2, 8, 8), torch.zeros(1, 2, 8, 8))
    with pytest.raises(RuntimeError, match="must live in HBM"):
        ops.ranks(torch.zeros(2, 3), torch.zeros(2, dtype=torch.int32))
    # both shared objects resolve to the same C-ABI library: the op library links it, it does not carry a second copy
    maps = open("/proc/self/maps").read()
    assert "libshoeprint_torch_ops.so" in maps and "libshoeprint_mi355x.so" in maps
