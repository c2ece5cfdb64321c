"""The roofline arithmetic bench.py reports must match the figures stated in SURVEY §8d / DESIGN §5."""
import importlib.util
import math
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_direct_form_flops_match_the_survey():
    # conv3_3 of a 512x256 print: cropped 124 x 60 maps, 256 channels -> 15.94 GFLOP per pair
    assert abs(bench.direct_pair_flops(124, 60, 124, 60) * 256 / 1e9 - 15.94) < 0.01
    # conv4_3: 60 x 28, 512 channels -> 1.626; conv5_3: 28 x 12, 512 -> 0.065
    assert abs(bench.direct_pair_flops(60, 28, 60, 28) * 512 / 1e9 - 1.626) < 0.002
    assert abs(bench.direct_pair_flops(28, 12, 28, 12) * 512 / 1e9 - 0.065) < 0.001


def test_fft_form_flops_match_design():
    per_channel = bench.fft_pair_flops((192, 96), 124, 60, 124)
    parts = (6 * 49 * 192, 48 * 5 * 192 * math.log2(192), 62 * 5 * 96 * math.log2(96), 2 * 124 * 60)
    assert abs(per_channel - sum(parts)) < 1e-6
    assert abs(per_channel * 256 / 1e9 - 0.1579) < 1e-4       # 0.1579 GFLOP per pair
    assert bench.PEAK_FP32_TFLOPS == 157.3


def test_clock_sampler_is_inert_without_a_matching_card():
    s = bench.ClockSampler("ffff:ff:1f.0")
    s.start()
    assert s.stop() is None


def test_matrix_core_form_flops():
    # ResNet50 layer3 shape, cropped 28 x 12, 1024 channels: 0.130 GFLOP per pair on in-map pixels (twice SURVEY's 512-channel
    # conv5_3 figure), 0.231 for the dense taps x positions product, 0.308 as issued with template rows padded to 16 taps
    assert abs(bench.direct_pair_flops(28, 12, 28, 12) * 1024 / 1e9 - 0.130) < 0.001
    assert abs(2.0 * 336 * 336 * 1024 / 1e9 - 0.2312) < 1e-4
    assert abs(2.0 * (28 * 16) * 336 * 1024 / 1e9 - 0.3083) < 1e-4
    assert bench.PEAK_BF16_TFLOPS == 2500.0
    assert bench.mfma_tile_steps(28, 12) == (231, 294)  # 63 of the 294 tile steps fall on all-zero fragments
