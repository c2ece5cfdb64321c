"""Seeded synthetic feature maps and shoeprint-like images (host twin).

Neither WVU2019 nor pretrained weights exist offline ("Mismatches"),
so tests and ``bench.py`` run on WVU2019-*shaped* synthetic data.  Everything is
derived from one counter-based generator (splitmix64) using integer arithmetic
and power-of-two scalings only, so the numpy twin here and the HIP generator in
``csrc/synth.hip`` produce bit-identical float32 values: the GPU box regenerates
inputs from a seed and only small *outputs* are committed as fixtures.

Feature model (post-ReLU activations, ~50 % zeros like VGG maps):

* gallery item ``g``:  relu(n_g[c,y,x])                         n = Irwin-Hall(4) noise
* query ``q`` matching gallery item ``m`` with shift (dy,dx):
      relu((3*n_m[c,y+dy,x+dx] + 2*n'_q[c,y,x]) / 4)            (noise only where the
                                                                shifted pixel is outside)
"""

from __future__ import annotations

import numpy as np

MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

STREAM_GALLERY = 1
STREAM_QUERY = 2
STREAM_SHIFT = 3
STREAM_IMAGE = 4
STREAM_WEIGHT = 5


def splitmix64(x: np.ndarray | int) -> np.ndarray:
    """One splitmix64 output for every 64-bit counter in ``x``."""
    with np.errstate(over="ignore"):
        z = (np.asarray(x, dtype=np.uint64) + _GOLD) & MASK
        z = ((z ^ (z >> np.uint64(30))) * _M1) & MASK
        z = ((z ^ (z >> np.uint64(27))) * _M2) & MASK
        return z ^ (z >> np.uint64(31))


def stream_key(seed: int, stream: int, item: int) -> np.uint64:
    """Key of one item's stream: two chained splitmix64 rounds over (seed, stream, item)."""
    with np.errstate(over="ignore"):
        k = splitmix64(np.uint64(seed))
        k = splitmix64((k ^ (np.uint64(stream) << np.uint64(56))) + np.uint64(item))
    return np.uint64(k)


def irwin_hall_int(key: np.uint64, idx: np.ndarray) -> np.ndarray:
    """Sum of the four 16-bit fields of splitmix64(key + idx), centred: an int32 in
    [-131070, 131070] with standard deviation ~37 837 (approximately normal)."""
    with np.errstate(over="ignore"):
        r = splitmix64((np.uint64(key) + idx.astype(np.uint64)) & MASK)
    f = np.uint64(0xFFFF)
    s = (r & f) + ((r >> np.uint64(16)) & f) + ((r >> np.uint64(32)) & f) + (r >> np.uint64(48))
    return s.astype(np.int64).astype(np.int32) - np.int32(131070)


def gallery_features(seed: int, g: int, c: int, h: int, w: int) -> np.ndarray:
    """float32 [c,h,w] post-ReLU features of gallery item ``g``."""
    n = irwin_hall_int(stream_key(seed, STREAM_GALLERY, g), np.arange(c * h * w, dtype=np.int64))
    return (np.maximum(n, 0).astype(np.float32) * np.float32(1.0 / 32768.0)).reshape(c, h, w)


def query_shift(seed: int, q: int, max_shift: int = 3) -> tuple[int, int]:
    r = int(splitmix64(stream_key(seed, STREAM_SHIFT, q)))
    span = 2 * max_shift + 1
    return (r & 0xFFFF) % span - max_shift, ((r >> 16) & 0xFFFF) % span - max_shift


def query_features(seed: int, q: int, match: int, c: int, h: int, w: int, max_shift: int = 3,
                   signal: int = 3, noise: int = 2) -> np.ndarray:
    """float32 [c,h,w] features of query ``q``: a shifted, noised copy of gallery item ``match``.

    ``signal``/``noise`` are the small integer mixing weights (3, 2 by default; e.g. 1, 6
    gives a hard set whose true matches are not all ranked first)."""
    dy, dx = query_shift(seed, q, max_shift)
    cc, yy, xx = np.meshgrid(np.arange(c), np.arange(h), np.arange(w), indexing="ij")
    sy, sx = yy + dy, xx + dx
    inside = (sy >= 0) & (sy < h) & (sx >= 0) & (sx < w)
    src = (cc * h + np.clip(sy, 0, h - 1)) * w + np.clip(sx, 0, w - 1)
    base = irwin_hall_int(stream_key(seed, STREAM_GALLERY, match), src.ravel().astype(np.int64)).reshape(c, h, w)
    noise_i = irwin_hall_int(stream_key(seed, STREAM_QUERY, q), np.arange(c * h * w, dtype=np.int64)).reshape(c, h, w)
    mix = np.where(inside, signal * base, 0) + noise * noise_i  # |mix| < 2^24: exact in float32
    return np.maximum(mix, 0).astype(np.float32) * np.float32(1.0 / 131072.0)


def default_matches(n_queries: int, n_gallery: int) -> np.ndarray:
    """Gallery index of every query's true match: spread over the gallery, distinct while Q <= G."""
    step = max(1, n_gallery // max(1, n_queries))
    return ((np.arange(n_queries, dtype=np.int64) * step) % n_gallery).astype(np.int32)


def dataset(seed: int, n_queries: int, n_gallery: int, c: int, h: int, w: int, signal: int = 3, noise: int = 2):
    """(query list, gallery list, matching ids) in the reference's list-of-arrays form."""
    matches = default_matches(n_queries, n_gallery)
    gallery = [gallery_features(seed, g, c, h, w) for g in range(n_gallery)]
    queries = [query_features(seed, q, int(matches[q]), c, h, w, signal=signal, noise=noise)
               for q in range(n_queries)]
    return queries, gallery, [int(m) for m in matches]


def shoeprint_image(seed: int, item: int, h: int = 512, w: int = 256) -> np.ndarray:
    """uint8 [h,w] shoeprint-like image: low-pass filtered noise thresholded to ~50 % ink
    with a soft edge (ridge/blob texture), for the extractor tests."""
    n = irwin_hall_int(stream_key(seed, STREAM_IMAGE, item), np.arange(h * w, dtype=np.int64))
    f = n.astype(np.float64).reshape(h, w)
    for axis, k in ((0, 9), (1, 9)):
        cs = np.cumsum(np.concatenate([np.zeros_like(np.take(f, [0], axis=axis)), f], axis=axis), axis=axis)
        lo = np.clip(np.arange(f.shape[axis]) - k // 2, 0, f.shape[axis])
        hi = np.clip(np.arange(f.shape[axis]) - k // 2 + k, 0, f.shape[axis])
        f = np.take(cs, hi, axis=axis) - np.take(cs, lo, axis=axis)
    f = f / (np.abs(f).max() + 1e-9)
    img = 127.5 + 127.5 * np.tanh(6.0 * f)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def vgg16_parameters(seed: int, conv_shapes) -> list[tuple[np.ndarray, np.ndarray]]:
    """Seeded He-normal weights [cout,cin,3,3] and small biases [cout] for every convolution
    (pretrained weights cannot be fetched offline)."""
    params = []
    for i, (cin, cout) in enumerate(conv_shapes):
        n = cout * cin * 9
        w = irwin_hall_int(stream_key(seed, STREAM_WEIGHT, 2 * i), np.arange(n, dtype=np.int64)).astype(np.float32)
        w = (w * np.float32(np.sqrt(2.0 / (cin * 9)) / 37837.0)).reshape(cout, cin, 3, 3)
        b = irwin_hall_int(stream_key(seed, STREAM_WEIGHT, 2 * i + 1), np.arange(cout, dtype=np.int64)).astype(np.float32)
        params.append((w, b * np.float32(0.05 / 37837.0)))
    return params


def vgg_parameters(seed: int, conv_shapes, bn_flags=None):
    """``vgg16_parameters`` plus, where ``bn_flags[i]`` is set, BatchNorm2d parameters (gamma ~ 1 +- 0.1, beta and
    running mean small, running variance in [0.6, 1.4]) for the convolution's BatchNorm layer."""
    params = vgg16_parameters(seed, conv_shapes)
    if not bn_flags:
        return params
    out = []
    for i, ((w, b), bn) in enumerate(zip(params, bn_flags)):
        if not bn:
            out.append((w, b))
            continue
        idx = np.arange(w.shape[0], dtype=np.int64)
        u = [irwin_hall_int(stream_key(seed, STREAM_WEIGHT, 1000 + 4 * i + k), idx).astype(np.float32) / np.float32(37837.0)
             for k in range(4)]  # ~N(0,1) each
        gamma = np.float32(1.0) + np.float32(0.1) * np.clip(u[0], -3, 3)
        beta = np.float32(0.05) * u[1]
        mean = np.float32(0.1) * u[2]
        var = np.float32(1.0) + np.float32(0.4) * np.tanh(u[3])
        out.append((w, b, gamma.astype(np.float32), beta.astype(np.float32), mean.astype(np.float32), var.astype(np.float32)))
    return out


def resnet_parameters(seed: int, conv_specs):
    """Seeded parameters of the build-defined ResNet50 extractor: for every (cin, cout, ksize, stride, role) a He-normal
    convolution weight [cout,cin,k,k], a small bias and BatchNorm2d (gamma, beta, running mean, running variance).  The
    last BatchNorm of a bottleneck (role 3) gets gamma ~ 0.5 so that activations stay O(1) through sixteen residual sums."""
    out = []
    for i, (cin, cout, ks, _stride, role) in enumerate(conv_specs):
        n = cout * cin * ks * ks
        w = irwin_hall_int(stream_key(seed, STREAM_WEIGHT, 5000 + 8 * i), np.arange(n, dtype=np.int64)).astype(np.float32)
        w = (w * np.float32(np.sqrt(2.0 / (cin * ks * ks)) / 37837.0)).reshape(cout, cin, ks, ks)
        idx = np.arange(cout, dtype=np.int64)
        u = [irwin_hall_int(stream_key(seed, STREAM_WEIGHT, 5000 + 8 * i + 1 + k), idx).astype(np.float32) / np.float32(37837.0)
             for k in range(5)]
        b = np.float32(0.05) * u[0]
        gamma = (np.float32(0.5) if role == 3 else np.float32(1.0)) * (np.float32(1.0) + np.float32(0.1) * np.clip(u[1], -3, 3))
        beta = np.float32(0.05) * u[2]
        mean = np.float32(0.1) * u[3]
        var = np.float32(1.0) + np.float32(0.4) * np.tanh(u[4])
        out.append(tuple(np.ascontiguousarray(t, dtype=np.float32) for t in (w, b, gamma, beta, mean, var)))
    return out


def effnet_parameters(seed: int, ops):
    """Seeded parameters of an EfficientNetV2 truncation (``Model.effnet_ops``): per convolution / depthwise layer a He-normal
    weight, a zero bias (torchvision's have none) and BatchNorm2d (gamma, beta, running mean, running variance); per
    squeeze-excitation (fc1 weight, bias, fc2 weight, bias).  Projections into a residual sum get gamma ~ 0.5."""
    def normal(i, k, n):
        return irwin_hall_int(stream_key(seed, STREAM_WEIGHT, 9000 + 8 * i + k), np.arange(n, dtype=np.int64)).astype(np.float32) / np.float32(37837.0)

    out = []
    for i, op in enumerate(ops):
        cin, cout, ks = op["cin"], op["cout"], op["ks"]
        if op["kind"] == 2:
            sq = op["sq"]
            w1 = (normal(i, 0, sq * cin) * np.float32(np.sqrt(1.0 / cin))).reshape(sq, cin, 1, 1)
            w2 = (normal(i, 1, cin * sq) * np.float32(np.sqrt(1.0 / sq))).reshape(cin, sq, 1, 1)
            out.append((w1, np.float32(0.1) * normal(i, 2, sq), w2, np.float32(1.0) + np.float32(0.3) * normal(i, 3, cin)))
            continue
        fan = ks * ks if op["kind"] == 1 else cin * ks * ks
        shape = (cin, 1, ks, ks) if op["kind"] == 1 else (cout, cin, ks, ks)
        n_out = shape[0]
        w = (normal(i, 0, int(np.prod(shape))) * np.float32(np.sqrt(2.0 / fan))).reshape(shape)
        u = [normal(i, 1 + k, n_out) for k in range(4)]
        gamma = (np.float32(0.5) if op["kind"] == 0 and op["act"] == 0 else np.float32(1.0)) * \
            (np.float32(1.0) + np.float32(0.1) * np.clip(u[0], -3, 3))
        out.append(tuple(np.ascontiguousarray(t, dtype=np.float32) for t in
                         (w, np.zeros(n_out, np.float32), gamma, np.float32(0.05) * u[1], np.float32(0.1) * u[2],
                          np.float32(1.0) + np.float32(0.4) * np.tanh(u[3]))))
    return out


def densenet_parameters(seed: int, ops):
    """Seeded parameters of a DenseNet_201 truncation (``Model.densenet_ops``), in torchvision's module order per layer:
    stem (conv0 weight, norm0 gamma, beta, mean, variance); dense 1x1 (norm1 x4, conv1 weight, norm2 x4); dense 3x3 (conv2
    weight); transition (norm x4, conv weight); closing BatchNorm (x4)."""
    def normal(i, k, n):
        return irwin_hall_int(stream_key(seed, STREAM_WEIGHT, 13000 + 16 * i + k), np.arange(n, dtype=np.int64)).astype(np.float32) / np.float32(37837.0)

    def bn(i, k, n):
        u = [normal(i, k + j, n) for j in range(4)]
        return (np.float32(1.0) + np.float32(0.1) * np.clip(u[0], -3, 3), np.float32(0.05) * u[1], np.float32(0.1) * u[2],
                np.float32(1.0) + np.float32(0.4) * np.tanh(u[3]))

    def conv(i, k, cout, cin, ks):
        return (normal(i, k, cout * cin * ks * ks) * np.float32(np.sqrt(2.0 / (cin * ks * ks)))).reshape(cout, cin, ks, ks)

    out = []
    for i, op in enumerate(ops):
        if op["kind"] == 0:
            out.append((conv(i, 0, 64, 3, 7),) + bn(i, 1, 64))
        elif op["kind"] == 1:
            out.append(bn(i, 0, op["cin"]) + (conv(i, 4, 128, op["cin"], 1),) + bn(i, 5, 128))
        elif op["kind"] == 2:
            out.append((conv(i, 0, 32, 128, 3),))
        elif op["kind"] == 3:
            out.append(bn(i, 0, op["cin"]) + (conv(i, 4, op["cout"], op["cin"], 1),))
        else:
            out.append(bn(i, 0, op["cin"]))
    return [tuple(np.ascontiguousarray(t, dtype=np.float32) for t in p) for p in out]


def bfloat16_bits(x: np.ndarray) -> np.ndarray:
    """float32 -> bfloat16 bit patterns (uint16), round to nearest even — the storage form the scorer
    accepts for bf16 features (numpy has no bfloat16 type)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    rounded = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return rounded.astype(np.uint16)


def from_bfloat16_bits(b: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)
