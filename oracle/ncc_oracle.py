"""CPU oracle for the query-vs-gallery NCC scorer and ranker.

TEST INFRASTRUCTURE ONLY.  This file is the *checker* for the HIP path: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  Nothing under ``shoeprint-image-retrieval_amd/``
imports it, and the product path raises if the HIP library is missing instead
of falling back to this code.

It restates, in numpy/scipy, the algorithm of the reference scorer
(``/root/reference/src/shoeprint_image_retrieval/similarity.py``):

* ``ncc_maps`` / ``normxcorr``      <- similarity.py:26-72  (normxcorr)
* ``get_similarity``                <- similarity.py:75-108
* ``similarity_matrix``             <- similarity.py:355-367 (float32 matrix, floor 0,
                                       max over transform variants)
* ``rank_true_match``               <- similarity.py:378-386 (_get_rank)
* ``compare_maps``                  <- similarity.py:129-227 (+ worker 287-375)
* ``apply_transformations`` / ``transform_variants``  <- similarity.py:230-284, 321-353
* ``cmp`` / ``cmp_all_line``        <- parse_results.py:4-35

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function
here against ``tests/golden/*.npz|json``, which ``oracle/make_golden.py``
produced by importing the real reference in the build container.

Two evaluation modes:

* ``precise=False`` (default) follows the reference's arithmetic types: float32
  mean subtraction, float32 (complex64 FFT) numerator, float64 box sums.  It is
  what the CPU baseline times ("port").
* ``precise=True`` evaluates the same closed form with a float64 numerator; it
  is the ground truth the HIP parity tests compare against.
"""

from __future__ import annotations

import multiprocessing as mp
from typing import Sequence

import numpy as np
from scipy.signal import fftconvolve

CROP = 2  # similarity.py:92-93 — two feature pixels are cut from every edge


def _centre(x: np.ndarray) -> np.ndarray:
    """Subtract the per-map mean in the array's own dtype (similarity.py:48-49)."""
    return x - x.mean(axis=(-2, -1), keepdims=True, dtype=x.dtype)


def _window_sums(img0: np.ndarray, th: int, tw: int) -> tuple[np.ndarray, np.ndarray]:
    """float64 sums of img0 and fl32(img0**2) over the th x tw window that the
    'same'-mode correlation places at every output pixel (similarity.py:57-59).

    Window for output (y, x) covers rows y - th//2 ... y - th//2 + th - 1 (zero
    outside the image), likewise for columns.  Evaluated exactly with summed-area
    tables instead of the reference's two FFT convolutions with a ones kernel.
    """
    h, w = img0.shape[-2:]
    sq = np.square(img0).astype(np.float64)  # np.square keeps float32, then widened
    lin = img0.astype(np.float64)

    def box(a: np.ndarray) -> np.ndarray:
        sat = np.zeros(a.shape[:-2] + (h + 1, w + 1), dtype=np.float64)
        sat[..., 1:, 1:] = a.cumsum(axis=-2).cumsum(axis=-1)
        y0 = np.clip(np.arange(h) - th // 2, 0, h)
        y1 = np.clip(np.arange(h) - th // 2 + th, 0, h)
        x0 = np.clip(np.arange(w) - tw // 2, 0, w)
        x1 = np.clip(np.arange(w) - tw // 2 + tw, 0, w)
        return (
            sat[..., y1[:, None], x1[None, :]]
            - sat[..., y0[:, None], x1[None, :]]
            - sat[..., y1[:, None], x0[None, :]]
            + sat[..., y0[:, None], x0[None, :]]
        )

    return box(lin), box(sq)


def ncc_maps(
    templates: np.ndarray,
    images: np.ndarray,
    *,
    precise: bool = False,
    box: str = "fft",
) -> np.ndarray:
    """Per-channel 'same'-mode NCC maps of a [C,th,tw] stack against a [C,h,w] stack.

    Follows similarity.py:48-70 channel by channel (vectorised over C):
    out[c,y,x] = num / sqrt(var * T) with
      num  = sum_{u,v} t0[c,u,v] * I0z[c, y+u-th//2, x+v-tw//2]      (:53-55)
      var  = S2 - S1^2 / (th*tw), negatives clamped to 0             (:57-65)
      T    = sum t0^2                                                (:67)
    and every non-finite quotient replaced by 0 (:70).

    ``box`` selects how S1/S2 are evaluated: "fft" = float64 FFT convolution with
    a ones kernel exactly as the reference does (speed-faithful), "sat" = exact
    float64 summed-area tables.
    """
    t0 = _centre(np.asarray(templates))
    i0 = _centre(np.asarray(images))
    th, tw = t0.shape[-2:]
    flipped = t0[..., ::-1, ::-1]
    if precise:
        num = fftconvolve(i0.astype(np.float64), flipped.astype(np.float64), mode="same", axes=(-2, -1))
    else:
        num = fftconvolve(i0, flipped, mode="same", axes=(-2, -1))
    if box == "fft" and not precise:
        ones = np.ones((1,) * (t0.ndim - 2) + (th, tw))  # float64, as np.ones in :50
        s2 = fftconvolve(np.square(i0), ones, mode="same", axes=(-2, -1))
        s1 = fftconvolve(i0, ones, mode="same", axes=(-2, -1))
    else:
        s1, s2 = _window_sums(i0, th, tw)
    var = s2 - np.square(s1) / float(th * tw)
    var[var < 0] = 0
    energy = np.square(t0).sum(axis=(-2, -1), keepdims=True, dtype=np.float64 if precise else t0.dtype)
    with np.errstate(divide="ignore", invalid="ignore"):
        out = num / np.sqrt(var * energy)
    out[~np.isfinite(out)] = 0
    return out


def normxcorr(template: np.ndarray, image: np.ndarray, mode: str = "same", *, precise: bool = False) -> np.ndarray:
    """2-D entry point with the reference's signature (similarity.py:26-31).

    Only ``mode="same"`` is ever used by the reference (similarity.py:104) and
    only that mode is restated.
    """
    if mode != "same":
        raise ValueError("only mode='same' is on the hot path (similarity.py:104)")
    return ncc_maps(template[None], image[None], precise=precise)[0]


def get_similarity(shoemark: np.ndarray, shoeprint: np.ndarray, *, precise: bool = False) -> np.floating:
    """Crop both stacks, sum the per-channel NCC maps, take the spatial maximum and
    divide by the channel count (similarity.py:92-108)."""
    mark = shoemark[:, CROP:-CROP, CROP:-CROP]
    prnt = shoeprint[:, CROP:-CROP, CROP:-CROP]
    summed = ncc_maps(mark, prnt, precise=precise).sum(axis=0)
    return summed.max() / len(mark)


# --------------------------------------------------------------------------- variants (f1)
def apply_transformations(maps: Sequence[np.ndarray], value: float, kind: str) -> list[np.ndarray]:
    """One rotated (NEAREST, no expand, zero fill) or resized (BICUBIC, int()-truncated
    size) copy of every query stack, channel by channel through Pillow mode-'F'
    images exactly as similarity.py:260-278 does."""
    from PIL import Image  # local import: only the f1 path needs Pillow

    out = []
    for stack in maps:
        chans = []
        for fmap in stack:
            img = Image.fromarray(fmap)
            if kind == "rotate":
                img = img.rotate(value)
            elif kind == "scale":
                img = img.resize((int(img.width * value), int(img.height * value)))
            else:
                raise ValueError(kind)
            chans.append(np.array(img))
        out.append(np.array(chans))
    return out


def transform_variants(
    maps: Sequence[np.ndarray],
    rotations: Sequence[float] | None,
    scales: Sequence[float] | None,
) -> list[list[np.ndarray]]:
    """Variant lists in the reference's order (similarity.py:321-353, 282).

    rotations only: [orig, rot...];  scales only: [orig, scale...];
    both: [orig] + every scale of [orig, rot...]  => 1 + (R+1)*S lists — the
    rotation-only variants are dropped by the reference (the second insert(0)
    puts the originals in front of the *scaled* lists only).
    """
    orig = list(maps)
    if rotations is None and scales is None:
        return [orig]
    if scales is None:
        return [orig] + [apply_transformations(orig, r, "rotate") for r in rotations]
    if rotations is None:
        return [orig] + [apply_transformations(orig, s, "scale") for s in scales]
    rotated = [orig] + [apply_transformations(orig, r, "rotate") for r in rotations]
    scaled = [apply_transformations(v, s, "scale") for v in rotated for s in scales]
    return [orig] + scaled


# --------------------------------------------------------------------------- matrix, ranks
def similarity_matrix(
    shoemark_maps: Sequence[np.ndarray],
    shoeprint_maps: Sequence[np.ndarray],
    rotations: Sequence[float] | None = None,
    scales: Sequence[float] | None = None,
    *,
    precise: bool = False,
) -> np.ndarray:
    """float32 [Q,G]: running maximum over variants, starting from 0.0
    (similarity.py:355-367 — negative similarities therefore clamp to 0)."""
    sims = np.zeros((len(shoemark_maps), len(shoeprint_maps)), dtype=np.float32)
    for variant in transform_variants(shoemark_maps, rotations, scales):
        for qi, mark in enumerate(variant):
            for gi, prnt in enumerate(shoeprint_maps):
                s = get_similarity(mark, prnt, precise=precise)
                if s > sims[qi, gi]:
                    sims[qi, gi] = s
    return sims


def rank_true_match(similarities: np.ndarray, match: int) -> int:
    """1-based position of ``match`` in the descending order of ``similarities``
    (similarity.py:381-386).

    The reference sorts with numpy's default (unstable) argsort and flips, so
    exact ties involving the true match are platform dependent there.  The
    oracle and the HIP ranker both use the rule a *stable* ascending argsort +
    flip would give: tied items with a larger index come first, i.e.
    rank = 1 + #{s_j > s_m} + #{j > m : s_j == s_m}.
    """
    s = np.asarray(similarities)
    if not 0 <= match < len(s):
        raise IndexError("matching shoeprint id is not in the gallery")  # :386 raises IndexError
    greater = int(np.count_nonzero(s > s[match]))
    tied_after = int(np.count_nonzero(s[match + 1 :] == s[match]))
    return 1 + greater + tied_after


def ranks_from_matrix(sims: np.ndarray, matching_pairs: Sequence[int]) -> np.ndarray:
    return np.array([rank_true_match(sims[q], matching_pairs[q]) for q in range(len(sims))], dtype=np.int32)


def _chunk_bounds(n_items: int, n_chunks: int) -> list[tuple[int, int]]:
    """Contiguous chunks, the first ``n_items % n_chunks`` one longer (similarity.py:146-157)."""
    base, extra = divmod(n_items, n_chunks)
    bounds, start = [], 0
    for i in range(n_chunks):
        end = start + base + (1 if i < extra else 0)
        bounds.append((start, end))
        start = end
    return bounds


_POOL_GALLERY: list[np.ndarray] = []


def _pool_rows(args):
    marks, rotations, scales = args
    return similarity_matrix(marks, _POOL_GALLERY, rotations, scales)


def compare_maps(
    shoemark_maps: Sequence[np.ndarray],
    shoeprint_maps: Sequence[np.ndarray],
    matching_pairs: Sequence[int],
    config: dict,
    *,
    return_matrix: bool = False,
):
    """Ranks of the true matches (similarity.py:129-227): the queries are split
    into ``n_processes`` contiguous chunks, each worker scores its chunk against
    the whole gallery, then every query's true match is ranked.

    Unlike the reference the variant count follows what the workers really
    produce, so rotations+scales does not hang.
    """
    global _POOL_GALLERY
    comp = config["comparison"]
    n_proc = max(1, int(comp["n_processes"]))
    rotations, scales = comp.get("rotations"), comp.get("scales")
    bounds = [b for b in _chunk_bounds(len(shoemark_maps), n_proc) if b[1] > b[0]]
    if n_proc == 1 or len(bounds) <= 1:
        sims = similarity_matrix(shoemark_maps, shoeprint_maps, rotations, scales)
    else:
        _POOL_GALLERY = list(shoeprint_maps)  # inherited by fork, like the shared Array (:164-176)
        ctx = mp.get_context("fork")
        with ctx.Pool(len(bounds)) as pool:
            parts = pool.map(_pool_rows, [(list(shoemark_maps[a:b]), rotations, scales) for a, b in bounds])
        _POOL_GALLERY = []
        sims = np.concatenate(parts, axis=0)
    ranks = ranks_from_matrix(sims, matching_pairs)
    return (ranks, sims) if return_matrix else ranks


# --------------------------------------------------------------------------- S-scores
def cmp(rankings: Sequence[int], p: int, total_shoeprints: int, total_shoemarks: int) -> float:
    """Fraction of queries whose true match is within the top p % of the gallery
    (parse_results.py:4-24)."""
    limit = (p * total_shoeprints) / 100
    return sum(1 for r in rankings if r <= limit) / total_shoemarks


def cmp_all_line(rankings: Sequence[int], total_shoeprints: int, total_shoemarks: int) -> str:
    """The line parse_results.py:27-35 prints."""
    vals = [cmp(rankings, p, total_shoeprints, total_shoemarks) * 100 for p in (1, 5, 10, 15, 20)]
    return "S1:{:.2f} S5:{:.2f} S10:{:.2f} S15:{:.2f} S20:{:.2f}".format(*vals)
