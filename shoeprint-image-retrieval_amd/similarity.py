"""Query-vs-gallery NCC scoring and ranking on MI355X — host mirror of the reference scorer.

Drop-in for ``src/shoeprint_image_retrieval/similarity.py`` of the reference:

* ``compare_maps(shoemark_maps, shoeprint_maps, matching_pairs, config)``  (similarity.py:129-134)
* ``get_similarity(shoemark, shoeprint)``                                  (similarity.py:75-78)
* ``normxcorr(template, image, mode="same")``                              (similarity.py:26-31)

with the same argument meaning, return types and error behaviour, computed by the HIP
library behind the C ABI of ``include/shoeprint_mi355x.h`` (see ``_lib.py``).  The reference
forks ``n_processes`` CPU workers over query chunks (similarity.py:146-197); here every
(query, gallery) pair is a workgroup of one kernel launch, so ``n_processes`` is accepted
and ignored.  There is no CPU fallback: without the library or a GPU these functions raise.

``NccScorer`` is the device-level interface used by ``bench.py`` and the multi-GPU driver:
it keeps features, prepared spectra and the score matrix resident in HBM.
"""

from __future__ import annotations

import ctypes as C
from typing import Any, Sequence

import numpy as np

from . import _lib
from ._lib import NCC_AUTO, NCC_DIRECT, NCC_FFT, NccShape

CROP = 2  # similarity.py:92-93
_METHODS = {"auto": NCC_AUTO, "fft": NCC_FFT, "direct": NCC_DIRECT, "fft_pow2": _lib.NCC_FFT_POW2, "mfma": _lib.NCC_MFMA}
_DTYPES = {np.dtype(np.float32): _lib.F32, np.dtype(np.float16): _lib.F16}


def _dtype_code(dtype) -> tuple[int, str]:
    """(spr_dtype, cache key) of a feature storage type: numpy float32 / float16, or bfloat16 given as the
    string "bfloat16", a torch.bfloat16, or uint16 bit patterns (numpy has no bfloat16)."""
    name = str(dtype).replace("torch.", "")
    if name in ("bfloat16", "bf16", "uint16", "<class 'numpy.uint16'>"):
        return _lib.BF16, "bf16"
    d = np.dtype(name if name in ("float32", "float16") else dtype)
    return _DTYPES[d], d.str


class _Plan:
    """Owns one spr_ncc_plan (one (query shape, gallery shape) class)."""

    def __init__(self, lib: _lib.Library, channels: int, q_hw, g_hw, crop: int, dtype: int, method: int):
        self.lib = lib
        shape = NccShape(channels, q_hw[0], q_hw[1], g_hw[0], g_hw[1], crop, dtype, method)
        handle = C.c_void_p()
        lib.check(lib.spr_ncc_plan_create(C.byref(shape), C.byref(handle)))
        self.handle = handle
        self.channels, self.q_hw, self.g_hw, self.crop = channels, tuple(q_hw), tuple(g_hw), crop
        self.method = lib.spr_ncc_plan_method(handle)
        rows, cols = C.c_int32(), C.c_int32()
        lib.check(lib.spr_ncc_plan_fft_size(handle, C.byref(rows), C.byref(cols)))
        self.fft_size = (rows.value, cols.value)
        self.query_item_bytes = lib.spr_ncc_query_bytes(handle, 1)
        self.gallery_item_bytes = lib.spr_ncc_gallery_bytes(handle, 1)

    def close(self):
        if self.handle:
            self.lib.spr_ncc_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass


def _as_item_list(maps) -> tuple[list | None, Any]:
    """Normalise the accepted inputs: returns (list of per-item arrays, None) for host lists /
    host arrays, or (None, device_array) for an already device-resident [N,C,h,w] batch."""
    if isinstance(maps, np.ndarray):
        if maps.ndim != 4:
            raise ValueError("a feature batch must be [N, C, h, w]")
        return list(maps), None
    if isinstance(maps, (list, tuple)):
        return list(maps), None
    return None, maps


class NccScorer:
    """Scores queries against a gallery with the HIP NCC kernels; everything stays in HBM.

    Parameters
    ----------
    device : backend object (default ``TorchDevice()``: PyTorch-ROCm on the current GPU)
    library: ``_lib.Library`` (default: the in-tree libshoeprint_mi355x.so)
    method : "auto" | "fft" | "direct" | "fft_pow2" | "mfma"  (pair-kernel choice, see the C header)
    max_prepared_bytes: HBM budget for the prepared form of one gallery chunk (default: a third
        of the free memory, at most 64 GiB); larger galleries are processed chunk by chunk.
    storage: HBM storage type of feature batches uploaded from host lists ("float32" | "float16" | "bfloat16");
        arithmetic is float32 / float64 whatever the storage.
    """

    def __init__(self, device=None, library: _lib.Library | None = None, method: str = "auto",
                 max_prepared_bytes: int | None = None, crop: int = CROP, storage: str = "float32"):
        self.lib = library or _lib.load_library()
        if device is None:
            from .device import TorchDevice

            device = TorchDevice()
        self.dev = device
        self.method = _METHODS[method]
        self.crop = crop
        self.max_prepared_bytes = max_prepared_bytes
        if storage not in ("float32", "float16", "bfloat16"):
            raise ValueError(f"unknown storage type {storage!r}")
        self.storage = storage
        self._plans: dict[tuple, _Plan] = {}
        self._variants = None

    # ------------------------------------------------------------------ plans
    def plan(self, channels: int, q_hw, g_hw, dtype=np.float32, crop: int | None = None) -> _Plan:
        crop = self.crop if crop is None else crop
        code, dkey = _dtype_code(dtype)
        key = (channels, tuple(q_hw), tuple(g_hw), dkey, crop, self.method)
        p = self._plans.get(key)
        if p is None:
            p = _Plan(self.lib, channels, q_hw, g_hw, crop, code, self.method)
            self._plans[key] = p
        return p

    def close(self):
        for p in self._plans.values():
            p.close()
        self._plans.clear()

    # ------------------------------------------------------------------ device-level steps
    def _budget(self) -> int:
        if self.max_prepared_bytes is not None:
            return int(self.max_prepared_bytes)
        return max(256 << 20, min(self.dev.free_bytes() // 3, 64 << 30))

    def prepare_queries(self, plan: _Plan, q_dev):
        n = self.dev.shape(q_dev)[0]
        out = self.dev.empty_bytes(max(1, plan.query_item_bytes * n))
        self.lib.check(self.lib.spr_ncc_prepare_queries(plan.handle, self.dev.ptr(q_dev), n, self.dev.ptr(out),
                                                        self.dev.stream()))
        return out

    def prepare_gallery(self, plan: _Plan, g_dev, out=None):
        n = self.dev.shape(g_dev)[0]
        if out is None:
            out = self.dev.empty_bytes(max(1, plan.gallery_item_bytes * n))
        self.lib.check(self.lib.spr_ncc_prepare_gallery(plan.handle, self.dev.ptr(g_dev), n, self.dev.ptr(out),
                                                        self.dev.stream()))
        return out

    def score_prepared(self, plan: _Plan, pq, nq: int, pg, ng: int, scores, ld: int, col0: int,
                       accumulate_max: bool = False):
        self.lib.check(self.lib.spr_ncc_score(plan.handle, self.dev.ptr(pq), nq, self.dev.ptr(pg), ng,
                                              self.dev.ptr(scores), ld, col0, 1 if accumulate_max else 0,
                                              self.dev.stream()))

    def gallery_chunk_items(self, plan: _Plan, n_gallery: int, share: int = 1) -> int:
        """Gallery items per prepared chunk; ``share`` = how many prepared forms of the chunk are alive at once (one per
        distinct query-variant shape: the 1/sigma map depends on the template size) and split the budget."""
        per_item = plan.gallery_item_bytes
        return int(max(1, min(n_gallery, self._budget() // max(1, share) // max(1, per_item), 65535)))

    def scores_device(self, q_dev, g_dev, scores=None, accumulate_max: bool = False, plan: _Plan | None = None):
        """[Q,G] float32 score matrix (device) of a uniform query batch [Q,C,h,w] against a uniform
        gallery batch [G,C,h',w'], both already in HBM.  One step of the hot path.  The storage type is
        taken from the buffers: float32, float16, or bfloat16 (torch.bfloat16 or uint16 bit patterns); the
        arithmetic is float32 / float64 as ever."""
        nq, c, qh, qw = self.dev.shape(q_dev)
        ng, c2, gh, gw = self.dev.shape(g_dev)
        if c != c2:
            raise ValueError(f"channel mismatch: queries {c}, gallery {c2}")
        if plan is None and scores is None and not accumulate_max and self._torch_ops() is not None:
            # north_star's named mechanism: the registered PyTorch-ROCm custom op (csrc/torch_ops.cpp), same entry points
            if str(q_dev.dtype) != str(g_dev.dtype):
                raise ValueError(f"storage type mismatch: queries {q_dev.dtype}, gallery {g_dev.dtype}")
            return self._torch_ops().ncc_scores(q_dev, g_dev, self.crop, _lib.METHOD_NAMES[self.method], self._budget())
        if plan is None:
            if str(q_dev.dtype) != str(g_dev.dtype):
                raise ValueError(f"storage type mismatch: queries {q_dev.dtype}, gallery {g_dev.dtype}")
            plan = self.plan(c, (qh, qw), (gh, gw), dtype=q_dev.dtype)
        if scores is None:
            scores = self.dev.zeros((nq, ng), np.float32)
        if nq == 0 or ng == 0:
            return scores
        pq = self.prepare_queries(plan, q_dev)
        chunk = self.gallery_chunk_items(plan, ng)
        pg = self.dev.empty_bytes(plan.gallery_item_bytes * chunk)
        for start in range(0, ng, chunk):
            n = min(chunk, ng - start)
            self.prepare_gallery(plan, self.dev.narrow0(g_dev, start, n), out=pg)
            for q0 in range(0, nq, 65535):
                qn = min(65535, nq - q0)
                self.score_prepared(plan, self._offset(pq, q0 * plan.query_item_bytes), qn, pg, n,
                                    self.dev.narrow0(scores, q0, qn), ng, start, accumulate_max)
        return scores

    def _offset(self, byte_buf, nbytes: int):
        return byte_buf if nbytes == 0 else self.dev.narrow0(byte_buf, nbytes, self.dev.shape(byte_buf)[0] - nbytes)

    def _torch_ops(self):
        """torch.ops.shoeprint_mi355x when this scorer runs the in-tree library on PyTorch-ROCm tensors and the op library
        is built (and not switched off with SPR_TORCH_OPS=0); None otherwise (emulation tests, explicit libraries)."""
        if getattr(self, "_ops_cache", False) is False:
            from . import _torch_ops

            ok = (self.dev.name == "hip" and self.lib is _lib.load_library() and _torch_ops.enabled())
            self._ops_cache = _torch_ops.load() if ok else None
        return self._ops_cache

    def ranks_device(self, scores, match_dev):
        nq, ng = self.dev.shape(scores)
        if nq > 0 and self._torch_ops() is not None and scores.is_contiguous() and match_dev.is_contiguous():
            return self._torch_ops().ranks(scores, match_dev)
        ranks = self.dev.zeros((max(nq, 1),), np.int32)
        self.lib.check(self.lib.spr_rank_true_match(self.dev.ptr(scores), ng, nq, ng, self.dev.ptr(match_dev),
                                                    self.dev.ptr(ranks), self.dev.stream()))
        return self.dev.narrow0(ranks, 0, nq)

    def ncc_maps_device(self, plan: _Plan, pq, pg):
        ih, iw = plan.g_hw[0] - 2 * plan.crop, plan.g_hw[1] - 2 * plan.crop
        out = self.dev.zeros((plan.channels, ih, iw), np.float32)
        self.lib.check(self.lib.spr_ncc_maps(plan.handle, self.dev.ptr(pq), self.dev.ptr(pg), self.dev.ptr(out),
                                             self.dev.stream()))
        return out

    # ------------------------------------------------------------------ list-of-arrays level
    def score_matrix(self, shoemark_maps, shoeprint_maps, accumulate_into=None, rotations=None,
                     scales=None) -> np.ndarray:
        """Host float32 [Q,G] matrix for the reference's list-of-arrays inputs, including ragged
        sets (items of different spatial size): items are grouped by shape and every
        (query shape, gallery shape) class is one plan.  With ``rotations`` / ``scales`` every query
        batch is expanded on the device into the reference's variant list (variants.py) and the
        matrix keeps the running maximum over variants (similarity.py:357-367)."""
        q_items, q_dev = _as_item_list(shoemark_maps)
        g_items, g_dev = _as_item_list(shoeprint_maps)
        if q_items is None and g_items is None and rotations is None and scales is None:
            scores = self.scores_device(q_dev, g_dev)
            return self.dev.to_host(scores)
        if q_items is None:
            q_items = list(self.dev.to_host(q_dev))
        if g_items is None:
            g_items = list(self.dev.to_host(g_dev))
        nq, ng = len(q_items), len(g_items)
        out = np.zeros((nq, ng), dtype=np.float32) if accumulate_into is None else accumulate_into
        if nq == 0 or ng == 0:
            return out
        q_groups = _group_by_shape(q_items)
        g_groups = _group_by_shape(g_items)
        from .variants import VariantBuilder

        if self._variants is None:
            self._variants = VariantBuilder(self.lib, self.dev)  # (keeps its resample tables on the device)
        builder = self._variants
        # Every query shape group is uploaded and expanded into its variant list ONCE; every gallery chunk is uploaded
        # ONCE and prepared once per plan (= per distinct variant shape: the 1/sigma map depends on the template size).
        q_side = {}
        for qshape, q_idx in q_groups.items():
            q_batch = self.dev.stack_to_device([q_items[i] for i in q_idx])
            by_shape: dict[tuple, list] = {}  # variants of one shape share a plan and the prepared gallery
            for v in builder.variants(q_batch, rotations, scales):
                by_shape.setdefault(tuple(self.dev.shape(v)[2:]), []).append(self.dev.astype_storage(v, self.storage))
            q_side[qshape] = by_shape
        for gshape, g_idx in g_groups.items():
            plans = {}
            for qshape, by_shape in q_side.items():
                if qshape[0] != gshape[0]:
                    raise ValueError(f"channel mismatch: query {qshape}, gallery {gshape}")
                for vs in by_shape:
                    plans[vs] = self.plan(qshape[0], vs, gshape[1:], dtype=self.storage)
            # every plan's prepared form of a chunk stays alive while the query groups are walked: they share the budget
            chunk = min(self.gallery_chunk_items(p, len(g_idx), share=len(plans)) for p in plans.values())
            self.last_chunk_items = chunk  # (read by the tests)
            for start in range(0, len(g_idx), chunk):
                idx = g_idx[start:start + chunk]
                g_batch = self.dev.astype_storage(self.dev.stack_to_device([g_items[i] for i in idx]), self.storage)
                prepared = {}  # plan -> prepared gallery chunk
                for qshape, q_idx in q_groups.items():
                    sub = self.dev.zeros((len(q_idx), len(idx)), np.float32)
                    for vs, vlist in q_side[qshape].items():
                        plan = plans[vs]
                        if vs not in prepared:
                            prepared[vs] = self.prepare_gallery(plan, g_batch)
                        for v in vlist:
                            pq = self.prepare_queries(plan, v)
                            self.score_prepared(plan, pq, len(q_idx), prepared[vs], len(idx), sub, len(idx), 0,
                                                accumulate_max=True)
                    sub_h = self.dev.to_host(sub)
                    block = out[np.ix_(q_idx, idx)]
                    out[np.ix_(q_idx, idx)] = np.maximum(block, sub_h)
        return out

    def multi_layer_scores_device(self, layers, out=None):
        """Device [Q,G] float32 mean over feature layers of the per-layer score matrices (e.g.
        conv3_3 + conv4_3 + conv5_3 maps of the same items; build-defined, the reference scores one layer).
        ``layers`` = [(q_dev [Q,C_l,h_l,w_l], g_dev [G,C_l,h_l,w_l]), ...] resident in HBM.  Nothing leaves the
        device: each layer's matrix is folded into the running mean by spr_scores_fuse."""
        layers = list(layers)
        if not layers:
            raise ValueError("no feature layers given")
        weight = 1.0 / len(layers)
        layer_scores = None
        for k, (q_dev, g_dev) in enumerate(layers):
            nq, ng = self.dev.shape(q_dev)[0], self.dev.shape(g_dev)[0]
            if out is None:
                out = self.dev.zeros((nq, ng), np.float32)
            if layer_scores is None:
                layer_scores = self.dev.zeros((nq, ng), np.float32)
            self.scores_device(q_dev, g_dev, scores=layer_scores)
            self.lib.check(self.lib.spr_scores_fuse(self.dev.ptr(out), self.dev.ptr(layer_scores), nq * ng,
                                                    0.0 if k == 0 else 1.0, weight, self.dev.stream()))
        return out

    def multi_layer_score_matrix(self, layers) -> np.ndarray:
        """Host copy of multi_layer_scores_device (one transfer, of the fused matrix)."""
        return self.dev.to_host(self.multi_layer_scores_device(layers))

    def ranks(self, scores_host: np.ndarray, matching_pairs: Sequence[int]) -> np.ndarray:
        nq, ng = scores_host.shape
        if nq == 0:
            return np.zeros(0, dtype=np.int32)
        s_dev = self.dev.to_device(np.ascontiguousarray(scores_host, dtype=np.float32))
        m_dev = self.dev.to_device(np.asarray(matching_pairs, dtype=np.int32))
        ranks = self.dev.to_host(self.ranks_device(s_dev, m_dev)).astype(np.int32, copy=True)
        if (ranks == 0).any():
            # the reference's np.where(...)[0][0] raises IndexError when the id is not in the gallery
            raise IndexError("index 0 is out of bounds for axis 0 with size 0")
        return ranks


def _group_by_shape(items) -> dict[tuple, list[int]]:
    groups: dict[tuple, list[int]] = {}
    for i, a in enumerate(items):
        if a.ndim != 3:
            raise ValueError("feature maps must be [C, h, w] (customtypes.py:11-14)")
        groups.setdefault(tuple(a.shape), []).append(i)
    return groups


# ---------------------------------------------------------------------------------------------
# The reference's call surface
# ---------------------------------------------------------------------------------------------
_default_scorer: NccScorer | None = None


def default_scorer() -> NccScorer:
    global _default_scorer
    if _default_scorer is None:
        _default_scorer = NccScorer()
    return _default_scorer


_config_scorers: dict[tuple, NccScorer] = {}


def scorer_from_config(config: dict, *, device=None, library: _lib.Library | None = None) -> NccScorer:
    """The scorer that ``[mi355x]`` of run.toml asks for: ``ncc_method`` ("auto" | "fft" | "fft_pow2" | "direct" | "mfma"),
    ``dtype`` (HBM storage type of the feature maps: "float32" | "float16" | "bfloat16") and ``max_prepared_gib`` (HBM
    budget of one prepared gallery chunk; 0 = automatic).  Reference files, which have no such table, get the defaults."""
    extra = config.get("mi355x") or {}
    method = extra.get("ncc_method", "auto") or "auto"
    storage = extra.get("dtype", "float32") or "float32"
    gib = float(extra.get("max_prepared_gib", 0.0) or 0.0)
    if method not in _METHODS:
        raise ValueError(f"[mi355x].ncc_method = {method!r}: expected one of {sorted(_METHODS)}")
    key = (method, storage, gib, id(device), id(library))
    if key == ("auto", "float32", 0.0, id(None), id(None)):
        return default_scorer()
    if key not in _config_scorers:
        _config_scorers[key] = NccScorer(device=device, library=library, method=method, storage=storage,
                                         max_prepared_bytes=int(gib * (1 << 30)) if gib > 0 else None)
    return _config_scorers[key]


def compare_maps(
    shoemark_maps: list[np.ndarray],
    shoeprint_maps: list[np.ndarray],
    matching_pairs: list[int],
    config: dict,
    *,
    scorer: NccScorer | None = None,
    progress: bool = False,
) -> np.ndarray:
    """Ranks (1-based, int32 [Q]) of every query's true match — reference similarity.py:129-227.

    ``matching_pairs[i]`` is the index into ``shoeprint_maps`` of query i's true match;
    ``config["comparison"]`` supplies ``n_processes`` (ignored: the GPU grid replaces the
    process pool), ``rotations`` and ``scales`` (query variants, see variants.py; with both set the
    reference builds 1 + (R+1)*S variant lists and — a latent defect, SURVEY §4 — then waits forever for
    (R+1)*(S+1) progress ticks; the variant lists are reproduced, the hang is not).
    """
    comp = config["comparison"]
    rotations, scales = comp.get("rotations"), comp.get("scales")
    scorer = scorer or scorer_from_config(config)
    scores = scorer.score_matrix(shoemark_maps, shoeprint_maps, rotations=rotations, scales=scales)
    ranks = scorer.ranks(scores, matching_pairs)
    if progress:
        for i, r in enumerate(ranks):
            print(f"Print {i} true match ranked {r}")  # similarity.py:375
    return ranks


def get_similarity(shoemark: np.ndarray, shoeprint: np.ndarray, *, scorer: NccScorer | None = None) -> np.floating[Any]:
    """max over positions of the channel-summed NCC maps, divided by the channel count
    (similarity.py:75-108); both stacks are cropped by 2 pixels per edge first."""
    scorer = scorer or default_scorer()
    mark = np.ascontiguousarray(shoemark, dtype=np.float32)
    prnt = np.ascontiguousarray(shoeprint, dtype=np.float32)
    plan = scorer.plan(mark.shape[0], mark.shape[1:], prnt.shape[1:])
    pq = scorer.prepare_queries(plan, scorer.dev.to_device(mark[None]))
    pg = scorer.prepare_gallery(plan, scorer.dev.to_device(prnt[None]))
    maps = scorer.dev.to_host(scorer.ncc_maps_device(plan, pq, pg)).astype(np.float64)
    return np.max(maps.sum(axis=0)) / mark.shape[0]


def normxcorr(template: np.ndarray, image: np.ndarray, mode: str = "same", *, scorer: NccScorer | None = None) -> np.ndarray:
    """Normalised cross-correlation map of ``template`` over ``image`` (similarity.py:26-72)."""
    if mode != "same":
        raise NotImplementedError("only mode='same' is on the hot path (similarity.py:104)")
    scorer = scorer or default_scorer()
    t = np.ascontiguousarray(template, dtype=np.float32)
    i = np.ascontiguousarray(image, dtype=np.float32)
    plan = scorer.plan(1, t.shape, i.shape, crop=0)
    pq = scorer.prepare_queries(plan, scorer.dev.to_device(t[None, None]))
    pg = scorer.prepare_gallery(plan, scorer.dev.to_device(i[None, None]))
    return scorer.dev.to_host(scorer.ncc_maps_device(plan, pq, pg))[0]
