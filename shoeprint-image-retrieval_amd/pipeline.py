"""Images -> multi-layer NCC score matrix in one pass, with the stages on separate HIP streams (BASELINE.json config 5:
"multi-layer (conv3 + conv4 + conv5) fused NCC ... with overlapped extract/score HIP streams").

The reference runs ONE block per size cluster (run.py:20, dataloader.py:109) and would repeat the whole extractor for
another block; the multi-layer score is build-defined (SURVEY "Mismatches": mean of the per-layer ``get_similarity``
values, each of which is oracle-pinned).  Here

* the extractor runs once per image batch and hands out the activations of every requested layer (``Model.
  extract_taps_device`` -> ``spr_vgg16_forward_taps``);
* gallery batch i+1 is extracted on the extractor stream while the prepare + score chains of batch i run, one stream per
  feature layer (they are independent until the fusion);
* per-layer score blocks stay in HBM and are folded into the mean by ``spr_scores_fuse``; the only device-to-host copy is
  the final [Q] rank vector (or the fused matrix, if asked for).

Hand-offs are HIP events (no host synchronisation); at most two batches of tapped activations are alive at a time.
"""

from __future__ import annotations

import numpy as np

VGG16_TAPS = {"conv3_3": 16, "conv4_3": 23, "conv5_3": 30}  # slice ends into vgg16().features, each behind a ReLU


class MultiLayerPipeline:
    def __init__(self, model, scorer, taps=(16, 23, 30), batch_size: int = 64, clahe: bool = True):
        if model.block != max(taps):
            raise ValueError(f"the model must be truncated at the deepest tap ({max(taps)}), not {model.block}")
        self.model, self.scorer, self.taps = model, scorer, tuple(taps)
        self.batch_size, self.clahe = int(batch_size), clahe
        self.dev, self.lib = scorer.dev, scorer.lib

    def _features(self, images_dev):
        x = self.model.clahe_device(images_dev) if self.clahe else images_dev
        return self.model.extract_taps_device(x, self.taps)

    def scores_device(self, query_images_dev, gallery_images_dev):
        """uint8 device batches [Q,H,W], [G,H,W] -> device float32 [Q,G]: mean over the tapped layers of the NCC scores."""
        dev, sc = self.dev, self.scorer
        nq, ng = dev.shape(query_images_dev)[0], dev.shape(gallery_images_dev)[0]
        nl = len(self.taps)
        main = dev.current()
        ext = dev.new_stream()
        lay = [dev.new_stream() for _ in range(nl)]
        # ---- query side, on the caller's stream: one extractor pass, one prepared set per layer
        q_taps = self._features(query_images_dev)
        plans, pqs = [], []
        bs = min(self.batch_size, max(1, ng))
        for qt in q_taps:
            _, c, h, w = dev.shape(qt)
            plan = sc.plan(c, (h, w), (h, w))
            plans.append(plan)
            pqs.append(sc.prepare_queries(plan, qt))
        pgs = [dev.empty_bytes(max(1, p.gallery_item_bytes * bs)) for p in plans]
        layer_scores = [dev.zeros((nq, max(ng, 1)), np.float32) for _ in range(nl)]
        fused = dev.zeros((nq, max(ng, 1)), np.float32)
        if ng == 0 or nq == 0:
            return dev.narrow1(fused, 0, ng) if hasattr(dev, "narrow1") else fused[:, :ng]
        if ext is not None:
            dev.wait_stream(ext, main)
            for s in lay:
                dev.wait_stream(s, main)
        # ---- gallery side: extract batch i on `ext` while the layer streams prepare + score batch i-1
        starts = list(range(0, ng, bs))
        taps_of, ev_extract, ev_scored = {}, {}, {}
        for i in range(len(starts) + 1):
            if i < len(starts):
                with dev.use(ext):
                    if i >= 2:  # back-pressure: the taps of batch i-2 are scored before batch i is extracted
                        for ev in ev_scored[i - 2]:
                            dev.wait_event(ext, ev)
                    n = min(bs, ng - starts[i])
                    taps_of[i] = self._features(dev.narrow0(gallery_images_dev, starts[i], n))
                    ev_extract[i] = dev.record_event(ext)
            if i >= 1:
                j = i - 1
                n = min(bs, ng - starts[j])
                ev_scored[j] = []
                for k in range(nl):
                    with dev.use(lay[k]):
                        dev.wait_event(lay[k], ev_extract[j])
                        dev.record_stream(taps_of[j][k], lay[k])
                        sc.prepare_gallery(plans[k], taps_of[j][k], out=pgs[k])
                        sc.score_prepared(plans[k], pqs[k], nq, pgs[k], n, layer_scores[k], ng, starts[j])
                        ev_scored[j].append(dev.record_event(lay[k]))
                del taps_of[j]
        # ---- fusion on the caller's stream, behind every layer stream
        if ext is not None:
            for s in lay:
                dev.wait_stream(main, s)
            dev.wait_stream(main, ext)
        w = 1.0 / nl
        for k, ls in enumerate(layer_scores):
            self.lib.check(self.lib.spr_scores_fuse(dev.ptr(fused), dev.ptr(ls), nq * ng, 0.0 if k == 0 else 1.0, w,
                                                    dev.stream()))
        return fused

    def ranks(self, query_images_dev, gallery_images_dev, matching_pairs) -> np.ndarray:
        """int32 [Q] ranks of the true matches (1-based) from the fused scores; the only device-to-host copy."""
        dev, sc = self.dev, self.scorer
        scores = self.scores_device(query_images_dev, gallery_images_dev)
        m_dev = dev.to_device(np.asarray(matching_pairs, dtype=np.int32))
        return dev.to_host(sc.ranks_device(scores, m_dev)).astype(np.int32)
