#!/usr/bin/env python3
"""Time the NCC prep kernels alone (gallery + query) for one or more library builds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from shoeprint_image_retrieval_amd import _lib, synth
from shoeprint_image_retrieval_amd.similarity import NccScorer
C, H, W, NG = 256, 128, 64, 512
for a in (sys.argv[1:] or [None]):
    sc = NccScorer(method="fft", library=_lib.load_library(a)); dev, lib = sc.dev, sc.lib
    g = dev.empty((NG, C, H, W), np.float32)
    lib.check(lib.spr_synth_gallery(dev.ptr(g), 0, NG, C, H, W, 1, dev.stream()))
    plan = sc.plan(C, (H, W), (H, W)); pg = dev.empty_bytes(plan.gallery_item_bytes * NG); pq = dev.empty_bytes(plan.query_item_bytes * NG)
    res = {}
    for name, fn in (("gallery", lambda: sc.prepare_gallery(plan, g, out=pg)),
                     ("query", lambda: lib.check(lib.spr_ncc_prepare_queries(plan.handle, dev.ptr(g), NG, dev.ptr(pq), dev.stream())))):
        ts = []
        for r in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            if r: ts.append(e0.elapsed_time(e1))
        res[name] = min(ts)
    print(f"{os.path.basename(a or 'in-tree'):24s} prep of {NG} items: gallery {res['gallery']:7.2f} ms ({res['gallery']/NG*1e3:6.1f} us/item)  query {res['query']:7.2f} ms")
