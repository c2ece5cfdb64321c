#!/usr/bin/env python3
"""Pair-kernel throughput for the VGG16 layer shapes of a 512x256 print: grid family A/B in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from shoeprint_image_retrieval_amd import synth
from shoeprint_image_retrieval_amd.similarity import NccScorer
NQ, NG = 32, 512
for name, (C, H, W) in {"conv3_3": (256, 128, 64), "conv4_3": (512, 64, 32), "conv5_3": (512, 32, 16)}.items():
    for method in ("fft", "fft_pow2"):
        sc = NccScorer(method=method); dev, lib = sc.dev, sc.lib
        g = dev.empty((NG, C, H, W), np.float32); q = dev.empty((NQ, C, H, W), np.float32)
        m = dev.to_device(synth.default_matches(NQ, NG))
        lib.check(lib.spr_synth_gallery(dev.ptr(g), 0, NG, C, H, W, 1, dev.stream()))
        lib.check(lib.spr_synth_queries(dev.ptr(q), 0, NQ, dev.ptr(m), C, H, W, 1, 3, 3, 2, dev.stream()))
        plan = sc.plan(C, (H, W), (H, W)); pq = sc.prepare_queries(plan, q); pg = sc.prepare_gallery(plan, g)
        scores = dev.zeros((NQ, NG), np.float32); ts = []
        for r in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); sc.score_prepared(plan, pq, NQ, pg, NG, scores, NG, 0); e1.record(); torch.cuda.synchronize()
            if r: ts.append(e0.elapsed_time(e1))
        print(f"{name} [{C},{H},{W}] {method:9s} grid {plan.fft_size}: {min(ts):7.2f} ms -> {NQ*NG/min(ts)*1e3:10.0f} pairs/s")
        del g, q, pq, pg
