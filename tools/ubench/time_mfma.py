#!/usr/bin/env python3
"""Kernel-only time of the matrix-core pair kernel at the config-3 shape: python time_mfma.py [library ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from shoeprint_image_retrieval_amd import _lib, synth
from shoeprint_image_retrieval_amd.similarity import NccScorer
C, H, W, NQ, NG = 1024, 32, 16, 64, 5120
for path in (sys.argv[1:] or [None]):
    lib = _lib.load_library(path) if path else _lib.load_library()
    sc = NccScorer(method="mfma", library=lib); dev = sc.dev
    g = dev.empty((NG, C, H, W), np.float32); q = dev.empty((NQ, C, H, W), np.float32)
    m = dev.to_device(synth.default_matches(NQ, NG))
    lib.check(lib.spr_synth_gallery(dev.ptr(g), 0, NG, C, H, W, 1234, dev.stream()))
    lib.check(lib.spr_synth_queries(dev.ptr(q), 0, NQ, dev.ptr(m), C, H, W, 1234, 3, 1, 60, dev.stream()))
    g = g.to(torch.bfloat16); q = q.to(torch.bfloat16)
    plan = sc.plan(C, (H, W), (H, W), dtype="bfloat16")
    pq = sc.prepare_queries(plan, q); pg = sc.prepare_gallery(plan, g)
    scores = dev.zeros((NQ, NG), np.float32)
    sc.score_prepared(plan, pq, NQ, pg, NG, scores, NG, 0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        sc.score_prepared(plan, pq, NQ, pg, NG, scores, NG, 0)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 3
    per = ms * 1e-3 / 20 / (C + 1) * 2.4e9  # 20 workgroups per CU in turn, channels + 1 periods each
    print(f"{os.path.basename(path) if path else 'shipped':24s} {ms:8.2f} ms  {NQ * NG / ms / 1e3:7.2f} M pairs/s  ~{per:6.0f} cycles per period (9408 of MFMA)")
