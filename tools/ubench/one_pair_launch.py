#!/usr/bin/env python3
"""One scoring call (plus one warm-up) of the NCC pair kernel on synthetic conv3_3-shaped maps: the program
tools/ubench/prof_pair.sh runs under rocprofv3.  Kernel choices come from the environment (SPR_NCC_SIX ...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from shoeprint_image_retrieval_amd import synth
from shoeprint_image_retrieval_amd.similarity import NccScorer
C, H, W = int(os.environ.get("TP_C", 256)), int(os.environ.get("TP_H", 128)), int(os.environ.get("TP_W", 64))
NQ, NG = int(os.environ.get("TP_Q", 32)), int(os.environ.get("TP_G", 256))
sc = NccScorer(method=os.environ.get("TP_METHOD", "fft")); dev, lib = sc.dev, sc.lib
g = dev.empty((NG, C, H, W), np.float32); q = dev.empty((NQ, C, H, W), np.float32)
m = dev.to_device(synth.default_matches(NQ, NG))
lib.check(lib.spr_synth_gallery(dev.ptr(g), 0, NG, C, H, W, 1234, dev.stream()))
lib.check(lib.spr_synth_queries(dev.ptr(q), 0, NQ, dev.ptr(m), C, H, W, 1234, 3, 3, 2, dev.stream()))
plan = sc.plan(C, (H, W), (H, W))
pq = sc.prepare_queries(plan, q); pg = sc.prepare_gallery(plan, g)
scores = dev.zeros((NQ, NG), np.float32)
for _ in range(2):
    sc.score_prepared(plan, pq, NQ, pg, NG, scores, NG, 0)
torch.cuda.synchronize()
print("done", plan.fft_size, float(dev.to_host(scores).max()))
