#!/usr/bin/env python3
"""Time the NCC pair kernel alone for one or more builds of the library (A/B in one process, interleaved
rounds — cdna guide rule 24).  usage: time_pair.py [lib.so ...]   (default: the in-tree build)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from shoeprint_image_retrieval_amd import _lib, synth
from shoeprint_image_retrieval_amd.similarity import NccScorer
C, H, W = int(os.environ.get("TP_C", 256)), int(os.environ.get("TP_H", 128)), int(os.environ.get("TP_W", 64))
NQ, NG = int(os.environ.get("TP_Q", 32)), int(os.environ.get("TP_G", 512))
# arguments: library paths (*.so) and/or method names (fft, fft_pow2, direct) for the in-tree build
args = sys.argv[1:] or ["fft"]
paths = [a for a in args]
# "method@VAR=value[,VAR=value]" sets environment variables while that scorer's plan is created (kernel choices
# such as SPR_NCC_SIX are read at plan creation)
def _scorer(a):
    return NccScorer(method="fft", library=_lib.load_library(a)) if a.endswith(".so") else \
        NccScorer(method=a.split("@")[0], library=_lib.load_library(None))
def _env(a):
    return dict(kv.split("=") for kv in a.split("@")[1].split(",")) if "@" in a else {}
scorers = [_scorer(a) for a in args]
sc0 = scorers[0]; dev = sc0.dev; lib = sc0.lib
g = dev.empty((NG, C, H, W), np.float32); q = dev.empty((NQ, C, H, W), np.float32)
m = dev.to_device(synth.default_matches(NQ, NG))
lib.check(lib.spr_synth_gallery(dev.ptr(g), 0, NG, C, H, W, 1234, dev.stream()))
lib.check(lib.spr_synth_queries(dev.ptr(q), 0, NQ, dev.ptr(m), C, H, W, 1234, 3, 3, 2, dev.stream()))
state = []
for sc, a in zip(scorers, args):
    old = {k: os.environ.get(k) for k in _env(a)}
    os.environ.update(_env(a))
    plan = sc.plan(C, (H, W), (H, W))
    for k, v in old.items():
        os.environ.pop(k) if v is None else os.environ.__setitem__(k, v)
    pq = sc.prepare_queries(plan, q); pg = sc.prepare_gallery(plan, g)
    scores = dev.zeros((NQ, NG), np.float32)
    state.append((sc, plan, pq, pg, scores))
res = {i: [] for i in range(len(scorers))}
for rnd in range(4):
    for i, (sc, plan, pq, pg, scores) in enumerate(state):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        old = {k: os.environ.get(k) for k in _env(args[i])}
        os.environ.update(_env(args[i]))
        e0.record(); sc.score_prepared(plan, pq, NQ, pg, NG, scores, NG, 0); e1.record(); torch.cuda.synchronize()
        for k, v in old.items():
            os.environ.pop(k) if v is None else os.environ.__setitem__(k, v)
        if rnd: res[i].append(e0.elapsed_time(e1))
for sc, plan, *_ in state:
    print("plan", plan.fft_size, "gallery item MB %.1f" % (plan.gallery_item_bytes / 1e6))
ref = dev.to_host(state[0][4])
for i, p in enumerate(paths):
    ms = np.array(res[i]); d = np.abs(dev.to_host(state[i][4]) - ref).max()
    print(f"{os.path.basename(p or 'in-tree'):28s} pair kernel min {ms.min():8.2f} ms  med {np.median(ms):8.2f} ms  -> {NQ*NG/ms.min()*1e3:9.0f} pairs/s   max|d vs first| {d:.2e}")
