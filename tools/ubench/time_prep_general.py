#!/usr/bin/env python3
"""Gallery preparation of the matrix-core general instance (33 x 16 templates on 32 x 16 maps, 1024 channels):
SPR_MFMA_PREP=0 (general kernel with tables) against the default (two channels per wave)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from shoeprint_image_retrieval_amd import _lib, synth
from shoeprint_image_retrieval_amd.similarity import NccScorer
C, NG = 1024, 4096
lib = _lib.load_library()
sc = NccScorer(method="mfma", library=lib); dev = sc.dev
g = dev.empty((NG, C, 32, 16), np.float32)
lib.check(lib.spr_synth_gallery(dev.ptr(g), 0, NG, C, 32, 16, 1234, dev.stream()))
g = g.to(torch.bfloat16)
plan = sc.plan(C, (33, 16), (32, 16), dtype="bfloat16")
times = []
for _ in range(6):  # (every call allocates its 22 GB output: the first ones wait for the allocator - the minimum is the kernels)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    pg = sc.prepare_gallery(plan, g)
    b.record(); torch.cuda.synchronize()
    times.append(a.elapsed_time(b))
    del pg
print(f"SPR_MFMA_PREP={os.environ.get('SPR_MFMA_PREP', '1')}: min {min(times):.2f} ms per {NG} items x {C} channels  (all: {[round(t, 1) for t in times]})")
