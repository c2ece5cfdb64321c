#!/usr/bin/env python3
"""Where one channel's cycles go in the six-wave pair kernel: a -DSPR_STAMPS build of the library (first argument)
records the shader clock at phase boundaries in one workgroup (diagnostic build: ~10 % slower, never shipped)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from shoeprint_image_retrieval_amd import _lib, synth
from shoeprint_image_retrieval_amd.similarity import NccScorer
lib = _lib.load_library(sys.argv[1])
sc = NccScorer(method="fft", library=lib); dev = sc.dev
C, H, W, NQ, NG = 256, 128, 64, 64, 512
g = dev.empty((NG, C, H, W), np.float32); q = dev.empty((NQ, C, H, W), np.float32)
m = dev.to_device(synth.default_matches(NQ, NG))
lib.check(lib.spr_synth_gallery(dev.ptr(g), 0, NG, C, H, W, 1234, dev.stream()))
lib.check(lib.spr_synth_queries(dev.ptr(q), 0, NQ, dev.ptr(m), C, H, W, 1234, 3, 3, 2, dev.stream()))
plan = sc.plan(C, (H, W), (H, W))
pq = sc.prepare_queries(plan, q); pg = sc.prepare_gallery(plan, g)
scores = dev.zeros((NQ, NG), np.float32)
for _ in range(3):
    sc.score_prepared(plan, pq, NQ, pg, NG, scores, NG, 0)
torch.cuda.synchronize()
P, CH = 12, 8
buf = (ctypes.c_ulonglong * (12 * CH * P))()
fn = lib.cdll.spr_debug_read_stamps; fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, len(buf)) == 0
st = np.array(buf, dtype=np.int64).reshape(12, CH, P)
d = np.diff(st, axis=2)  # [wave, channel, phase]
names = ["A: product + 12-point stage", "B2 wait", "product B, exchange A, 12-point stage B", "16-point stage + stores A",
         "exchange B", "16-point stage + stores B", "B1 wait", "row: image reads + pre-twist", "row: 16-point stage + exchange writes",
         "row: exchange reads, 3-point stage, accumulate"]
print("cycles per channel (mean over 8 channels), per wave; total =", (st[:, 1:, 0] - st[:, :-1, 0]).mean())
for i, n in enumerate(names):
    print(f"{n:28s} mean {d[:, :, i].mean():8.0f}   per wave: " + " ".join(f"{v:6.0f}" for v in d[:, :, i].mean(axis=1)))
