#!/usr/bin/env python3
"""Where the cycles of one gallery prep workgroup (one channel of one item, six-wave layout) go: a -DSPR_PREP_STAMPS build
of the library (first argument) records the shader clock at its phase boundaries; also times the whole launch."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from shoeprint_image_retrieval_amd import _lib
from shoeprint_image_retrieval_amd.similarity import NccScorer
lib = _lib.load_library(sys.argv[1])
sc = NccScorer(method="fft", library=lib); dev = sc.dev
C, H, W, NG = 256, 128, 64, 1500
g = dev.empty((NG, C, H, W), np.float32)
lib.check(lib.spr_synth_gallery(dev.ptr(g), 0, NG, C, H, W, 1234, dev.stream()))
plan = sc.plan(C, (H, W), (H, W))
pg = dev.empty_bytes(plan.gallery_item_bytes * NG)
for _ in range(2):
    sc.prepare_gallery(plan, g, out=pg)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(3):
    sc.prepare_gallery(plan, g, out=pg)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 3
print(f"gallery prep of {NG} items: {ms:.2f} ms = {NG * C / ms / 1e3:.1f} k channel-workgroups/ms, "
      f"{(plan.gallery_item_bytes + C * H * W * 4) * NG / ms / 1e6:.0f} GB/s of compulsory traffic")
try:
    fn = lib.cdll.spr_debug_read_prep_stamps
except AttributeError:
    sys.exit(0)
buf = (ctypes.c_ulonglong * 16)()
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, 16) == 0
st = np.array(buf, dtype=np.int64)
names = ["load + centre", "template energy / dead flag", "1/sigma map (float64 tables)", "row transforms", "column transforms + stores"]
for i, n in enumerate(names):
    print(f"{n:32s} {st[i + 1] - st[i]:8d} cycles")
print(f"{'total':32s} {st[5] - st[0]:8d} cycles")
print(f"  1/sigma: row scans {st[7] - st[2]}, column scans {st[8] - st[7]}, window sums + stores {st[3] - st[8]}")
