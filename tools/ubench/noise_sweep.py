#!/usr/bin/env python3
"""How hard are the synthetic queries?  True-match rank statistics on conv3_3-sized maps for several (signal, noise)
mixing weights of the query generator (picks the bench's "hard" setting)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from shoeprint_image_retrieval_amd import synth
from shoeprint_image_retrieval_amd.similarity import NccScorer
C, H, W, NQ, NG = 256, 128, 64, 32, 512
sc = NccScorer(method="fft"); dev, lib = sc.dev, sc.lib
g = dev.empty((NG, C, H, W), np.float32); q = dev.empty((NQ, C, H, W), np.float32)
matches = synth.default_matches(NQ, NG); m = dev.to_device(matches)
lib.check(lib.spr_synth_gallery(dev.ptr(g), 0, NG, C, H, W, 1234, dev.stream()))
for sig, noi in ((3, 2), (1, 6), (1, 16), (1, 32), (1, 64), (1, 128), (1, 255)):
    lib.check(lib.spr_synth_queries(dev.ptr(q), 0, NQ, dev.ptr(m), C, H, W, 1234, 3, sig, noi, dev.stream()))
    s = sc.scores_device(q, g)
    r = dev.to_host(sc.ranks_device(s, m)); sh = dev.to_host(s)
    tm = sh[np.arange(NQ), matches]; other = np.sort(sh, axis=1)[:, -2]
    print(f"signal {sig} noise {noi}: rank1 {np.mean(r == 1):.2f} mean rank {r.mean():.1f} max {r.max()}  true-match score {tm.mean():.4f}  runner-up {other.mean():.4f}")
