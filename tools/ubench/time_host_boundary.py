#!/usr/bin/env python3
"""compare_maps from HOST lists (the reference's call surface): pairs/s including the H2D copies of the
feature maps, next to the same work with the maps already resident in HBM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from shoeprint_image_retrieval_amd import synth
from shoeprint_image_retrieval_amd.similarity import NccScorer, compare_maps
C, H, W, NQ, NG = 256, 128, 64, 50, 500
sc = NccScorer(method="fft"); dev, lib = sc.dev, sc.lib
g = dev.empty((NG, C, H, W), np.float32); q = dev.empty((NQ, C, H, W), np.float32)
m = synth.default_matches(NQ, NG); md = dev.to_device(m)
lib.check(lib.spr_synth_gallery(dev.ptr(g), 0, NG, C, H, W, 1234, dev.stream()))
lib.check(lib.spr_synth_queries(dev.ptr(q), 0, NQ, dev.ptr(md), C, H, W, 1234, 3, 3, 2, dev.stream()))
gh, qh = list(dev.to_host(g)), list(dev.to_host(q))
cfg = {"comparison": {"n_processes": 1, "rotations": None, "scales": None}}
compare_maps(qh[:2], gh[:4], [0, 1], cfg, scorer=sc)  # warm-up (plans, kernels)
t0 = time.perf_counter(); r_host = compare_maps(qh, gh, [int(v) for v in m], cfg, scorer=sc); t_host = time.perf_counter() - t0
torch.cuda.synchronize(); t0 = time.perf_counter()
s = sc.scores_device(q, g); r_dev = dev.to_host(sc.ranks_device(s, md)); t_dev = time.perf_counter() - t0
assert (np.asarray(r_host) == r_dev).all()
print(f"Q={NQ} x G={NG} conv3_3 maps ({(NQ + NG) * C * H * W * 4 / 1e9:.1f} GB of features): host lists {NQ * NG / t_host:9.0f} pairs/s "
      f"({t_host * 1e3:.0f} ms), HBM-resident {NQ * NG / t_dev:9.0f} pairs/s ({t_dev * 1e3:.0f} ms)")
