#!/usr/bin/env python3
"""Time the VGG16 extractor (features[:block]) on a batch of synthetic 512x256 prints."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from shoeprint_image_retrieval_amd import network
N, H, W, BLOCK = int(os.environ.get("TE_N", 32)), 512, 256, int(os.environ.get("TE_BLOCK", 16))
DTYPE, MODEL = os.environ.get("TE_DTYPE", "float32"), os.environ.get("TE_MODEL", "VGG16")
cfg = {"model": {"type": MODEL, "clahe_clip_limit": 2.0, "clahe_tile_grid_size": [8, 8]}, "mi355x": {"extractor_dtype": DTYPE}}
m = network.Model(cfg, BLOCK)
imgs = torch.randint(0, 256, (N, H, W), dtype=torch.uint8, device="cuda")
for _ in range(2): out = m.extract_device(imgs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): out = m.extract_device(imgs)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
gflop = {("VGG16", 16): 48.77, ("VGG16", 23): 72.93, ("VGG16", 30): 80.18, ("ResNet50", 7): 17.13, ("EfficientNetV2_M", 6): 19.0}.get((MODEL, BLOCK), 0.0)
print(f"{MODEL} {DTYPE} block {BLOCK}: {N} images {ms:.2f} ms -> {N/ms*1e3:.1f} images/s, {gflop*N/ms:.2f} TFLOP/s (conv MACs x2), out {tuple(out.shape)}")
