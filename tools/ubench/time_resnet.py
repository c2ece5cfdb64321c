#!/usr/bin/env python3
"""Images/s of the build-defined ResNet50-layer3 extractor at 512x256 (17.13 GFLOP per image)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from shoeprint_image_retrieval_amd import network
B = int(os.environ.get("TR_B", 32))
from shoeprint_image_retrieval_amd import _lib
lib = _lib.load_library(os.environ["SPR_LIB"]) if os.environ.get("SPR_LIB") else None  # A/B runs: another build of the library
m = network.Model({"model": {"type": "ResNet50", "clahe_clip_limit": 2.0, "clahe_tile_grid_size": [8, 8]}}, 7, library=lib)
imgs = torch.randint(0, 256, (B, 512, 256), dtype=torch.uint8, device="cuda")
m.extract_device(imgs); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): out = m.extract_device(imgs)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"ResNet50 layer3, batch {B}: {ms:.2f} ms -> {B / ms * 1e3:.1f} images/s = {17.13 * B / ms:.1f} TFLOP/s ({17.13 * B / ms / 157.3:.1%} of the fp32 MFMA peak), out {tuple(out.shape)}")
