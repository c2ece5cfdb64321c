#!/usr/bin/env python3
"""Images/s of the EfficientNetV2_M extractor (features[:block], the reference's run.toml default model) at 512x256, with
the flops of the real (unpadded) layers and of the layers as issued (channel counts padded to 64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from shoeprint_image_retrieval_amd import network
B = int(os.environ.get("TR_B", 32))
for block in (4, 6):
    m = network.Model({"model": {"type": "EfficientNetV2_M", "clahe_clip_limit": 2.0, "clahe_tile_grid_size": [8, 8]}}, block)
    h, w, real, issued = 512, 256, 0.0, 0.0
    for op in m.effnet_ops():
        if op["kind"] == 2:
            continue
        if op["stride"] == 2:
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        if op["kind"] == 1:
            real += 2 * 9 * op["cin"] * h * w; issued += 2 * 9 * op["cin_p"] * h * w
        else:
            real += 2 * op["ks"] ** 2 * op["cin"] * op["cout"] * h * w
            issued += 2 * op["ks"] ** 2 * op["cin_p"] * op["cout_p"] * h * w
    imgs = torch.randint(0, 256, (B, 512, 256), dtype=torch.uint8, device="cuda")
    m.extract_device(imgs); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): out = m.extract_device(imgs)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"EfficientNetV2_M features[:{block}], batch {B}: {ms:.2f} ms -> {B / ms * 1e3:.1f} images/s; {real / 1e9:.2f} GFLOP per image "
          f"({issued / 1e9:.2f} issued with padded channels) = {real * B / ms / 1e9:.1f} TFLOP/s real, {issued * B / ms / 1e9:.1f} issued "
          f"({issued * B / ms / 1e9 / 157.3:.1%} of the fp32 MFMA peak), {len(m.effnet_ops())} layers, out {tuple(out.shape)}")
    m.close()

# DenseNet_201, all of `features` (block 12)
m = network.Model({"model": {"type": "DenseNet_201", "clahe_clip_limit": 2.0, "clahe_tile_grid_size": [8, 8]}}, 12)
h, w = 128, 64  # behind conv0 (256 x 128) and pool0
flops = 2 * 147 * 64 * 256 * 128
issued = flops
for op in m.densenet_ops():
    if op["kind"] == 1:
        flops += 2 * op["cin"] * 128 * h * w; issued += 2 * op["cin"] * 128 * h * w
    elif op["kind"] == 2:
        flops += 2 * 9 * 128 * 32 * h * w; issued += 2 * 9 * 128 * 64 * h * w
    elif op["kind"] == 3:
        flops += 2 * op["cin"] * op["cout"] * h * w; issued += 2 * op["cin"] * op["cout"] * h * w
        h, w = h // 2, w // 2
imgs = torch.randint(0, 256, (B, 512, 256), dtype=torch.uint8, device="cuda")
m.extract_device(imgs); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): out = m.extract_device(imgs)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print(f"DenseNet_201 features[:12], batch {B}: {ms:.2f} ms -> {B / ms * 1e3:.1f} images/s; {flops / 1e9:.2f} GFLOP per image "
      f"({issued / 1e9:.2f} issued) = {flops * B / ms / 1e9:.1f} TFLOP/s real, {issued * B / ms / 1e9:.1f} issued "
      f"({issued * B / ms / 1e9 / 157.3:.1%} of the fp32 MFMA peak), {len(m.densenet_ops())} layers, out {tuple(out.shape)}")
