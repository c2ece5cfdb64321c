#!/usr/bin/env python3
"""Config 5 with the extractor in the loop, for a rocprofv3 --kernel-trace run: Q queries x G gallery IMAGES (512x256) ->
VGG16 features[:30] with conv3_3 / conv4_3 / conv5_3 taps -> three NCC scoring chains on their own streams -> fused ranks.
    rocprofv3 --kernel-trace --output-format csv -d DIR -o x -- python3 tools/ubench/pipeline_trace.py
    python3 tools/ubench/pipeline_trace.py --summarise DIR OUT.json    (overlap of the extractor with the scoring kernels)"""
import csv, glob, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def run():
    import numpy as np, torch
    from shoeprint_image_retrieval_amd import network, pipeline
    from shoeprint_image_retrieval_amd.similarity import NccScorer
    Q, G, B = int(os.environ.get("PT_Q", 32)), int(os.environ.get("PT_G", 256)), int(os.environ.get("PT_B", 64))
    model = network.Model({"model": {"type": "VGG16", "clahe_clip_limit": 2.0, "clahe_tile_grid_size": [8, 8]}}, 30)
    sc = NccScorer(method="fft")
    gen = torch.Generator(device="cuda").manual_seed(7)
    gal = torch.randint(0, 256, (G, 512, 256), dtype=torch.uint8, device="cuda", generator=gen)
    qry = gal[:Q].clone()
    pipe = pipeline.MultiLayerPipeline(model, sc, taps=(16, 23, 30), batch_size=B)
    pipe.ranks(qry[:4], gal[:8], list(range(4)))  # warm-up (plans, kernels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ranks = pipe.ranks(qry, gal, list(range(Q)))
    dt = time.perf_counter() - t0
    print(json.dumps({"queries": Q, "gallery_images": G, "batch": B, "seconds": round(dt, 3), "images_per_s": round((Q + G) / dt, 1),
                      "pairs_per_s_x3_layers": round(Q * G / dt, 1), "rank1": float(np.mean(ranks == 1))}))


def summarise(d, out):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    ev = []
    for r in rows:
        name = r["Kernel_Name"]
        kind = "extract" if ("conv_" in name or "clahe" in name) else ("score" if ("pair" in name or "prep_fft" in name) else "other")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, r.get("Queue_Id", r.get("Stream_Id", "?"))))
    ev.sort()
    t_last = max(e[1] for e in ev)
    # only the big run (after the warm-up): from the first extractor kernel of the last 80 % of the trace
    cut = ev[0][0] + 0  # keep all; the warm-up is tiny
    def union(kind):
        iv = sorted((a, b) for a, b, k, _ in ev if k == kind and a >= cut)
        out_iv, cur = [], None
        for a, b in iv:
            if cur and a <= cur[1]: cur[1] = max(cur[1], b)
            else:
                if cur: out_iv.append(tuple(cur))
                cur = [a, b]
        if cur: out_iv.append(tuple(cur))
        return out_iv
    ex, scv = union("extract"), union("score")
    def total(iv): return sum(b - a for a, b in iv)
    def inter(x, y):
        i = j = 0; t = 0
        while i < len(x) and j < len(y):
            a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
            if a < b: t += b - a
            if x[i][1] < y[j][1]: i += 1
            else: j += 1
        return t
    both = inter(ex, scv)
    res = {"command": "rocprofv3 --kernel-trace -- python3 tools/ubench/pipeline_trace.py",
           "kernels": len(ev), "queues": sorted({e[3] for e in ev}),
           "extractor_busy_ms": round(total(ex) / 1e6, 2), "scoring_busy_ms": round(total(scv) / 1e6, 2),
           "both_busy_ms": round(both / 1e6, 2), "span_ms": round((t_last - ev[0][0]) / 1e6, 2),
           "extractor_time_overlapped_by_scoring": round(both / max(1, total(ex)), 3),
           "note": "busy = union of kernel intervals of that stage; both = time with an extractor kernel AND a scoring kernel in flight"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2], sys.argv[3])
    else:
        run()
