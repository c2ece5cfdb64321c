#!/usr/bin/env python3
"""Condense rocprofv3 output directories into profiles/<tag>_rocprof_summary.json + <tag>_kernel_stats.csv.

    python tools/summarise_profile.py r01 gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write [pairs_per_launch]

stats dir : rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-parity-sample
fetch/write dirs : rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (own passes) -- python3 bench.py --steps 1 --warmup 0 ...
"""
import csv
import glob
import json
import os
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::|spr::|void ", "", name)
    return re.split(r"[<(]", name)[0].strip()


def _db_rows(d, query):
    import sqlite3
    for f in glob.glob(os.path.join(d, "**", "*results.db"), recursive=True):
        yield from sqlite3.connect(f).execute(query)


def kernel_stats(d):
    rows = {}
    for name, calls, total_ns, avg_ns in _db_rows(d, "select name, total_calls, total_duration*1000, average*1000 from top_kernels"):
        k = short(name)  # the rocpd view reports microseconds
        e = rows.setdefault(k, {"kernel": k, "calls": 0, "total_ms": 0.0, "variants": []})
        e["calls"] += calls; e["total_ms"] += total_ns / 1e6
        e["variants"].append({"name": name[:160], "calls": calls, "avg_ms": avg_ns / 1e6})
    for f in ([] if rows else glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            e = rows.setdefault(k, {"kernel": k, "calls": 0, "total_ms": 0.0, "variants": []})
            e["calls"] += int(r["Calls"]); e["total_ms"] += float(r["TotalDurationNs"]) / 1e6
            e["variants"].append({"name": r["Name"][:160], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6})
    total = sum(e["total_ms"] for e in rows.values()) or 1.0
    out = sorted(rows.values(), key=lambda e: -e["total_ms"])
    for e in out:
        e["avg_ms"] = round(e["total_ms"] / e["calls"], 4); e["pct"] = round(100 * e["total_ms"] / total, 4)
        e["total_ms"] = round(e["total_ms"], 3)
    return out


def counter_sums(d, counter):
    sums = {}
    for name, disp, value in _db_rows(d, f"select kernel_name, dispatch_id, value from counters_collection where counter_name = '{counter}'"):
        e = sums.setdefault(short(name), {"sum": 0.0, "dispatches": set()})
        e["sum"] += float(value); e["dispatches"].add(disp)
    for f in ([] if sums else glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            e = sums.setdefault(short(r["Kernel_Name"]), {"sum": 0.0, "dispatches": set()})
            e["sum"] += float(r["Counter_Value"]); e["dispatches"].add(r["Dispatch_Id"])
    return {k: {"sum_kib": v["sum"], "dispatches": len(v["dispatches"])} for k, v in sums.items()}


def main():
    tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
    pairs = float(sys.argv[5]) if len(sys.argv) > 5 else 150000.0
    stats = kernel_stats(stats_dir)
    fetch, write = counter_sums(fetch_dir, "FETCH_SIZE"), counter_sums(write_dir, "WRITE_SIZE")
    summary = {"kernel_stats (rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-parity-sample)":
               [{k: e[k] for k in ("kernel", "calls", "avg_ms", "total_ms", "pct")} for e in stats],
               "FETCH_SIZE (KiB, summed over dispatches; own rocprofv3 --pmc pass, bench.py --steps 1 --warmup 0)": fetch,
               "WRITE_SIZE (KiB, summed over dispatches; own rocprofv3 --pmc pass, bench.py --steps 1 --warmup 0)": write}
    pair = next((k for k in ("pair6_kernel", "pair_fft_kernel") if k in fetch and k in write), None)
    if pair:
        import hashlib
        f = fetch[pair]; w = write[pair]
        fb = f["sum_kib"] * 1024 / f["dispatches"]; wb = w["sum_kib"] * 1024 / w["dispatches"]
        lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shoeprint-image-retrieval_amd",
                           "libshoeprint_mi355x.so")
        summary["pair kernel traffic per launch"] = {
            "kernel": pair, "lib_sha16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16],
            "FETCH_SIZE_bytes_raw": fb, "FETCH_SIZE_bytes_x2_gfx950_correction": 2 * fb, "WRITE_SIZE_bytes": wb,
            "hbm_bytes_per_launch": 2 * fb + wb, "hbm_bytes_per_pair": (2 * fb + wb) / pairs, "pairs_per_launch": pairs,
            "note": "FETCH_SIZE on gfx950 reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section): "
                    "doubled. Counters are memory-side L2 requests and include Infinity-Cache hits.  bench.py reports this figure "
                    "as roofline.traffic only while lib_sha16 matches the library it runs (the counters cannot be read in-run)."}
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    json.dump(summary, open(os.path.join(root, f"{tag}_rocprof_summary.json"), "w"), indent=1)
    with open(os.path.join(root, f"{tag}_kernel_stats.csv"), "w", newline="") as fh:
        wr = csv.writer(fh); wr.writerow(["kernel_variant", "calls", "avg_ms"])
        for e in stats:
            for v in e["variants"]:
                wr.writerow([v["name"], v["calls"], round(v["avg_ms"], 4)])
    print(json.dumps(summary.get("pair kernel traffic per launch", {}), indent=1))
    for e in stats[:6]:
        print(e["kernel"], e["calls"], e["avg_ms"], e["pct"])


if __name__ == "__main__":
    main()
