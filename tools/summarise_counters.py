#!/usr/bin/env python3
"""Per-kernel averages of the counters collected by tools/profile_counters.sh -> profiles/<tag>_sq_counters.json"""
import collections, csv, glob, json, os, re, sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|spr::|void ", "", name)
    return re.split(r"[<(]", name)[0].strip()


def main():
    tag, d = sys.argv[1], sys.argv[2]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(lambda: collections.defaultdict(set))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    out = {}
    for k in acc:
        if not any(s in k for s in ("pair6", "pair_fft", "pair_mfma", "prep_fft", "prep_mfma", "conv_mfma", "conv_first")):
            continue
        c = {name: acc[k][name] / len(n[k][name]) for name in acc[k]}
        derived = {}
        if "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"]:
            # SQ_BUSY_CYCLES counts per shader engine (32 of them x ... ) - ratios against wave cycles are the portable part
            pass
        if c.get("SQ_WAVE_CYCLES"):
            w = c["SQ_WAVE_CYCLES"]
            for name in ("SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
                if name in c:
                    derived[name + "/SQ_WAVE_CYCLES"] = round(c[name] / w, 4)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            derived["lds_bank_conflict_rate"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
        if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
            derived["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        out[k] = {"per_dispatch": {a: c[a] for a in sorted(c)}, "derived": derived}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_sq_counters.json")
    json.dump({"command": "rocprofv3 --pmc <group> -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-sample "
                          "(one pass per counter group, tools/profile_counters.sh)", "kernels": out}, open(path, "w"), indent=1)
    for k, v in out.items():
        print(k, v["derived"])


if __name__ == "__main__":
    main()
