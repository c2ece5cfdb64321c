# per-kernel time of the config-3 bench step (rocprofv3 kernel trace).  usage (GPU box): bash tools/ubench/stats_config3.sh
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/c3_stats
rm -rf $O; mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o x -- python3 $R/bench.py --config 3 --no-cpu-baseline --no-extractor --steps 2 --warmup 1 > $O/run.log 2>&1 || { echo failed; tail -5 $O/run.log; exit 1; }
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
import json, re
out = []
for r in rows[:8]:
    print(r["Name"][:70], r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 2), "ms total", round(float(r["AverageNs"]) / 1e6, 3), "ms avg")
    out.append({"kernel": re.split(r"[<(]", re.sub(r"\(anonymous namespace\)::|spr::|void ", "", r["Name"]))[0].strip(),
                "calls": int(r["Calls"]), "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 3),
                "avg_ms": round(float(r["AverageNs"]) / 1e6, 4)})
json.dump({"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --config 3 --no-cpu-baseline --no-extractor --steps 2 --warmup 1 "
                      "(three steps in the trace: one warm-up + two timed)", "kernels": out},
          open("gpurun_out/r03_config3_kernel_stats.json", "w"), indent=1)
PY
