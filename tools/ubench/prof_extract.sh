# rocprofv3 kernel trace of one extractor timing run: prof_extract.sh <tag> [TE_MODEL=.. TE_BLOCK=.. TE_DTYPE=..] -> gpurun_out/pe_<tag>/stats.txt
export TMPDIR=/tmp
R=$PWD
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
OUT=$R/gpurun_out/pe_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o x -- python3 $R/tools/ubench/time_extract.py > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY' > $OUT/stats.txt
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.1f} share {float(r["TotalDurationNs"])/tot:6.3f}')
PY
cat $OUT/run.log | tail -1; cat $OUT/stats.txt
