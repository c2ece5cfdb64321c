# SQ / TCC counters of one extractor timing run: pmc_extract.sh <tag> [TE_MODEL=.. TE_BLOCK=.. TE_DTYPE=..] -> gpurun_out/pm_<tag>/summary.txt
# (own --pmc passes, no trace domains; the program after "--" is python3 itself)
export TMPDIR=/tmp
R=$PWD
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
OUT=$R/gpurun_out/pm_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $OUT/p$i -o x -- python3 $R/tools/ubench/time_extract.py > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done
python3 - "$OUT" <<'PY' > $OUT/summary.txt
import collections, csv, glob, os, sys
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "spr::" not in k: continue
        k = k.split("spr::(anonymous namespace)::")[-1].split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
    c = {a: acc[k][a] / len(n[k][a]) for a in acc[k]}
    print(k, "dispatches", max(len(v) for v in n[k].values()))
    for a in sorted(c): print(f"  {a:32s} {c[a]:.4g}")
    w = c.get("SQ_WAVE_CYCLES", 0)
    if w:
        for a in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
            if a in c: print(f"  {a}/WAVE_CYCLES = {c[a]/w:.3f}")
    if c.get("SQ_LDS_IDX_ACTIVE"): print(f"  bank conflict rate = {c.get('SQ_LDS_BANK_CONFLICT',0)/c['SQ_LDS_IDX_ACTIVE']:.3f}")
    if c.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in c: print(f"  MFMA busy / SQ busy = {c['SQ_VALU_MFMA_BUSY_CYCLES']/c['SQ_BUSY_CYCLES']:.3f}")
    if "TCC_HIT_sum" in c: print(f"  L2 hit = {c['TCC_HIT_sum']/(c['TCC_HIT_sum']+c['TCC_MISS_sum']):.3f}")
PY
head -120 $OUT/summary.txt
