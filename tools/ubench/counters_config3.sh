# SQ counter passes of the matrix-core pair kernel (bench config 3, a quarter of the gallery).  usage (GPU box): bash tools/ubench/counters_config3.sh <tag>
export TMPDIR=/tmp
R=$PWD
TAG=${1:-mfma}
O=$R/gpurun_out/${TAG}_c3
rm -rf $O; mkdir -p $O
cd /tmp
B="python3 $R/bench.py --config 3 --gallery-per-gpu 2560 --no-cpu-baseline --no-extractor --steps 1 --warmup 0"
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $O/sq/p$i -o x -- $B > $O/sq$i.log 2>&1 || { echo "sq pass $i failed"; tail -5 $O/sq$i.log; exit 1; }
done
cd $R
python3 tools/summarise_counters.py ${TAG}_config3 $O/sq
