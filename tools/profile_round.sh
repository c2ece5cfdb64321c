# One round's rocprofv3 evidence of `python bench.py` (default workload): kernel stats, HBM-side traffic (FETCH_SIZE and
# WRITE_SIZE in passes of their own), SQ / TCC counter groups.  usage: bash tools/profile_round.sh r02   (on the GPU box)
# Counter passes carry no trace domains; the program after "--" is python3 itself.
export TMPDIR=/tmp
R=$PWD
TAG=$1
O=$R/gpurun_out/${TAG}_prof
rm -rf $O; mkdir -p $O
cd /tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extractor --no-secondary"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o x -- $B > $O/stats.log 2>&1 || { echo stats failed; tail -3 $O/stats.log; exit 1; }
echo "stats done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o x -- $B --steps 1 --warmup 0 > $O/fetch.log 2>&1 || { echo fetch failed; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o x -- $B --steps 1 --warmup 0 > $O/write.log 2>&1 || { echo write failed; exit 1; }
echo "traffic done"
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $O/sq/p$i -o x -- $B --steps 1 --warmup 0 > $O/sq$i.log 2>&1 || { echo "sq pass $i failed"; exit 1; }
done
echo "counters done"
cd $R
python3 tools/summarise_profile.py $TAG $O/stats $O/fetch $O/write 150000
python3 tools/summarise_counters.py $TAG $O/sq
