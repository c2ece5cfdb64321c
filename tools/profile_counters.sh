# SQ / TCC counters of one bench step (own --pmc passes, kernel-trace off), summarised by tools/summarise_counters.py
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/pmc_final
mkdir -p $OUT
cd /tmp
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $ctr --output-format csv -d $OUT/p$i -o x -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-sample > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; exit 1; }
done
echo done
