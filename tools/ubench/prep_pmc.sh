# prep kernels: times, then SQ counters (own --pmc passes)
export TMPDIR=/tmp
R=$PWD
python tools/ubench/time_prep.py 2>/dev/null | tail -1
cd /tmp
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $ctr | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_prep/$tag -o x -- python3 $R/tools/ubench/time_prep.py > $R/gpurun_out/pmc_prep/$tag.log 2>&1 || exit 1
done
