# L2 hit/miss and memory-side fetch of the pair kernel, tile schedule vs team schedule (own --pmc passes)
export TMPDIR=/tmp TP_Q=100 TP_G=1500
R=$PWD
cd /tmp
for cfg in "0 0 0" "1 256 32" "1 2048 8"; do
  set -- $cfg
  export SPR_NCC_TEAM=$1 SPR_NCC_TEAM_POLLS=$2 SPR_NCC_TEAM_EVERY=$3
  for ctr in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
    tag="t$1_p$2_e$3_$(echo $ctr | cut -c1-7)"
    timeout -k 10 240 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_team/$tag -o x -- python3 $R/tools/ubench/time_pair.py fft > $R/gpurun_out/pmc_team/$tag.log 2>&1 || exit 1
  done
done
