# A/B of the pair kernel's schedules at full size: tile (default) vs team (SPR_NCC_TEAM=1), kernel time only
set -e
export TP_Q=100 TP_G=1500
for cfg in "0 0 0" "1 256 32" "1 2048 8" "1 256 0"; do
  set -- $cfg
  echo "TEAM=$1 POLLS=$2 EVERY=$3"
  SPR_NCC_TEAM=$1 SPR_NCC_TEAM_POLLS=$2 SPR_NCC_TEAM_EVERY=$3 timeout -k 10 200 python tools/ubench/time_pair.py fft 2>/dev/null | tail -1
done
