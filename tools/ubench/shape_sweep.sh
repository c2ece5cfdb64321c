# pair-kernel rate on the other feature shapes of SURVEY §8 (conv4_3, conv5_3, ResNet50 layer3)
set -e
for cfg in "512 64 32 64 1024" "512 32 16 64 4096" "1024 32 16 64 4096"; do
  set -- $cfg
  echo "C=$1 H=$2 W=$3 Q=$4 G=$5"
  TP_C=$1 TP_H=$2 TP_W=$3 TP_Q=$4 TP_G=$5 timeout -k 10 200 python tools/ubench/time_pair.py ${METHODS:-fft} 2>&1 | tail -4
done
