# pair / prep rate on maps that only the workspace ("big") mode covers: conv3_3 of an 800x400 print
export TP_C=256 TP_H=200 TP_W=100 TP_Q=16 TP_G=64
timeout -k 10 300 python tools/ubench/time_pair.py fft 2>&1 | tail -2
