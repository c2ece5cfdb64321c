# first contact of the matrix-core scorer with the hardware: its parity tests, then the config-3 bench at several noise levels
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "matrix_core" > gpurun_out/mfma_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/mfma_tests.log
for n in ${NOISES:-40}; do
timeout -k 10 300 python bench.py --config 3 --steps 2 --warmup 1 --no-extractor --no-cpu-baseline --noise $n > gpurun_out/mfma_bench3_n$n.log 2>&1 || exit 1
python - <<PY
import json
d=json.loads(open("gpurun_out/mfma_bench3_n$n.log").read().strip().splitlines()[-1])
print("noise $n", d["value"], d["rank1"], d["mAP"], d["mean_rank"], d["roofline"]["avg_launch_ms"], d["roofline"]["issued_frac_of_peak"])
PY
done
