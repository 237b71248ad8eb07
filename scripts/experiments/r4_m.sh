cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
for k in "prefetched" "bench_step or prefetched" "stage_drift or prefetched" "parity_mode or prefetched" "2048-2048_k1 or prefetched"; do
  timeout 300 python -m pytest tests/test_config2_gpu.py tests/test_data_gpu.py -m gpu -q -x -k "$k" --durations=3 2>&1 | grep "prefetched\|passed" | tr '\n' ' '; echo " <= $k"
done
