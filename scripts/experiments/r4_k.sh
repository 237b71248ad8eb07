cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_k; mkdir -p $O
timeout 1300 python -m pytest tests -m gpu -q --timeout 300 --durations=15 > $O/pytest_gpu.log 2>&1
grep -n "passed\|failed\|Timeout\|timeout" $O/pytest_gpu.log | head -20
grep -n "FAILED\|ERROR" $O/pytest_gpu.log | head -20
