# round-4 evidence, part A: the full GPU suite (per-test timeout 600 s from tests/conftest.py)
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp && export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-2}
O=gpurun_out/final_r4; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $O/pytest_gpu.log 2>&1; tail -n 16 $O/pytest_gpu.log | cut -c1-200
