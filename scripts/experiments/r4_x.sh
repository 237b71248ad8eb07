cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_x; mkdir -p $O
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; tail -n 5 $O/pytest.log | cut -c1-300
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-dp-path 2>/dev/null | cut -c1-200
