cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_ac; mkdir -p $O
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_backward_units_gpu.py tests/test_engine_gpu.py tests/test_model_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log | cut -c1-300
for i in 1 2; do python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-dp-path --no-roofline-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"; done
