python -m pytest tests/test_kernels_gpu.py tests/test_backward_units_gpu.py tests/test_config2_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -n 3
for rep in 1 2 3; do
for v in 0 1; do
echo "tiled=$v $(IEEE_POOLED_TILED=$v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-distmat --no-fp32 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d["value"])')"
done
done
