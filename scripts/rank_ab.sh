python -m pytest tests/test_evaluator_gpu.py tests/test_engine_r2_gpu.py -x -q -m gpu 2>&1 | tail -n 4
for a in 1 2; do
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print({k:v for k,v in d["distmat"].items() if "rank" in k and "roof" not in k})'
done
