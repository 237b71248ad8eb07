#!/bin/bash
# A/B of the batched weight-gradient reduction (IEEE_WGRAD_BATCH) + the tests that exercise it
mkdir -p gpurun_out/batch
python -m pytest tests/test_conv_gpu.py tests/test_config2_gpu.py tests/test_backward_units_gpu.py tests/test_model_gpu.py tests/test_dp_gpu.py -x -q -m gpu > gpurun_out/batch/tests.log 2>&1
tail -n 5 gpurun_out/batch/tests.log
for rep in 1 2 3; do
  for b in 0 1 2; do
    echo "batch=$b rep=$rep: $(IEEE_WGRAD_BATCH=$b python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-distmat --no-fp32 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d["value"], d["roofline"].get("wgrad"))')"
  done
done | tee gpurun_out/batch/ab.log
