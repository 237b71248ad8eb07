# round 6, call h: no-epilogue conv bound (rebuilt), ranking probe after the revert, evaluator goldens
mkdir -p gpurun_out/r6_h
bash scripts/experiments/r6_f.sh > gpurun_out/r6_h/noepi.log 2>&1; tail -60 gpurun_out/r6_h/noepi.log
timeout 600 python scripts/rank_probe.py > gpurun_out/r6_h/rank.txt 2>&1; grep -v amdgpu gpurun_out/r6_h/rank.txt
timeout 600 python -m pytest tests/test_evaluator_gpu.py -q 2>&1 | tail -2
