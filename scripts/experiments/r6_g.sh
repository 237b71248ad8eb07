# round 6, call g: device-side range guard (skip + degrade) tests, evaluator goldens with the two-pass ranking kernel,
# ranking probe, no-epilogue conv bound, A/B that the guard costs nothing
mkdir -p gpurun_out/r6_g
timeout 1800 python -m pytest tests/test_trajectory_gpu.py tests/test_engine_gpu.py tests/test_evaluator_gpu.py tests/test_kernels_gpu.py -q > gpurun_out/r6_g/pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r6_g/pytest.txt
tail -6 gpurun_out/r6_g/pytest.txt
timeout 600 python scripts/rank_probe.py > gpurun_out/r6_g/rank.txt 2>&1; grep -v amdgpu gpurun_out/r6_g/rank.txt
bash scripts/experiments/r6_f.sh > gpurun_out/r6_g/noepi.log 2>&1; tail -45 gpurun_out/r6_g/noepi.log
timeout 1200 bash scripts/ab5.sh 3 "IEEE_NOP=1" > gpurun_out/r6_g/ab.txt 2>&1; cat gpurun_out/r6_g/ab.txt
