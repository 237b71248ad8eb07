# round 6, call b: chained weight gradients (previous reduction as the next launch's prologue) -- parity, then interleaved A/B
mkdir -p gpurun_out/r6_b
timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_config2_gpu.py tests/test_model_gpu.py tests/test_engine_gpu.py -x -q > gpurun_out/r6_b/pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r6_b/pytest.txt
tail -5 gpurun_out/r6_b/pytest.txt
timeout 1500 bash scripts/ab5.sh 4 "IEEE_WGRAD_CHAIN=0" "IEEE_WGRAD_CHAIN=1" > gpurun_out/r6_b/ab.txt 2>&1
cat gpurun_out/r6_b/ab.txt
