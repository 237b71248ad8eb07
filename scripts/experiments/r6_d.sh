# round 6, call d: new tests (pair GEMM, persistent stem, optimizer shadow), the degrade probe, A/B of stem walk and shadow
mkdir -p gpurun_out/r6_d
timeout 900 python scripts/experiments/r6_degrade_probe.py > gpurun_out/r6_d/degrade_probe.txt 2>&1
cat gpurun_out/r6_d/degrade_probe.txt | grep -v Warning | tail -8
timeout 1800 python -m pytest tests/test_kernels_gpu.py tests/test_conv_gpu.py tests/test_engine_gpu.py tests/test_engine_r2_gpu.py tests/test_model_gpu.py tests/test_data_gpu.py tests/test_dp_gpu.py -q > gpurun_out/r6_d/pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r6_d/pytest.txt
tail -8 gpurun_out/r6_d/pytest.txt
timeout 1500 bash scripts/ab5.sh 4 "IEEE_STEM_WALK=1 IEEE_SGD_SHADOW=0" "IEEE_STEM_WALK=4 IEEE_SGD_SHADOW=0" "IEEE_STEM_WALK=4 IEEE_SGD_SHADOW=1" > gpurun_out/r6_d/ab.txt 2>&1
cat gpurun_out/r6_d/ab.txt
