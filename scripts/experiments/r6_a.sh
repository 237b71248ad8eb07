# round 6, call a: in-launch weight-gradient fold -- parity, then interleaved A/B against the reduction-launch path
mkdir -p gpurun_out/r6_a
timeout 1500 python -m pytest tests/test_config2_gpu.py -x -q -k "conv_kernels or wgrad_fold" > gpurun_out/r6_a/pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r6_a/pytest.txt
tail -5 gpurun_out/r6_a/pytest.txt
timeout 1200 bash scripts/ab5.sh 4 "IEEE_WGRAD_FOLD=0" "IEEE_WGRAD_FOLD=1" > gpurun_out/r6_a/ab.txt 2>&1
cat gpurun_out/r6_a/ab.txt
