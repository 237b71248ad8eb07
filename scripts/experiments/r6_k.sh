# round 6, call k: bench-leg tests; timing-only upper bound of consumer-side BatchNorm (applies off the launch stream)
mkdir -p gpurun_out/r6_k
timeout 1200 python -m pytest tests/test_bench_gpu.py -q 2>&1 | tail -4
IEEE_BN_STRICT=0 timeout 1500 bash scripts/ab5.sh 4 "IEEE_NOP=1" "IEEE_DBG_BN_ASIDE=1" > gpurun_out/r6_k/ab.txt 2>&1
cat gpurun_out/r6_k/ab.txt
