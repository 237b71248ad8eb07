# round 6, call o: dual-issue k-loop in the weight-gradient TN core (IEEE_WGRAD_PIPE=6): parity at config 2's shapes, serialized table, step A/B
mkdir -p gpurun_out/r6_o
IEEE_WGRAD_PIPE=6 timeout 900 python -m pytest tests/test_config2_gpu.py -x -q -k "conv_kernels_at_config2_shapes or b64_bf16_engine_step" 2>&1 | tail -3
LEAN="--no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-dp-path --no-config5"
run() { env $2 IEEE_PROFILE_DUMP=gpurun_out/r6_o/$1.csv timeout 600 python bench.py --steps 6 --warmup 3 $LEAN > gpurun_out/r6_o/$1.json 2> gpurun_out/r6_o/$1.err; }
run base IEEE_NOP=1
run pipe6 IEEE_WGRAD_PIPE=6
python scripts/experiments/r6_variants_compare.py base=gpurun_out/r6_o/base.csv pipe6=gpurun_out/r6_o/pipe6.csv 2>&1 | grep -E "kind|wgrad|total" 
timeout 1500 bash scripts/ab5.sh 4 "IEEE_NOP=1" "IEEE_WGRAD_PIPE=6" "IEEE_WGRAD_PIPE=6 IEEE_WGRAD_LDS=64" > gpurun_out/r6_o/ab.txt 2>&1
cat gpurun_out/r6_o/ab.txt
