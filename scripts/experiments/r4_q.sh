cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_q; mkdir -p $O
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_model_gpu.py tests/test_backward_units_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; tail -n 12 $O/pytest.log | cut -c1-300
run() { env $1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-dp-path 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3), 'in-situ', round(d['roofline']['achieved'],1), 'serialized', round(d['roofline']['serialized_achieved'],1), 'loss', d['config']['loss_last_step'])"; }
for i in 1 2 3; do for v in "IEEE_BN_TOTALS_TILES=0" "IEEE_BN_TOTALS_TILES=64" "IEEE_BN_TOTALS_TILES=256" "IEEE_BN_TOTALS_TILES=1024"; do run "$v"; done; done | tee $O/ab_totals.txt
