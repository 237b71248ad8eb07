cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_aa; mkdir -p $O
PROBE_REPLICAS=4 python scripts/bn_totals_probe.py 2>&1 | grep "^M=" | tee $O/probe.txt
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_backward_units_gpu.py tests/test_engine_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log | cut -c1-300
cp ieee_amd/libieee_amd.so /tmp/after.so; cp ieee_amd/libieee_amd_before.so /tmp/before.so
for i in 1 2 3 4; do for v in before after; do cp /tmp/$v.so ieee_amd/libieee_amd.so; python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-dp-path --no-roofline-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['ms_per_step'],3), 'loss', d['config']['loss_last_step'])"; done; done | tee $O/ab.txt
