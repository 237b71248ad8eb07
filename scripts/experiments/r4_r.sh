cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_r; mkdir -p $O
timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q -x -k "totals or batchnorm" > $O/pytest.log 2>&1; tail -n 5 $O/pytest.log | cut -c1-300
run() { env $1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-dp-path 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3), 'in-situ', round(d['roofline']['achieved'],1), 'serialized', round(d['roofline']['serialized_achieved'],1), 'loss', d['config']['loss_last_step'])"; }
for i in 1 2; do for v in "IEEE_BN_TOTALS_TILES=0" "IEEE_BN_TOTALS_TILES=64" "IEEE_BN_TOTALS_TILES=256" "IEEE_BN_TOTALS_TILES=1024"; do run "$v"; done; done | tee $O/ab_totals.txt
export IEEE_BN_TOTALS_TILES=256
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 20 --warmup 5 --no-roofline-pass --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-dp-path > $O/trace.log 2>&1
find $O/trace -name "*_kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_totals256.csv
head -25 $O/kernel_stats_totals256.csv | cut -c1-160
rm -rf $O/trace
