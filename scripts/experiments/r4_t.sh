cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_t; mkdir -p $O
run() { env $1 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-dp-path --no-roofline-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3), 'loss', d['config']['loss_last_step'])"; }
for i in 1 2 3 4; do for v in "IEEE_BN_TOTALS_TILES=0" "IEEE_BN_TOTALS_TILES=64" "IEEE_BN_TOTALS_TILES=256" "IEEE_BN_TOTALS_TILES=256 IEEE_BN_TOTALS_BLOCKS=512" "IEEE_BN_TOTALS_TILES=256 IEEE_BN_TOTALS_BLOCKS=2048"; do run "$v"; done; done | tee $O/ab_totals.txt
export IEEE_BN_TOTALS_TILES=256
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -n 8 $O/pytest.log | cut -c1-300
