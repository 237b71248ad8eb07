cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_ad; mkdir -p $O
run() { env $1 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-dp-path --no-roofline-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3))"; }
for i in 1 2 3; do for v in "IEEE_X=0" "IEEE_DS_SUMS=1" "IEEE_EW_BLOCKS=2048" "IEEE_BN_FIXED=15"; do run "$v"; done; done | tee $O/ab.txt
