# interleaved A/B of environments on one box, plain timed steps only:
#   bash scripts/ab5.sh ROUNDS "ENV_A=.. ENV_A2=.." "ENV_B=.." ["ENV_C=.." ...]     ("-" = no variables)
R=$1; shift
run() { env $1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-dp-path --no-config5 --no-roofline-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s' % '$1', round(d['value'],1), round(d['ms_per_step'],3))"; }
for i in $(seq $R); do for e in "$@"; do if [ "$e" = "-" ]; then run "IEEE_NOP=1"; else run "$e"; fi; done; done
