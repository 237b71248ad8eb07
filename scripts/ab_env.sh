# interleaved A/B of two environments on one box: bash scripts/ab_env.sh "ENV_A=.." "ENV_B=.." [rounds]
A=$1; B=$2; R=${3:-3}
run() { env $1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-distmat --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3), 'gather', round(d['roofline']['achieved'],1), 'wgrad', round(d['roofline']['wgrad']['achieved'],1))"; }
for i in $(seq $R); do run "$A"; run "$B"; done
