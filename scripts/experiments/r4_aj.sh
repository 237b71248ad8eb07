# (record of an experiment: IEEE_SIDE_CUS was a throw-away patch of net.hip -- side stream through hipExtStreamCreateWithCUMask -- that is not in the tree)
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_aj; mkdir -p $O
run() { env $1 timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-dp-path --no-roofline-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3), d['config']['loss_last_step'])"; }
for i in 1 2; do for v in "IEEE_X=0" "IEEE_SIDE_CUS=224" "IEEE_SIDE_CUS=192" "IEEE_SIDE_CUS=160" "IEEE_SIDE_CUS=128" "IEEE_SIDE_CUS=192 IEEE_WGRAD_LDS=32"; do run "$v"; done; done | tee $O/ab.txt
