cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_ai; mkdir -p $O
run() { env $1 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-dp-path --no-roofline-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3))"; }
for i in 1 2 3; do for v in "IEEE_X=0" "IEEE_BRANCH_ASYNC=0" "IEEE_OPT_OVERLAP=0" "IEEE_SIDE_PRIO=0" "GPU_MAX_HW_QUEUES=3" "GPU_MAX_HW_QUEUES=1" "IEEE_WGRAD_LDS=32"; do run "$v"; done; done | tee $O/ab.txt
