cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_y; mkdir -p $O
run() { env $1 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-roofline-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['dp_path']; print('$1', round(d['ms_per_step'],3), 'dp plain', round(p['plain_ms_per_step'],3), 'staged', round(p['staged_ms_per_step'],3), 'overhead', round(p['dp_path_overhead_ms'],3))"; }
for i in 1 2 3; do for v in "IEEE_X=0" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=6"; do run "$v"; done; done | tee $O/ab.txt
