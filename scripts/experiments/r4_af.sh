cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_af; mkdir -p $O
run() { echo "== $1"; env $1 timeout 300 python scripts/dp_order_probe.py 2>&1 | grep "^round" | tail -1; }
plain() { env $1 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-roofline-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['dp_path']; print('   bench $1: step', round(d['ms_per_step'],3), 'dp leg plain', round(p['plain_ms_per_step'],3), 'staged', round(p['staged_ms_per_step'],3), 'unoverlapped', round(p['unoverlapped_dp_ms_per_step'],3))"; }
for q in 1 2 3 4; do
  run "PROBE_PRECREATE=0 GPU_MAX_HW_QUEUES=$q"
  run "PROBE_PRECREATE=1 GPU_MAX_HW_QUEUES=$q"
  plain "GPU_MAX_HW_QUEUES=$q"
done
