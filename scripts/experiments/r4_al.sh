cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_al; mkdir -p $O
run() { echo "== $1"; env $1 timeout 300 python scripts/dp_order_probe.py 2>&1 | grep "^round 2"; }
for pre in 0 1; do
  run "PROBE_PRECREATE=$pre GPU_MAX_HW_QUEUES=1 PROBE_FAKE_COMM_US=0 IEEE_COMM_PRIO=-1"
  run "PROBE_PRECREATE=$pre GPU_MAX_HW_QUEUES=1 PROBE_FAKE_COMM_US=150 IEEE_COMM_PRIO=-1"
  run "PROBE_PRECREATE=$pre GPU_MAX_HW_QUEUES=1 PROBE_FAKE_COMM_US=600 IEEE_COMM_PRIO=-1"
  run "PROBE_PRECREATE=$pre GPU_MAX_HW_QUEUES=2 PROBE_FAKE_COMM_US=600"
done
# two ranks on one GPU over gloo, the settings a torchrun job gets by default (WORLD_SIZE=2 -> 1 queue, comm at high priority)
IEEE_DIST_BACKEND=gloo IEEE_FORCE_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-roofline-pass 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin.read().splitlines() if x.startswith(chr(123))]; d=json.loads(l[-1]); print(len(l), d['n_gpus'], round(d['value'],1), round(d['ms_per_step'],1), d['dp_calibration'])"
