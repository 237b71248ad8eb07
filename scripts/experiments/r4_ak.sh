cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_ak; mkdir -p $O
run() { echo "== $1"; env $1 timeout 300 python scripts/dp_order_probe.py 2>&1 | grep "^round 2"; }
for pre in 0 1; do
  run "PROBE_PRECREATE=$pre GPU_MAX_HW_QUEUES=2 PROBE_FAKE_COMM_US=300 IEEE_COMM_PRIO=-1"
  run "PROBE_PRECREATE=$pre GPU_MAX_HW_QUEUES=2 PROBE_FAKE_COMM_US=0 IEEE_COMM_PRIO=-1"
  run "PROBE_PRECREATE=$pre GPU_MAX_HW_QUEUES=1 PROBE_FAKE_COMM_US=300 IEEE_COMM_PRIO=-1"
  run "PROBE_PRECREATE=$pre GPU_MAX_HW_QUEUES=1 PROBE_FAKE_COMM_US=300"
done
