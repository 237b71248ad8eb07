cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_i; mkdir -p $O
python -m pytest tests/test_fixbase_gpu.py tests/test_data_gpu.py -m gpu -q 2>&1 | tail -n 4
python scripts/loader_probe.py --workers 8,16,32 --steps 40 > $O/loader_ring.json 2> $O/loader_ring.err; tail -n 3 $O/loader_ring.err; python - <<PY
import json
d=json.load(open("$O/loader_ring.json")); print(d["resident_step_ms"]); print(json.dumps(d["per_workers"], indent=0)); print(d["recommended_workers"], d["gpu_step_stops_waiting_at_workers"])
PY
run() { env $1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-dp-path 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3), 'in-situ', round(d['roofline']['achieved'],1), 'serialized', round(d['roofline']['serialized_achieved'],1), 'wgrad', round(d['roofline']['wgrad']['achieved'],1), round(d['roofline']['wgrad']['serialized_achieved'],1))"; }
for i in 1 2; do for v in "X=0" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=2" "GPU_MAX_HW_QUEUES=16"; do run "$v"; done; done | tee $O/ab_queues.txt
