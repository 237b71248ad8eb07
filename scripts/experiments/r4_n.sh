cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_n; mkdir -p $O
timeout 600 python -m pytest tests/test_config2_gpu.py tests/test_data_gpu.py -m gpu -q -x -k "bench_step or prefetched or decode_loader" --durations=4 2>&1 | grep "s call\|passed\|failed" 
timeout 600 python scripts/loader_probe.py --workers 8,16 --steps 40 > $O/loader.json 2> $O/loader.err; tail -n 3 $O/loader.err; python - <<PY
import json
d=json.load(open("$O/loader.json")); print(d["resident_step_ms"]); print(json.dumps(d["per_workers"], indent=0)); print(d["recommended_workers"], d["gpu_step_stops_waiting_at_workers"])
PY
IEEE_LOADER_START=fork timeout 600 python scripts/loader_probe.py --workers 8 --steps 40 > $O/loader_fork.json 2> $O/loader_fork.err; python - <<PY
import json
d=json.load(open("$O/loader_fork.json")); print("fork:", json.dumps(d["per_workers"], indent=0))
PY
