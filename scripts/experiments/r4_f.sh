cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_f; mkdir -p $O
python scripts/box_diag.py 2>&1 | grep -v amdgpu.ids | tee $O/box_diag.txt
python -m pytest tests/test_fixbase_gpu.py tests/test_data_gpu.py -m gpu -q 2>&1 | tail -n 3
python scripts/loader_probe.py --workers 8,16,32 --steps 30 > $O/loader_prefetch.json 2> $O/loader_prefetch.err; python - <<PY
import json
d=json.load(open("$O/loader_prefetch.json")); print(d["resident_step_ms"]); print(json.dumps(d["per_workers"], indent=0))
PY
python scripts/loader_probe.py --workers 8,16,32 --steps 30 --prefetch 0 > $O/loader_sync.json 2> $O/loader_sync.err; python - <<PY
import json
d=json.load(open("$O/loader_sync.json")); print(d["resident_step_ms"]); print(json.dumps(d["per_workers"], indent=0))
PY
