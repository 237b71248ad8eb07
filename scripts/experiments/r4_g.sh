cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_g; mkdir -p $O
python scripts/box_diag.py 2>&1 | grep -v amdgpu.ids | grep "round\|plain" | tee $O/box_diag.txt
IEEE_EVENT_RIDE=0 python scripts/box_diag.py 2>&1 | grep -v amdgpu.ids | grep "round\|plain" | tee $O/box_diag_noride.txt
python -m pytest tests/test_fixbase_gpu.py -m gpu -q -x 2>&1 | grep -v "^  \|Warning" | tail -n 30 | cut -c1-1500
python scripts/loader_probe.py --workers 8,16 --steps 30 > $O/loader_prefetch.json 2> $O/loader_prefetch.err; python - <<PY
import json
d=json.load(open("$O/loader_prefetch.json")); print(d["resident_step_ms"]); print(json.dumps(d["per_workers"], indent=0))
PY
