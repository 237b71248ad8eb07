# round 4, call E: fixed tests, loader probe, short bench line with the new legs
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_e; mkdir -p $O
python -m pytest tests/test_trajectory_gpu.py tests/test_fixbase_gpu.py tests/test_data_gpu.py -m gpu -q -s > $O/pytest_new.log 2>&1; grep -n "passed\|failed\|Error\|error\|tail on the native\|20 steps\|frozen head\|max |smoothed\|loss .* ->" $O/pytest_new.log | cut -c1-600
python scripts/loader_probe.py --workers 4,8,16,32 --steps 30 --prefetch 0 > $O/loader_sync.json 2> $O/loader_sync.err; tail -n 45 $O/loader_sync.json; tail -n 3 $O/loader_sync.err
python scripts/loader_probe.py --workers 4,8,16,32 --steps 30 > $O/loader_prefetch.json 2> $O/loader_prefetch.err; tail -n 45 $O/loader_prefetch.json; tail -n 3 $O/loader_prefetch.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --no-loader > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print(json.dumps(d["roofline"], indent=0)[:2500]); print(d.get("dp_path")); print(d["distmat"].get("fp32_d2304"))
PY
tail -n 3 $O/bench.err
