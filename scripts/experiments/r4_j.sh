cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_j; mkdir -p $O
python -m pytest tests/test_fixbase_gpu.py tests/test_data_gpu.py tests/test_engine_gpu.py -m gpu -q 2>&1 | tail -n 4
python scripts/loader_probe.py --workers 8,16,32 --steps 40 > $O/loader_ring.json 2> $O/loader_ring.err; tail -n 3 $O/loader_ring.err; python - <<PY
import json
d=json.load(open("$O/loader_ring.json")); print(d["resident_step_ms"]); print(json.dumps(d["per_workers"], indent=0)); print(d["recommended_workers"], d["gpu_step_stops_waiting_at_workers"])
PY
run() { env $1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-dp-path 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3), 'in-situ', round(d['roofline']['achieved'],1), 'serialized', round(d['roofline']['serialized_achieved'],1), 'wgrad', round(d['roofline']['wgrad']['achieved'],1), round(d['roofline']['wgrad']['serialized_achieved'],1))"; }
for i in 1 2 3; do for v in "X=0" "IEEE_BENCH_HIPRIO=1"; do run "$v"; done; done | tee $O/ab_prio.txt
bash scripts/in_situ_stats.sh $O/in_situ r04 > $O/in_situ.log 2>&1; tail -n 3 $O/in_situ.log
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/in_situ/r04_kernel_stats_in_situ.csv")))
fam=lambda names: (sum(float(r["TotalDurationNs"]) for r in rows if any(k in r["Name"] for k in names)), sum(int(r["Calls"]) for r in rows if any(k in r["Name"] for k in names)))
for nm,names in (("fwd+dgrad",("conv_gather_kernel","conv3x3_patch_kernel","stem_conv_kernel")),("wgrad+reduce",("conv_wgrad_kernel","conv3x3_wgrad_patch_kernel","stem_wgrad_kernel","wgrad_reduce"))):
    ns,c=fam(names); print(nm, "ms/step", ns/25e6, "launches/step", c/25, "avg us", ns/c/1e3, "TFLOP/s", (64*61.0617e9 if nm[0]=="f" else 64*30.762e9)/(ns/25*1e-9)/1e12)
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace*.csv" -size +8M -delete; du -sh $O
