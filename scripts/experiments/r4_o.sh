cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['loader']; print('$1', round(d['ms_per_step'],2), {w:(round(v['with_train_step_triples_per_s']),round(v['frac_of_resident_step_rate'],3),round(v['host_wait_for_batch_ms_per_step'],1)) for w,v in l['per_workers'].items()}, round(l['resident_step_ms'],2))"; }
python bench.py --no-cpu-baseline --no-fp32 --no-distmat --no-dp-path --loader-workers 8,16 2>/dev/null | show "no-dp-path"
python bench.py --no-cpu-baseline --no-fp32 --no-distmat --loader-workers 8,16 2>/dev/null | show "with-dp-path"
python bench.py --no-cpu-baseline --no-fp32 --no-distmat --no-dp-path --no-roofline-pass --loader-workers 8,16 2>/dev/null | show "no-dp-no-roofline"
python scripts/loader_probe.py --workers 8,16 --steps 30 2>/dev/null | python -c "import sys,json; l=json.load(sys.stdin); print('standalone', {w:(round(v['with_train_step_triples_per_s']),round(v['frac_of_resident_step_rate'],3),round(v['host_wait_for_batch_ms_per_step'],1)) for w,v in l['per_workers'].items()}, round(l['resident_step_ms'],2))"
