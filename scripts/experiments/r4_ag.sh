cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_ag; mkdir -p $O
for pre in 0 1; do PROBE_PRECREATE=$pre timeout 300 python scripts/dp_order_probe.py 2>&1 | grep "^round\|^GPU_MAX" | tail -2; done
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -n 3 $O/bench.time | head -1
python -c "
import json; d=json.load(open('$O/bench.json')); p=d['dp_path']; l=d['loader']['per_workers']
print(d['value'], d['ms_per_step'], 'dp', p['plain_ms_per_step'], p['staged_ms_per_step'], p['unoverlapped_dp_ms_per_step'], 'loader', {w:round(v['frac_of_resident_step_rate'],3) for w,v in l.items()}, 'fp32', d['fp32_parity_mode']['value'], 'distmat', d['distmat']['fp32']['ms'], d['distmat']['rank_kernels_ms'])"
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -n 2 $O/pytest.log | cut -c1-200
