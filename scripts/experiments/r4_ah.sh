cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_ah; mkdir -p $O
for i in 1 2; do for q in 2 3 4; do
  GPU_MAX_HW_QUEUES=$q timeout 400 python scripts/loader_probe.py --workers 8,16,32 --steps 40 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('queues $q:', 'resident %.2f ms' % d['resident_step_ms'], {w:(round(v['with_train_step_triples_per_s']), round(v['frac_of_resident_step_rate'],3), round(v['host_wait_for_batch_ms_per_step'],2)) for w,v in d['per_workers'].items()})"
done; done | tee $O/loader.txt
