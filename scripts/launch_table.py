"""summarise gpurun_out/launches.csv (IEEE_PROFILE_DUMP) by layer shape"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/launches.csv')))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = len(rows) // steps
rows = rows[-n:]
tot = sum(float(r['us']) for r in rows)
print('conv launches per step', n, 'total conv us %.0f' % tot)
agg = collections.OrderedDict()
for r in rows:
    k = (r['unit'].split(' ', 1)[1], r['kind'])
    agg.setdefault(k, []).append((float(r['us']), float(r['gflop'])))
top = int(sys.argv[3]) if len(sys.argv) > 3 else 14
for k, v in sorted(agg.items(), key=lambda kv: -sum(x[0] for x in kv[1]))[:top]:
    us = sum(x[0] for x in v)
    gf = sum(x[1] for x in v)
    # gflop / us = 1e9 flop / 1e-6 s = 1e15 flop/s -> x 1e3 for TFLOP/s
    print('%-30s %-7s n=%2d %8.1f us %7.1f TFLOP/s %5.1f%%' % (k[0], k[1], len(v), us, gf / us * 1e3, 100 * us / tot))
