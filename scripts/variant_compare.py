"""compare per-launch conv tables (scripts/experiments/variant_scan.sh) and print, per (layer shape, kind), each variant's time"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/scan'
kinds = sys.argv[2].split(',') if len(sys.argv) > 2 else ['fwd', 'dgrad', 'wgrad']
tabs = {}
for f in sorted(glob.glob(os.path.join(d, '*.csv'))):
    rows = list(csv.DictReader(open(f)))
    agg = collections.OrderedDict()
    for r in rows:
        k = (r['unit'].split(' ', 1)[1], r['kind'])
        a = agg.setdefault(k, [0.0, 0])
        a[0] += float(r['us']); a[1] += 1
    tabs[os.path.basename(f)[:-4]] = agg
names = list(tabs)
base = tabs[names[0]] if 'base' not in tabs else tabs['base']
steps = 4
print('%-34s %-6s %3s ' % ('shape', 'kind', 'n') + ' '.join('%9s' % n[:9] for n in names))
tot = {n: 0.0 for n in names}
best_tot = 0.0
for k, (us, cnt) in base.items():
    if k[1] not in kinds:
        continue
    line = '%-34s %-6s %3d ' % (k[0], k[1], cnt // steps)
    vals = []
    for n in names:
        v = tabs[n].get(k, [float('nan'), 1])[0] / steps
        vals.append(v); tot[n] += v
        line += ' %9.1f' % v
    best_tot += min(vals)
    line += '   best=' + names[vals.index(min(vals))]
    print(line)
print('%-34s %-6s %3s ' % ('TOTAL us/step', '', '') + ' '.join('%9.0f' % tot[n] for n in names), '  best-per-shape %.0f' % best_tot)
