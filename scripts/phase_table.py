"""Where the launch stream's time goes in one training step of a rocprofv3 --kernel-trace CSV (tuning aid).
usage: python scripts/phase_table.py TRACE_DIR [step index from the end, default 2]
Prints, for the stream with the most launches (the caller's stream): busy time, the gaps between consecutive kernels
(total and histogram), and kernel time by family; then the same family table for the other streams."""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'nchw_to_nhwc' in r['Kernel_Name']]
step = rows[idx[-back - 1]:idx[-back]]
t0 = int(step[0]['Start_Timestamp'])
t1 = max(int(r['End_Timestamp']) for r in step)


def fam(n):
    m = re.search(r'ieee(?:::|\d+)([a-z_0-9]+?)(?:_kernel|I|<|\()', n.replace('_ZN4ieee', 'ieee'))
    s = n.split('(')[0].replace('void ', '').replace('ieee::', '')
    m2 = re.match(r'_ZN4ieee(\d+)(.*)', n)
    if m2:
        L = int(m2.group(1))
        s = m2.group(2)[:L]
    return s.replace('_kernel', '').split('<')[0]


streams = collections.Counter(r.get('Stream_Id', r.get('Queue_Id')) for r in step)
key = 'Stream_Id' if 'Stream_Id' in step[0] else 'Queue_Id'
main_id = streams.most_common(1)[0][0]
print('step span %.1f us, %d launches on %d streams' % ((t1 - t0) / 1e3, len(step), len(streams)))
for sid, cnt in streams.most_common():
    ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in step if r[key] == sid)
    busy = sum(e - s for s, e, _ in ks) / 1e3
    gaps = [ks[i + 1][0] - max(k[1] for k in ks[:i + 1][-8:]) for i in range(len(ks) - 1)]
    gaps = [g / 1e3 for g in gaps if g > 0]
    print('\nstream %s%s: %d launches, busy %.0f us, gaps %.0f us (%d gaps; <2us %d, 2-5us %d, 5-20us %d, >20us %d: %.0f us)'
          % (sid, ' (launch stream)' if sid == main_id else '', cnt, busy, sum(gaps), len(gaps), sum(g < 2 for g in gaps),
             sum(2 <= g < 5 for g in gaps), sum(5 <= g < 20 for g in gaps), sum(g >= 20 for g in gaps), sum(g for g in gaps if g >= 20)))
    agg = collections.defaultdict(lambda: [0.0, 0])
    for s, e, n in ks:
        a = agg[fam(n)]
        a[0] += (e - s) / 1e3
        a[1] += 1
    for k, (us, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:22]:
        print('   %-34s n=%3d %8.1f us  avg %6.1f' % (k[:34], n, us, us / n))
