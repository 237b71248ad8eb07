"""main-stream stalls and the side-stream tail of one timed-region step of a rocprofv3 kernel trace (tuning aid)"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'nchw_to_nhwc' in r['Kernel_Name']]
step = rows[idx[k]:idx[k + 1]]
t0 = int(step[0]['Start_Timestamp'])
streams = sorted(set(r['Stream_Id'] for r in step))
main_id = max(streams, key=lambda s: sum(1 for r in step if r['Stream_Id'] == s))
main = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in step if r['Stream_Id'] == main_id)
side = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in step if r['Stream_Id'] != main_id)
print('step %.0f us; main busy %.0f us; side busy %.0f us' % ((int(step[-1]['End_Timestamp']) - t0) / 1e3,
      sum(e - s for s, e, _ in main) / 1e3, sum(e - s for s, e, _ in side) / 1e3))
prev = main[0][1]
tot = 0
for s, e, n in main[1:]:
    if s - prev > 20000:
        tot += s - prev
        print('main idle %.0f us before %s at t=%.0f' % ((s - prev) / 1e3, n.split('(')[0][-44:], (s - t0) / 1e3))
    prev = max(prev, e)
print('main idle total (gaps > 20 us): %.0f us' % (tot / 1e3))
