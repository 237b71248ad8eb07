"""print one training step of a rocprofv3 --kernel-trace CSV as a timeline (tuning aid)"""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0, 1e12)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'nchw_to_nhwc' in r['Kernel_Name']]
step = rows[idx[-2]:idx[-1]]
t0 = int(step[0]['Start_Timestamp'])
tot = 0
for r in step:
    nm = r['Kernel_Name']
    m = re.match(r'_ZN4ieee(\d+)', nm)
    short = nm.split('(')[0].replace('void ieee::', '').replace('ieee::', '')
    if m:
        L = int(m.group(1)); i = nm.index(m.group(1)) + len(m.group(1)); short = nm[i:i + L] + nm[i + L:i + L + 26]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    t = (int(r['Start_Timestamp']) - t0) / 1e3
    tot += d
    if lo <= t <= hi:
        print('%-48s %7.1f us  wgs %6d  t=%8.1f q%s' % (short[:48], d, int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // int(r['Workgroup_Size_X']), t, r['Queue_Id']))
print('step span us', (int(step[-1]['End_Timestamp']) - t0) / 1e3, 'kernel sum', tot)
