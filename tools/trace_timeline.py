"""Prints the launch timeline of the last frame in a rocprofv3 kernel trace (t_kernel_trace.csv):
start, end, duration in ms relative to the frame's first kernel, hardware queue, kernel name."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'extend_camera' in r['Kernel_Name']]
start = idx[-1]
t0 = int(rows[start]['Start_Timestamp'])
for r in rows[start:]:
    s = int(r['Start_Timestamp']) - t0
    e = int(r['End_Timestamp']) - t0
    name = r['Kernel_Name'].replace('void ', '').split('(')[0][:70]
    print("%9.3f %9.3f %8.3f  q%s  %s" % (s / 1e6, e / 1e6, (e - s) / 1e6, r.get('Queue_Id'), name))
