"""Timeline of the last full step in a rocprofv3 kernel trace: start / end offsets (us), queue, name.
usage: python tools/trace_timeline.py <kernel_trace.csv> [marker substring of the step's last kernel]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else 'adamw_kernel'
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
pick = len(ends) // 2
lo, hi = ends[pick - 1] + 1, ends[pick]
t0 = int(rows[lo]['Start_Timestamp'])
busy = 0
last_end = t0
for r in rows[lo:hi + 1]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print(f"{s / 1e3:8.1f} {e / 1e3:8.1f} {(e - s) / 1e3:6.1f}  q{r['Queue_Id']}  {r['Kernel_Name'][:70]}")
print('step wall us', (int(rows[hi]['End_Timestamp']) - t0) / 1e3)
