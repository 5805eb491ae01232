"""Where the GPU waits for the host: idle gaps between consecutive kernels of one step in a rocprofv3 kernel trace,
summed per (kernel before the gap -> kernel after it).  usage: python tools/trace_gaps.py <kernel_trace.csv> [marker]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else 'adamw_kernel'
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
# steps end with the LAST optimizer launch of a run of them
last = [i for j, i in enumerate(ends) if j + 1 == len(ends) or ends[j + 1] - i > 50]
pick = len(last) // 2
lo, hi = last[pick - 1] + 1, last[pick]
t0 = int(rows[lo]['Start_Timestamp'])
busy, gaps, big = 0, collections.Counter(), []
prev_end, prev_name = t0, 'step start'
for k, r in enumerate(rows[lo:hi + 1]):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s > prev_end:
        g = (s - prev_end) / 1e3
        if g > 15:
            gaps[(prev_name[:60], r['Kernel_Name'][:60])] += g
            big.append(((prev_end - t0) / 1e3, g, prev_name[:50], r['Kernel_Name'][:50]))
    busy += max(0, e - max(s, prev_end))
    if e > prev_end:
        prev_end, prev_name = e, r['Kernel_Name']
wall = (prev_end - t0) / 1e3
print(f'step wall {wall / 1e3:.2f} ms, busy {busy / 1e6:.2f} ms ({100 * busy / 1e3 / wall:.0f} %), launches {hi - lo + 1}')
print('gaps > 15 us, by position:')
acc = 0
for at, g, a, b in big:
    acc += g
    if g > 300:
        print(f'  at {at / 1e3:7.2f} ms: {g / 1e3:6.2f} ms   {a}  ->  {b}')
print(f'sum of gaps > 15 us: {acc / 1e3:.2f} ms in {len(big)} gaps')
