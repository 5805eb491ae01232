"""Idle time between consecutive graph replays in a rocprofv3 kernel trace: from the end of the step's last kernel to
the start of the next step's first.  usage: python tools/trace_step_gap.py <kernel_trace.csv> <last-kernel substring>"""
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
mark = sys.argv[2]
gaps, walls, prev_start = [], [], None
for i, r in enumerate(rows[:-1]):
    if mark in r['Kernel_Name']:
        end = max(int(x['End_Timestamp']) for x in rows[max(0, i - 6):i + 1])
        nxt = int(rows[i + 1]['Start_Timestamp'])
        gaps.append((nxt - end) / 1e3)
        if prev_start is not None:
            walls.append((nxt - prev_start) / 1e3)
        prev_start = nxt
gaps = gaps[len(gaps) // 4:]
walls = walls[len(walls) // 4:]
print('steps', len(gaps), 'median gap us', round(statistics.median(gaps), 2), 'median step period us', round(statistics.median(walls), 2))
