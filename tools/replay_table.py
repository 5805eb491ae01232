"""Per-step kernel table of a rocprofv3 kernel trace cut down to the HIP-graph replays.

usage: python tools/replay_table.py <kernel_trace.csv> [min_us]

bench.py's default run = eager warm-up and probe launches + W + K replays of ONE captured graph + eager roofline-probe
steps.  Only the replays are the timed region.  All kernels of one hipGraphLaunch carry the launch's correlation id, so a
replay is a correlation id with many dispatches; should the profiler number them one by one, the fallback finds the
longest stretch of the trace in which the kernel-name sequence repeats with a fixed period.  The table averages over
the replays only: its sum is <= the measured step (the span from a replay's first kernel start to its last kernel end
is printed beside it)."""
import collections
import csv
import sys


def replay_groups(rows):
    by = collections.OrderedDict()
    for r in rows:
        by.setdefault(r['Correlation_Id'], []).append(r)
    groups = [g for g in by.values() if len(g) >= 8]
    if groups:
        size = collections.Counter(len(g) for g in groups).most_common(1)[0][0]
        sig = collections.Counter(tuple(r['Kernel_Name'] for r in g) for g in groups if len(g) == size).most_common(1)[0][0]
        groups = [g for g in groups if tuple(r['Kernel_Name'] for r in g) == sig]
        if len(groups) >= 3:
            return groups, 'correlation id'
    # fallback: longest periodic stretch of kernel names
    names = [r['Kernel_Name'] for r in rows]
    best = (0, 0, 0)
    for period in range(8, 200):
        run, start = 0, 0
        for i in range(period, len(names)):
            if names[i] == names[i - period]:
                run += 1
                if run // period > best[0]:
                    best = (run // period, period, i - run - period + 1)
            else:
                run = 0
    reps, period, start = best
    if reps < 3:
        return [], 'none'
    return [rows[start + j * period:start + (j + 1) * period] for j in range(reps + 1)], 'periodic kernel-name sequence'


def main():
    path = sys.argv[1]
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    rows = [r for r in csv.DictReader(open(path)) if r['Kind'] == 'KERNEL_DISPATCH']
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    groups, how = replay_groups(rows)
    if not groups:
        print('no graph replays found in', path)
        return 1
    n = len(groups)
    per = collections.OrderedDict()
    for g in groups:
        for r in g:
            t = per.setdefault(r['Kernel_Name'], [0, 0.0])
            t[0] += 1
            t[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    spans = sorted((int(g[-1]['End_Timestamp']) - int(g[0]['Start_Timestamp'])) / 1e3 for g in groups)
    starts = [int(g[0]['Start_Timestamp']) for g in groups]
    gaps = sorted((b - a) / 1e3 for a, b in zip(starts, starts[1:]))
    tot = 0.0
    print(f'{n} graph replays of {len(groups[0])} kernels (found by {how}); eager launches dropped: {len(rows) - n * len(groups[0])}')
    print(f'{"kernel":92s} calls  avg us  us/step')
    for name, (calls, us) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        tot += us / n
        if us / n >= min_us:
            print(f'{name[:92]:92s} {calls / n:5.1f} {us / calls:7.1f} {us / n:8.1f}')
    print(f'kernel us per replay: {tot:.1f}   first-start..last-end span: median {spans[len(spans) // 2]:.1f} us   '
          f'replay period (start to start): median {gaps[len(gaps) // 2]:.1f} us' if gaps else '')
    return 0


if __name__ == '__main__':
    sys.exit(main())
