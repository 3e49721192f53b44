"""rocprofv3 --kernel-trace CSV -> per kernel: launches, average duration, average idle time in front of it (the gap
between the end of the previous kernel on the device and its start), over the last `--tail` fraction of the trace.

    python tools/trace_gaps.py DIR_OR_CSV [--tail 0.5]

Answers "where does a cycle go" for graph-replayed paths, where HIP events cannot be placed between the kernels."""
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r'\(.*$', '', name).strip().replace('void ', '')
    return re.sub(r'<.*$', '', name)[:60]


def main():
    src = sys.argv[1]
    tail = float(sys.argv[sys.argv.index('--tail') + 1]) if '--tail' in sys.argv else 0.5
    paths = [src] if src.endswith('.csv') else glob.glob(os.path.join(src, '**', '*kernel_trace.csv'), recursive=True)
    rows = []
    for p in paths:
        with open(p) as f:
            for r in csv.DictReader(f):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
    rows.sort()
    rows = rows[int(len(rows) * (1 - tail)):]
    stat = {}
    prev_end = None
    for s, e, n in rows:
        d = stat.setdefault(n, [0, 0, 0, 0])
        d[0] += 1
        d[1] += e - s
        if prev_end is not None:
            d[2] += max(0, s - prev_end)
            d[3] += 1
        prev_end = max(prev_end or 0, e)
    span = rows[-1][1] - rows[0][0]
    busy = sum(v[1] for v in stat.values())
    print('%d kernels over %.3f ms; kernels %.3f ms (%.1f %%), idle %.3f ms' %
          (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6))
    print('%-62s %8s %10s %10s %10s' % ('kernel', 'n', 'avg us', 'gap us', 'total ms'))
    for n, v in sorted(stat.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        print('%-62s %8d %10.2f %10.2f %10.3f' % (n, v[0], v[1] / v[0] / 1e3, v[2] / max(1, v[3]) / 1e3,
                                                  (v[1] + v[2]) / 1e6))


if __name__ == '__main__':
    main()
