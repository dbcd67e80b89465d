#!/usr/bin/env python3
"""Cut a rocprofv3 kernel trace to bench.py's timed region and aggregate per kernel.
usage: summarize_trace.py <kernel_trace.csv> <marker.json> <out.csv>"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name[:150]


def main():
    trace, marker, out = sys.argv[1:4]
    rows = list(csv.DictReader(open(trace)))
    sk = 'Start_Timestamp' if 'Start_Timestamp' in rows[0] else [k for k in rows[0] if 'start' in k.lower()][0]
    ek = 'End_Timestamp' if 'End_Timestamp' in rows[0] else [k for k in rows[0] if 'end' in k.lower()][0]
    nk = 'Kernel_Name' if 'Kernel_Name' in rows[0] else [k for k in rows[0] if 'name' in k.lower()][0]
    lo, hi = min(int(r[sk]) for r in rows), max(int(r[ek]) for r in rows)
    m = json.load(open(marker))
    clock = None
    for c, v in m['start'].items():
        if lo <= v <= hi:
            clock = c
            break
    if clock is None:
        print('no clock of the marker falls inside the trace', lo, hi, m, file=sys.stderr)
        sys.exit(2)
    t0, t1 = m['start'][clock], m['end'][clock]
    agg = defaultdict(lambda: [0, 0])
    busy = 0
    for r in rows:
        s, e = int(r[sk]), int(r[ek])
        if s >= t0 and e <= t1:
            a = agg[short(r[nk])]
            a[0] += 1
            a[1] += e - s
            busy += e - s
    steps = m.get('steps', 1)
    wall = t1 - t0
    with open(out, 'w') as fh:
        fh.write(f'# clock {clock}; timed region {wall / 1e6:.1f} ms over {steps} steps; sum of kernel durations {busy / 1e6:.1f} ms '
                 f'({100 * busy / wall:.1f}% of wall); {sum(a[0] for a in agg.values())} dispatches\n')
        fh.write('kernel,calls_per_step,ms_per_step,avg_us,percent_of_kernel_time\n')
        for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            fh.write(f'"{name}",{n / steps:.1f},{t / 1e6 / steps:.3f},{t / 1e3 / n:.2f},{100 * t / busy:.2f}\n')
    print(open(out).read()[:6000])


if __name__ == '__main__':
    main()
