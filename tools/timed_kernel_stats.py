#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of `bench.py`: per-kernel statistics over the LAST `steps` launches of each fcpp kernel, i.e.
the timed steps only (bench.py's placement calibration and warm-up launch the same kernels before them).
    timed_kernel_stats.py <kernel_trace.csv> <steps> > profiles/..._timed.csv"""
import csv
import sys
from collections import defaultdict

rows, steps = list(csv.DictReader(open(sys.argv[1]))), int(sys.argv[2])
by = defaultdict(list)
for r in rows:
    by[r['Kernel_Name'].split('(')[0]].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
print('"Name","CallsInTrace","TimedCalls","AverageNs","MinNs","MaxNs"')
for name, v in sorted(by.items()):
    if 'fcpp::k_plan' not in name and 'k_reduce_stats' not in name and 'k_quiet_run_stats' not in name:
        continue
    v.sort()
    d = [e - s for s, e in v[-steps:]]
    print(f'"{name}",{len(v)},{len(d)},{sum(d) / len(d):.1f},{min(d)},{max(d)}')
