#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of `bench.py`: per-kernel statistics over the LAST `steps` launches of each fcpp kernel, i.e.
the timed steps only (bench.py's placement calibration and warm-up launch the same kernels before them).
    timed_kernel_stats.py <kernel_trace.csv> <steps> [skip_last] > profiles/..._timed.csv
skip_last: launches BEHIND the timed steps that are not them (since round 5 bench.py's headline ends with the sustained regions -- two plan calls
in flight on two streams, (REPS_SHORT + 1) x K = 520 launches at the driver's K = 20 -- whose kernels overlap and take longer)."""
import csv
import sys
from collections import defaultdict

rows, steps = list(csv.DictReader(open(sys.argv[1]))), int(sys.argv[2])
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
by = defaultdict(list)
for r in rows:
    by[r['Kernel_Name'].split('(')[0]].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
print('"Name","CallsInTrace","TimedCalls","AverageNs","MinNs","MaxNs"')
for name, v in sorted(by.items()):
    if 'fcpp::k_plan' not in name and 'k_reduce_stats' not in name and 'k_quiet_run_stats' not in name:
        continue
    v.sort()
    sel = v[-(steps + skip):-skip] if skip and 'k_plan_sparse_fields' in name else v[-steps:]
    d = [e - s for s, e in sel]
    print(f'"{name}",{len(v)},{len(d)},{sum(d) / len(d):.1f},{min(d)},{max(d)}')
