#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc counter_collection CSV:   pmc_summary.py <..._counter_collection.csv> [more.csv ...]
Prints one row per (kernel, counter): dispatches, mean value per dispatch."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0, 0.0])
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r.get('Kernel_Name', '').split('(')[0]
        if 'fcpp' not in name:
            continue
        key = (name, r['Counter_Name'])
        acc[key][0] += 1
        acc[key][1] += float(r['Counter_Value'])
print('"Kernel","Counter","Dispatches","MeanPerDispatch"')
for (name, ctr), (n, s) in sorted(acc.items()):
    print(f'"{name}","{ctr}",{n},{s / n:.1f}')
