#!/usr/bin/env python3
"""Per-DISPATCH counters and durations of one kernel from a rocprofv3 run with --pmc ... --kernel-trace:
   pmc_by_dispatch.py <dir> <kernel substring>      -> one row per dispatch in launch order: duration (us), counters"""
import csv
import glob
import sys
from collections import OrderedDict, defaultdict

d, pat = sys.argv[1], sys.argv[2]
cc = glob.glob(f'{d}/**/*counter_collection.csv', recursive=True)
kt = glob.glob(f'{d}/**/*kernel_trace.csv', recursive=True)
dur = {}
for p in kt:
    for r in csv.DictReader(open(p)):
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
rows = OrderedDict()
names = []
for p in cc:
    for r in csv.DictReader(open(p)):
        if pat not in r['Kernel_Name']:
            continue
        k = int(r['Dispatch_Id'])
        rows.setdefault(k, {})[r['Counter_Name']] = float(r['Counter_Value'])
        if r['Counter_Name'] not in names:
            names.append(r['Counter_Name'])
print('dispatch', 'us', *names)
for k in sorted(rows):
    print(k, f"{dur.get(str(k), float('nan')):.1f}", *[f'{rows[k].get(n, float("nan")):.0f}' for n in names])
