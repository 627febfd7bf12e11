#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of tools/trace_create.py: the timeline of ONE plan call -- for every kernel of a call its start
relative to the call's first kernel (k_plan_fields) and its duration, medians over the calls of the trace.
    call_timeline.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'fcpp' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))


def short(n):
    return n.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').replace('fcpp::', '').strip()


calls, cur = [], None
for r in rows:
    n = short(r['Kernel_Name'])
    # (a large batch's planner and counting pass run in chunks: a call begins with the first planner launch after a fill pass)
    if n.startswith('k_plan_fields') and (cur is None or any(x[0].startswith('k_tile_fields<true') for x in cur)):
        cur = []
        calls.append(cur)
    if cur is not None:
        cur.append((n, int(r['Start_Timestamp']), int(r['End_Timestamp'])))
calls = [c for c in calls[2:] if len(c) == len(calls[-1])]       # (the first calls of a process: templates, allocations)
if not calls:
    sys.exit('no calls found')


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


print(f'{len(calls)} calls of {len(calls[0])} kernels; microseconds, medians')
print(f'{"kernel":44s} {"start":>8s} {"dur":>8s} {"gap before":>10s}')
for k in range(len(calls[0])):
    st = med([c[k][1] - c[0][1] for c in calls]) / 1e3
    du = med([c[k][2] - c[k][1] for c in calls]) / 1e3
    gap = med([c[k][1] - c[k - 1][2] for c in calls]) / 1e3 if k else 0.0
    print(f'{calls[0][k][0][:44]:44s} {st:8.1f} {du:8.1f} {gap:10.1f}')
print(f'{"end of the last kernel":44s} {med([c[-1][2] - c[0][1] for c in calls]) / 1e3:8.1f}')
