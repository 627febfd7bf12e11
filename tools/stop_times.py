#!/usr/bin/env python3
"""Durations of the dispatches of one kernel from a rocprofv3 --kernel-trace run of tools/diag_sparse_stop.py, three per cut, in launch order:
   stop_times.py <dir> <kernel substring>"""
import csv, glob, sys
d, pat = sys.argv[1], sys.argv[2]
rows = []
for p in glob.glob(f'{d}/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if pat in r['Kernel_Name']:
            rows.append((int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
rows.sort()
cuts = ['kernel entry only (-2)', 'pack read, no tile, no reduction (-5)', 'no tile, with reduction / span (-1)', 'complete tiles, no reduction / span (-4)', 'first point (-3)'] + \
       [f'after section {k}' for k in range(9)] + ['complete (99)']
durs = [u for _, u in rows]
# (the first dispatches are the batch's own warm-up: the last 3 x len(cuts) are the cuts)
durs = durs[-3 * len(cuts):]
for i, c in enumerate(cuts):
    g = durs[3 * i:3 * i + 3]
    print(f'{c:48s} ' + ' '.join(f'{u:7.1f}' for u in g) + ' us')
