#!/usr/bin/env python3
"""The section table of the wave-tile kernels: vector / scalar / LDS instructions per wavefront of every section of sparse_tile2
(fcpp_sparse2_fn.h), from a `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES` run of tools/diag_sparse_stop.py on the
-DFCPP_DIAG_SPARSE build: that tool cuts the tile function off after each section in turn, three steps per cut, so the difference between
consecutive groups of dispatches is the section's instructions.
    sparse_sections.py <rocprof dir> <kernel substring> [points per step]"""
import csv
import glob
import sys
from collections import OrderedDict

d, pat = sys.argv[1], sys.argv[2]
points = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
rows = OrderedDict()
for p in glob.glob(f'{d}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if pat not in r['Kernel_Name']:
            continue
        rows.setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
ids = sorted(rows)
cuts = [-2, -5, -1, -4, -3] + list(range(9)) + [99]
names = {-2: 'kernel entry only', -5: 'pack / records read, no tile, no reduction', -1: 'the same + the field\'s reduction', -4: 'complete tiles, no reduction',
         -3: 'first point of every lane', 0: 'second point (section 0: points)', 1: 'chords, lengths', 2: 'curvature (atan2)', 3: 'clamp, u0', 4: 'sweeps',
         5: 'final speed (sqrt)', 6: 'geofence, obstacles', 7: 'metrics', 8: 'stores, counts', 99: 'complete (with the reduction / span)'}
per = len(ids) // len(cuts)
assert per >= 1, (len(ids), 'dispatches')
ids = ids[len(ids) - per * len(cuts):]          # (the warm-up step, if any, comes first)
tab = []
for k, c in enumerate(cuts):
    grp = [rows[i] for i in ids[k * per:(k + 1) * per]]
    w = sum(g['SQ_WAVES'] for g in grp) / len(grp)
    tab.append((c, w, *[sum(g.get(n, 0.0) for g in grp) / len(grp) for n in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS')]))
print(f'{pat}: {per} dispatches per cut, {tab[-1][1]:.0f} wavefronts per launch')
print(f'{"cut after":52s} {"VALU/wave":>10s} {"SALU/wave":>10s} {"LDS/wave":>9s}   {"+VALU":>7s} {"+SALU":>7s}')
order = [-2, -5, -1, -3, 0, 1, 2, 3, 4, 5, 6, 7, 8]
by = {t[0]: t for t in tab}
prev = None
for c in order + [-4, 99]:
    t = by[c]
    v, s, l = t[2] / t[1], t[3] / t[1], t[4] / t[1]
    if c in (-4, 99) or prev is None:
        print(f'{names[c]:52s} {v:10.0f} {s:10.0f} {l:9.0f}')
    else:
        print(f'{names[c]:52s} {v:10.0f} {s:10.0f} {l:9.0f}   {v - prev[0]:7.0f} {s - prev[1]:7.0f}')
    if c not in (-4, 99):
        prev = (v, s)
if points:
    t = by[99]
    print(f'complete: {t[2] / points:.2f} vector instructions (wavefront-wide) per 64 output points = {t[2] * 64 / points / 64:.2f}; per output point x 64 lanes: {t[2] * 64 / points:.1f}')
