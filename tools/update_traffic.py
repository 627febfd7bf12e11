#!/usr/bin/env python3
"""profiles/traffic.json from two rocprofv3 --pmc passes (WRITE_SIZE, FETCH_SIZE) of tools/prof_run.py:
    update_traffic.py <write_counter_collection.csv> <fetch_counter_collection.csv> [fields spacing turn]
HBM bytes per launch = WRITE_SIZE KiB x 1024 + 2 x FETCH_SIZE KiB x 1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md, HBM)."""
import collections
import csv
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
wcsv, fcsv = sys.argv[1], sys.argv[2]
fields, spacing, turn = (sys.argv[3:6] + ['1024', '0.1', '1'])[:3] if len(sys.argv) > 3 else ('1024', '0.1', '1')


def collect(path, counter):
    """mean per launch of every kernel INSTANCE, then the instances of one kernel (k_plan_quiet<14,..> for straights / turns and
    k_plan_quiet<16,..> for the mixed chunks run once each per step) added up under the kernel's base name"""
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            agg[r['Kernel_Name'].split('(')[0].replace('void ', '').replace('fcpp::', '')].append(float(r['Counter_Value']))
    out = collections.defaultdict(float)
    for k, v in agg.items():
        out[k.split('<')[0]] += sum(v) / len(v)
    return dict(out)


w, f = collect(wcsv, 'WRITE_SIZE'), collect(fcsv, 'FETCH_SIZE')
out, detail = {}, {}
for k in sorted(set(w) | set(f)):
    if not k.startswith('k_'):
        continue
    wb, fb = w.get(k, 0.0) * 1024, 2 * f.get(k, 0.0) * 1024
    out[f'{k}|fields={fields}|spacing={spacing}|turn={turn}'] = wb + fb
    detail[k] = {'WRITE_SIZE_KB_per_launch': w.get(k, 0.0), 'FETCH_SIZE_KB_per_launch': f.get(k, 0.0), 'hbm_bytes_per_launch': wb + fb,
                 'write_bytes': wb, 'fetch_bytes_corrected_x2': fb}
out['_notes'] = {
    'source': 'rocprofv3 --kernel-trace --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes) -- python3 tools/prof_run.py --steps 2, round 1',
    'units': 'bytes per kernel launch; WRITE_SIZE*1024 + 2*FETCH_SIZE*1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md section HBM)',
    'detail': detail}
json.dump(out, open(os.path.join(REPO, 'profiles', 'traffic.json'), 'w'), indent=1)
for k, v in detail.items():
    print(k, f"{v['hbm_bytes_per_launch'] / 1e9:.3f} GB  (write {v['write_bytes'] / 1e9:.3f}, fetch {v['fetch_bytes_corrected_x2'] / 1e9:.3f})")
