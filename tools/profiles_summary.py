#!/usr/bin/env python3
"""gpurun_out/prof_r02/<config>/ (tools/collect_profiles.sh) -> profiles/: per configuration
    r02_<config>_kernel_stats.csv : the rocprofv3 --kernel-trace --stats summary rows of the fcpp kernels (calls, total / average ns)
    r02_<config>_counters.csv     : per kernel and counter, the mean value per launch (WRITE_SIZE, FETCH_SIZE, SQ_*)
and profiles/traffic.json: HBM bytes per launch = WRITE_SIZE KiB x 1024 + 2 x FETCH_SIZE KiB x 1024 (the gfx950 FETCH_SIZE correction
of MI355X_MICROARCH.md, section HBM), keyed '<bench stage name>|<config>' as bench.py looks it up."""
import collections
import csv
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, 'gpurun_out', 'prof_r02')
DST = os.path.join(REPO, 'profiles')
STAGE = {'k_plan_quiet<16, true>': 'k_plan_quiet_spans', 'k_plan_quiet<14, false>': 'k_plan_quiet', 'k_plan_sparse': 'k_plan_sparse',
         'k_plan_fused<3>': 'k_plan_fused', 'k_reduce_stats': 'k_reduce_stats'}


def short(name):
    return name.split('(')[0].replace('void ', '').replace('fcpp::', '').strip()


traffic = {}
notes = {}
for cdir in sorted(glob.glob(os.path.join(SRC, '*'))):
    cfg = os.path.basename(cdir)
    stats = glob.glob(os.path.join(cdir, 'stats', '**', '*kernel_stats.csv'), recursive=True)
    if stats:
        rows = [r for r in csv.DictReader(open(stats[0])) if 'fcpp' in r['Name']]
        with open(os.path.join(DST, f'r02_{cfg}_kernel_stats.csv'), 'w', newline='') as fh:
            w = csv.writer(fh)
            w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev'])
            for r in rows:
                w.writerow([short(r['Name']), r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs'], r['StdDev']])
    acc = collections.defaultdict(lambda: [0, 0.0])
    for path in glob.glob(os.path.join(cdir, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(path)):
            if 'fcpp' in r.get('Kernel_Name', ''):
                key = (short(r['Kernel_Name']), r['Counter_Name'])
                acc[key][0] += 1
                acc[key][1] += float(r['Counter_Value'])
    if acc:
        with open(os.path.join(DST, f'r02_{cfg}_counters.csv'), 'w', newline='') as fh:
            w = csv.writer(fh)
            w.writerow(['Kernel', 'Counter', 'Dispatches', 'MeanPerDispatch'])
            for (k, c), (n, s) in sorted(acc.items()):
                w.writerow([k, c, n, f'{s / n:.1f}'])
        for k in sorted({k for k, _ in acc}):
            if k in STAGE and (k, 'WRITE_SIZE') in acc and (k, 'FETCH_SIZE') in acc:
                wkb = acc[(k, 'WRITE_SIZE')][1] / acc[(k, 'WRITE_SIZE')][0]
                fkb = acc[(k, 'FETCH_SIZE')][1] / acc[(k, 'FETCH_SIZE')][0]
                traffic[f'{STAGE[k]}|{cfg}'] = wkb * 1024 + 2 * fkb * 1024
                notes[f'{STAGE[k]}|{cfg}'] = {'WRITE_SIZE_KB_per_launch': wkb, 'FETCH_SIZE_KB_per_launch': fkb}
    print(cfg, 'stats' if stats else 'NO stats', len(acc), 'counter rows')
traffic['_notes'] = {
    'source': 'rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE, separate passes, program directly after `--` (tools/collect_profiles.sh), round 2',
    'units': 'bytes per kernel launch = WRITE_SIZE*1024 + 2*FETCH_SIZE*1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md section HBM)',
    'detail': notes}
json.dump(traffic, open(os.path.join(DST, 'traffic.json'), 'w'), indent=1)
