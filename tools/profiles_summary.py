#!/usr/bin/env python3
"""gpurun_out/prof_<round>/<config>/ (tools/collect_profiles.sh) -> profiles/: per configuration
    <round>_<config>_kernel_stats.csv : the rocprofv3 --kernel-trace --stats summary rows of the fcpp kernels (calls, total / average ns)
    <round>_<config>_counters.csv     : per kernel and counter, the mean value per launch (WRITE_SIZE, FETCH_SIZE, SQ_*)
profiles/valu.json (vector instructions per launch, SQ_INSTS_VALU) and profiles/traffic.json: HBM bytes per launch = WRITE_SIZE KiB x 1024 + 2 x FETCH_SIZE KiB x 1024 (the gfx950 FETCH_SIZE correction
of MI355X_MICROARCH.md, section HBM), keyed '<bench stage name>|<config>' as bench.py looks it up."""
import collections
import csv
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get('FCPP_ROUND', 'r05')
SRC = os.path.join(REPO, 'gpurun_out', 'prof_' + ROUND)
DST = os.path.join(REPO, 'profiles')
STAGE_PREFIX = (('k_plan_quiet<16', 'k_plan_quiet_spans'), ('k_plan_quiet<14', 'k_plan_quiet'), ('k_plan_sparse_fields', 'k_plan_sparse_fields'), ('k_plan_sparse', 'k_plan_sparse'),
                ('k_plan_fused<', 'k_plan_fused'), ('k_reduce_stats', 'k_reduce_stats'))


def stage_of(kernel):
    for prefix, stage in STAGE_PREFIX:
        if kernel.startswith(prefix):
            return stage
    return None


def short(name):
    return name.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').replace('fcpp::', '').strip()


traffic = {}
notes = {}
valu = {}
for cdir in sorted(glob.glob(os.path.join(SRC, '*'))):
    cfg = os.path.basename(cdir)
    # kernel statistics over the LAST `timed_steps` dispatches of every kernel of the trace (tools/prof_cfg.py prints the count): the
    # warm-up step and the placement calibration of cfg2_0.1 / cfg5 come before them.  Same columns as rocprofv3's own *_kernel_stats.csv.
    stats = glob.glob(os.path.join(cdir, 'stats', '**', '*kernel_trace.csv'), recursive=True)
    if stats:
        steps = 10
        log = os.path.join(cdir, 'stats.log')
        if os.path.exists(log):
            for line in open(log, errors='replace'):
                if 'timed_steps' in line:
                    steps = int(line.split('timed_steps')[1].split()[0])
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(stats[0])):
            if 'fcpp' in r['Kernel_Name']:
                per[short(r['Kernel_Name'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        sel = {k: (v[-steps:] if len(v) > steps else v) for k, v in per.items()}
        grand = sum(sum(v) for v in sel.values()) or 1
        with open(os.path.join(DST, f'{ROUND}_{cfg}_kernel_stats.csv'), 'w', newline='') as fh:
            w = csv.writer(fh)
            w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev'])
            for k, v in sorted(sel.items(), key=lambda kv: -sum(kv[1])):
                mean = sum(v) / len(v)
                sd = (sum((x - mean) ** 2 for x in v) / len(v)) ** 0.5
                w.writerow([k, len(v), sum(v), f'{mean:.1f}', f'{100 * sum(v) / grand:.2f}', min(v), max(v), f'{sd:.1f}'])
    acc = collections.defaultdict(lambda: [0, 0.0])
    for path in glob.glob(os.path.join(cdir, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(path)):
            if 'fcpp' in r.get('Kernel_Name', ''):
                key = (short(r['Kernel_Name']), r['Counter_Name'])
                acc[key][0] += 1
                acc[key][1] += float(r['Counter_Value'])
    if acc:
        with open(os.path.join(DST, f'{ROUND}_{cfg}_counters.csv'), 'w', newline='') as fh:
            w = csv.writer(fh)
            w.writerow(['Kernel', 'Counter', 'Dispatches', 'MeanPerDispatch'])
            for (k, c), (n, s) in sorted(acc.items()):
                w.writerow([k, c, n, f'{s / n:.1f}'])
        for k in sorted({k for k, _ in acc}):
            if stage_of(k) and (k, 'WRITE_SIZE') in acc and (k, 'FETCH_SIZE') in acc:
                wkb = acc[(k, 'WRITE_SIZE')][1] / acc[(k, 'WRITE_SIZE')][0]
                fkb = acc[(k, 'FETCH_SIZE')][1] / acc[(k, 'FETCH_SIZE')][0]
                key = f'{stage_of(k)}|{cfg}'
                if key in traffic:            # (a stage launched in several classes, e.g. k_reduce_stats<8> and <64>: the first one is what bench.py times)
                    continue
                traffic[key] = wkb * 1024 + 2 * fkb * 1024
                notes[key] = {'kernel': k, 'WRITE_SIZE_KB_per_launch': wkb, 'FETCH_SIZE_KB_per_launch': fkb}
        # vector instructions per launch (all wavefronts): bench.py's valu_frac = this x 4 cycles / (1024 SIMDs x clock x its own kernel time)
        for k in sorted({k for k, _ in acc}):
            if stage_of(k) and (k, 'SQ_INSTS_VALU') in acc and f'{stage_of(k)}|{cfg}' not in valu:
                valu[f'{stage_of(k)}|{cfg}'] = acc[(k, 'SQ_INSTS_VALU')][1] / acc[(k, 'SQ_INSTS_VALU')][0]
    print(cfg, 'stats' if stats else 'NO stats', len(acc), 'counter rows')
traffic['_notes'] = {
    'source': 'rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE, separate passes, program directly after `--` (tools/collect_profiles.sh), ' + ROUND,
    'units': 'bytes per kernel launch = WRITE_SIZE*1024 + 2*FETCH_SIZE*1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md section HBM)',
    'detail': notes}
json.dump(traffic, open(os.path.join(DST, 'traffic.json'), 'w'), indent=1)
valu['_notes'] = {'source': 'rocprofv3 --pmc SQ_INSTS_VALU ... (tools/collect_profiles.sh), ' + ROUND, 'units': 'vector instructions per kernel launch, summed over its wavefronts'}
json.dump(valu, open(os.path.join(DST, 'valu.json'), 'w'), indent=1)
