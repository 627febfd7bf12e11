#!/usr/bin/env python3
"""tools/DESIGN.template.md + one kept bench record -> DESIGN.md: the table of section 6 is filled from profiles/<round>_bench_detail.json
(the detail file bench.py wrote beside the line kept as profiles/<round>_bench.json), profiles/<round>_arena_live.json and
profiles/traffic.json, so that every figure of the document comes from one run.  Usage: python tools/fill_design.py [round]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1] if len(sys.argv) > 1 else 'r05'
d = json.load(open(os.path.join(REPO, 'profiles', f'{ROUND}_bench_detail.json')))
cfg = {e['name']: e for e in d['configs']}


def g(v, n=4):
    return f'{v:.{n}g}'


def sci(v):
    m, e = f'{v:.3e}'.split('e')
    return f'{m}·10^{int(e)}'.replace('^10', '¹⁰').replace('^11', '¹¹').replace('^9', '⁹').replace('^8', '⁸').replace('^7', '⁷').replace('^6', '⁶')


def row(e):
    """plan call ms, fresh points/s, step ms, step frac"""
    rf = e.get('roofline') or {}
    ms_f = e.get('ms_fresh') or (e.get('end_to_end') or {}).get('ms', 0.0)
    v_f = e.get('value_fresh') or (e.get('end_to_end') or {}).get('points_per_s', 0.0)
    return g(ms_f), sci(v_f), g(e['ms_per_step']), g(rf.get('step_frac', 0.0), 3)


sub = {}
hk = d['roofline']['all_kernels_ms']['k_plan_sparse_fields']
sub['H_MS'], sub['H_V'], sub['H_S'], sub['H_F'] = g(d['ms_per_step']), sci(d['value']), g(d['ms_step']), g(d['roofline']['step_frac'], 3)
sub['SUS_MS'], sub['SUS_V'] = g(d.get('ms_sustained', 0.0)), sci(d.get('value_sustained', 0.0))
sub['H_K'] = f'{hk * 1e3:.1f} µs ({d["roofline"]["frac"]:.2f})'
for key, name in (('C', 'cfg1_clothoid'), ('X', 'cfg1_x16384'), ('R', 'cfg2_ref'), ('5', 'cfg2_0.5'), ('1', 'cfg2_0.1'), ('D', 'cfg1_clothoid_dense'), ('3', 'cfg3'),
                  ('A', 'cfg3_avoid'), ('P', 'cfg5')):
    sub[key + '_MS'], sub[key + '_V'], sub[key + '_S'], sub[key + '_F'] = row(cfg[name])
g4 = cfg['cfg4']
sub['G_MS'], sub['G_V'] = g(g4.get('ms_total', g4.get('ms_per_step'))), sci(g4['value'])
cb = d['cpu_baseline']
sub['CPU'] = f'{sci(cb["value"])} ({sci(cb["single_core_value"])} on one thread)'
sub['PYL'], sub['NPY'] = sci(cb['python_loops_value']), sci(cb['numpy_value'])
tj = json.load(open(os.path.join(REPO, 'profiles', 'traffic.json')))
pts = {'k_plan_sparse_fields|cfg1': 6926336, 'k_plan_quiet_spans|cfg5': 240011750, 'k_plan_quiet|cfg2_0.1': None}
tr = []
for k, p in pts.items():
    kern, c = k.split('|')
    if p is None:
        p = cfg[c]['roofline']['all_kernels_points'][kern]
    tr.append(f'{kern} on {c} ×{tj[k] / (36 * p):.3f}')
sub['TRAFFIC'] = ', '.join(tr)
s = open(os.path.join(REPO, 'tools', 'DESIGN.template.md')).read()
for k, v in sub.items():
    s = s.replace(f'@{k}@', v)
assert '@' not in s.replace('@ 2025', ''), [w for w in s.split() if w.startswith('@')]
open(os.path.join(REPO, 'DESIGN.md'), 'w').write(s)
print(len(s.encode()), 'bytes')
