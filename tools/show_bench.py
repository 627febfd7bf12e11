#!/usr/bin/env python3
"""Print the figures of a bench.py JSON line (or of bench_detail.json) as a table: headline and every entry of `configs`.
    show_bench.py <bench.json>"""
import json
import sys

d = json.load(open(sys.argv[1]))


def show(e, name):
    r = e.get('roofline') or {}
    ms = e.get('ms_per_step', e.get('ms_total', e.get('ms_501_evaluations', 0.0))) or 0.0
    kp = r.get('all_kernels_points') or {}
    ks = {k: (round(v * 1e3, 1), kp.get(k)) for k, v in (r.get('all_kernels_ms') or {}).items()}
    e2e = e.get('value_end_to_end')
    print(f"{name:22s} ms={ms:9.4f} value={e['value']:.3e} dom={r.get('kernel')} frac={r.get('frac') or 0:.3f} "
          f"step_frac={r.get('step_frac') or 0:.3f} end_to_end={(f'{e2e:.3e}' if e2e else '-')} create_ms={(e.get('setup_ms') or {}).get('create', '-')} "
          f"cpu={(e.get('cpu_baseline') or {}).get('value')}")
    if ks:
        print(' ' * 22, 'kernels (us, points):', ks)


show(d, 'HEADLINE')
for k in ('value_step', 'ms_step', 'value_sustained', 'ms_sustained', 'value_clothoid'):
    if d.get(k) is not None:
        print(f"{'':22s} {k} = {d[k]:.5g}")
cfgs = d.get('configs', [])
if isinstance(cfgs, dict):                    # the compact line: a row per configuration under `columns`
    cols = cfgs.get('columns', [])
    print(f"{'':22s} " + ' '.join(f'{c:>18s}' for c in cols))
    for name, row in cfgs.items():
        if name != 'columns':
            print(f'{name:22s} ' + ' '.join(f'{(v if v is not None else "-"):>18.5g}' if isinstance(v, (int, float)) else f'{str(v if v is not None else "-"):>18s}' for v in row))
else:
    for e in cfgs:
        show(e, e['name'])
