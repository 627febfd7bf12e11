#!/usr/bin/env python3
"""Print the figures of a bench.py JSON line as a table: headline and every entry of `configs`.   show_bench.py <bench.json>"""
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
for e in d.get('configs', []):
    show(e, e['name'])
