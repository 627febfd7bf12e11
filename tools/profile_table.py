#!/usr/bin/env python3
"""profiles/r03_<config>_{kernel_stats,counters}.csv -> one row per (configuration, kernel): time, waves, vector / scalar / memory
instructions per wave, wave life, share of wave-cycles spent waiting, vector-ALU utilisation and resident waves per SIMD.
    VALU busy = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / (SQ_BUSY_CYCLES / 32 shader-engine instances);  waves per SIMD = SQ_WAVE_CYCLES x 4
    / 1024 / the same busy time (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles, MI355X_MICROARCH.md)."""
import csv
import glob
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
print(f"{'config':10s} {'kernel':34s} {'us':>8s} {'waves':>9s} {'VALU/w':>7s} {'SALU/w':>7s} {'SMEM/w':>7s} {'VMEMrd/w':>8s} {'LDS/w':>6s} {'cycles/w':>9s} {'wait%':>6s} {'VALUbusy%':>9s} {'waves/SIMD':>10s}")
for path in sorted(glob.glob(os.path.join(REPO, 'profiles', os.environ.get('FCPP_ROUND', 'r05') + '_*_counters.csv'))):
    cfg = os.path.basename(path)[4:-13]
    rows = {(r['Kernel'], r['Counter']): float(r['MeanPerDispatch']) for r in csv.DictReader(open(path))}
    st = {r['Name']: float(r['AverageNs']) / 1e3 for r in csv.DictReader(open(path.replace('_counters', '_kernel_stats')))}
    for k in sorted({k for k, _ in rows}):
        g = lambda c: rows.get((k, c), 0.0)
        w = g('SQ_WAVES')
        if w < 64:
            continue
        busy = max(g('SQ_BUSY_CYCLES') / 32, 1.0)
        print(f"{cfg:10s} {k:34s} {st.get(k, 0):8.1f} {w:9.0f} {g('SQ_INSTS_VALU') / w:7.0f} {g('SQ_INSTS_SALU') / w:7.0f} {g('SQ_INSTS_SMEM') / w:7.1f} "
              f"{g('SQ_INSTS_VMEM_RD') / w:8.1f} {g('SQ_INSTS_LDS') / w:6.1f} {g('SQ_WAVE_CYCLES') / w * 4:9.0f} {100 * g('SQ_WAIT_ANY') / max(g('SQ_WAVE_CYCLES'), 1):6.0f} "
              f"{100 * g('SQ_ACTIVE_INST_VALU') * 4 / 1024 / busy:9.0f} {g('SQ_WAVE_CYCLES') * 4 / 1024 / busy:10.1f}")
