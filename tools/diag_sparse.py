#!/usr/bin/env python3
"""Diagnostic only: where the cycles of a wave tile go.  Needs the -DFCPP_DIAG_SPARSE build (`make -C field_coverage_path_planning_amd/csrc
diag-sparse` -> build/libfcpp_diag_sparse.so), which records shader-clock cycles per section of sparse_tile2 (fcpp_sparse2_fn.h) for every
wave tile.  A wavefront shares its SIMD with the others resident there, so a section's cycles are its share of the wave's life, not
its instruction count.  Usage: FCPP_LIBRARY=build/libfcpp_diag_sparse.so python tools/diag_sparse.py [headline|cfg2_ref|cfg5]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from field_coverage_path_planning_amd import _lib, engine as E, workloads as WL  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'headline'
table = {'headline': lambda: E.FieldTable.from_rectangles(WL.cfg1_batch(4096)),
         'cfg2_ref': lambda: E.FieldTable.from_rectangles(WL.cfg2_rectangles()),
         'cfg5': lambda: E.FieldTable.from_vertices(WL.cfg5_parallelograms())}[which]()
lib = _lib.load()
fn = lib.fcpp_diag_sparse
fn.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
fn.restype = ctypes.c_longlong
torch.cuda.set_stream(torch.cuda.Stream())
b = E.Batch(table, E.make_vehicle(), E.make_options())
bufs = b.alloc()
rows = np.zeros((1 << 16, 16), dtype=np.uint32)
for _ in range(3):
    b.run(bufs)
torch.cuda.synchronize()
assert fn(rows.ctypes.data, rows.shape[0]) >= 0
b.run(bufs)
torch.cuda.synchronize()
n = fn(rows.ctypes.data, rows.shape[0])
assert n > 0
r = rows[:min(n, rows.shape[0]), :13].astype(np.float64)
names = ['points: decode, primitives, templates', 'chords, lengths', 'curvature (atan2)', 'clamp, u0', 'sweeps', 'final speed (sqrt)',
         'geofence, obstacles', 'metrics', 'stores, counts', '  (before points) wave-tile record arrives', '  (before points) field record arrives',
         '  (in stores) output pointers arrive', '  (in stores) stores issued']
total = r[:, :9].sum(axis=1) + r[:, 9] + r[:, 10]
print(which, 'wave tiles in one step', n, 'cycles per wave tile: mean', round(total.mean()), 'median', round(float(np.median(total))))
for k, nm in enumerate(names):
    print(f'  {nm:40s} mean {r[:, k].mean():8.0f}  median {np.median(r[:, k]):8.0f} cycles  {100.0 * r[:, k].sum() / total.sum():5.1f} %')

# the schedule: wave tiles in the order they started (cols 13, 14 = start / end of the tile in units of 16 cycles)
full = rows[:min(n, rows.shape[0])].astype(np.int64)
t0 = full[:, 13].min()
start, end = (full[:, 13] - t0) * 16, (full[:, 14] - t0) * 16
order = np.argsort(start)
print('kernel span (first start to last end)', int(end.max()), 'cycles')
nb = 8
for b in range(nb):
    idx = order[b * len(order) // nb:(b + 1) * len(order) // nb]
    print(f'  tiles {b}/{nb} by start: start {int(start[idx].min()):7d}..{int(start[idx].max()):7d}  life mean {int((end[idx] - start[idx]).mean()):6d}'
          f'  points {int(r[idx, 0].mean()):6d}  stores issued {int(r[idx, 12].mean()):6d}  compute {int(r[idx, 1:8].sum(axis=1).mean()):6d}')
