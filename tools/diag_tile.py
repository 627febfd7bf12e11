#!/usr/bin/env python3
"""Diagnostic only: where the time of the device tiler's counting pass goes for one field of the headline batch.  Needs the
-DFCPP_DIAG_TILE build (build/libfcpp_diag_tile.so, see HISTORY.md) selected with FCPP_LIBRARY; prints the phase stamps (10 ns ticks)
of field 1000's wavefront: entry, field read, staging, then per wave tile: window + back halo + candidates + record."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from field_coverage_path_planning_amd import engine as E, _lib as L

LH = np.tile(np.array([[500.0, 200.0]]), (4096, 1))
table = E.FieldTable.from_rectangles(LH)
veh, opt = E.make_vehicle(), E.make_options()
for r in range(5):
    b = E.Batch(table, veh, opt)
    torch.cuda.synchronize()
    out = (C.c_uint64 * 40)()
    rc = L.load().fcpp_diag_tile_stamps(out)
    st = list(out)
    out2 = (C.c_uint64 * 40)()
    L.load().fcpp_diag_fill_stamps(out2)
    fs = list(out2)
    out3 = (C.c_uint64 * 16)()
    L.load().fcpp_diag_plan_stamps(out3)
    ps = list(out3)
    b.close()
names = {0: 'entry', 1: 'field read', 2: 'staged', 36: 'cut done', 37: 'decisions', 38: 'counts written'}
for k in range(4):
    names[3 + 4 * k] = f'tile {k}: window'; names[4 + 4 * k] = f'tile {k}: back halo'; names[5 + 4 * k] = f'tile {k}: candidates'; names[6 + 4 * k] = f'tile {k}: record'
names.update({10: 'closed cut: begin', 11: 'closed cut: primitive ends', 12: 'closed cut: primitive records', 13: 'closed cut: tiles cut', 14: 'closed cut: tile records'})
names.update({15: 'closed cut: loads issued', 16: 'closed cut: templates staged', 17: 'closed cut: first points'})
names.update({24: 'round 2: begin', 25: 'round 2: search', 26: 'round 2: point', 27: 'round 2: distance', 28: 'round 2: inside'})
t0, prev = st[0], st[0]
for k in sorted(names, key=lambda k: st[k]):
    if st[k]:
        print(f'{names[k]:24s} +{(st[k] - prev) * 10:6d} ns   at {(st[k] - t0) * 10:6d} ns')
        prev = st[k]

print('fill pass of the same field:')
fn = {0: 'entry', 1: 'field read', 2: 'cut / kept tiles / span', 3: 'field + primitives copied', 4: 'entries, work record, connectors', 5: 'pack', 6: 'junction + span statistics', 7: 'slots, totals, info'}
prev = fs[0]
for k in sorted(fn):
    print(f'{fn[k]:36s} +{(fs[k] - prev) * 10:6d} ns   at {(fs[k] - fs[0]) * 10:6d} ns')
    prev = fs[k]

print('the planner (k_plan_fields16) of the same field:')
pn = {0: 'entry', 1: 'field read, convexity', 2: 'bounds, corner angles, shape, start corner', 3: 'mitres, inset by the headland width', 4: 'rotation (atan2, sincos), frame', 5: 'layer 1 sizes and record',
      6: 'geofence half-planes', 7: 'loops 0-3: inset, corner exits, reverse fill', 8: 'failure vote, prefix over the row', 9: 'primitives written', 10: 'loop over', 11: 'record finished'}
prev = ps[0]
for k in sorted(pn):
    if ps[k]:
        print(f'{pn[k]:48s} +{(ps[k] - prev) * 10:6d} ns   at {(ps[k] - ps[0]) * 10:6d} ns')
        prev = ps[k]
