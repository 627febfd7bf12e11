#!/usr/bin/env python3
"""fcpp_validate (include/fcpp.h) on a caller's paths: the cfg3 path (5000 x 2000 m field, 32 obstacles, 0.05 m: 6.3e7 points) planned once,
then validated against the field polygon and the obstacle polygons as a caller's own path -- points/s, with the call's own synchronisation
and the copy of the statistics back (what the entry point does).  One JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from field_coverage_path_planning_amd import engine as E, workloads as WL  # noqa: E402

torch.cuda.set_stream(torch.cuda.Stream())
(L3, H3), obst = WL.cfg3_field()
veh = E.make_vehicle()
b = E.Batch(E.FieldTable.from_specs([E.FieldSpec(field_length=L3, field_width=H3, obstacles=obst)]), veh, E.make_options(1, 0.05))
res = b.run()
torch.cuda.synchronize()
x, y, v = res.x, res.y, res.v
field = [[(0.0, 0.0), (L3, 0.0), (L3, H3), (0.0, H3)]]
offs = np.array([0, x.numel()], dtype=np.int64)
flags, st = E.validate(x, y, v, veh, field_polygons=field, obstacles=obst, offsets=offs)
same = int((flags.to(torch.int64) & 0x70).ne(res.flagseg.to(torch.int64) & 0x70).sum())      # a_lat / outside / obstacle bits of the planner's own flags
ts = []
for _ in range(7):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    E.validate(x, y, v, veh, field_polygons=field, obstacles=obst, offsets=offs)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
dt = sorted(ts)[3]
print(json.dumps({'metric': 'fcpp_validate points/s (cfg3 path as a caller\'s path: field polygon + 32 obstacle polygons)', 'points': int(x.numel()),
                  'ms': round(dt * 1e3, 3), 'value': x.numel() / dt, 'bytes_per_point': 28, 'GB_per_s': 28 * x.numel() / dt / 1e9,
                  'flag_words_differing_from_the_planner': same, 'n_in_obstacle': int(st['n_in_obstacle'][0]), 'n_outside': int(st['n_outside'][0])}))
