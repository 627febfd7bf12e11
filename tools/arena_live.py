#!/usr/bin/env python3
"""Three batches alive at once in the context's output arena (cfg3, cfg5, cfg2 at 0.5 m): each one's step alone -- its arrays the only
live allocation of the arena -- and with all three alive, plus the device memory the process holds.  Usage: python tools/arena_live.py"""
import gc
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from field_coverage_path_planning_amd import engine as E, workloads as WL  # noqa: E402

torch.cuda.set_stream(torch.cuda.Stream())
ctx = E.get_context()
free0 = torch.cuda.mem_get_info()[0]
t0 = time.perf_counter()
ctx.reserve_outputs(24.0, 24.0)
reserve_ms = (time.perf_counter() - t0) * 1e3
veh = E.make_vehicle()
(L3, H3), obst = WL.cfg3_field()
mk = {'cfg3': lambda: E.Batch(E.FieldTable.from_specs([E.FieldSpec(field_length=L3, field_width=H3, obstacles=obst)]), veh, E.make_options(1, 0.05)),
      'cfg5': lambda: E.Batch(E.FieldTable.from_vertices(WL.cfg5_parallelograms()), veh, E.make_options()),
      'cfg2_0.5': lambda: E.Batch(E.FieldTable.from_rectangles(WL.cfg2_rectangles()), veh, E.make_options(1, 0.5))}
batches = {k: f() for k, f in mk.items()}


def ms(b, bufs, n=20):
    for _ in range(3):
        b.run(bufs)
    torch.cuda.synchronize()
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n):
            b.run(bufs)
        torch.cuda.synchronize()
        t.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(t)[2]


out = {'reserve_ms': round(reserve_ms, 1), 'arena': ctx.outputs_info()}
solo = {}
for k, b in batches.items():
    bufs = b.alloc()
    assert b.layout['layout'] == 'arena', b.layout
    solo[k] = ms(b, bufs)
    del bufs
    gc.collect()
assert ctx.outputs_info()[2] == 0
held = {k: b.alloc() for k, b in batches.items()}
together = {k: ms(batches[k], held[k]) for k in batches}
out['live_bytes_per_lane'] = ctx.outputs_info()[2]
out['arrays_GB'] = {k: round(36 * b.total_points / 1e9, 2) for k, b in batches.items()}
out['ms_solo'] = {k: round(v, 4) for k, v in solo.items()}
out['ms_all_alive'] = {k: round(v, 4) for k, v in together.items()}
out['ratio'] = {k: round(together[k] / solo[k], 3) for k in batches}
out['device_GiB_held_by_process'] = round((free0 - torch.cuda.mem_get_info()[0]) / 2**30, 1)
print(json.dumps(out))
