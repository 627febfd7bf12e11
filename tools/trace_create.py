#!/usr/bin/env python3
"""A plan call end to end, repeated: wall time of create / alloc / run / sync, median over the repetitions after the first.
Usage: python tools/trace_create.py [reps] [headline|cfg2_ref|cfg2_0.5|cfg2_0.1|cfg3|cfg3_avoid|cfg5|cfg1_clothoid|cfg1_clothoid_dense]   (FCPP_NO_PIN=1: pageable field records)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from field_coverage_path_planning_amd import engine as E
from field_coverage_path_planning_amd import workloads as WL

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
what = sys.argv[2] if len(sys.argv) > 2 else 'headline'
opt = E.make_options()
if what in ('cfg2_ref', 'cfg2_0.5', 'cfg2_0.1'):
    table = E.FieldTable.from_rectangles(WL.cfg2_rectangles())
    if what != 'cfg2_ref':
        opt = E.make_options(1, float(what[5:]))
elif what in ('cfg3', 'cfg3_avoid'):
    (L_, H_), obst_ = WL.cfg3_field()
    table = E.FieldTable.from_specs([E.FieldSpec(field_length=L_, field_width=H_, obstacles=obst_)])
    opt = E.make_options(1, 0.05, avoid_obstacles=(what == 'cfg3_avoid'))
elif what.startswith('n') and what[1:].isdigit():          # n<k>: k fields of 500 x 200 m (the small-batch rule: FCPP_SMALL_BATCH)
    table = E.FieldTable.from_rectangles(np.tile(np.array([[500.0, 200.0]]), (int(what[1:]), 1)) + np.arange(int(what[1:]))[:, None] * 0.37)
elif what.startswith('x') and what[1:].isdigit():          # x<k>: k copies of the 500 x 200 m field (cfg1_x16384 of bench.py)
    table = E.FieldTable.from_rectangles(np.tile(np.array([[500.0, 200.0]]), (int(what[1:]), 1)))
elif what == 'cfg1_clothoid_dense':
    table = E.FieldTable.from_rectangles(np.tile(np.array([[500.0, 200.0]]), (4096, 1)))
    opt = E.make_options(1, 0.1)
elif what == 'cfg5':
    table = E.FieldTable.from_vertices(WL.cfg5_parallelograms())
else:
    table = E.FieldTable.from_rectangles(np.tile(np.array([[500.0, 200.0]]), (4096, 1)))
    if what == 'cfg1_clothoid':
        opt = E.make_options(1, 0.0)
if os.environ.get('FCPP_RECORDS') == 'device':          # (the records in device memory: FieldTable.to_device())
    table.to_device()
elif not os.environ.get('FCPP_NO_PIN'):
    table.pin()
if os.environ.get('FCPP_ARENA', '1') != '0':
    E.get_context().reserve_outputs(lane_gib=24.0, pitch_gib=24.0)
veh = E.make_vehicle()
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
rows = []
batch = None
one_call = os.environ.get('FCPP_ONE_CALL', '1') != '0'
closes = []
for r in range(reps):
    if batch is not None:
        tc0 = time.perf_counter()
        batch.close()
        closes.append((time.perf_counter() - tc0) * 1e3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if one_call:
        batch, res = E.Batch.plan(table, veh, opt)
        t1 = t2 = t3 = time.perf_counter()
    else:
        batch = E.Batch(table, veh, opt)
        t1 = time.perf_counter()
        bufs = batch.alloc()
        t2 = time.perf_counter()
        batch.run(bufs, mode=1)
        t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    res = bufs = None
    rows.append([(t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3])
a = np.array(rows[1:])
med = np.median(a, axis=0)
print(("one call " if one_call else "three calls ") + "%s: median ms: create %.4f alloc %.4f run_enqueue %.4f sync %.4f total %.4f" % ((what,) + tuple(med)))
print("setup_times", batch.setup_times(), 'points', batch.total_points, 'close() of the previous batch, median ms: %.4f' % float(np.median(closes)))
