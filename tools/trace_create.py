#!/usr/bin/env python3
"""The headline plan call (4096 fields of 500 x 200 m) end to end, repeated: wall time of create / alloc+run+sync, median over the
repetitions after the first.  Usage: python tools/trace_create.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from field_coverage_path_planning_amd import engine as E

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
LH = np.tile(np.array([[500.0, 200.0]]), (4096, 1))
table = E.FieldTable.from_rectangles(LH)
if not os.environ.get('FCPP_NO_PIN'):
    table.pin()
veh, opt = E.make_vehicle(), E.make_options()
rows = []
batch = None
for r in range(reps):
    if batch is not None:
        batch.close()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    batch = E.Batch(table, veh, opt)
    t1 = time.perf_counter()
    bufs = batch.alloc()
    t2 = time.perf_counter()
    batch.run(bufs, mode=1)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    rows.append([(t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3])
a = np.array(rows[1:])
med = np.median(a, axis=0)
print("median ms: create %.4f alloc %.4f run_enqueue %.4f sync %.4f total %.4f" % tuple(med))
print("setup_times", batch.setup_times())
