#!/usr/bin/env python3
"""Step time of the headline batch (4096 x 500x200 m, reference sampling) with and without per-kernel event timing, and the host
time of one Batch.run call: shows whether bench.py's timed region is disturbed by its own instrumentation."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from field_coverage_path_planning_amd import engine as E, workloads as WL  # noqa: E402

b = E.Batch(WL.specs_from_lh(E, WL.cfg1_batch(4096)), E.make_vehicle(), E.make_options())
bufs = b.alloc()
for prof in (False, True, False, True):
    b.set_profiling(prof)
    for _ in range(20):
        b.run(bufs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        b.run(bufs)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if prof:
        b.stage_times()
    print(f'profiling={prof}: {dt / 200 * 1e6:.1f} us per step, host enqueue {t_host / 200 * 1e6:.1f} us per step')
