#!/usr/bin/env python3
"""Diagnostic: the step of the headline batch (and of cfg1 x 16384) on a batch that is set up -- regions of K steps between two synchronisations,
median -- for the library FCPP_LIBRARY names (A/B of kernel variants: one call per variant, alternating, on the same box).
    python tools/step_probe.py [tag]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from field_coverage_path_planning_amd import engine as E

tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get('FCPP_LIBRARY', 'libfcpp.so'))
E.get_context().reserve_outputs(lane_gib=24.0, pitch_gib=24.0)
veh, opt = E.make_vehicle(), E.make_options()
out = []
for n, K, R in ((4096, 200, 15), (16384, 50, 9)):
    table = E.FieldTable.from_rectangles(np.tile(np.array([[500.0, 200.0]]), (n, 1))).to_device()
    b, res = E.Batch.plan(table, veh, opt)
    bufs = (res.x, res.y, res.kappa, res.v, res.flagseg, res.stats_raw)
    for _ in range(20):
        b.run(bufs)
    dts = []
    for _ in range(R):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            b.run(bufs)
        torch.cuda.synchronize()
        dts.append((time.perf_counter() - t0) / K * 1e6)
    out.append(f'{n} fields: step {np.median(dts):.2f} us (min {min(dts):.2f})')
    del res, bufs
    b.close()
print(f'{tag}: ' + ' | '.join(out), flush=True)
