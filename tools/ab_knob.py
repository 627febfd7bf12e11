#!/usr/bin/env python3
"""A/B of launch-time tuning knobs on IDENTICAL memory: one process, one batch, one set of output arrays; the knob (an environment
variable the launchers read at every launch, csrc/fcpp_devfn.h tune_int) is flipped between interleaved groups of runs.
    ab_knob.py <workload> <stage> [create:]VAR=V1,V2,...        workload: cfg1 | cfg1_dense | cfg2_ref | cfg2_0.5 | cfg2_0.1 | cfg3 | cfg5
(create:VAR = a knob the tiler reads when the batch is created: one batch per value, same arrays)
Prints min / median of the stage's per-launch time (HIP events of the dispatch) and of the whole step per value."""
import os
os.environ.setdefault('FCPP_TUNE', '1')      # the knobs are live only in a process started with FCPP_TUNE=1 (fcpp_device.h)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from field_coverage_path_planning_amd import engine as E  # noqa: E402
from field_coverage_path_planning_amd import workloads as WL  # noqa: E402

wl, stage, knob = sys.argv[1], sys.argv[2], sys.argv[3]
var, vals = knob.split('=', 1)
vals = vals.split(',')
veh = E.make_vehicle()
if wl == 'cfg1':
    specs, opt = WL.specs_from_lh(E, WL.cfg1_batch(4096)), E.make_options()
elif wl == 'cfg1_dense':
    specs, opt = WL.specs_from_lh(E, WL.cfg1_batch(4096)), E.make_options(1, 0.1)
elif wl == 'cfg5':
    specs, opt = WL.specs_from_vertices(E, WL.cfg5_parallelograms()), E.make_options()
elif wl == 'cfg3':
    (L3, H3), obst = WL.cfg3_field()
    specs, opt = [E.FieldSpec(field_length=L3, field_width=H3, obstacles=obst)], E.make_options(1, 0.05)
elif wl.startswith('cfg2'):
    sp = {'cfg2_ref': None, 'cfg2_0.5': 0.5, 'cfg2_0.1': 0.1, 'cfg2_0.25': 0.25, 'cfg2_1.0': 1.0}[wl]
    specs, opt = WL.specs_from_lh(E, WL.cfg2_rectangles()), (E.make_options() if sp is None else E.make_options(1, sp))
else:
    raise SystemExit(__doc__)
create = var.startswith('create:')          # a knob read when the batch is created (the tiler's): one batch per value
if create:
    var = var[7:]
    batches = {}
    for v in vals:
        os.environ[var] = v
        batches[v] = E.Batch(specs, veh, opt)
    del os.environ[var]
    b = batches[vals[0]]
else:
    b = E.Batch(specs, veh, opt)
bufs = b.alloc()
res = {v: [] for v in vals}
step = {v: [] for v in vals}
for rnd in range(8):
    for v in vals:
        if create:
            b = batches[v]
        else:
            os.environ[var] = v
        b.run(bufs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            b.run(bufs)
        torch.cuda.synchronize()
        step[v].append((time.perf_counter() - t0) / 10 * 1e3)
        b.set_profiling(True)
        for _ in range(5):
            b.run(bufs)
        t, _ = b.stage_times()
        b.set_profiling(False)
        res[v].append(t[stage])
for v in vals:
    print(f'{var}={v:8s} {stage} min {min(res[v]) * 1e3:9.1f} median {float(np.median(res[v])) * 1e3:9.1f} us | step min {min(step[v]):.4f} median {float(np.median(step[v])):.4f} ms')
