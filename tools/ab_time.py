#!/usr/bin/env python3
"""A/B timing helper: per-kernel HIP-event times of the bench workload, min and median over several rounds of 10 runs,
for one or more pipeline modes interleaved in ONE process (cdna_hip_programming.md rule 24)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from field_coverage_path_planning_amd import engine as E  # noqa: E402

modes = [int(m) for m in (sys.argv[1:] or ['1'])]
rng = np.random.default_rng(1024)
LH = rng.uniform(100.0, 1000.0, size=(1024, 2))
specs = [E.FieldSpec(field_length=float(x), field_width=float(y)) for x, y in LH]
b = E.Batch(specs, E.make_vehicle(), E.make_options(1, 0.1))
bufs = b.alloc()
for m in modes:
    b.run(bufs, mode=m)
torch.cuda.synchronize()
res = {m: [] for m in modes}
for rnd in range(6):
    for m in modes:
        b.set_profiling(True)
        for _ in range(10):
            b.run(bufs, mode=m)
        t, _ = b.stage_times()
        b.set_profiling(False)
        res[m].append(t)
for m in modes:
    keys = res[m][0].keys()
    print('mode', m, {k: (round(min(r[k] for r in res[m]), 3), round(float(np.median([r[k] for r in res[m]])), 3)) for k in keys if k},
          'total(min,median)', round(min(sum(r.values()) for r in res[m]), 3), round(float(np.median([sum(r.values()) for r in res[m]])), 3))
