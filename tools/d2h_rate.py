#!/usr/bin/env python3
"""The headline batch WITH the device-to-host copy of its results ("report both with / without D2H"): one step of
cfg1 x 4096 followed by copies of x, y, kappa, v, flagseg into pinned host buffers (and into pageable ones).  Prints one JSON line.
Never `value` of bench.py: the ABI hands out device pointers."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from field_coverage_path_planning_amd import engine as E  # noqa: E402
from field_coverage_path_planning_amd import workloads as WL  # noqa: E402

torch.cuda.set_stream(torch.cuda.Stream())
b = E.Batch(WL.specs_from_lh(E, WL.cfg1_batch(4096)), E.make_vehicle(), E.make_options())
bufs = b.alloc()
res = b.run(bufs)
arrays = [res.x, res.y, res.kappa, res.v, res.flagseg]
nbytes = sum(a.numel() * a.element_size() for a in arrays)
out = {'points': b.total_points, 'bytes': nbytes}
for kind in ('pinned', 'pageable'):
    host = [torch.empty(a.shape, dtype=a.dtype, pin_memory=(kind == 'pinned')) for a in arrays]
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = b.run(bufs)
        for h, a in zip(host, arrays):
            h.copy_(a, non_blocking=(kind == 'pinned'))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    out[kind] = {'ms_step_plus_d2h': dt * 1e3, 'points_per_s': b.total_points / dt, 'GB_per_s': nbytes / dt / 1e9}
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    b.run(bufs)
torch.cuda.synchronize()
out['device_only_ms'] = (time.perf_counter() - t0) / 100 * 1e3
print(json.dumps(out))
