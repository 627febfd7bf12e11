#!/usr/bin/env python3
"""Tuning: order in which k_plan_quiet walks its tiles (FCPP_QUIET_STRIPES=K,C) x how the outputs are allocated."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from field_coverage_path_planning_amd import engine as E  # noqa: E402

rng = np.random.default_rng(1024)
LH = rng.uniform(100.0, 1000.0, size=(1024, 2))
specs = [E.FieldSpec(field_length=float(x), field_width=float(y)) for x, y in LH]
dev = torch.device('cuda', 0)
SZ = [8, 8, 8, 8, 4]
DT = [torch.float64] * 4 + [torch.int32]
batches = {}
for st in (sys.argv[1:] or ['1,64', '8,64', '64,64', '8,1024', '64,8', '1024,8']):
    os.environ['FCPP_QUIET_STRIPES'] = st
    batches[st] = E.Batch(specs, E.make_vehicle(), E.make_options(1, 0.1))
n = next(iter(batches.values())).total_points
stats = torch.zeros((1024, 13), dtype=torch.int64, device=dev)
slab = torch.empty(5 * 8 * n + (1 << 20), dtype=torch.uint8, device=dev)
out, off = [], 0
for k in range(5):
    out.append(slab[off:off + SZ[k] * n].view(DT[k]))
    off += (SZ[k] * n + 4095) // 4096 * 4096
layouts = {'slab': tuple(out) + (stats,), 'separate': next(iter(batches.values())).alloc()}
ref = None
for lname, bufs in layouts.items():
    for st, b in batches.items():
        r = b.run(bufs)
        torch.cuda.synchronize()
        chk = float(r.x.sum().item()) + float(r.v.sum().item())
        ref = chk if ref is None else ref
        b.set_profiling(True)
        for _ in range(10):
            b.run(bufs)
        t, _ = b.stage_times()
        b.set_profiling(False)
        print(f'{lname:9s} stripes {st:>8s}: quiet {t["k_plan_quiet"]:.3f} ms  (checksum equal: {chk == ref})', flush=True)
