#!/usr/bin/env python3
"""Tuning: how does the way the five output arrays are allocated change k_plan_quiet's time?  (one process)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from field_coverage_path_planning_amd import engine as E  # noqa: E402

rng = np.random.default_rng(1024)
LH = rng.uniform(100.0, 1000.0, size=(1024, 2))
specs = [E.FieldSpec(field_length=float(x), field_width=float(y)) for x, y in LH]
b = E.Batch(specs, E.make_vehicle(), E.make_options(1, 0.1))
n = b.total_points
dev = torch.device('cuda', 0)
stats = torch.zeros((b.n_fields, 13), dtype=torch.int64, device=dev)
SZ = [8, 8, 8, 8, 4]
DT = [torch.float64] * 4 + [torch.int32]


def timeit(bufs, label):
    b.run(bufs)
    torch.cuda.synchronize()
    b.set_profiling(True)
    for _ in range(10):
        b.run(bufs)
    st, _ = b.stage_times()
    b.set_profiling(False)
    ptrs = [t.data_ptr() for t in bufs[:5]]
    print(f'{label:44s} quiet {st["k_plan_quiet"]:.3f} ms  fused {st["k_plan_fused"]:.3f}   ptr>>30: {[p >> 30 for p in ptrs]}', flush=True)


def group(sizes_groups, oversize=0):
    """arrays grouped into allocations: [[0,1],[2,3,4]] ..."""
    out = [None] * 5
    keep = []
    for g in sizes_groups:
        tot = sum(SZ[k] * n + 4096 for k in g) + oversize if oversize >= 0 else (-oversize) << 30
        buf = torch.empty(tot, dtype=torch.uint8, device=dev)
        keep.append(buf)
        off = 0
        for k in g:
            out[k] = buf[off:off + SZ[k] * n].view(DT[k])
            off += (SZ[k] * n + 4095) // 4096 * 4096
    return tuple(out) + (stats,), keep


for rnd in range(2):
    for label, g, over in (('5 separate, exact', [[0], [1], [2], [3], [4]], 0), ('5 separate, +1 GiB each', [[0], [1], [2], [3], [4]], 1 << 30),
                           ('5 separate, +2 GiB each', [[0], [1], [2], [3], [4]], 2 << 30), ('5 separate, +4 GiB each', [[0], [1], [2], [3], [4]], 4 << 30),
                           ('5 separate, +8 GiB each', [[0], [1], [2], [3], [4]], 8 << 30),
                           ('5 separate, 16 GiB each', [[0], [1], [2], [3], [4]], -16)):
        bufs, keep = group(g, over)
        timeit(bufs, label)
        del bufs, keep
        torch.cuda.empty_cache()
