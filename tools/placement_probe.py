#!/usr/bin/env python3
"""Placement probe: K complete sets of output arrays for one batch (cfg5 by default, or cfg2 at 0.1 m), all alive at once; the batch's
step timed on each (per-kernel HIP events).  Run plain, with PYTORCH_HIP_ALLOC_CONF=expandable_segments:True, and under
rocprofv3 --pmc (every set gets exactly one warm-up and REPS timed steps, in set order, so dispatches map to sets by position).
usage: placement_probe.py [cfg5|cfg2] [K] [REPS] [churn]    churn=1: allocate / free a few odd-sized tensors between sets"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from field_coverage_path_planning_amd import engine as E, workloads as WL  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'cfg5'
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
churn = len(sys.argv) > 4 and sys.argv[4] == '1'
torch.cuda.set_stream(torch.cuda.Stream())
if which == 'cfg5':
    table, opt = E.FieldTable.from_vertices(WL.cfg5_parallelograms()), E.make_options()
else:
    table, opt = E.FieldTable.from_rectangles(WL.cfg2_rectangles()), E.make_options(1, 0.1)
b = E.Batch(table, E.make_vehicle(), opt)
print(f'{which}: {b.total_points} points, alloc conf {os.environ.get("PYTORCH_HIP_ALLOC_CONF", "-")}', flush=True)
sets, junk = [], []
rng = np.random.default_rng(3)
for k in range(K):
    if churn:
        tmp = [torch.empty(int(rng.integers(1 << 20, 1 << 28)), dtype=torch.uint8, device='cuda') for _ in range(6)]
        junk.append(tmp[::2])
        del tmp
        torch.cuda.empty_cache()
    sets.append(b.alloc())
rows = []
for k, s in enumerate(sets):
    b.run(s)
    torch.cuda.synchronize()
    b.set_profiling(True)
    for _ in range(REPS):
        b.run(s)
    st, _ = b.stage_times()
    b.set_profiling(False)
    dom = max(st, key=st.get)
    rows.append({'set': k, 'kernels_ms': {n: round(v, 4) for n, v in st.items() if v > 0}, 'ptr_x': hex(s[0].data_ptr()), 'ptr_fs': hex(s[4].data_ptr())})
    print(json.dumps(rows[-1]), flush=True)
tot = [sum(r['kernels_ms'].values()) for r in rows]
print(f'sum of kernels per set: min {min(tot):.3f} max {max(tot):.3f} ms  spread {max(tot) / min(tot):.3f}')
