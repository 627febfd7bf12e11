#!/usr/bin/env python3
"""Placement map: a small batch (its five output arrays packed into a ~1.3 GiB window) slid through ONE large slab in steps; the span
kernel's time at every position.  Shows whether the speed classes are REGIONS of the allocation (physical placement) and how large.
usage: placement_map.py [slab GiB] [step GiB] [fields]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from field_coverage_path_planning_amd import engine as E, workloads as WL  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 200
STEP = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
NF = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
torch.cuda.set_stream(torch.cuda.Stream())
b = E.Batch(E.FieldTable.from_vertices(WL.cfg5_parallelograms(NF)), E.make_vehicle(), E.make_options())
n = b.total_points
dev = torch.device('cuda', 0)
stats = torch.zeros((b.n_fields, 13), dtype=torch.int64, device=dev)
SZ = [8, 8, 8, 8, 4]
DT = [torch.float64] * 4 + [torch.int32]
MiB = 1 << 20
A = ((8 * n + 2 * MiB - 1) // (2 * MiB)) * 2 * MiB
free, total = torch.cuda.mem_get_info()
G = min(G, int(free / 2**30) - 4)
slab = torch.empty(G << 30, dtype=torch.uint8, device=dev)
print(f'{n} points, window {5 * A / 2**30:.2f} GiB, slab {G} GiB at {slab.data_ptr():#x}', flush=True)


def span_ms(off, reps=5):
    bufs = tuple(slab[off + k * A: off + k * A + SZ[k] * n].view(DT[k]) for k in range(5)) + (stats,)
    b.run(bufs)
    torch.cuda.synchronize()
    b.set_profiling(True)
    for _ in range(reps):
        b.run(bufs)
    st, _ = b.stage_times()
    b.set_profiling(False)
    return st['k_plan_quiet_spans']


row = []
off = 0
while off + 5 * A <= (G << 30):
    row.append((off / 2**30, span_ms(off)))
    off += int(STEP * 2**30)
pts = b.stage_points()['k_plan_quiet_spans']
print('GiB: us (TB/s)')
for o, ms in row:
    print(f'{o:7.1f}: {ms * 1e3:7.1f}  {36 * pts / ms / 1e9:5.2f}', flush=True)
