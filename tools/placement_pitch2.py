#!/usr/bin/env python3
"""Placement probe 5: a small batch (8192 cfg5 fields: five arrays of 0.25 GiB) inside one large slab, span kernel vs the PITCH between
the arrays in 1 GiB steps -- where is the transition between the slow and the fast class?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from field_coverage_path_planning_amd import engine as E, workloads as WL  # noqa: E402

torch.cuda.set_stream(torch.cuda.Stream())
b = E.Batch(E.FieldTable.from_vertices(WL.cfg5_parallelograms(8192)), E.make_vehicle(), E.make_options())
n = b.total_points
stats = torch.zeros((b.n_fields, 13), dtype=torch.int64, device='cuda')
free, total = torch.cuda.mem_get_info()
G = int(free / 2**30) - 6
slab = torch.empty(G << 30, dtype=torch.uint8, device='cuda')
print(f'slab {G} GiB at {slab.data_ptr():#x}; total device memory {total / 2**30:.2f} GiB, free {free / 2**30:.2f} GiB; arrays {8 * n / 2**30:.3f} GiB', flush=True)
SZ, DT = [8, 8, 8, 8, 4], [torch.float64] * 4 + [torch.int32]


def run(offs):
    bufs = tuple(slab[o: o + SZ[k] * n].view(DT[k]) for k, o in enumerate(offs)) + (stats,)
    b.run(bufs)
    torch.cuda.synchronize()
    b.set_profiling(True)
    for _ in range(5):
        b.run(bufs)
    st, _ = b.stage_times()
    b.set_profiling(False)
    return st['k_plan_quiet_spans']


run([k << 30 for k in range(5)])
pts = b.stage_points()['k_plan_quiet_spans']


for p in [0.5, 1, 2, 4, 6, 8, 10, 12, 13, 14, 15, 16, 17, 18, 19, 20, 22, 24, 28, 32, 40, 48]:
    P = int(p * 2**30)
    if 4 * P + 8 * n > (G << 30):
        continue
    ms = run([k * P for k in range(5)])
    print(f'pitch {p:5.1f} GiB: {ms * 1e3:7.1f} us  {36 * pts / ms / 1e9:5.2f} TB/s', flush=True)
# two arrays close, three far (which pairs matter?)
P = 32 << 30
for label, offs in (('x,y adjacent; others 32 GiB apart', [0, 1 << 30, P, 2 * P, 3 * P]), ('x,y,kappa adjacent', [0, 1 << 30, 2 << 30, P, 2 * P]),
                    ('all at 32 GiB', [0, P, 2 * P, 3 * P, 4 * P]), ('all adjacent, at 100 GiB', [(100 << 30) + (k << 30) for k in range(5)])):
    print(f'{label}: {run(offs) * 1e3:7.1f} us', flush=True)
