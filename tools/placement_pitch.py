#!/usr/bin/env python3
"""Placement probe 4: the five output arrays of cfg5 inside ONE large slab at a pitch of P GiB (P from the arrays' own size up to 56 GiB):
does the span kernel's speed class depend on how far apart in device memory the five streams lie?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from field_coverage_path_planning_amd import engine as E, workloads as WL  # noqa: E402

torch.cuda.set_stream(torch.cuda.Stream())
b = E.Batch(E.FieldTable.from_vertices(WL.cfg5_parallelograms()), E.make_vehicle(), E.make_options())
n = b.total_points
stats = torch.zeros((b.n_fields, 13), dtype=torch.int64, device='cuda')
free, total = torch.cuda.mem_get_info()
G = int(free / 2**30) - 6
slab = torch.empty(G << 30, dtype=torch.uint8, device='cuda')
print(f'slab {G} GiB at {slab.data_ptr():#x}', flush=True)
SZ, DT = [8, 8, 8, 8, 4], [torch.float64] * 4 + [torch.int32]
for pitch_gib in (2.25, 4, 8, 16, 24, 32, 33.5, 36, 40, 48, 56):
    P = int(pitch_gib * 2**30) // 4096 * 4096
    if 4 * P + 8 * n > (G << 30):
        continue
    for shift in (0, 3 << 30):
        if shift + 4 * P + 8 * n > (G << 30):
            continue
        bufs = tuple(slab[shift + k * P: shift + k * P + SZ[k] * n].view(DT[k]) for k in range(5)) + (stats,)
        b.run(bufs)
        torch.cuda.synchronize()
        b.set_profiling(True)
        for _ in range(3):
            b.run(bufs)
        st, _ = b.stage_times()
        b.set_profiling(False)
        print(f'pitch {pitch_gib:6.2f} GiB shift {shift >> 30} GiB: spans {st["k_plan_quiet_spans"]:.3f} ms  sparse {st["k_plan_sparse"]:.3f} ms', flush=True)
