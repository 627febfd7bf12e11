#!/usr/bin/env python3
"""Placement probe 2: the five output arrays carved out of ONE slab at chosen offsets -- does the span kernel's speed class follow the
arrays' RELATIVE offsets (skews between the streams) or the region of the slab they lie in (a common shift)?
usage: placement_probe2.py [slab GiB] [n_slabs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from field_coverage_path_planning_amd import engine as E, workloads as WL  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 40
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.cuda.set_stream(torch.cuda.Stream())
b = E.Batch(E.FieldTable.from_vertices(WL.cfg5_parallelograms()), E.make_vehicle(), E.make_options())
n = b.total_points
dev = torch.device('cuda', 0)
stats = torch.zeros((b.n_fields, 13), dtype=torch.int64, device=dev)
SZ = [8, 8, 8, 8, 4]
DT = [torch.float64] * 4 + [torch.int32]
MiB = 1 << 20
A = ((8 * n + 64 * MiB - 1) // (64 * MiB)) * 64 * MiB          # array pitch: a multiple of 64 MiB


def span_ms(bufs, reps=3):
    b.run(bufs)
    torch.cuda.synchronize()
    b.set_profiling(True)
    for _ in range(reps):
        b.run(bufs)
    st, _ = b.stage_times()
    b.set_profiling(False)
    return st['k_plan_quiet_spans']


def carve(slab, shift, skew):
    out = []
    for k in range(5):
        off = shift + k * A + k * skew
        out.append(slab[off:off + SZ[k] * n].view(DT[k]))
    return tuple(out) + (stats,)


for s in range(NS):
    slab = torch.empty(G << 30, dtype=torch.uint8, device=dev)
    print(f'slab {s}: {G} GiB at {slab.data_ptr():#x}, array pitch {A / MiB:.0f} MiB', flush=True)
    room = (G << 30) - 5 * A
    for skew in (0, 4096, 65536, 256 * 1024, MiB, 2 * MiB, 3 * MiB, 8 * MiB, 32 * MiB, 128 * MiB + 4096):
        if 4 * skew > room:
            continue
        print(f'  skew {skew / MiB:9.4f} MiB  shift 0: {span_ms(carve(slab, 0, skew)):.3f} ms', flush=True)
    for shift in (0, 2 * MiB, 64 * MiB, 1 << 30, 4 << 30, 8 << 30, 16 << 30, 24 << 30):
        if shift > room:
            continue
        print(f'  skew 0, shift {shift / (1 << 30):7.3f} GiB: {span_ms(carve(slab, shift, 0)):.3f} ms', flush=True)
    del slab
    torch.cuda.empty_cache()
