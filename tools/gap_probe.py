#!/usr/bin/env python3
"""Tuning: ONE slab holding the five output arrays with a coarse gap between them -> k_plan_quiet time per gap (one process)."""
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
dists = [7.6, 16.0]
bases = [0.0, 8.0, 16.0, 24.0, 33.0]      # GiB: where the first array starts (distance 7.6 only)
ballast = torch.empty(int(float(os.environ.get('BALLAST_GIB', '0')) * (1 << 30)) + 8, dtype=torch.uint8, device=dev)
print('ballast GiB', ballast.numel() >> 30)
slab = torch.empty(int(4 * max(dists) * (1 << 30) + 8 * n + (1 << 20)), dtype=torch.uint8, device=dev)
print('slab base mod 16 GiB:', slab.data_ptr() % (16 << 30) / (1 << 30))
for rnd in range(2):
    for gap, base in [(7.6, bb) for bb in bases] + [(16.0, 0.0)]:
        out = []
        for k in range(5):
            off = k * (int(gap * (1 << 30)) // 4096 * 4096) + int(base * (1 << 30))
            out.append(slab[off:off + SZ[k] * n].view(DT[k]))
        bufs = tuple(out) + (stats,)
        b.run(bufs)
        torch.cuda.synchronize()
        b.set_profiling(True)
        for _ in range(10):
            b.run(bufs)
        st, _ = b.stage_times()
        b.set_profiling(False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            for a in out:
                a.fill_(1)
        e1.record()
        torch.cuda.synchronize()
        fill_ms = e0.elapsed_time(e1) / 3
        d = out[1].data_ptr() - out[0].data_ptr()
        print(f'base {base:5.1f} GiB, distance {gap:5.2f} GiB: quiet {st["k_plan_quiet"]:.3f} ms   torch fill of the same arrays {fill_ms:.3f} ms   (array distance {d / (1 << 30):.4f} GiB)', flush=True)
