#!/usr/bin/env python3
"""Tuning: does the relative placement of the five output arrays change k_plan_quiet's time?  One big allocation, arrays carved
out of it with different paddings between them, timed in one process."""
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
big = torch.empty(5 * 8 * n + (64 << 20) * 8, dtype=torch.uint8, device=dev)
stats = torch.zeros((b.n_fields, 13), dtype=torch.int64, device=dev)
print('base address mod 2 MiB:', big.data_ptr() % (2 << 20))


def carve(align):
    off, out = 0, []
    for k, (dt, sz) in enumerate([(torch.float64, 8)] * 4 + [(torch.int32, 4)]):
        out.append(big[off:off + sz * n].view(dt))
        off += sz * n
        off = (off + align - 1) // align * align
    return tuple(out) + (stats,)


pads = [256, 4096, 65536, 1 << 20, 2 << 20, 32 << 20]
for rnd in range(2):
    for pad in pads:
        bufs = carve(pad)
        b.run(bufs)
        torch.cuda.synchronize()
        b.set_profiling(True)
        for _ in range(10):
            b.run(bufs)
        st, _ = b.stage_times()
        b.set_profiling(False)
        print(f'pad {pad:>10d}: quiet {st["k_plan_quiet"]:.3f} ms  fused {st["k_plan_fused"]:.3f} ms  (x base mod 1MiB {bufs[0].data_ptr() % (1 << 20)}, y-x mod 1MiB {(bufs[1].data_ptr() - bufs[0].data_ptr()) % (1 << 20)})')
# separate allocations, as Batch.alloc() makes them
bufs = b.alloc()
b.run(bufs); torch.cuda.synchronize()
b.set_profiling(True)
for _ in range(10):
    b.run(bufs)
st, _ = b.stage_times()
print(f'separate allocations: quiet {st["k_plan_quiet"]:.3f} ms; y-x mod 1MiB {(bufs[1].data_ptr() - bufs[0].data_ptr()) % (1 << 20)}')
