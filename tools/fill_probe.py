#!/usr/bin/env python3
"""Calibration: what does a plain device fill of the same output arrays achieve on THIS box / allocation, next to k_plan_quiet?"""
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
bufs = b.alloc()
res = b.run(bufs)
torch.cuda.synchronize()
arrs = [res.x, res.y, res.kappa, res.v, res.flagseg]
nbytes = sum(a.numel() * a.element_size() for a in arrs)


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def fill_all():
    for a in arrs:
        a.fill_(1)


t_fill = timed(fill_all)
t_one = timed(lambda: arrs[0].fill_(1.0))
b.set_profiling(True)
for _ in range(10):
    b.run(bufs)
st, _ = b.stage_times()
q, g = b.point_split()
print(f'fill of the 5 output arrays ({nbytes/1e9:.1f} GB): {t_fill:.3f} ms = {nbytes/t_fill/1e6:.0f} GB/s;  one 8 GB array: {arrs[0].numel()*8/t_one/1e6:.0f} GB/s')
print(f'k_plan_quiet: {st["k_plan_quiet"]:.3f} ms for {q*36/1e9:.1f} GB = {q*36/st["k_plan_quiet"]/1e6:.0f} GB/s; k_plan_fused {st["k_plan_fused"]:.3f} ms; total {sum(st.values()):.3f} ms')
