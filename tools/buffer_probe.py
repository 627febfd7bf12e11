#!/usr/bin/env python3
"""Tuning: is the write speed a property of each BUFFER (where its pages live)?  12 candidate buffers of one array's size are
timed one by one (torch fill, single stream), then k_plan_quiet runs with the four fastest and with the four slowest as x, y, kappa, v."""
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
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bufs = [torch.empty(n, dtype=torch.float64, device=dev) for _ in range(K)]
fsb = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3)]
stats = torch.zeros((1024, 13), dtype=torch.int64, device=dev)


def fill_ms(t, reps=4):
    t.fill_(1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        t.fill_(1)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def quiet_ms(sel, fs):
    s = (bufs[sel[0]], bufs[sel[1]], bufs[sel[2]], bufs[sel[3]], fs, stats)
    b.run(s)
    torch.cuda.synchronize()
    b.set_profiling(True)
    for _ in range(10):
        b.run(s)
    t, _ = b.stage_times()
    b.set_profiling(False)
    return t['k_plan_quiet']


for rnd in range(2):
    ms = [fill_ms(t) for t in bufs]
    order = np.argsort(ms)
    print('fill ms per buffer:', ' '.join(f'{m:.3f}' for m in ms))
    fs_ms = [fill_ms(t) for t in fsb]
    fbest, fworst = fsb[int(np.argmin(fs_ms))], fsb[int(np.argmax(fs_ms))]
    print(f'k_plan_quiet with the 4 fastest buffers: {quiet_ms(order[:4], fbest):.3f} ms;  4 slowest: {quiet_ms(order[-4:], fworst):.3f} ms;  '
          f'first 4 allocated: {quiet_ms([0, 1, 2, 3], fsb[0]):.3f} ms; last 4: {quiet_ms([K - 4, K - 3, K - 2, K - 1], fsb[2]):.3f} ms')
