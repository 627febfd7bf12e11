#!/usr/bin/env python3
"""What a SHORT timed region of the headline step is made of (bench.py at the driver's K = 20 steps of ~0.06 ms): the region per step for
K = 20 / 200 / 2000 with the per-kernel events on a sample of its steps (as bench.py runs it), on one step per region and off; the
kernel's own duration from those events; and the host time to enqueue a step.  Usage: python tools/short_region.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from field_coverage_path_planning_amd import engine as E, workloads as WL  # noqa: E402

torch.cuda.set_stream(torch.cuda.Stream())
wa = torch.randn(4096, 4096, device='cuda')
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    for _ in range(10):
        wa = (wa @ wa).clamp_(-1.0, 1.0)
    torch.cuda.synchronize()
del wa
batch = E.Batch(E.FieldTable.from_rectangles(WL.cfg1_batch(4096)), E.make_vehicle(), E.make_options())
bufs = batch.alloc()
for _ in range(5):
    batch.run(bufs)
torch.cuda.synchronize()


def region(k, reps=9):
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(k):
            batch.run(bufs)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        t.append(((t2 - t0) / k * 1e6, (t1 - t0) / k * 1e6))
    t.sort()
    return {'us_per_step': round(t[len(t) // 2][0], 2), 'min': round(t[0][0], 2), 'enqueue_us_per_step': round(t[len(t) // 2][1], 2)}


out = {}
for k in (20, 200, 2000):
    for name, every in (('sampled', max(8, k // 8)), ('one', k), ('off', 0)):
        if every:
            batch.set_profiling(True, every=every)
        r = region(k)
        if every:
            kern, runs = batch.stage_times()
            r['kernel_us'] = {a: round(b * 1e3, 2) for a, b in kern.items() if b}
            r['prof_runs'] = runs
            batch.set_profiling(False)
        out[f'K{k}_{name}'] = r
        print(f'K{k}_{name}', json.dumps(r), flush=True)
# bench.py's own region: HIP events around the K launches, five regions back to back -- each region's time, in order
for trial in range(3):
    t = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(20):
            batch.run(bufs)
        e1.record()
        torch.cuda.synchronize()
        t.append(((time.perf_counter() - t0) / 20 * 1e6, e0.elapsed_time(e1) / 20 * 1e3))
    print('K20 with events around the region, 5 in a row (wall, events):', [(round(a, 1), round(b, 1)) for a, b in t], flush=True)
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(20):
            batch.run(bufs)
        torch.cuda.synchronize()
        t.append((time.perf_counter() - t0) / 20 * 1e6)
    print('K20 without events, 5 in a row:', [round(a, 1) for a in t], flush=True)
# where the five arrays lie: plain tensors (above), the context's arena (five lanes 24 GiB apart), plain tensors again with the arena held
ctx = E.get_context()
ctx.reserve_outputs(24.0, 24.0)
for layout in ('arena', 'plain'):
    b2 = batch.alloc(layout=layout)
    for _ in range(5):
        batch.run(b2)
    torch.cuda.synchronize()
    keep = bufs
    bufs = b2
    for k in (20, 2000):
        print(f'layout {layout} (arena reserved) K{k}', json.dumps(region(k)), [hex(t.data_ptr()) for t in b2[:5]], flush=True)
    bufs = keep
    del b2
# back to back regions without a pause, as bench.py's REPS do, vs. after an idle gap
for gap in (0.0, 0.05, 0.5):
    t = []
    for _ in range(7):
        time.sleep(gap)
        t0 = time.perf_counter()
        for _ in range(20):
            batch.run(bufs)
        torch.cuda.synchronize()
        t.append((time.perf_counter() - t0) / 20 * 1e6)
    print(f'K20 after {gap} s idle', [round(v, 1) for v in t], flush=True)
# what brings the rate back after an idle gap: 50 ms of matrix products, or of a trivial elementwise kernel, right before the region
wa = torch.randn(4096, 4096, device='cuda')
wb = torch.zeros(1 << 20, device='cuda')
wc = torch.zeros(1 << 24, device='cuda', dtype=torch.float64)


def busy(kind, seconds):
    global wa
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(4):
            if kind == 'matmul':
                wa = (wa @ wa).clamp_(-1.0, 1.0)
            elif kind == 'sin64':
                wc.sin_()
            elif kind == 'own':
                for _ in range(50):
                    batch.run(bufs)
            else:
                wb.add_(1.0)
        torch.cuda.synchronize()


for kind, seconds in (('own', 0.005), ('own', 0.02), ('own', 0.1), ('sin64', 0.02), ('sin64', 0.1), ('matmul', 0.05)):
    t = []
    for _ in range(5):
        time.sleep(0.5)
        busy(kind, seconds)
        t0 = time.perf_counter()
        for _ in range(20):
            batch.run(bufs)
        torch.cuda.synchronize()
        t.append((time.perf_counter() - t0) / 20 * 1e6)
    print(f'K20 after 0.5 s idle + {seconds} s of {kind}', [round(v, 1) for v in t], flush=True)
