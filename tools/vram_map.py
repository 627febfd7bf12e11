#!/usr/bin/env python3
"""Tuning: write-bandwidth map of device memory.  One big slab; a 2 GiB window is filled at successive offsets (torch fill,
single stream) and, every 8 GiB, by k_plan_quiet with all five output arrays packed into a 41 GB window starting there."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

dev = torch.device('cuda', 0)
free, total = torch.cuda.mem_get_info()
print(f'free {free / 2**30:.1f} GiB of {total / 2**30:.1f} GiB')
G = int(sys.argv[1]) if len(sys.argv) > 1 else 224
slab = torch.empty(G << 30, dtype=torch.uint8, device=dev)
print(f'slab {G} GiB at VA {slab.data_ptr():#x}')
W = 2 << 30


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


row = []
for off in range(0, G << 30, 4 << 30):
    w = slab[off:off + W].view(torch.float64)
    ms = timed(lambda: w.fill_(1.0))
    row.append(W / ms / 1e6)
print('fill GB/s per 4 GiB step:', ' '.join(f'{v:.0f}' for v in row))
