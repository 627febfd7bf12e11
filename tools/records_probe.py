#!/usr/bin/env python3
"""Diagnostic: the fresh plan call of a batch with its field records in PINNED host memory against the same records in DEVICE memory
(FieldTable.to_device(): nothing crosses PCIe before the first kernel), regions interleaved in one process; and what the drain at the end
of a call costs (torch.cuda.synchronize() against the stream's own synchronize).
    python tools/records_probe.py [fields] [calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from field_coverage_path_planning_amd import engine as E

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
LH = np.tile(np.array([[500.0, 200.0]]), (n, 1))
tables = {'pinned': E.FieldTable.from_rectangles(LH).pin(), 'device': E.FieldTable.from_rectangles(LH).to_device()}
E.get_context().reserve_outputs(lane_gib=24.0, pitch_gib=24.0)
veh, opt = E.make_vehicle(), E.make_options()
cur = torch.cuda.current_stream()
drains = {'device_sync': torch.cuda.synchronize, 'stream_sync': cur.synchronize}


def region(table, drain):
    batch = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        if batch is not None:
            batch.close()
        batch, res = E.Batch.plan(table, veh, opt)
        drain()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    batch.close()
    return dt / calls * 1e3


for name in tables:
    region(tables[name], torch.cuda.synchronize)
for rep in range(4):
    row = []
    for dname, drain in drains.items():
        for name, t in tables.items():
            row.append(f'{name}/{dname} {region(t, drain):.4f}')
    print('ms per call: ' + ' | '.join(row), flush=True)
