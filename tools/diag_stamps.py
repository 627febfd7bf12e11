#!/usr/bin/env python3
"""Diagnostic only: run the bench workload with a -DFCPP_DIAG_STAMPS build (FCPP_LIBRARY=build/libfcpp_diagN.so) and print
the mean shader cycles each phase of k_plan_fused takes in wave N."""
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
for _ in range(2):
    res = b.run(bufs, mode=13)
torch.cuda.synchronize()
st = res.stats()
ntiles = sum((i.n_main + i.n_head + 2047) // 2048 for i in b.info)
names = [('main_len_m', 'load tile+field / halo'), ('main_time_pre_s', 'decode+generate'), ('main_time_s', 'exchange+barrier1'),
         ('head_len_m', 'd/kappa/geofence'), ('head_time_pre_s', 'store x,y,kappa'), ('head_time_s', 'clamp+scan(+barrier2)'),
         ('n_viol', 'prev exchange(+barrier3)'), ('n_outside', 'metrics'), ('n_in_obstacle', 'store v,fs + reduce(+barrier4)'),
         ('n_adjusted', 'TOTAL')]
print(os.environ.get('FCPP_LIBRARY'), 'tiles', ntiles)
for k, label in names:
    print(f'{label:36s} {st[k].sum() / ntiles:10.0f} cycles')
