#!/usr/bin/env python3
"""Diagnostic only: run the bench workload with a -DFCPP_DIAG_STAMPS build (FCPP_LIBRARY=build/libfcpp_diag0.so) and print
the mean shader cycles each phase of k_plan_fused takes per (non-quiet) tile.  In that build the metrics of the
non-quiet tiles are REPLACED by time stamps, so its results are not valid plans."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from field_coverage_path_planning_amd import engine as E  # noqa: E402

rng = np.random.default_rng(1024)
LH = rng.uniform(100.0, 1000.0, size=(1024, 2))
specs = [E.FieldSpec(field_length=float(x), field_width=float(y)) for x, y in LH]
sparse = os.environ.get('FCPP_DIAG_SPARSE')          # the reference's sampling: every general tile is a headland
if sparse:
    LH = rng.uniform(100.0, 1000.0, size=(16384, 2))
    specs = [E.FieldSpec(field_length=float(x), field_width=float(y)) for x, y in LH]
b = E.Batch(specs, E.make_vehicle(), E.make_options() if sparse else E.make_options(1, 0.1))
bufs = b.alloc()
for _ in range(2):
    res = b.run(bufs, mode=int(os.environ.get('FCPP_MODE', '1')))
torch.cuda.synchronize()
st = res.stats()
q, g = b.point_split()
ntiles = int(os.environ.get('FCPP_GENERAL_TILES', '0')) or max(1, g // 440)   # general tiles (approx. if not given)
names = [('max_kappa', '  load tile + field -> scalars'), ('max_alat', '  backward halo'), ('max_jump', '  forward halo'),
         ('main_len_m', 'load tile+field, both halos'), ('main_time_pre_s', 'decode+generate'), ('main_time_s', 'neighbour exchange'),
         ('head_len_m', 'd/kappa/geofence'), ('head_time_pre_s', 'store x,y,kappa'), ('head_time_s', 'clamp+scan'),
         ('n_viol', 'prev exchange'), ('n_outside', 'metrics'), ('n_in_obstacle', 'store v,fs + reduce'),
         ('n_adjusted', 'TOTAL')]
print(os.environ.get('FCPP_LIBRARY'), 'tiles', ntiles)
for k, label in names:
    val = st[k].mean() if k.startswith('max_') else st[k].sum() / ntiles     # max_* fields: per-field MAXIMUM over tiles, mean over fields
    print(f'{label:36s} {val:10.0f} cycles')
