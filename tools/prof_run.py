#!/usr/bin/env python3
"""Small driver for rocprofv3 counter passes: a few steps of the bench workload (fused kernel by default)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from field_coverage_path_planning_amd import engine as E  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--fields', type=int, default=1024)
ap.add_argument('--spacing', type=float, default=0.1)
ap.add_argument('--turn-model', type=int, default=1)
ap.add_argument('--mode', type=int, default=1)
ap.add_argument('--steps', type=int, default=3)
a = ap.parse_args()
rng = np.random.default_rng(1024)
LH = rng.uniform(100.0, 1000.0, size=(a.fields, 2))
specs = [E.FieldSpec(field_length=float(x), field_width=float(y)) for x, y in LH]
b = E.Batch(specs, E.make_vehicle(), E.make_options(a.turn_model, a.spacing))
bufs = b.alloc()
for _ in range(a.steps):
    b.run(bufs, mode=a.mode)
torch.cuda.synchronize()
print('points', b.total_points)
