#!/usr/bin/env python3
"""Latency of the drop-in planner for ONE field (the reference's own case, 500 x 200 m: 0.046 s published, README_en.md:206-208):
constructor, plan_complete_coverage (incl. coverage_rate), verify_all_corners_coverage, verify_curvature_constraints."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from field_coverage_path_planning_amd.multi_layer_planner_v3 import TwoLayerPathPlannerV37, VehicleParams
pl = TwoLayerPathPlannerV37(VehicleParams(), field_length=500, field_width=200)
r = pl.plan_complete_coverage()
import torch; torch.cuda.synchronize()
for k in range(3):
    t0 = time.perf_counter(); pl = TwoLayerPathPlannerV37(VehicleParams(), field_length=500, field_width=200); t1 = time.perf_counter()
    r = pl.plan_complete_coverage(); t2 = time.perf_counter()
    c = pl.verify_all_corners_coverage(r['headland']); t3 = time.perf_counter()
    v = pl.verify_curvature_constraints(r['main_work']['path'], r['main_work']['speeds']); t4 = time.perf_counter()
    print(f'ctor {1e3*(t1-t0):.1f} ms, plan {1e3*(t2-t1):.1f} ms (total_time {r["total_time"]*1e3:.1f}), corners {1e3*(t3-t2):.1f} ms, verify {1e3*(t4-t3):.1f} ms, coverage_rate {r["headland"]["stats"]["coverage_rate"]:.4f}, points {len(r["main_work"]["path"])}+{len(r["headland"]["path"])}')
