import sys, time
sys.path.insert(0, '/root/repo')
from field_coverage_path_planning_amd.multi_layer_planner_v3 import TwoLayerPathPlannerV37, VehicleParams
import torch
vp = VehicleParams()
def once():
    t0 = time.perf_counter()
    p = TwoLayerPathPlannerV37(vp, field_length=500, field_width=200); p.verbose = False
    t1 = time.perf_counter()
    r = p.plan_complete_coverage()
    t2 = time.perf_counter()
    r = p.plan_complete_coverage()
    t3 = time.perf_counter()
    p.close()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3
for _ in range(5): once()
import numpy as np
a = np.array([once() for _ in range(60)])
print('median ms: ctor %.3f plan %.3f plan again %.3f' % tuple(np.median(a, axis=0)))
