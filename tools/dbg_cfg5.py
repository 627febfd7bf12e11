import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from field_coverage_path_planning_amd import engine as E, workloads as WL
V = WL.cfg5_parallelograms(4096)
b = E.Batch(WL.specs_from_vertices(E, V), E.make_vehicle(), E.make_options())
print(b.stage_points() if hasattr(b,'_last_mode') else '')
b.run()
print(b.stage_points())
