"""`multi_layer_planner_v3_optimized` is imported by the reference's test/test_v37_complete.py:15 but does
not exist in the reference tree; it re-exports the planner so that script resolves."""
from .multi_layer_planner_v3 import *  # noqa: F401,F403
from .multi_layer_planner_v3 import TwoLayerPathPlannerV37, VehicleParams  # noqa: F401
