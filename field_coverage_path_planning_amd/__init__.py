"""field_coverage_path_planning_amd -- MI355X-native hot path of qwagrox/field-coverage-path-planning.

Layout (only what the hot path needs):
    csrc/                       HIP kernels + C ABI (libfcpp.so, declared in include/fcpp.h)
    _lib.py                     ctypes binding of the C ABI (fails loudly without the .so)
    engine.py                   batches, device buffers (torch tensors as plain memory), operators
    multi_layer_planner_v3.py   mirror of the reference's planner surface (VehicleParams, TwoLayerPathPlannerV37)
    genetic_algorithm_solver.py mirror of GAConfig / GeneticAlgorithmSolver (fitness on the GPU)
    sharding.py                 one-process-per-GPU field sharding + final gather (RCCL via torch.distributed)
"""
from .multi_layer_planner_v3 import (TwoLayerPathPlannerV35, TwoLayerPathPlannerV36,  # noqa: F401
                                     TwoLayerPathPlannerV37, TwoLayerPlannerV35, TwoLayerPlannerV36,
                                     VehicleParams)

__version__ = '0.1.0'
