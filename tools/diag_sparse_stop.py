#!/usr/bin/env python3
"""Diagnostic only (-DFCPP_DIAG_SPARSE build): the headline batch with sparse_tile2 cut off at the kernel's start (-2), before the first point (-1; -5: and without the field's reduction; -4: complete tiles without the field's reduction), after the lanes' first point (-3), after section 0, 1, ... 8, and complete, three steps
each, in this order -- under `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES` the differences between consecutive groups of
dispatches of k_plan_sparse_fields are the instructions of each section.  Results of the cut-off runs are not valid plans."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from field_coverage_path_planning_amd import _lib, engine as E, workloads as WL  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'headline'
table = {'headline': lambda: E.FieldTable.from_rectangles(WL.cfg1_batch(4096)),
         'cfg2_ref': lambda: E.FieldTable.from_rectangles(WL.cfg2_rectangles()),
         'cfg5': lambda: E.FieldTable.from_vertices(WL.cfg5_parallelograms())}[which]()
lib = _lib.load()
lib.fcpp_diag_sparse_stop.argtypes = [ctypes.c_int]
lib.fcpp_diag_sparse_stop.restype = ctypes.c_int
torch.cuda.set_stream(torch.cuda.Stream())
b = E.Batch(table, E.make_vehicle(), E.make_options())
bufs = b.alloc()
for stop in [-2, -5, -1, -4, -3] + list(range(9)) + [99]:
    torch.cuda.synchronize()
    assert lib.fcpp_diag_sparse_stop(stop) == 0
    for _ in range(3):
        b.run(bufs)
torch.cuda.synchronize()
print('done')
