#!/usr/bin/env python3
"""A few steps of one BASELINE configuration, for rocprofv3 passes (the program goes directly after `--`):
    rocprofv3 --kernel-trace --stats -d out -o name --output-format csv -- python3 tools/prof_cfg.py cfg5 --steps 5
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU ... -d out -o name --output-format csv -- python3 tools/prof_cfg.py cfg1 --steps 3
configs: cfg1 (4096 x 500x200, arcs, reference sampling), cfg1_clothoid, cfg2_ref, cfg2_0.5, cfg2_0.1, cfg3, cfg5."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from field_coverage_path_planning_amd import engine as E, workloads as WL  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('config')
ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--mode', type=int, default=1)
ap.add_argument('--placement', type=int, default=1, help='> 1: candidate sets of output arrays (Batch.alloc best_of, opt-in); default 1: Batch.alloc() with its layout rule, as bench.py')
a = ap.parse_args()
c = a.config
if c.startswith('cfg1'):
    specs, opt = WL.specs_from_lh(E, WL.cfg1_batch(4096)), E.make_options(1 if 'clothoid' in c else 0, 0.1 if 'dense' in c else 0.0)
elif c.startswith('cfg2'):
    sp = {'cfg2_ref': (0, 0.0), 'cfg2_0.5': (1, 0.5), 'cfg2_0.1': (1, 0.1)}[c]
    specs, opt = WL.specs_from_lh(E, WL.cfg2_rectangles()), E.make_options(*sp)
elif c == 'cfg3':
    (L, H), obst = WL.cfg3_field()
    specs, opt = [E.FieldSpec(field_length=L, field_width=H, obstacles=obst)], E.make_options(1, 0.05)
elif c == 'cfg5':
    specs, opt = WL.specs_from_vertices(E, WL.cfg5_parallelograms()), E.make_options()
else:
    raise SystemExit('unknown config ' + c)
b = E.Batch(specs, E.make_vehicle(), opt)
# the same output placement as bench.py gives the configuration (setup; its launches precede the `steps` timed ones in the trace --
# tools/profiles_summary.py takes the kernel statistics from the LAST `steps` dispatches of every kernel)
placement = a.placement
if placement > 1 and a.mode == 1:
    bufs = b.alloc(best_of=placement, include=[b.alloc()])
else:
    bufs = b.alloc()
b.run(bufs, mode=a.mode)
torch.cuda.synchronize()
for _ in range(a.steps):
    b.run(bufs, mode=a.mode)
torch.cuda.synchronize()
print(c, 'points', b.total_points, 'stage points', b.stage_points(), 'timed_steps', a.steps, 'layout', getattr(b, 'layout', None), 'placement', getattr(b, 'placement', None))
