"""Child of tests/test_gpu_field_work.py (started with FCPP_TUNE=1): the same batch planned with fields of up to four wave tiles planned
and reduced by one workgroup each (k_plan_sparse_fields, the default) and with every wave tile in k_plan_sparse and every field in
k_reduce_stats (FCPP_FIELD_WORK=0, read when a batch is created).  Writes both results."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from field_coverage_path_planning_amd import engine as E  # noqa: E402
from tests import test_gpu_parity as T  # noqa: E402

out = sys.argv[1]
rng = np.random.default_rng(77)
specs = [E.FieldSpec(field_length=float(a), field_width=float(b)) for a, b in rng.uniform(90.0, 700.0, size=(300, 2))]
more, _ = T._random_fields(4242, 60, para=True, with_obstacles=True)       # skewed fields (points outside), obstacles
specs += more
veh, opt = E.make_vehicle(), E.make_options()
res = {}
for tag, val in (('work', '1'), ('open', '0')):
    os.environ['FCPP_FIELD_WORK'] = val
    b = E.Batch(specs, veh, opt)
    bufs = b.alloc()
    for t in (tag, tag + '2'):       # the second step goes into the same arrays: the flag counts of the first were reset
        rr = b.run(bufs)
        torch.cuda.synchronize()
        res[t + '_x'], res[t + '_y'] = rr.x.cpu().numpy(), rr.y.cpu().numpy()
        res[t + '_k'], res[t + '_v'], res[t + '_f'] = rr.kappa.cpu().numpy(), rr.v.cpu().numpy(), rr.flagseg.cpu().numpy()
        res[t + '_s'] = rr.stats_raw.cpu().numpy()
    res[tag + '_classes'] = np.array(b.reduce_classes(), dtype=np.int64)
    b.close()
np.savez(out, **res)
print('field work worker OK')
