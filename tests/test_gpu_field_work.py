"""k_plan_sparse_fields (a workgroup plans the wave tiles of a field and reduces the field) against the path it replaced for such fields
(k_plan_sparse + k_reduce_stats): the points must be identical bit for bit -- the same tile function plans them --, the statistics equal
up to the order of their sums, the counts exactly; a second step into the same arrays gives the same (the flag counts the streaming
kernels leave in the runs' slots are reset by whoever reduces).  The knob is read when a batch is created and is live only in a process
started with FCPP_TUNE=1: the comparison runs in a child."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_field_work_equals_open_path(tmp_path):
    out = str(tmp_path / 'fw.npz')
    env = dict(os.environ, FCPP_TUNE='1')
    p = subprocess.run([sys.executable, os.path.join(REPO, 'tests', '_field_work_worker.py'), out], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and 'field work worker OK' in p.stdout, (p.stdout[-3000:], p.stderr[-3000:])
    g = np.load(out)
    # the default sends most of these fields to k_plan_sparse_fields, the knob none
    assert g['work_classes'].sum() < g['open_classes'].sum() and g['open_classes'].sum() == 360
    for a in ('work', 'open'):
        for k in 'xykvf':
            assert np.array_equal(g[f'{a}_{k}'], g[f'{a}2_{k}']), (a, k)           # a re-run is bit-identical
        assert np.array_equal(g[f'{a}_s'], g[f'{a}2_s']), a
    for k in 'xykvf':
        assert np.array_equal(g[f'work_{k}'], g[f'open_{k}']), k
    sw, so = g['work_s'], g['open_s']                      # (n_fields, 13) int64 words: nine doubles, four counters
    assert np.array_equal(sw[:, 9:], so[:, 9:])
    assert int(so[:, 10].sum()) > 0                        # some skewed field does leave its polygon: the flag counts are exercised
    fw, fo = sw[:, :9].view(np.float64), so[:, :9].view(np.float64)
    np.testing.assert_allclose(fw[:, :6], fo[:, :6], rtol=1e-13, atol=0.0)
    assert np.array_equal(fw[:, 6:], fo[:, 6:])            # maxima do not depend on the order
