import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_kernels():
    import numpy as np
    return np.load(os.path.join(GOLDEN, 'golden_kernels.npz'))


@pytest.fixture(scope='session')
def golden_ga():
    import numpy as np
    return np.load(os.path.join(GOLDEN, 'golden_ga.npz'))


@pytest.fixture(scope='session')
def golden_cover():
    """corner grid verification run by the reference (tools/gen_golden.py tier_cover)"""
    import numpy as np
    return np.load(os.path.join(GOLDEN, 'golden_cover.npz'))


@pytest.fixture(scope='session')
def golden_plans():
    import numpy as np
    return np.load(os.path.join(GOLDEN, 'golden_plans.npz'))


@pytest.fixture(scope='session')
def golden_mfp():
    """MultiFieldPlannerV38 run by the reference itself (tools/gen_golden.py tier_mfp): distance matrix and best connections"""
    import numpy as np
    return np.load(os.path.join(GOLDEN, 'golden_mfp.npz'))
