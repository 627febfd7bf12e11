"""The ctypes stub of INTEGRATION.md section B is the binding a maintainer of the reference would write: it is executed here as it
stands (only the library path is made absolute) and must plan the reference's own 500 x 200 m field."""
import os
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_stub_runs(golden_plans):
    import torch  # noqa: F401  (one HIP runtime per process: torch before libfcpp.so)
    text = open(os.path.join(REPO, 'INTEGRATION.md')).read()
    code = re.search(r'## B\..*?```python\n(.*?)```', text, re.S).group(1)
    lib = os.path.join(REPO, 'field_coverage_path_planning_amd', 'libfcpp.so')
    assert 'C.CDLL("libfcpp.so")' in code
    ns = {}
    exec(compile(code.replace('C.CDLL("libfcpp.so")', f'C.CDLL({lib!r})'), 'INTEGRATION.md', 'exec'), ns)
    assert ns['total'].value == 1256 + 435                                      # README_en.md:206-207
    g = golden_plans
    want = np.concatenate([g['cfg1_500x200/main_path'][:, 0], g['cfg1_500x200/head_path'][:, 0]])
    assert np.abs(ns['x'] - want).max() <= 1e-9
