"""Hypothesis-driven GPU tests of the standalone operators (fcpp_speed_plan / fcpp_curvature / fcpp_verify, the staged kernels behind
them): arbitrary ragged batches -- empty, one- and two-point paths, zero-length steps, repeated points, speed jumps, vehicles with tiny
and large accelerations, paths that cross tile (512) and spine-block boundaries -- against the oracle path by path, plus the properties
any valid speed plan has (never above the input, idempotent)."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import oracle as orc
from field_coverage_path_planning_amd import engine as E

pytestmark = pytest.mark.gpu


@st.composite
def ragged_batches(draw):
    n_paths = draw(st.integers(1, 9))
    lens = [draw(st.one_of(st.integers(0, 4), st.integers(5, 80), st.sampled_from([511, 512, 513, 1024, 1025, 1600]))) for _ in range(n_paths)]
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    total = int(sum(lens))
    step = draw(st.sampled_from([0.05, 0.4, 3.0]))
    xy = np.cumsum(rng.normal(0, step, size=(total, 2)), axis=0) if total else np.zeros((0, 2))
    for _ in range(draw(st.integers(0, 6))):               # repeated points and steps below the 1e-6 threshold
        if total >= 2:
            k = int(rng.integers(1, total))
            xy[k] = xy[k - 1] + (0.0 if rng.random() < 0.5 else 4e-7)
    v = rng.choice([0.3, 2.5, 4.0, 9.0, 15.0, 28.0], size=total)
    a_lon = draw(st.sampled_from([0.05, 1.5, 3.0]))
    a_lat = draw(st.sampled_from([0.5, 2.0]))
    return lens, xy, v, a_lon, a_lat


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])
@given(ragged_batches())
def test_speed_plan_on_arbitrary_ragged_batches(case):
    lens, xy, v, a_lon, a_lat = case
    offs = np.cumsum([0] + lens)
    vp = [3.2, 8.0, 9.0, 15.0, 4.0, a_lat, a_lon, 0.85]
    veh, oveh = E.make_vehicle(max_lateral_accel=a_lat, max_longitudinal_accel=a_lon), orc.Vehicle.make(vp)
    for clamp in (True, False):
        got, nadj = E.speed_plan(xy[:, 0], xy[:, 1], v, veh, clamp=clamp, offsets=offs)
        got = got.cpu().numpy()
        for k, (a, b) in enumerate(zip(offs[:-1], offs[1:])):
            if b == a:
                continue
            if clamp:
                want, adj = orc.speed_limit(xy[a:b], v[a:b], oveh)
                assert int(nadj[k]) == adj, (k, lens)
            else:
                want = orc.smooth_speed_profile(xy[a:b], v[a:b], a_lon)
            np.testing.assert_allclose(got[a:b], want, rtol=0, atol=1e-9, err_msg=f'path {k} of {lens}, clamp {clamp}')
        assert (got <= v + 1e-12).all()                                    # a plan never raises a speed
        again, nadj2 = E.speed_plan(xy[:, 0], xy[:, 1], got, veh, clamp=clamp, offsets=offs)
        np.testing.assert_allclose(again.cpu().numpy(), got, rtol=0, atol=1e-9)      # ... and is a fixed point
    kap = E.curvature(xy[:, 0], xy[:, 1], offsets=offs).cpu().numpy()
    for a, b in zip(offs[:-1], offs[1:]):
        for i in range(a + 1, b - 1):
            if i - a < 3 or b - i < 4 or (i - a) % 97 == 0:              # ends of every path and a sample of its interior
                assert abs(kap[i] - orc.curvature(xy[i - 1], xy[i], xy[i + 1])) <= 1e-9 * max(1.0, abs(kap[i])), (i, a, b)
        if b > a:
            assert kap[a] == 0.0 and kap[b - 1] == 0.0
