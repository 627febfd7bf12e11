"""Property tests (hypothesis) of the reformulation the kernels rest on ("Sweeps as scans"):
the reference's two sequential sweeps (MLP:538-589, restated in oracle/fcpp_oracle.c: orc_smooth_speed_profile) equal
    u = min(forward scan of u0, backward scan of u0),   u = (v / 3.6)^2,
where a point is the map u -> min(c, u + w), w = 2a|dp| (+inf for a skipped step), and maps compose associatively
    (c, w) o (c', w') = (min(c', c + w'), w + w').
Checked on random paths with zero-length steps, repeated points and speed jumps, for the sequential form of the scans, for an arbitrary
bracketing of the composition (what the tile / spine / look-back levels of the kernels amount to) and for the fixed point of the sparse
kernel's relaxation rounds."""
import numpy as np
from hypothesis import given, settings, strategies as st

import oracle as orc

INF = float('inf')


def _couplings(xy, a_lon):
    d = np.sqrt(np.sum(np.diff(xy, axis=0) ** 2, axis=1))
    w = 2.0 * a_lon * d
    w[d < 1e-6] = INF                         # MLP:560-561, 576-577: a skipped step couples nothing
    return np.concatenate([[INF], w])          # w[i]: coupling of segment (i - 1, i)


def _compose(first, then):
    """the map `first` followed by `then`: u -> min(c2, min(c1, u + w1) + w2)"""
    (c1, w1), (c2, w2) = first, then
    return (min(c2, c1 + w2), w1 + w2)


def _scan(u0, w):
    """inclusive scan of the maps (u0[i], w[i]) from the left, applied to u = +inf"""
    out = np.empty_like(u0)
    acc = (INF, 0.0)
    for i in range(len(u0)):
        acc = _compose(acc, (u0[i], w[i]))
        out[i] = acc[0]
    return out


def _tree(maps):
    """the same composition with an arbitrary (balanced) bracketing"""
    if len(maps) == 1:
        return maps[0]
    m = len(maps) // 2
    return _compose(_tree(maps[:m]), _tree(maps[m:]))


@st.composite
def paths(draw):
    n = draw(st.integers(2, 60))
    steps = draw(st.lists(st.one_of(st.just(0.0), st.floats(1e-7, 5e-7), st.floats(0.01, 30.0)), min_size=n - 1, max_size=n - 1))
    ang = draw(st.lists(st.floats(-3.2, 3.2), min_size=n - 1, max_size=n - 1))
    xy = np.zeros((n, 2))
    for i in range(1, n):
        xy[i] = xy[i - 1] + steps[i - 1] * np.array([np.cos(ang[i - 1]), np.sin(ang[i - 1])])
    v = np.array(draw(st.lists(st.floats(0.05, 30.0), min_size=n, max_size=n)))
    a_lon = draw(st.floats(0.05, 3.0))
    return xy, v, a_lon


@settings(max_examples=300, deadline=None)
@given(paths())
def test_two_sweeps_equal_min_of_two_scans(case):
    xy, v, a_lon = case
    ref = orc.smooth_speed_profile(xy, v, a_lon)                     # the reference's loops
    u0 = (v / 3.6) ** 2
    w = _couplings(xy, a_lon)
    fwd = _scan(u0, w)
    bwd = _scan(u0[::-1], np.concatenate([[INF], w[1:][::-1]]))[::-1]
    u = np.minimum(fwd, bwd)
    got = np.where(u < u0, np.sqrt(u) * 3.6, v)                      # untouched points keep their value exactly
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)
    assert np.array_equal(got[u >= u0], ref[u >= u0])


@settings(max_examples=200, deadline=None)
@given(paths())
def test_composition_is_associative_up_to_rounding(case):
    xy, v, a_lon = case
    u0, w = (v / 3.6) ** 2, _couplings(xy, a_lon)
    maps = [(INF, 0.0)] + [(float(u0[i]), float(w[i])) for i in range(len(u0))]
    seq = (INF, 0.0)
    for m in maps[1:]:
        seq = _compose(seq, m)
    tre = _tree(maps)
    for a, b in zip(seq, tre):
        assert a == b or abs(a - b) <= 1e-12 * max(abs(a), abs(b))


@settings(max_examples=200, deadline=None)
@given(paths())
def test_relaxation_rounds_reach_the_same_fixed_point(case):
    """k_plan_sparse: u_i = min(u_i, u_(i-1) + w_i, u_(i+1) + w_(i+1)) repeated until nothing moves"""
    xy, v, a_lon = case
    u0, w = (v / 3.6) ** 2, _couplings(xy, a_lon)
    u = u0.copy()
    wn = np.concatenate([w[1:], [INF]])
    for _ in range(len(u) + 1):
        left = np.concatenate([[INF], u[:-1]]) + w
        right = np.concatenate([u[1:], [INF]]) + wn
        nu = np.minimum(u, np.minimum(left, right))
        if np.array_equal(nu, u):
            break
        u = nu
    ref = (orc.smooth_speed_profile(xy, v, a_lon) / 3.6) ** 2
    np.testing.assert_allclose(u, ref, rtol=1e-12, atol=1e-14)
