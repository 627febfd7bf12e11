"""csrc/fcpp_math.h: the setup's transcendentals in plain IEEE operations (the same bits on the host and on the GPU).  Accuracy against
numpy on the host (no GPU needed); the GPU half of the claim -- device == host bit for bit -- is tests/test_gpu_devplan.py."""
import ctypes as C

import numpy as np

from field_coverage_path_planning_amd import _lib as L


def _call(fn, a, b=None):
    lib = L.load()
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), dtype=np.float64)
    o0, o1 = np.empty_like(a), np.empty_like(a)
    L.check(lib.fcpp_debug_math(fn, a.size, C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(o0.ctypes.data), C.c_void_p(o1.ctypes.data)))
    return o0, o1


def _ulps(got, want):
    return np.abs(got - want) / np.spacing(np.abs(want))


def test_sincos_accuracy_and_exact_cases():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-np.pi, np.pi, 200000), rng.uniform(-1e5, 1e5, 100000), rng.uniform(-1e-3, 1e-3, 10000),
                        np.array([0.0, np.pi / 2, -np.pi / 2, np.pi, -np.pi, np.pi / 4, 1e-300])])
    s, c = _call(0, x)
    xl = x.astype(np.longdouble)
    assert _ulps(s, np.sin(xl).astype(np.float64)).max() <= 1.0
    assert _ulps(c, np.cos(xl).astype(np.float64)).max() <= 1.0
    s0, c0 = _call(0, np.array([0.0]))
    assert s0[0] == 0.0 and c0[0] == 1.0                     # an unrotated field stays exactly unrotated
    # odd / even symmetry, exactly (the frame of layer 1 is reached with -rotation)
    s1, c1 = _call(0, -x)
    assert np.array_equal(s1, -s) and np.array_equal(c1, c)


def test_atan2_acos_hypot_accuracy():
    rng = np.random.default_rng(2)
    y, x = rng.uniform(-1e3, 1e3, 200000), rng.uniform(-1e3, 1e3, 200000)
    a, _ = _call(1, y, x)
    assert _ulps(a, np.arctan2(y.astype(np.longdouble), x.astype(np.longdouble)).astype(np.float64)).max() <= 1.5
    assert _call(1, np.array([0.0]), np.array([500.0]))[0][0] == 0.0
    c = np.concatenate([rng.uniform(-1, 1, 200000), np.array([0.0, 1.0, -1.0, 0.5, -0.5])])
    ac, _ = _call(2, c)
    assert np.abs(ac - np.arccos(c.astype(np.longdouble)).astype(np.float64)).max() <= 1e-15 * np.pi * 2
    assert ac[-5] == np.arccos(0.0)                         # a right angle is exactly the platform's pi / 2
    h, _ = _call(3, y, x)
    assert _ulps(h, np.hypot(y.astype(np.longdouble), x.astype(np.longdouble)).astype(np.float64)).max() <= 1.0


def test_rotation_functions_are_correctly_rounded():
    """Round 5: the rotation's angle, sine and cosine (fc_atan2_cr, fc_sincos_cr) and the edge lengths (fc_hypot) in double-double, rounded
    once -- against mpmath at 300 bits, bit for bit.  (The platform libm is correctly rounded in ~99.9 % of arguments: the library now
    agrees with it -- and so with the oracle and, on the fragile exact-multiple fields, with the reference -- wherever the platform is.)"""
    mpmath = __import__('pytest').importorskip('mpmath')
    mpmath.mp.prec = 300
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(-np.pi, np.pi, 1500), rng.uniform(-1e5, 1e5, 300), rng.uniform(-1e-3, 1e-3, 200), np.array([np.pi / 2, -np.pi, 0.5, 1e-300])])
    s, c = _call(4, x)
    assert np.array_equal(s, np.array([float(mpmath.sin(mpmath.mpf(float(a)))) for a in x]))
    assert np.array_equal(c, np.array([float(mpmath.cos(mpmath.mpf(float(a)))) for a in x]))
    s0, c0 = _call(4, np.array([0.0]))
    assert s0[0] == 0.0 and c0[0] == 1.0
    s1, c1 = _call(4, -x)
    assert np.array_equal(s1, -s) and np.array_equal(c1, c)
    y, xx = rng.uniform(-1e3, 1e3, 2000), rng.uniform(-1e3, 1e3, 2000)
    a, _ = _call(5, y, xx)
    assert np.array_equal(a, np.array([float(mpmath.atan2(mpmath.mpf(float(p)), mpmath.mpf(float(q)))) for p, q in zip(y, xx)]))
    assert _call(5, np.array([0.0]), np.array([500.0]))[0][0] == 0.0 and _call(5, np.array([3.0]), np.array([0.0]))[0][0] == np.pi / 2
    h, _ = _call(3, y, xx)
    assert np.array_equal(h, np.array([float(mpmath.sqrt(mpmath.mpf(float(p)) ** 2 + mpmath.mpf(float(q)) ** 2)) for p, q in zip(y, xx)]))


def test_corner_angle_is_correctly_rounded_where_a_decision_hangs_on_it():
    """fc_acos within 1e-6 degrees of 60, 89 and 91 degrees (MLP:1043 the reverse fill of a corner of 60 degrees and more; MLP:224-235 the
    rectangle test): the correctly rounded acos, so that a field DRAWN with a corner of exactly 60 degrees falls on the reference's side."""
    mpmath = __import__('pytest').importorskip('mpmath')
    mpmath.mp.prec = 300
    rng = np.random.default_rng(12)
    cs = []
    for deg in (60.0, 89.0, 91.0, 120.0):
        c0 = np.cos(np.deg2rad(deg))
        cs.append(np.nextafter(c0, 1.0) + np.arange(-40, 41) * np.spacing(c0))
        if deg != 120.0:
            cs.append(c0 + rng.uniform(-1.5e-8, 1.5e-8, 300))
    c = np.concatenate(cs + [np.array([0.5, -0.5])])
    a, _ = _call(2, c)
    want = np.array([float(mpmath.acos(mpmath.mpf(float(v)))) for v in c])
    near = np.zeros(c.size, bool)
    for deg in (60.0, 89.0, 91.0):
        near |= np.abs(np.degrees(want) - deg) < 0.9e-6
    assert near.sum() > 800
    assert np.array_equal(a[near], want[near])
    assert _ulps(a, want).max() <= 2.0
    assert a[-2] * (180.0 / np.pi) >= 60.0                   # acos(0.5) in degrees: the platform's 60.00000000000001
