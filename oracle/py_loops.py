"""oracle/py_loops.py -- TEST INFRASTRUCTURE (tests/, bench.py's cpu_baseline leg): never imported by the product.

The per-point stages of the hot path -- curvature, curvature clamp, forward / backward speed sweeps, metrics, verifier (SURVEY.md 8a rows
10-14) -- restated in Python the two ways SURVEY.md 8d asks the CPU baseline to be timed:

  *_loops : one Python iteration per path point with numpy scalars, the way the reference spends its time
            (/root/reference/multi_layer_planner_v3.py:490-504 clamp loop, :513-536 curvature, :558-587 sweeps, :1383-1408 verifier loop).
            Same float64 operations in the same order => the reference's values to the last bit where numpy's scalar functions are used.
  *_numpy : the same stages as whole-array numpy expressions; the two sweeps -- sequential recurrences in the reference -- as segmented
            running minima (u_i = 2 a S_i + min_{j <= i in segment}(u0_j - 2 a S_j), u = (v / 3.6)^2, S = arc length,
            a skipped step |dp| < 1e-6 opens a new segment).  Algebraically equal, not bit-equal: tolerance 1e-9 km/h in the tests.

Own code: written from the algorithm, checked against tests/golden/golden_kernels.npz (outputs of the reference itself,
tools/gen_golden.py) in tests/test_py_loops.py.  `vehicle` is any object with the VehicleParams attribute names (MLP:29-39).
"""
import numpy as np


# ---- per-point Python loops (the reference's cost model) ----------------------------------------------------------------------------
def curvature_loops(p1, p2, p3):
    """three-point curvature, MLP:513-536"""
    dx1, dy1 = p2 - p1
    dx2, dy2 = p3 - p2
    ds1 = np.sqrt(dx1 ** 2 + dy1 ** 2)
    ds2 = np.sqrt(dx2 ** 2 + dy2 ** 2)
    if ds1 < 1e-6 or ds2 < 1e-6:
        return 0.0
    dth = np.arctan2(dy2, dx2) - np.arctan2(dy1, dx1)
    dth = np.arctan2(np.sin(dth), np.cos(dth))
    return abs(2 * dth / (ds1 + ds2))


def smooth_loops(path, speeds, a_lon):
    """forward then backward sweep, MLP:538-589"""
    n = len(path)
    if n < 2:
        return speeds
    out = speeds.copy()
    for i in range(1, n):
        d = np.linalg.norm(path[i] - path[i - 1])
        if d < 1e-6:
            continue
        cap = np.sqrt((out[i - 1] / 3.6) ** 2 + 2 * a_lon * d) * 3.6
        if out[i] > cap:
            out[i] = cap
    for i in range(n - 2, -1, -1):
        d = np.linalg.norm(path[i + 1] - path[i])
        if d < 1e-6:
            continue
        cap = np.sqrt((out[i + 1] / 3.6) ** 2 + 2 * a_lon * d) * 3.6
        if out[i] > cap:
            out[i] = cap
    return out


def speed_plan_loops(path, speeds, vehicle):
    """curvature clamp + sweeps, MLP:467-511 -> (speeds, number of clamped points)"""
    n = len(path)
    if n < 3:
        return speeds, 0
    out = speeds.copy()
    a_lat, sf = vehicle.max_lateral_accel, vehicle.safety_factor
    adjusted = 0
    for i in range(1, n - 1):
        k = curvature_loops(path[i - 1], path[i], path[i + 1])
        if k > 1e-6:
            v_max = np.sqrt(a_lat / k) * sf * 3.6
            if out[i] > v_max:
                out[i] = v_max
                adjusted += 1
    return smooth_loops(path, out, vehicle.max_longitudinal_accel), adjusted


def verify_loops(path, speeds, vehicle):
    """verify_curvature_constraints, MLP:1373-1424 -> [max kappa, max a_lat, violations, violation rate %, max |d kappa|, pass]"""
    n = len(path)
    if n < 3:
        return np.array([0.0, 0.0, 0.0, 0.0, 0.0, 1.0])
    kap, alat = [], []
    for i in range(1, n - 1):
        k = curvature_loops(path[i - 1], path[i], path[i + 1])
        kap.append(k)
        alat.append((speeds[i] / 3.6) ** 2 * k)
    kap, alat = np.array(kap), np.array(alat)
    viol = int(np.sum(alat > vehicle.max_lateral_accel))
    rate = viol / len(alat) * 100
    jump = float(np.max(np.abs(np.diff(kap)))) if len(kap) > 1 else 0.0
    return np.array([float(np.max(kap)), float(np.max(alat)), float(viol), rate, jump, float(rate < 5)])


def metrics(path, speeds):
    """path length and work time, MLP:1290-1311 (whole-array numpy in the reference too)"""
    if len(path) < 2:
        return 0.0, 0.0
    d = np.sqrt(np.sum(np.diff(path, axis=0) ** 2, axis=1))
    ms = np.maximum((speeds[:-1] + speeds[1:]) / 2 / 3.6, 0.1)
    return float(np.sum(d)), float(np.sum(d / ms))


# ---- whole-array numpy ------------------------------------------------------------------------------------------------------------
def curvature_numpy(path):
    """kappa of every point (0 at both ends), MLP:513-536 over the whole path"""
    n = len(path)
    kap = np.zeros(n)
    if n < 3:
        return kap
    d = np.diff(path, axis=0)
    ds = np.sqrt(d[:, 0] ** 2 + d[:, 1] ** 2)
    th = np.arctan2(d[:, 1], d[:, 0])
    dth = th[1:] - th[:-1]
    dth = np.arctan2(np.sin(dth), np.cos(dth))
    ok = (ds[:-1] >= 1e-6) & (ds[1:] >= 1e-6)
    with np.errstate(divide='ignore', invalid='ignore'):
        kap[1:-1] = np.where(ok, np.abs(2 * dth / (ds[:-1] + ds[1:])), 0.0)
    return kap


def _sweep(u, w, skip):
    """u_i <- min(u_i, u_(i-1) + w_i) along the array, nothing carried across a skipped step: per segment the running minimum of u - S,
    S = the couplings summed from the segment's first point"""
    S = np.cumsum(np.where(skip, 0.0, w))
    S = S - np.maximum.accumulate(np.where(skip, S, 0.0))      # (S never decreases: the last segment start's value carries forward)
    seg = np.cumsum(skip)                                      # segment number of every point (a skipped step opens one)
    t = u - S
    # one minimum.accumulate for all segments: segment s lifted by (last - s) * big, so that nothing of an earlier segment is ever the minimum
    big = (t.max() - t.min()) + 1.0
    lift = (seg[-1] - seg) * big
    return np.minimum.accumulate(t + lift) - lift + S


def smooth_numpy(path, speeds, a_lon):
    n = len(path)
    if n < 2:
        return speeds
    d = np.sqrt(np.sum(np.diff(path, axis=0) ** 2, axis=1))
    skip = np.concatenate([[True], d < 1e-6])             # step (i-1, i) skipped; the first point opens a segment
    w = np.concatenate([[0.0], 2 * a_lon * d])
    u = (speeds / 3.6) ** 2
    f = _sweep(u, w, skip)
    # backward: the same on the reversed arrays (step (i, i+1) belongs to point i there)
    skip_b = np.concatenate([[True], (d < 1e-6)[::-1]])
    w_b = np.concatenate([[0.0], (2 * a_lon * d)[::-1]])
    b = _sweep(f[::-1], w_b, skip_b)[::-1]
    out = speeds.copy()
    low = b < u
    out[low] = np.sqrt(b[low]) * 3.6
    return out


def speed_plan_numpy(path, speeds, vehicle):
    n = len(path)
    if n < 3:
        return speeds, 0
    kap = curvature_numpy(path)
    out = speeds.copy()
    with np.errstate(divide='ignore'):
        v_max = np.sqrt(vehicle.max_lateral_accel / np.where(kap > 1e-6, kap, 1.0)) * vehicle.safety_factor * 3.6
    hit = (kap > 1e-6) & (out > v_max)
    out[hit] = v_max[hit]
    return smooth_numpy(path, out, vehicle.max_longitudinal_accel), int(hit.sum())


def verify_numpy(path, speeds, vehicle):
    n = len(path)
    if n < 3:
        return np.array([0.0, 0.0, 0.0, 0.0, 0.0, 1.0])
    kap = curvature_numpy(path)[1:-1]
    alat = (speeds[1:-1] / 3.6) ** 2 * kap
    viol = int(np.sum(alat > vehicle.max_lateral_accel))
    rate = viol / len(alat) * 100
    jump = float(np.max(np.abs(np.diff(kap)))) if len(kap) > 1 else 0.0
    return np.array([float(kap.max()), float(alat.max()), float(viol), rate, jump, float(rate < 5)])


# ---- the timed leg of bench.py -------------------------------------------------------------------------------------------------------
def stages(path, speeds, vehicle, variant):
    """rows a10-a14 on one path: speed plan (curvature + clamp + sweeps), metrics, verifier.  -> (speeds, stats)"""
    if variant == 'loops':
        v, _ = speed_plan_loops(path, speeds, vehicle)
        st = verify_loops(path, v, vehicle)
    else:
        v, _ = speed_plan_numpy(path, speeds, vehicle)
        st = verify_numpy(path, v, vehicle)
    metrics(path, v)
    return v, st


def time_stages(path, speeds, vehicle, variant, budget_s):
    """-> (points per second, calls, seconds): `stages` repeated on one path for about budget_s seconds, one core"""
    import time
    calls, t = 0, 0.0
    while t < budget_s or calls < 1:
        t0 = time.perf_counter()
        stages(path, speeds, vehicle, variant)
        t += time.perf_counter() - t0
        calls += 1
    return calls * len(path) / t, calls, t
