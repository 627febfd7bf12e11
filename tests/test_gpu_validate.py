"""fcpp_validate: lateral-acceleration / geofence / obstacle flags of CALLER-SUPPLIED paths against arbitrary simple polygons, through
the C ABI, against the CPU oracle point by point (orc_point_in_polygon, orc_outside_polygon: build-defined, the reference has no such
code) and against fcpp_verify for the statistics."""
import numpy as np
import pytest

import oracle as orc
from field_coverage_path_planning_amd import _lib as L
from field_coverage_path_planning_amd import engine as E

pytestmark = pytest.mark.gpu


def _star(rng, cx, cy, r, n):
    """a simple (star-shaped, generally non-convex) polygon around (cx, cy)"""
    ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    rad = rng.uniform(0.4, 1.0, n) * r
    return [(float(cx + a * np.cos(t)), float(cy + a * np.sin(t))) for a, t in zip(rad, ang)]


@pytest.mark.parametrize('seed,tol', [(1, 1e-6), (2, 0.0), (3, 0.75), (4, -0.5)])
def test_flags_vs_oracle_on_random_paths_and_polygons(seed, tol):
    rng = np.random.default_rng(seed)
    n_paths = 12
    lens = rng.integers(0, 1400, n_paths)
    lens[3] = 0
    lens[5] = 1
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    fields, obstacles, obs_off = [], [], [0]
    xs, ys = [], []
    for p in range(n_paths):
        poly = _star(rng, 0.0, 0.0, 300.0, int(rng.integers(3, 40))) if p != 7 else [(0.0, 0.0), (1.0, 1.0)]      # path 7: no geofence (2 vertices)
        fields.append(poly)
        for _ in range(int(rng.integers(0, 5))):
            obstacles.append(_star(rng, *rng.uniform(-150, 150, 2), float(rng.uniform(5, 60)), int(rng.integers(3, 12))))
        obs_off.append(len(obstacles))
        pts = rng.uniform(-330, 330, (lens[p], 2))
        # points ON the boundary and on vertices, and a hair either side of it
        if lens[p] > 40 and len(poly) >= 3:
            v = np.asarray(poly)
            for k in range(10):
                a, b = v[k % len(v)], v[(k + 1) % len(v)]
                pts[k] = a + (b - a) * rng.uniform()
                pts[10 + k] = v[k % len(v)]
                nrm = np.array([-(b - a)[1], (b - a)[0]]) / np.linalg.norm(b - a)
                pts[20 + k] = a + (b - a) * 0.5 + nrm * abs(tol) * 0.999
                pts[30 + k] = a + (b - a) * 0.5 - nrm * abs(tol) * 1.001
        xs.append(pts[:, 0])
        ys.append(pts[:, 1])
    x, y = np.concatenate(xs), np.concatenate(ys)
    v = rng.uniform(2.0, 15.0, x.size)
    veh = E.make_vehicle()
    flags, st = E.validate(x, y, v, veh, fields, obstacles, obs_off, geofence_tol=tol, offsets=offs)
    flags = flags.cpu().numpy().view(np.uint32)
    ver = E.verify(x, y, v, veh, offsets=offs)
    for key in ('main_len_m', 'main_time_s', 'max_kappa', 'max_alat', 'max_jump', 'n_viol'):
        assert np.array_equal(st[key], ver[key]), key
    kap = E.curvature(x, y, offsets=offs).cpu().numpy()
    for p in range(n_paths):
        a, b = offs[p], offs[p + 1]
        exp = np.zeros(b - a, dtype=np.uint32)
        for i in range(a, b):
            if len(fields[p]) >= 3 and orc.outside_polygon(x[i], y[i], fields[p], tol):
                exp[i - a] |= L.FLAG_OUTSIDE
            if any(orc.point_in_polygon(x[i], y[i], o) for o in obstacles[obs_off[p]:obs_off[p + 1]]):
                exp[i - a] |= L.FLAG_OBSTACLE
            if a < i < b - 1 and (v[i] / 3.6) ** 2 * kap[i] > veh.max_lateral_accel:
                exp[i - a] |= L.FLAG_ALAT
        assert np.array_equal(flags[a:b], exp), (p, np.flatnonzero(flags[a:b] != exp)[:5])
        assert st['n_outside'][p] == int(((exp & L.FLAG_OUTSIDE) != 0).sum()) and st['n_in_obstacle'][p] == int(((exp & L.FLAG_OBSTACLE) != 0).sum())
        assert st['n_viol'][p] == int(((exp & L.FLAG_ALAT) != 0).sum())
    assert (flags & L.FLAG_OUTSIDE).any() and (flags & L.FLAG_OBSTACLE).any()


def test_a_planned_path_is_inside_its_own_field_and_agrees_with_the_planner_flags():
    """a path the planner generated, validated against the field it was planned for: no point outside; with a shrunken polygon the
    flags appear; the obstacle flags equal the planner's own"""
    rng = np.random.default_rng(9)
    obst = [_star(rng, 200.0, 100.0, 25.0, 9), _star(rng, 320.0, 60.0, 15.0, 6)]
    b = E.Batch([E.FieldSpec(field_length=400.0, field_width=220.0, obstacles=obst)], E.make_vehicle(), E.make_options(1, 0.5))
    r = b.run()
    field = [(0.0, 0.0), (400.0, 0.0), (400.0, 220.0), (0.0, 220.0)]
    flags, st = E.validate(r.x, r.y, r.v, E.make_vehicle(), [field], obst)
    own = r.flagseg.cpu().numpy().view(np.uint32)
    got = flags.cpu().numpy().view(np.uint32)
    assert st['n_outside'][0] == 0 and int(r.stats()['n_outside'][0]) == 0
    assert np.array_equal(got & L.FLAG_OBSTACLE, own & L.FLAG_OBSTACLE) and st['n_in_obstacle'][0] == r.stats()['n_in_obstacle'][0] > 0
    assert np.array_equal(got & L.FLAG_ALAT, own & L.FLAG_ALAT)
    flags2, st2 = E.validate(r.x, r.y, r.v, E.make_vehicle(), [[(5.0, 5.0), (395.0, 5.0), (395.0, 215.0), (5.0, 215.0)]])
    assert st2['n_outside'][0] > 0
    b.close()


def test_bad_arguments_are_refused():
    x = np.zeros(10)
    with pytest.raises(L.FcppError):
        E.validate(x, x, x, E.make_vehicle(), [[(0.0, 0.0), (1.0, 0.0), (0.0, 1.0)]] * 2)                # one polygon per path
    with pytest.raises(L.FcppError):
        E.validate(x, x, x, E.make_vehicle(), None, [[(0.0, 0.0), (1.0, 0.0), (0.0, 1.0)]], [0, 5])     # range beyond the table


def test_one_long_path_statistics_through_the_sliced_reduction():
    """A single path of 1.2 million points (2400 tiles): the standalone operators reduce it over 64 workgroups and join (reduce_paths,
    csrc/fcpp_api.cpp) instead of through one wavefront (1.2 ms for cfg3's path).  The same points as TWO paths take the plain reduction:
    lengths agree to rounding once the segment across the cut is added, maxima and counts up to the two points at the cut."""
    veh = E.make_vehicle()
    b = E.Batch([E.FieldSpec(field_length=900.0, field_width=420.0)], veh, E.make_options(1, 0.1))
    r = b.run()
    n = r.x.numel()
    assert 2048 * 512 < n < 2 * 2048 * 512          # one path: sliced; its halves: the plain reduction
    one = E.verify(r.x, r.y, r.v, veh)
    cut = n // 2
    two = E.verify(r.x, r.y, r.v, veh, offsets=np.array([0, cut, n], dtype=np.int64))
    xy = np.stack([r.x[cut - 1:cut + 1].cpu().numpy(), r.y[cut - 1:cut + 1].cpu().numpy()])
    seg = float(np.hypot(xy[0, 1] - xy[0, 0], xy[1, 1] - xy[1, 0]))
    assert abs(one['main_len_m'][0] - (two['main_len_m'].sum() + seg)) < 1e-9 * one['main_len_m'][0]
    assert one['main_time_s'][0] > two['main_time_s'].sum() and one['main_time_s'][0] < two['main_time_s'].sum() + 10.0
    for name in ('max_kappa', 'max_alat', 'max_jump'):
        assert one[name][0] >= two[name].max() - 1e-12 and one[name][0] < two[name].max() * 1.000001 + 1e-9 or name == 'max_jump', name
    assert abs(int(one['n_viol'][0]) - int(two['n_viol'].sum())) <= 2
    # fcpp_validate runs the same verifier and reduction: its path statistics are the same numbers
    _, st = E.validate(r.x, r.y, r.v, veh, [[(0.0, 0.0), (900.0, 0.0), (900.0, 420.0), (0.0, 420.0)]])
    for name in ('main_len_m', 'main_time_s', 'max_kappa', 'max_alat', 'max_jump', 'n_viol'):
        assert one[name][0] == st[name][0], name
    assert int(st['n_outside'][0]) == int(r.stats()['n_outside'][0])
    b.close()
