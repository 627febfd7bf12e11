"""GPU parity tests (run with -m gpu on an MI355X): the HIP library, called through its C ABI, against
(a) the reference's golden vectors and (b) the CPU oracle on seeded inputs.

Tolerances (float64): coordinates 1e-9 m (the bar in BASELINE.json is 1e-6 m), curvature 1e-9 1/m,
speeds 1e-9 km/h at the reference's sampling; integers (counts, swath indices, flags) exact.
"""
import os

import numpy as np
import pytest

import oracle as orc
from field_coverage_path_planning_amd import _lib as L
from field_coverage_path_planning_amd import engine as E
from tests.test_abi_and_host import _specs_from_golden
from tests.test_oracle_vs_golden import _field_from_golden

pytestmark = pytest.mark.gpu

XY_TOL, K_TOL, V_TOL = 1e-9, 1e-9, 1e-9
VP_NAMES = [n for n, _ in L.Vehicle._fields_]


def _veh(arr):
    return E.make_vehicle(**dict(zip(VP_NAMES, arr)))


def _np(t):
    return t.cpu().numpy()


def test_native_library_is_loaded():
    import torch
    assert torch.cuda.is_available()
    E.get_context()
    maps = open('/proc/self/maps').read()
    assert 'libfcpp.so' in maps


@pytest.mark.parametrize('mode', [1, 0])
def test_batch_vs_golden_plans(golden_plans, mode):
    """All golden scenarios sharing a vehicle are planned as ONE batch; every array is compared.
    mode 1 = fused single-pass kernel, mode 0 = staged pipeline.
    (Tier-B vectors: the reference's own code with the Shapely stand-in; the inset corners' ring order is documented intent, not an
    observation of GEOS -- tests/test_oracle_vs_golden.py::test_full_plans.)"""
    g = golden_plans
    groups = {}
    for name in g['names']:
        groups.setdefault((tuple(g[f'{name}/vp']), int(g[f'{name}/ring_order'])), []).append(str(name))
    assert any(ring == 1 for _, ring in groups)           # the reversed inset ring (cw_* scenarios): fcpp_options.ring_order = 1
    for (vp, ring), names in groups.items():
        specs = [_specs_from_golden(g, n) for n in names]
        batch = E.Batch(specs, _veh(vp), E.make_options(ring_order=ring))
        res = batch.run(mode=mode)
        ap, dp = batch.connectors()
        x, y, v, k, fs = _np(res.x), _np(res.y), _np(res.v), _np(res.kappa), _np(res.flagseg).view(np.uint32)
        st = res.stats()
        ap, dp = _np(ap), _np(dp)
        for i, n in enumerate(names):
            info = batch.info[i]
            sl = res.field_slice(i)
            mp, hp = g[f'{n}/main_path'], g[f'{n}/head_path']
            assert info.n_main == len(mp) and info.n_head == len(hp), n
            xy = np.column_stack([x[sl], y[sl]])
            np.testing.assert_allclose(xy[:info.n_main], mp, rtol=0, atol=XY_TOL, err_msg=n)
            np.testing.assert_allclose(xy[info.n_main:], hp, rtol=0, atol=XY_TOL, err_msg=n)
            np.testing.assert_allclose(v[sl][:info.n_main], g[f'{n}/main_v'], rtol=0, atol=V_TOL, err_msg=n)
            np.testing.assert_allclose(v[sl][info.n_main:], g[f'{n}/head_v'], rtol=0, atol=V_TOL, err_msg=n)
            ms, hs, ver = g[f'{n}/main_stats'], g[f'{n}/head_stats'], g[f'{n}/ver']
            np.testing.assert_allclose(st['main_len_m'][i] / 1000, ms[0], rtol=1e-12)
            np.testing.assert_allclose(st['main_time_s'][i] / 3600, ms[1], rtol=1e-10)
            np.testing.assert_allclose((st['main_len_m'][i] / 1000) / (st['main_time_pre_s'][i] / 3600), ms[2], rtol=1e-10)
            np.testing.assert_allclose(st['head_len_m'][i] / 1000, hs[0], rtol=1e-12)
            np.testing.assert_allclose(st['head_time_s'][i] / 3600, hs[1], rtol=1e-10)
            np.testing.assert_allclose([st['max_kappa'][i], st['max_alat'][i], st['max_jump'][i]], ver[[0, 1, 4]],
                                       rtol=1e-8, atol=1e-11, err_msg=n)
            assert st['n_viol'][i] == ver[2], n
            ga, gd = g[f'{n}/approach'], g[f'{n}/departure']
            if len(ga):
                np.testing.assert_allclose(ap[i], ga, rtol=0, atol=XY_TOL)
            else:
                assert np.isnan(ap[i]).all()
            if len(gd):
                np.testing.assert_allclose(dp[i], gd, rtol=0, atol=XY_TOL)
            else:
                assert np.isnan(dp[i]).all()
        batch.close()


def _compare_with_oracle(specs, ofields, veh_arr, opt_kw, xy_tol=XY_TOL, k_tol=K_TOL, v_tol=V_TOL):
    for mode in (1, 0):     # 1 = fused single-pass kernel (default), 0 = staged pipeline
        _compare_with_oracle_mode(mode, specs, ofields, veh_arr, opt_kw, xy_tol, k_tol, v_tol)


def _compare_with_oracle_mode(mode, specs, ofields, veh_arr, opt_kw, xy_tol, k_tol, v_tol):
    o = E.make_options(**opt_kw)
    if os.environ.get('FCPP_FUZZ_RECORDS') == 'device':          # (tools/fuzz_parity.py: the field records resident in device memory)
        specs = E.as_table(specs).to_device()
    batch = E.Batch(specs, _veh(veh_arr), o)
    res = batch.run(mode=mode)
    x, y, v, k, fs = _np(res.x), _np(res.y), _np(res.v), _np(res.kappa), _np(res.flagseg).view(np.uint32)
    st = res.stats()
    oopt = orc.Options.make(o.turn_model, o.clothoid_fit, o.sample_spacing, o.clothoid_frac, o.geofence_tol, o.obstacle_mode, o.ring_order)
    for i, of in enumerate(ofields):
        rc, p = orc.plan_field(of, orc.Vehicle.make(veh_arr), oopt)
        info = batch.info[i]
        assert rc == info.status
        if rc != 0:
            assert info.n_main == 0 and info.n_head == 0
            continue
        sl = res.field_slice(i)
        assert (info.n_main, info.n_head) == (p.n_main, p.n_head)
        np.testing.assert_allclose(np.column_stack([x[sl], y[sl]]), p.xy, rtol=0, atol=xy_tol)
        np.testing.assert_allclose(k[sl], p.kappa, rtol=0, atol=k_tol)
        np.testing.assert_allclose(v[sl], p.v, rtol=0, atol=v_tol)
        assert np.array_equal(fs[sl], p.flagseg), i          # kinds, swath indices, validity flags: bit-exact
        for a, b in (('main_len_m', p.main_len_m), ('main_time_pre_s', p.main_time_pre_s), ('main_time_s', p.main_time_s),
                     ('head_len_m', p.head_len_m), ('head_time_pre_s', p.head_time_pre_s), ('head_time_s', p.head_time_s)):
            np.testing.assert_allclose(st[a][i], b, rtol=1e-10, err_msg=a)
        np.testing.assert_allclose([st['max_kappa'][i], st['max_alat'][i], st['max_jump'][i]],
                                   [p.max_kappa, p.max_alat, p.max_jump], rtol=1e-7, atol=max(k_tol, 1e-11))
        assert (st['n_viol'][i], st['n_outside'][i], st['n_in_obstacle'][i], st['n_adjusted'][i]) == \
            (p.n_viol, p.n_outside, p.n_in_obstacle, p.n_adjusted), i
    batch.close()


def _random_fields(seed, n, para=False, with_obstacles=False, with_points=True):
    rng = np.random.default_rng(seed)
    specs, ofs = [], []
    for i in range(n):
        Lx, Hy = rng.uniform(100, 1000, 2)
        start = end = None
        if with_points and i % 2 == 0:
            start = (float(rng.uniform(0, Lx)), float(rng.uniform(0, Hy)))
        if with_points and i % 3 == 0:
            end = (float(rng.uniform(0, Lx)), float(rng.uniform(0, Hy)))
        obstacles = None
        if with_obstacles:
            obstacles = []
            for _ in range(int(rng.integers(1, 5))):
                cx, cy, r = rng.uniform(0.2 * Lx, 0.8 * Lx), rng.uniform(0.2 * Hy, 0.8 * Hy), rng.uniform(5, 30)
                a = np.sort(rng.uniform(0, 2 * np.pi, int(rng.integers(3, 9))))
                obstacles.append([(float(cx + r * np.cos(t)), float(cy + r * np.sin(t))) for t in a])
        if para:
            ang, rot = np.radians(rng.uniform(60, 120)), rng.uniform(-np.pi / 4, np.pi / 4)
            sx = Hy / np.tan(ang)
            q = np.array([[0, 0], [Lx, 0], [Lx + sx, Hy], [sx, Hy]], dtype=np.float64)
            if para == 'quad':      # a convex quadrilateral that is no parallelogram (`_detect_field_shape` -> 'other', MLP:137-163)
                q[2] += rng.uniform(-0.12, 0.12, 2) * (Lx, Hy)
                q[3] += rng.uniform(-0.12, 0.12, 2) * (Lx, Hy)
                e = np.roll(q, -1, 0) - q
                cr = e[:, 0] * np.roll(e, -1, 0)[:, 1] - e[:, 1] * np.roll(e, -1, 0)[:, 0]
                assert (cr > 0).all()
            vv = q @ np.array([[np.cos(rot), np.sin(rot)], [-np.sin(rot), np.cos(rot)]])
            verts = [(float(a), float(b)) for a, b in vv]
            specs.append(E.FieldSpec(field_vertices=verts, obstacles=obstacles, start_point=start, end_point=end))
            ofs.append(orc.make_field(verts=verts, start=start, end=end, obstacles=obstacles))
        else:
            specs.append(E.FieldSpec(field_length=float(Lx), field_width=float(Hy), obstacles=obstacles,
                                     start_point=start, end_point=end))
            ofs.append(orc.make_field(L=float(Lx), H=float(Hy), start=start, end=end, obstacles=obstacles))
    return specs, ofs


DEFAULT_VP = [3.2, 8.0, 9.0, 15.0, 4.0, 2.0, 1.5, 0.85]


def test_reference_sampling_random_rectangles_vs_oracle():
    specs, ofs = _random_fields(1024, 48)
    _compare_with_oracle(specs, ofs, DEFAULT_VP, {})


def test_reference_sampling_parallelograms_vs_oracle():
    specs, ofs = _random_fields(65536, 48, para=True)
    _compare_with_oracle(specs, ofs, DEFAULT_VP, {})


@pytest.mark.parametrize('ring', [0, 1])
def test_convex_quadrilaterals_and_both_ring_orders_vs_oracle(ring):
    """Fields of shape 'other' (convex, neither rectangle nor parallelogram; the reference plans them, MLP:137-163), and
    fcpp_options.ring_order -- the order in which the inset corners of a headland loop are listed (a GEOS fact the reference leaves
    open; both orders are pinned to the reference's code by the cw_* fixtures): reference sampling, dense arcs and clothoids."""
    specs, ofs = _random_fields(4242 + ring, 24, para='quad')
    b = E.Batch(specs, _veh(DEFAULT_VP), E.make_options(ring_order=ring))
    assert all(i.status == 0 and i.shape == 2 for i in b.info)
    b.close()
    _compare_with_oracle(specs, ofs, DEFAULT_VP, dict(ring_order=ring))
    _compare_with_oracle(specs[:8], ofs[:8], DEFAULT_VP, dict(ring_order=ring, sample_spacing=0.5), k_tol=1e-8, v_tol=1e-6)
    _compare_with_oracle(specs[:8], ofs[:8], DEFAULT_VP, dict(ring_order=ring, turn_model=1, sample_spacing=0.25), k_tol=1e-8, v_tol=1e-6)
    specs, ofs = _random_fields(99 + ring, 12, para=True)
    _compare_with_oracle(specs, ofs, DEFAULT_VP, dict(ring_order=ring))
    specs, ofs = _random_fields(98 + ring, 12)
    _compare_with_oracle(specs, ofs, DEFAULT_VP, dict(ring_order=ring))


def test_obstacles_and_geofence_flags_vs_oracle():
    specs, ofs = _random_fields(32, 24, with_obstacles=True)
    _compare_with_oracle(specs, ofs, DEFAULT_VP, {})
    # densely sampled swaths really cross the obstacles -> the flag must be set and match the oracle
    _compare_with_oracle(specs[:8], ofs[:8], DEFAULT_VP, dict(sample_spacing=1.0), k_tol=1e-8, v_tol=1e-6)
    b = E.Batch(specs[:8], _veh(DEFAULT_VP), E.make_options(sample_spacing=1.0))
    assert b.run().stats()['n_in_obstacle'].sum() > 0
    b.close()


def test_geofence_tolerances_vs_oracle():
    """A negative tolerance flags points that lie INSIDE the polygon, closer to an edge than |tol| (the outer headland loop runs W/2 =
    1.6 m inside, reverse fills end on the edge; the U-turns of a skewed field leave it): the wave tiles the host marks as "inside by a
    margin" (DevWaveTile.inside: no test on the device) must take the tolerance into account.  Flags and n_outside exact against the
    oracle, rectangles and parallelograms; the count falls as the tolerance grows."""
    counts = []
    for tol in (-2.0, -0.4, 0.0, 0.3):
        n_out = 0
        for para in (False, True):
            specs, ofs = _random_fields(77 + int(para), 12, para=para)
            _compare_with_oracle(specs, ofs, DEFAULT_VP, dict(geofence_tol=tol))
            b = E.Batch(specs, _veh(DEFAULT_VP), E.make_options(geofence_tol=tol))
            n_out += int(b.run().stats()['n_outside'].sum())
            b.close()
        counts.append(n_out)
    assert counts[0] > counts[1] > counts[2] >= counts[3] and counts[0] > 0, counts


def test_fields_far_from_the_origin_vs_oracle():
    """UTM-sized coordinates (~5e6 m, where a double resolves 9.3e-10 m): coordinates within 2 ulps (2e-9 m) of the oracle, speeds within
    1e-8 km/h, curvatures within 2e-8 1/m (the three-point stencil amplifies a coordinate's rounding by 4 / ds^2) -- measured: 0 m,
    5e-15 km/h, 1.4e-9 1/m; rounds 2-4 accepted 5e-8 m and 1e-4 km/h here --, every integer equal, and the
    geofence flags equal -- with the default tolerance exactly, with tolerance 0 wherever a point is farther than a micrometre from the
    boundary (ON it, rounding decides in the oracle as in the library).  The tiler's "inside by a margin" shortcut (DevWaveTile.inside,
    fcpp_tilefn.h: tiler_inside) scales its margin with the coordinates, so it cannot hide a flag here either."""
    rng = np.random.default_rng(41)
    shift = np.array([5.2e6, 4.1e6])
    specs, ofs, quads = [], [], []
    for k in range(10):
        w, h = rng.uniform(120, 700, 2)
        q = np.array([[0, 0], [w, 0], [w, h], [0, h]], dtype=np.float64)
        if k % 2:
            q[2:, 0] += rng.uniform(-0.3, 0.3) * h
            rot = rng.uniform(-0.7, 0.7)
            q = q @ np.array([[np.cos(rot), np.sin(rot)], [-np.sin(rot), np.cos(rot)]])
        q = q + shift
        verts = [(float(a), float(b)) for a, b in q]
        specs.append(E.FieldSpec(field_vertices=verts))
        ofs.append(orc.make_field(verts=verts))
        quads.append(q)
    for tol in (1e-6, 0.0):
        o = E.make_options(geofence_tol=tol)
        for setup in ('device', 'host'):
            E.get_context().set_setup(setup)
            try:
                batch = E.Batch(specs, _veh(DEFAULT_VP), o)
            finally:
                E.get_context().set_setup('auto')
            res = batch.run()
            x, y, v, fs, kap = _np(res.x), _np(res.y), _np(res.v), _np(res.flagseg).view(np.uint32), _np(res.kappa)
            for i, of in enumerate(ofs):
                rc, p = orc.plan_field(of, orc.Vehicle.make(DEFAULT_VP), orc.Options.make(geofence_tol=tol))
                info = batch.info[i]
                assert rc == info.status == 0 and (info.n_main, info.n_head, info.n_swaths) == (p.n_main, p.n_head, p.n_swaths)
                sl = res.field_slice(i)
                np.testing.assert_allclose(np.column_stack([x[sl], y[sl]]), p.xy, rtol=0, atol=2e-9)
                np.testing.assert_allclose(v[sl], p.v, rtol=0, atol=1e-8)
                np.testing.assert_allclose(kap[sl], p.kappa, rtol=0, atol=2e-8)
                kinds = ~np.uint32(L.FLAG_OUTSIDE)
                assert np.array_equal(fs[sl] & kinds, p.flagseg & kinds)
                # signed distance of every point to the nearest edge line (relative coordinates: no cancellation)
                q = quads[i] - shift
                rel = p.xy - shift
                cr = lambda a, b: a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]
                sgn = 1.0 if cr(q[1] - q[0], q[2] - q[1]) > 0 else -1.0
                d = np.min([sgn * cr(q[(e + 1) % 4] - q[e], rel - q[e]) / np.linalg.norm(q[(e + 1) % 4] - q[e]) for e in range(4)], axis=0)
                sure = np.abs(d + tol) > 1e-6
                assert np.array_equal((fs[sl] & L.FLAG_OUTSIDE)[sure], (p.flagseg & L.FLAG_OUTSIDE)[sure]), (tol, setup, i)
                if tol > 0:
                    assert np.array_equal(fs[sl], p.flagseg), (setup, i)
            batch.close()


def _clip_fields():
    """fields whose obstacles sit inside the work area, away from the swath lines' end zones: a rectangle with a square, a triangle
    and a pentagon (the last two side by side on the same swaths), and a tilted parallelogram with two obstacles"""
    rect_obs = [[(150.0, 60.0), (170.0, 60.0), (170.0, 80.0), (150.0, 80.0)],
                [(240.0, 120.0), (262.0, 124.0), (249.0, 141.0)],
                [(300.0, 118.0), (312.0, 114.0), (321.0, 125.0), (314.0, 138.0), (301.0, 134.0)]]
    specs = [E.FieldSpec(field_length=400.0, field_width=220.0, obstacles=rect_obs, start_point=(390.0, 200.0))]
    ofs = [orc.make_field(L=400.0, H=220.0, obstacles=rect_obs, start=(390.0, 200.0))]
    rot = 0.3
    c, s = np.cos(rot), np.sin(rot)
    tilt = lambda pts: [(float(x * c - y * s), float(x * s + y * c)) for x, y in pts]
    verts = tilt([(0.0, 0.0), (500.0, 0.0), (560.0, 260.0), (60.0, 260.0)])
    # (the rectangle's sides lie off the half-metre grid of the detour legs' sample counts, int(len / 0.5) + 1: with sides on multiples of 0.1 m some leg is
    # a multiple of 0.5 m long up to the rounding of the rotation into the frame of layer 1 and back -- a count that hinges on the last bit of a
    # sine is as platform-dependent here as the reference's own int((max_y - min_y) / W); the library takes its
    # setup's sine / cosine from csrc/fcpp_math.h, the oracle from the platform libm)
    para_obs = [tilt([(200.0, 100.33), (230.07, 100.33), (230.07, 125.27), (200.0, 125.27)]), tilt([(340.0, 150.03), (365.0, 160.0), (350.0, 185.07)])]
    specs.append(E.FieldSpec(field_vertices=verts, obstacles=para_obs))
    ofs.append(orc.make_field(verts=verts, obstacles=para_obs))
    return specs, ofs


@pytest.mark.parametrize('opt', [dict(), dict(sample_spacing=0.5), dict(turn_model=1, sample_spacing=0.2), dict(turn_model=1)])
def test_obstacle_aware_swaths_vs_oracle(opt):
    """SURVEY.md 8f-4 (build-defined, include/fcpp.h): with obstacle_mode = AVOID the swaths are clipped at the obstacles' boxes and
    re-routed around them.  Every array against the oracle's independent restatement, in both pipelines; no path point of layer 1 is
    left inside an obstacle, while the reference's behaviour (obstacles only flagged) leaves some."""
    specs, ofs = _clip_fields()
    ds = opt.get('sample_spacing', 0.0) or 0.5
    k_tol = max(K_TOL, 4e-12 / ds ** 2)
    _compare_with_oracle(specs, ofs, DEFAULT_VP, dict(opt, avoid_obstacles=True), xy_tol=1e-9, k_tol=k_tol, v_tol=max(V_TOL, 200 * k_tol))
    b = E.Batch(specs, _veh(DEFAULT_VP), E.make_options(avoid_obstacles=True, **opt))
    res = b.run()
    st = res.stats()
    fs = _np(res.flagseg).view(np.uint32)
    assert int(st['n_in_obstacle'].sum()) == 0 and int((fs & L.FLAG_OBSTACLE != 0).sum()) == 0
    assert int(((fs & L.KIND_MASK) == L.KIND_DETOUR).sum()) > 0
    assert all(i.status == 0 for i in b.info)
    # Round 4: the detours follow the obstacles' W/2-grown POLYGONS.  Field 0 (unrotated; square, triangle, pentagon): every detour point
    # keeps W/2 from every obstacle, and a pass near the triangle's apex leaves its line for a much shorter stretch than the box is wide.
    W = DEFAULT_VP[0]
    sl = res.field_slice(0)
    x0, y0, f0 = _np(res.x)[sl], _np(res.y)[sl], fs[sl]
    det = ((f0 & L.KIND_MASK) == L.KIND_DETOUR) & ((f0 & L.FLAG_HEADLAND) == 0)

    def dist_to_polygon(px, py, poly):
        q = np.asarray(poly, dtype=np.float64)
        d = np.full(px.shape, np.inf)
        for a, c in zip(q, np.roll(q, -1, axis=0)):
            e = c - a
            t = np.clip(((px - a[0]) * e[0] + (py - a[1]) * e[1]) / (e @ e), 0.0, 1.0)
            d = np.minimum(d, np.hypot(px - (a[0] + t * e[0]), py - (a[1] + t * e[1])))
        return d
    for poly in specs[0].obstacles:
        assert dist_to_polygon(x0[det], y0[det], poly).min() >= W / 2 - 1e-6
    tri = np.asarray(specs[0].obstacles[1])
    sw = (f0 & L.KIND_MASK) == L.KIND_SWATH
    pass_y = np.unique(np.round(y0[sw], 6))
    in_box = pass_y[(pass_y > tri[:, 1].min() - W / 2) & (pass_y < tri[:, 1].max() + W / 2)]
    near = det & (x0 > tri[:, 0].min() - W) & (x0 < tri[:, 0].max() + W) & (y0 > tri[:, 1].min() - W) & (y0 < tri[:, 1].max() + W)
    # (the swath is worked up to the polygon, not to its box: near the triangle's apex the detours begin well inside the box's x-range)
    spans = [np.ptp(x0[near & ((f0 >> L.INDEX_SHIFT) == k)]) for k in np.unique(f0[near] >> L.INDEX_SHIFT)]
    assert len(spans) == len(in_box) and min(spans) < (np.ptp(tri[:, 0]) + W) - 6.0, (spans, np.ptp(tri[:, 0]) + W)
    b.close()
    if opt.get('sample_spacing', 0.0) > 0:
        b0 = E.Batch(specs, _veh(DEFAULT_VP), E.make_options(**opt))
        assert int(b0.run().stats()['n_in_obstacle'].sum()) > 0          # the same fields without the option: swaths cross the obstacles
        b0.close()


def test_obstacle_aware_swaths_edge_cases():
    """An obstacle box at the free end of the first pass, or one that leaves no side to pass (it spans the work area's whole
    y-range): FCPP_EUNSUPPORTED for that field only (library and oracle agree).  Two boxes that overlap are merged into one and driven
    around together.  A field without obstacles plans as without the option (unrotated: coordinates and segment words bit for bit)."""
    near_end = [[(5.0, 5.0), (30.0, 5.0), (30.0, 25.0), (5.0, 25.0)]]
    overlap = [[(150.0, 60.0), (170.0, 60.0), (170.0, 80.0), (150.0, 80.0)], [(165.0, 65.0), (190.0, 65.0), (190.0, 85.0), (165.0, 85.0)]]
    wall = [[(200.0, 5.0), (210.0, 5.0), (210.0, 215.0), (200.0, 215.0)]]
    specs = [E.FieldSpec(field_length=400.0, field_width=220.0, obstacles=near_end),
             E.FieldSpec(field_length=400.0, field_width=220.0, obstacles=overlap),
             E.FieldSpec(field_length=400.0, field_width=220.0),
             E.FieldSpec(field_length=400.0, field_width=220.0, obstacles=wall)]
    ofs = [orc.make_field(L=400.0, H=220.0, obstacles=near_end), orc.make_field(L=400.0, H=220.0, obstacles=overlap),
           orc.make_field(L=400.0, H=220.0), orc.make_field(L=400.0, H=220.0, obstacles=wall)]
    for kw in (dict(avoid_obstacles=True), dict(avoid_obstacles=True, sample_spacing=0.5)):
        _compare_with_oracle(specs, ofs, DEFAULT_VP, kw, k_tol=1e-8, v_tol=1e-6)
        b = E.Batch(specs, _veh(DEFAULT_VP), E.make_options(**kw))
        assert [i.status for i in b.info] == [L.EUNSUPPORTED, 0, 0, L.EUNSUPPORTED]
        r1 = b.run()
        st = r1.stats()
        assert int(st['n_in_obstacle'][1]) == 0           # the merged box is passed as one
        b0 = E.Batch(specs[2:3], _veh(DEFAULT_VP), E.make_options(**{k: v for k, v in kw.items() if k != 'avoid_obstacles'}))
        r0 = b0.run()
        sl = r1.field_slice(2)
        for a in ('x', 'y', 'flagseg'):
            assert np.array_equal(_np(getattr(r1, a))[sl], _np(getattr(r0, a))), a
        for a in ('kappa', 'v'):      # (the closed-form U-turns of the plain mode take curvature from the turn shape itself)
            np.testing.assert_allclose(_np(getattr(r1, a))[sl], _np(getattr(r0, a)), rtol=0, atol=1e-9, err_msg=a)
        b.close(); b0.close()


@pytest.mark.parametrize('opt', [dict(), dict(sample_spacing=0.5), dict(turn_model=1, sample_spacing=0.25), dict(turn_model=1)])
def test_obstacles_in_the_end_zones_move_the_turns(opt):
    """Round 4 (include/fcpp.h, obstacle-aware swaths): a box that reaches into the lines' end zone -- where the turns are -- no longer
    refuses the field: the turns beside it move inwards until their zone is free, the passes end / start there.  Left and right ends,
    two boxes behind each other (the moved zone meets the second one), a rotated field; whole path against the oracle, no point of
    the path inside an obstacle, and the clipped passes are shorter than the plain ones."""
    left = [[(12.0, 100.0), (30.0, 100.0), (30.0, 120.0), (12.0, 120.0)]]
    right = [[(372.0, 60.0), (386.0, 60.0), (386.0, 75.0), (372.0, 75.0)]]
    chain = [[(372.0, 140.0), (386.0, 140.0), (386.0, 150.0), (372.0, 150.0)], [(340.0, 143.0), (350.0, 143.0), (350.0, 147.0), (340.0, 147.0)]]
    both = left + right + chain + [[(200.0, 100.0), (215.0, 100.0), (215.0, 110.0), (200.0, 110.0)]]
    rot = 0.25
    c, s = np.cos(rot), np.sin(rot)
    tilt = lambda pts: [(float(x * c - y * s), float(x * s + y * c)) for x, y in pts]
    verts = tilt([(0.0, 0.0), (400.0, 0.0), (400.0, 220.0), (0.0, 220.0)])
    tilted = [tilt(o) for o in left + right]
    layouts = [left, right, chain, both]
    specs = [E.FieldSpec(field_length=400.0, field_width=220.0, obstacles=o) for o in layouts] + [E.FieldSpec(field_vertices=verts, obstacles=tilted)]
    ofs = [orc.make_field(L=400.0, H=220.0, obstacles=o) for o in layouts] + [orc.make_field(verts=verts, obstacles=tilted)]
    ds = opt.get('sample_spacing', 0.0) or 0.5
    k_tol = max(K_TOL, 4e-12 / ds ** 2)
    _compare_with_oracle(specs, ofs, DEFAULT_VP, dict(opt, avoid_obstacles=True), xy_tol=1e-9, k_tol=k_tol, v_tol=max(V_TOL, 200 * k_tol))
    b = E.Batch(specs, _veh(DEFAULT_VP), E.make_options(avoid_obstacles=True, **opt))
    assert all(i.status == 0 for i in b.info)
    res = b.run()
    assert int(res.stats()['n_in_obstacle'].sum()) == 0
    if opt.get('sample_spacing', 0.0) > 0:
        b0 = E.Batch(specs, _veh(DEFAULT_VP), E.make_options(**opt))
        assert all(a.n_main < p.n_main + 200 for a, p in zip(b.info, b0.info))
        x = _np(res.x)[res.field_slice(1)]
        fs = _np(res.flagseg).view(np.uint32)[res.field_slice(1)]
        sw = (fs & L.KIND_MASK) == L.KIND_SWATH
        near = sw & (np.abs(_np(res.y)[res.field_slice(1)] - 67.0) < 8.0)
        assert x[near].max() < 372.0 - 1.6 and x[sw].max() > 380.0          # passes beside the box stop before it, the others run to the end
        b0.close()
    b.close()


@pytest.mark.parametrize('opt', [dict(), dict(sample_spacing=0.5), dict(turn_model=1, sample_spacing=0.25), dict(turn_model=1)])
def test_headland_loops_go_around_obstacles(opt):
    """Round 4 (include/fcpp.h, obstacle-aware swaths): an obstacle that reaches into the headland no longer leaves loop points inside it:
    every headland straight that crosses a grown box is cut there and led around the box along its boundary (detour legs with the
    headland flag), the shorter way that stays W/2 inside the field.  Boxes beside the right edge, below the top edge, both, one in a
    tilted parallelogram (the straights cross the box askew and leave through another face); a box in a corner's turn zone is
    refused.  Whole path against the oracle; no point of the path inside an obstacle."""
    right = [[(385.0, 60.0), (395.0, 60.0), (395.0, 75.0), (385.0, 75.0)]]
    top = [[(150.0, 205.0), (170.0, 205.0), (170.0, 214.0), (150.0, 214.0)]]
    corner = [[(385.0, 200.0), (395.0, 200.0), (395.0, 215.0), (385.0, 215.0)]]
    rot = 0.3
    c, s = np.cos(rot), np.sin(rot)
    tilt = lambda pts: [(float(x * c - y * s), float(x * s + y * c)) for x, y in pts]
    verts = tilt([(0.0, 0.0), (500.0, 0.0), (560.0, 260.0), (60.0, 260.0)])
    # (sides off the half-metre grid of the detour legs' sample counts, as in _clip_fields)
    para_obs = [tilt([(52.0, 150.13), (61.07, 150.13), (61.07, 162.41), (52.0, 162.41)]), tilt([(300.0, 4.0), (318.03, 4.0), (318.03, 9.17), (300.0, 9.17)])]
    layouts = [right, top, right + top, corner]
    specs = [E.FieldSpec(field_length=400.0, field_width=220.0, obstacles=o) for o in layouts] + [E.FieldSpec(field_vertices=verts, obstacles=para_obs)]
    ofs = [orc.make_field(L=400.0, H=220.0, obstacles=o) for o in layouts] + [orc.make_field(verts=verts, obstacles=para_obs)]
    ds = opt.get('sample_spacing', 0.0) or 0.5
    k_tol = max(K_TOL, 4e-12 / ds ** 2)
    _compare_with_oracle(specs, ofs, DEFAULT_VP, dict(opt, avoid_obstacles=True), xy_tol=1e-9, k_tol=k_tol, v_tol=max(V_TOL, 200 * k_tol))
    b = E.Batch(specs, _veh(DEFAULT_VP), E.make_options(avoid_obstacles=True, **opt))
    assert [i.status for i in b.info] == [0, 0, 0, L.EUNSUPPORTED, 0]
    res = b.run()
    st = res.stats()
    fs = _np(res.flagseg).view(np.uint32)
    assert int(st['n_in_obstacle'].sum()) == 0 and int((fs & L.FLAG_OBSTACLE != 0).sum()) == 0
    for k in (0, 1, 2, 4):          # detour legs that belong to the headland
        f = fs[res.field_slice(k)]
        assert int((((f & L.KIND_MASK) == L.KIND_DETOUR) & ((f & L.FLAG_HEADLAND) != 0)).sum()) > 0, k
    if opt.get('sample_spacing', 0.0) > 0:
        b0 = E.Batch(specs, _veh(DEFAULT_VP), E.make_options(**opt))
        assert int(b0.run().stats()['n_in_obstacle'].sum()) > 0          # the plain mode drives its loops through them
        b0.close()
    b.close()


def test_detours_never_cross_another_obstacle():
    """Found by review in round 2: an obstacle just above another one -- the detour around the first ran through the second and the
    field still came back OK.  Grown boxes that overlap are now merged, a leg runs on the boundary of its own (merged) box, and a side
    that would leave the work area is not taken: either the field is refused or no path point lies inside an obstacle.  The reviewer's
    layout, variations of it, and a few hundred random pairs of nearby obstacles, library against oracle."""
    A = [(240.0, 23.5), (260.0, 23.5), (260.0, 24.4), (240.0, 24.4)]
    Bq = [(245.0, 25.9), (255.0, 25.9), (255.0, 26.4), (245.0, 26.4)]
    rng = np.random.default_rng(11)
    layouts = [[A, Bq], [Bq, A], [A, [(x + 30.0, y) for x, y in Bq]], [A, [(x, y + 1.0) for x, y in Bq]], [A, [(x, y - 4.0) for x, y in Bq]]]
    for _ in range(40):
        cx, cy = rng.uniform(80, 420), rng.uniform(30, 170)
        obs = []
        for _k in range(int(rng.integers(2, 5))):
            ox, oy, w, h = cx + rng.uniform(-25, 25), cy + rng.uniform(-12, 12), rng.uniform(2, 20), rng.uniform(0.5, 8)
            obs.append([(ox, oy), (ox + w, oy), (ox + w, oy + h), (ox, oy + h)])
        layouts.append(obs)
    specs = [E.FieldSpec(field_length=500.0, field_width=200.0, obstacles=o) for o in layouts]
    ofs = [orc.make_field(L=500.0, H=200.0, obstacles=o) for o in layouts]
    for kw in (dict(avoid_obstacles=True), dict(avoid_obstacles=True, sample_spacing=0.4)):
        _compare_with_oracle(specs, ofs, DEFAULT_VP, kw, k_tol=1e-8, v_tol=1e-6)
        b = E.Batch(specs, _veh(DEFAULT_VP), E.make_options(**kw))
        st = b.run().stats()
        ok = np.array([i.status for i in b.info]) == 0
        assert ok[:5].all() and ok.sum() >= len(layouts) // 2
        assert int(st['n_in_obstacle'][ok].sum()) == 0
        b.close()


def test_error_fields_inside_a_batch():
    specs = [E.FieldSpec(field_length=500.0, field_width=200.0), E.FieldSpec(field_length=15.0, field_width=200.0),
             E.FieldSpec(field_length=100.0, field_width=80.0)]
    ofs = [orc.make_field(L=500.0, H=200.0), orc.make_field(L=15.0, H=200.0), orc.make_field(L=100.0, H=80.0)]
    _compare_with_oracle(specs, ofs, DEFAULT_VP, {})


@pytest.mark.parametrize('opt', [
    dict(sample_spacing=0.5), dict(sample_spacing=0.1), dict(turn_model=1, sample_spacing=0.25),
    dict(turn_model=1, sample_spacing=0.1, clothoid_frac=1.0), dict(turn_model=1, sample_spacing=0.2, clothoid_frac=0.0),
    dict(turn_model=1, sample_spacing=0.3, clothoid_frac=0.4, clothoid_fit=0), dict(turn_model=1)])
def test_dense_and_clothoid_sampling_vs_oracle(opt):
    """Build-defined modes (no reference counterpart): oracle = independent long-double Fresnel quadrature.
    Curvature from a 3-point stencil amplifies coordinate rounding by ~4/ds^2, hence the scaled tolerances."""
    specs, ofs = _random_fields(7, 6, with_points=True)
    specs2, ofs2 = _random_fields(8, 3, para=True)
    ds = opt.get('sample_spacing', 0.0) or 0.5
    k_tol = max(K_TOL, 4e-12 / ds ** 2)
    _compare_with_oracle(specs + specs2, ofs + ofs2, DEFAULT_VP, opt, xy_tol=1e-9, k_tol=k_tol, v_tol=max(V_TOL, 200 * k_tol))


@pytest.mark.parametrize('vp', [
    [3.2, 8.0, 9.0, 15.0, 15.0, 2.0, 1.5, 0.85],     # turn speed 15 km/h: the clamp binds on every turn sample
    [3.2, 8.0, 9.0, 15.0, 4.0, 2.0, 0.1, 0.85],      # a_lon 0.1: the sweeps bind across the jump from a turn's end to the next line
    [3.2, 8.0, 30.0, 15.0, 4.0, 2.0, 1.5, 0.85],     # work speed 30 km/h: the clamp binds on the first point of every line
])
def test_turns_that_are_not_closed_form_vs_oracle(vp):
    """The closed-form U-turn path (quiet runs of kind 3, swath lines quiet end to end) is only taken when every turn sample and
    the next line's first point keep their nominal speed and nothing propagates across the jump; otherwise the general kernel
    plans the turns.  Same comparison against the oracle in both cases (dense clothoid and dense arcs)."""
    specs, ofs = _random_fields(21, 5, with_points=True)
    specs2, ofs2 = _random_fields(22, 2, para=True)
    for opt in (dict(turn_model=1, sample_spacing=0.2), dict(sample_spacing=0.25)):
        ds = opt['sample_spacing']
        k_tol = max(K_TOL, 4e-12 / ds ** 2)
        _compare_with_oracle(specs + specs2, ofs + ofs2, vp, opt, xy_tol=1e-9, k_tol=k_tol, v_tol=max(V_TOL, 200 * k_tol))


def test_other_vehicles_vs_oracle():
    specs, ofs = _random_fields(99, 12)
    _compare_with_oracle(specs, ofs, [2.5, 6.0, 12.0, 14.0, 5.0, 1.2, 0.7, 0.9], {})
    _compare_with_oracle(specs, ofs, [4.0, 5.0, 9.0, 20.0, 3.0, 2.5, 3.0, 0.8], dict(sample_spacing=0.4))


def test_standalone_speed_planner_vs_golden(golden_kernels):
    g = golden_kernels
    offs = g['sp_offsets']
    x, y = g['sp_path'][:, 0], g['sp_path'][:, 1]
    out, nadj = E.speed_plan(x, y, g['sp_v_in'], _veh(g['vp_default']), clamp=True, offsets=offs)
    np.testing.assert_allclose(_np(out), g['sp_v_out'], rtol=0, atol=V_TOL)
    out2, _ = E.speed_plan(x, y, g['sp_v_in'], _veh(g['vp2']), clamp=True, offsets=offs)
    np.testing.assert_allclose(_np(out2), g['sp2_v_out'], rtol=0, atol=V_TOL)
    sm, _ = E.speed_plan(x, y, g['sp_v_in'], _veh(g['vp_default']), clamp=False, offsets=offs)
    np.testing.assert_allclose(_np(sm), g['sp_v_smooth_only'], rtol=0, atol=V_TOL)
    # path 0 has 3 points; a 2-point path must come back untouched by clamp=True (MLP:480-481)
    o2, _ = E.speed_plan([0.0, 1.0], [0.0, 0.0], [0.1, 15.0], _veh(g['vp_default']), clamp=True)
    assert _np(o2).tolist() == [0.1, 15.0]
    o3, _ = E.speed_plan([0.0, 1.0], [0.0, 0.0], [0.1, 15.0], _veh(g['vp_default']), clamp=False)
    assert _np(o3)[1] < 15.0
    # adjusted counts equal the oracle's
    veh = orc.Vehicle.make(g['vp_default'])
    want = [orc.speed_limit(g['sp_path'][offs[k]:offs[k + 1]], g['sp_v_in'][offs[k]:offs[k + 1]], veh)[1]
            for k in range(len(offs) - 1)]
    assert _np(nadj).tolist() == want


def test_standalone_curvature_and_verify_vs_golden(golden_kernels):
    g = golden_kernels
    tri = g['curv_tri']
    offs = np.arange(0, 3 * len(tri) + 1, 3)
    k = _np(E.curvature(tri[:, :, 0].ravel(), tri[:, :, 1].ravel(), offsets=offs)).reshape(-1, 3)
    np.testing.assert_allclose(k[:, 1], g['curv_kappa'], rtol=1e-11, atol=1e-13)
    assert (k[:, 0] == 0).all() and (k[:, 2] == 0).all()
    so = g['sp_offsets']
    st = E.verify(g['sp_path'][:, 0], g['sp_path'][:, 1], g['sp_v_out'], _veh(g['vp_default']), offsets=so)
    ref = g['ver_stats']
    np.testing.assert_allclose(st['max_kappa'], ref[:, 0], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(st['max_alat'], ref[:, 1], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(st['max_jump'], ref[:, 4], rtol=1e-10, atol=1e-12)
    assert st['n_viol'].tolist() == ref[:, 2].astype(int).tolist()
    np.testing.assert_allclose(st['main_len_m'], g['len_m'], rtol=1e-12)
    np.testing.assert_allclose(st['main_time_s'], g['time_s'], rtol=1e-12)
    st15 = E.verify(g['sp_path'][:, 0], g['sp_path'][:, 1], np.full(len(g['sp_path']), 15.0), _veh(g['vp_default']), offsets=so)
    assert st15['n_viol'][1:].tolist() == g['ver15_stats'][:, 2].astype(int).tolist()


def test_long_single_path_sweeps_vs_oracle():
    """One path of 300k points (147 tiles): exercises the cross-tile spine of the min-plus scan."""
    rng = np.random.default_rng(5)
    n = 300_000
    th = np.cumsum(rng.normal(0, 0.05, n))
    seg = rng.uniform(0.01, 0.06, n)                   # ~3.5 cm steps: constraints reach across many tiles
    xy = np.cumsum(np.column_stack([seg * np.cos(th), seg * np.sin(th)]), axis=0)
    for j in rng.integers(1, n, size=50):
        xy[j] = xy[j - 1]
    v = rng.choice([0.5, 2.5, 4.0, 9.0, 15.0, 40.0], size=n, p=[0.01, 0.04, 0.1, 0.4, 0.4, 0.05])
    veh = orc.Vehicle.make(max_longitudinal_accel=0.05)
    want, _ = orc.speed_limit(xy, v, veh)
    got, _ = E.speed_plan(xy[:, 0], xy[:, 1], v, _veh([3.2, 8.0, 9.0, 15.0, 4.0, 2.0, 0.05, 0.85]), clamp=True)
    np.testing.assert_allclose(_np(got), want, rtol=0, atol=1e-9)


def test_three_level_spine_and_cached_tilings_vs_oracle():
    """Standalone speed planner on 1.2 M points in 3 paths (2 346 tiles: the spine takes its three-level form, blocks of 2 048 tile
    aggregates) with a 0.02 m/s^2 vehicle, so that constraints cross tiles AND spine blocks; then the same operator on other path
    sets: offsets handed over on the host (tile table cached in the context, rebuilt when the offsets change) and as a device tensor
    (read back by the library)."""
    import torch
    rng = np.random.default_rng(11)
    n = 1_200_000
    th = np.cumsum(rng.normal(0, 0.03, n))
    seg = rng.uniform(0.01, 0.05, n)
    xy = np.cumsum(np.column_stack([seg * np.cos(th), seg * np.sin(th)]), axis=0)
    for j in rng.integers(1, n, size=80):
        xy[j] = xy[j - 1]
    v = rng.choice([0.5, 2.5, 4.0, 9.0, 15.0, 40.0], size=n, p=[0.002, 0.018, 0.08, 0.45, 0.4, 0.05])
    offs = np.array([0, 500_123, 500_125, n], dtype=np.int64)            # (the middle path has 2 points: returned unchanged)
    vp = [3.2, 8.0, 9.0, 15.0, 4.0, 2.0, 0.02, 0.85]
    veh = orc.Vehicle.make(max_longitudinal_accel=0.02)
    want = np.concatenate([orc.speed_limit(xy[a:b], v[a:b], veh)[0] if b - a >= 3 else v[a:b] for a, b in zip(offs[:-1], offs[1:])])
    got, _ = E.speed_plan(xy[:, 0], xy[:, 1], v, _veh(vp), clamp=True, offsets=offs)
    np.testing.assert_allclose(_np(got), want, rtol=0, atol=1e-9)
    got2, _ = E.speed_plan(xy[:, 0], xy[:, 1], v, _veh(vp), clamp=True, offsets=offs)          # cached tile table
    assert np.array_equal(_np(got2), _np(got))
    offs_b = np.array([0, 300_000, n], dtype=np.int64)                                          # other offsets: rebuilt
    want_b = np.concatenate([orc.speed_limit(xy[a:b], v[a:b], veh)[0] for a, b in zip(offs_b[:-1], offs_b[1:])])
    got_b, _ = E.speed_plan(xy[:, 0], xy[:, 1], v, _veh(vp), clamp=True, offsets=torch.as_tensor(offs_b, device='cuda'))
    np.testing.assert_allclose(_np(got_b), want_b, rtol=0, atol=1e-9)
    k = E.curvature(xy[:1000, 0], xy[:1000, 1], offsets=[0, 400, 1000])
    k2 = E.curvature(xy[:1000, 0], xy[:1000, 1], offsets=torch.as_tensor([0, 400, 1000], device='cuda'))
    assert np.array_equal(_np(k), _np(k2)) and float(_np(k)[399]) == 0.0 and float(_np(k)[400]) == 0.0


def test_fresnel_vs_mpmath_table():
    import os
    tab = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'fresnel_table.npz'))
    c, s = E.fresnel(tab['t'])
    np.testing.assert_allclose(_np(c), tab['C'], rtol=0, atol=2e-15)
    np.testing.assert_allclose(_np(s), tab['S'], rtol=0, atol=2e-15)


def test_ga_fitness_bit_exact(golden_ga):
    g = golden_ga
    for tag in ('n10', 'n128', 'n129', 'n33'):
        d, f = E.ga_fitness(g[f'{tag}_routes'], g[f'{tag}_D'], order_mode=0)
        assert np.array_equal(_np(d), g[f'{tag}_dist']), tag       # left-to-right sum: bit-exact
        assert np.array_equal(_np(f), g[f'{tag}_fit']), tag
        d1, _ = E.ga_fitness(g[f'{tag}_routes'], g[f'{tag}_D'], order_mode=1)
        np.testing.assert_allclose(_np(d1), g[f'{tag}_dist'], rtol=1e-13)


def test_planner_class_end_to_end(golden_plans):
    from field_coverage_path_planning_amd.multi_layer_planner_v3 import TwoLayerPathPlannerV37, VehicleParams
    g = golden_plans
    pl = TwoLayerPathPlannerV37(VehicleParams(max_headland_speed_kmh=14.0), field_length=500, field_width=200,
                                start_point=(10, 10), end_point=(490, 190))
    r = pl.plan_complete_coverage()
    n = 'v351_start_end'
    assert r['version'] == 'V3.5.1' and len(r['features']) == 5
    np.testing.assert_allclose(r['main_work']['path'], g[f'{n}/main_path'], rtol=0, atol=XY_TOL)
    np.testing.assert_allclose(r['headland']['speeds'], g[f'{n}/head_v'], rtol=0, atol=V_TOL)
    np.testing.assert_allclose(r['approach_path'], g[f'{n}/approach'], rtol=0, atol=XY_TOL)
    np.testing.assert_allclose(r['departure_path'], g[f'{n}/departure'], rtol=0, atol=XY_TOL)
    np.testing.assert_allclose([r['main_work']['stats'][k] for k in ('path_length_km', 'time_hours', 'avg_speed_kmh')],
                               g[f'{n}/main_stats'], rtol=1e-10)
    allp = np.vstack([r['main_work']['path'], r['headland']['path']])
    alls = np.concatenate([r['main_work']['speeds'], r['headland']['speeds']])
    ver = pl.verify_curvature_constraints(allp, alls)
    gv = g[f'{n}/ver']
    assert ver['accel_violations'] == gv[2] and ver['pass'] == bool(gv[5])
    np.testing.assert_allclose([ver['max_curvature'], ver['max_lateral_accel'], ver['max_jump']], gv[[0, 1, 4]], rtol=1e-9)
    assert abs(pl._calculate_path_length(r['approach_path']) - 11.9) < 0.05           # doc/V3.5.1:109-111
    with pytest.raises(ValueError):
        TwoLayerPathPlannerV37(VehicleParams(), field_length=15, field_width=200).plan_complete_coverage()


def test_ga_solver_end_to_end():
    from field_coverage_path_planning_amd.genetic_algorithm_solver import GAConfig, GeneticAlgorithmSolver
    rng = np.random.default_rng(42)
    pts = rng.uniform(0, 100, size=(24, 2))
    D = np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1))
    s = GeneticAlgorithmSolver(GAConfig(population_size=120, max_generations=150, convergence_threshold=40), seed=3)
    route, stats = s.solve(D, verbose=False)
    assert sorted(route) == list(range(24)) and route[0] == 0                 # GA:118-120
    assert stats['best_distance'] == pytest.approx(orc.ga_distance([route], D)[0], rel=1e-12)
    random_mean = np.mean(orc.ga_distance(np.array([rng.permutation(24) for _ in range(200)]), D))
    assert stats['best_distance'] < 0.6 * random_mean                          # it optimises
    assert s._calculate_distance(route, D) == orc.ga_distance([route], D)[0]   # bit-exact single-tour entry point
    assert s._calculate_fitness(route, D) == orc.ga_fitness([route], D)[0]


def test_sharded_api_single_rank_matches_batch():
    from field_coverage_path_planning_amd import sharding as S
    specs, _ = _random_fields(11, 9)
    res = S.plan_sharded(specs, _veh(DEFAULT_VP), E.make_options(1, 0.5))
    assert res.block == (0, 9) and res.stats_all.shape == (9, 13)
    b = E.Batch(specs, _veh(DEFAULT_VP), E.make_options(1, 0.5))
    assert np.array_equal(_np(b.run().stats_raw), _np(res.stats_all))          # deterministic reductions
    b.close()


@pytest.mark.parametrize('opt', [dict(turn_model=1, sample_spacing=0.2), dict(), dict(sample_spacing=0.1)])
def test_results_do_not_depend_on_batch_composition(opt):
    """A field planned alone or inside a batch (= on another rank of a sharded job) gives bit-identical arrays and statistics:
    the chunks of the quiet runs follow the position in the batch arrays, their results must not (dense runs, mixed chunks, spans)."""
    specs, _ = _random_fields(21, 5, para=True)
    o = E.make_options(**opt)
    b = E.Batch(specs, _veh(DEFAULT_VP), o)
    r = b.run()
    for i in (0, 3):
        b1 = E.Batch([specs[i]], _veh(DEFAULT_VP), o)
        r1 = b1.run()
        sl = r.field_slice(i)
        for name in ('x', 'y', 'kappa', 'v', 'flagseg'):
            assert np.array_equal(_np(getattr(r, name))[sl], _np(getattr(r1, name))), (i, name)
        assert np.array_equal(_np(r.stats_raw)[i], _np(r1.stats_raw)[0])
        b1.close()
    b.close()


@pytest.mark.parametrize('opt', [dict(sample_spacing=0.25), dict(turn_model=1, sample_spacing=0.1), dict()])
def test_fused_equals_staged_over_many_tile_alignments(opt):
    """The fused kernel takes shortcuts at tile edges (closed-form halos, quiet tiles); the staged pipeline has none.
    600 fields whose sizes shift every primitive boundary through all alignments relative to the 512-point tiles."""
    specs = [E.FieldSpec(field_length=100.0 + 0.37 * i, field_width=60.0 + 0.11 * i,
                         start_point=(5.0, 5.0) if i % 4 == 0 else None) for i in range(600)]
    b = E.Batch(specs, _veh(DEFAULT_VP), E.make_options(**opt))
    r1 = b.run(mode=1)
    keep = [t.clone() for t in (r1.x, r1.y, r1.kappa, r1.v, r1.flagseg, r1.stats_raw)]
    r0 = b.run(mode=0)
    assert torch_max_abs(keep[0], r0.x) == 0 and torch_max_abs(keep[1], r0.y) == 0     # same arithmetic: bit-equal
    assert torch_max_abs(keep[2], r0.kappa) <= 1e-9
    assert torch_max_abs(keep[3], r0.v) <= 1e-9
    assert bool((keep[4] == r0.flagseg).all())
    s1, s0 = E.BatchResult(b, *keep).stats(), r0.stats()
    for k in ('main_len_m', 'main_time_s', 'main_time_pre_s', 'head_len_m', 'head_time_s', 'head_time_pre_s'):
        np.testing.assert_allclose(s1[k], s0[k], rtol=1e-11, err_msg=k)
    for k in ('max_kappa', 'max_alat', 'max_jump'):
        np.testing.assert_allclose(s1[k], s0[k], rtol=1e-8, atol=1e-11, err_msg=k)
    for k in ('n_viol', 'n_outside', 'n_in_obstacle', 'n_adjusted'):
        assert np.array_equal(s1[k], s0[k]), k
    b.close()


def torch_max_abs(a, b):
    return float((a - b).abs().max())


def test_edge_case_fields_vs_oracle():
    """Tiny / degenerate geometries: reversed swath lines (line_end_x < line_start_x), single pass, fields barely larger
    than the headland, one headland loop (W > R), start point on the centre lines (tie-breaks), points out of bounds."""
    cases = [dict(L=33.0, H=40.0), dict(L=40.0, H=17.5), dict(L=17.2, H=300.0), dict(L=60.0, H=18.5),
             dict(L=200.0, H=100.0, start=(100.0, 50.0)), dict(L=200.0, H=100.0, start=(100.0, 50.0000001)),
             dict(L=120.0, H=80.0, start=(500.0, 10.0), end=(-1.0, 5.0)), dict(L=120.0, H=80.0, start=(0.0, 0.0), end=(120.0, 80.0)),
             dict(L=1000.0, H=16.1)]
    specs = [E.FieldSpec(field_length=c['L'], field_width=c['H'], start_point=c.get('start'), end_point=c.get('end')) for c in cases]
    ofs = [orc.make_field(L=c['L'], H=c['H'], start=c.get('start'), end=c.get('end')) for c in cases]
    _compare_with_oracle(specs, ofs, DEFAULT_VP, {})
    _compare_with_oracle(specs, ofs, DEFAULT_VP, dict(sample_spacing=0.3), k_tol=1e-7, v_tol=1e-5)
    # (the 17.2 m wide field: lines run against the jump from the previous turn, every line start is clamped -> no closed-form turns)
    _compare_with_oracle(specs, ofs, DEFAULT_VP, dict(turn_model=1, sample_spacing=0.3), k_tol=1e-7, v_tol=1e-5)
    _compare_with_oracle(specs, ofs, DEFAULT_VP, dict(turn_model=1, sample_spacing=50.0))         # spacing >> every primitive
    _compare_with_oracle(specs[4:8], ofs[4:8], [5.0, 4.5, 9.0, 15.0, 4.0, 2.0, 1.5, 0.85], {})     # W > R: one headland loop
    _compare_with_oracle(specs[:5], ofs[:5], [0.8, 8.0, 9.0, 15.0, 4.0, 2.0, 1.5, 0.85], {})       # 10 headland loops


def test_empty_and_all_error_batches():
    b = E.Batch([], _veh(DEFAULT_VP))
    r = b.run()
    assert b.total_points == 0 and r.x.numel() == 0 and r.stats_raw.shape == (0, 13)
    b.close()
    b = E.Batch([E.FieldSpec(field_length=10.0, field_width=10.0), E.FieldSpec(field_length=15.9, field_width=500.0)], _veh(DEFAULT_VP))
    assert [i.status for i in b.info] == [L.EINVAL, L.EINVAL] and b.total_points == 0
    r = b.run()
    assert r.x.numel() == 0 and int(_np(r.stats_raw).sum()) == 0
    ap, dp = b.connectors()
    assert np.isnan(_np(ap)).all()
    b.close()


def test_standalone_operators_on_ragged_paths():
    """CSR with empty, 1-point, 2-point paths between ordinary ones; duplicates at path ends; v_out aliasing v_in."""
    rng = np.random.default_rng(11)
    lens = [0, 1, 2, 3, 700, 0, 513, 1, 512, 1025]
    offs = np.cumsum([0] + lens)
    xy = np.cumsum(rng.normal(0, 0.4, size=(offs[-1], 2)), axis=0)
    xy[7] = xy[6]
    xy[offs[-1] - 1] = xy[offs[-1] - 2]
    v = rng.choice([2.5, 4.0, 9.0, 15.0], size=offs[-1])
    veh = orc.Vehicle.make()
    want = np.concatenate([orc.speed_limit(xy[a:b], v[a:b], veh)[0] if b - a >= 1 else np.zeros(0)
                           for a, b in zip(offs[:-1], offs[1:])])
    got, nadj = E.speed_plan(xy[:, 0], xy[:, 1], v, _veh(DEFAULT_VP), clamp=True, offsets=offs)
    np.testing.assert_allclose(_np(got), want, rtol=0, atol=1e-9)
    want_sm = np.concatenate([orc.smooth_speed_profile(xy[a:b], v[a:b], 1.5) for a, b in zip(offs[:-1], offs[1:])])
    got_sm, _ = E.speed_plan(xy[:, 0], xy[:, 1], v, _veh(DEFAULT_VP), clamp=False, offsets=offs)
    np.testing.assert_allclose(_np(got_sm), want_sm, rtol=0, atol=1e-9)
    st = E.verify(xy[:, 0], xy[:, 1], want, _veh(DEFAULT_VP), offsets=offs)
    for k, (a, b) in enumerate(zip(offs[:-1], offs[1:])):
        ref = orc.verify(xy[a:b], want[a:b], veh) if b - a >= 1 else np.array([0, 0, 0, 0, 0, 1.0])
        assert st['n_viol'][k] == ref[2]
        np.testing.assert_allclose([st['max_kappa'][k], st['max_alat'][k], st['max_jump'][k]], ref[[0, 1, 4]], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(st['main_len_m'][k], orc.path_length(xy[a:b]) if b - a >= 2 else 0.0, rtol=1e-12, atol=0)
    k = _np(E.curvature(xy[:, 0], xy[:, 1], offsets=offs))
    for a, b in zip(offs[:-1], offs[1:]):
        if b - a >= 1:
            assert k[a] == 0 and k[b - 1] == 0
    with pytest.raises(L.FcppError):
        E.speed_plan(xy[:, 0], xy[:, 1], v, _veh(DEFAULT_VP), offsets=[0, 5, 3, offs[-1]])     # decreasing offsets
    d, f = E.ga_fitness(np.zeros((3, 1), dtype=np.int32), np.zeros((1, 1)))
    assert _np(d).tolist() == [0.0, 0.0, 0.0] and np.allclose(_np(f), 1e6)


def test_output_layouts_give_the_same_plan():
    """Batch.alloc(layout=...) and alloc(best_of=K) only choose WHERE the output arrays lie (five write streams
    far apart in device memory run a class faster than the same streams back to back); the plan is the same bit for bit.  Also the C
    ABI's fcpp_outputs_alloc: one allocation, the arrays a pitch apart."""
    import ctypes as C
    import torch
    specs, _ = _random_fields(5, 6)
    b = E.Batch(specs, _veh(DEFAULT_VP), E.make_options(1, 0.2))
    r0 = b.run(b.alloc(layout='plain'))
    assert b.layout['layout'] == 'plain'
    keep = [t.clone() for t in (r0.x, r0.y, r0.kappa, r0.v, r0.flagseg, r0.stats_raw)]
    small = b.alloc()                                   # a small batch: 'auto' stays plain
    assert b.layout['layout'] == 'plain' and 36 * b.total_points < b.SPREAD_MIN_BYTES
    spread = b.alloc(layout='spread')
    assert b.layout['layout'] == 'spread' and b.layout['pitch_GiB'] >= 1.0
    assert spread[1].data_ptr() - spread[0].data_ptr() == int(b.layout['pitch_GiB'] * 2**30) == spread[4].data_ptr() - spread[3].data_ptr()
    r1 = b.run(spread)
    for a, c in zip(keep, (r1.x, r1.y, r1.kappa, r1.v, r1.flagseg, r1.stats_raw)):
        assert bool((a == c).all())
    del spread, r1
    torch.cuda.empty_cache()
    mine = b.alloc(layout='plain')
    bufs = b.alloc(best_of=3, include=[mine, small])
    p = b.placement
    assert p['probe'] == 'step' and len(p['step_ms']) == 5 and 0 <= p['chosen'] < 5
    r2 = b.run(bufs)
    for a, c in zip(keep, (r2.x, r2.y, r2.kappa, r2.v, r2.flagseg, r2.stats_raw)):
        assert bool((a == c).all())
    # the C ABI's allocator: five pointers one pitch apart inside one allocation
    ptrs = [C.c_void_p() for _ in range(5)]
    pitch = 3 << 30
    L.check(b.lib.fcpp_outputs_alloc(b.ctx.handle, b.total_points, pitch, *[C.byref(q) for q in ptrs]))
    assert all(ptrs[k + 1].value - ptrs[k].value == pitch for k in range(4))
    stats = torch.zeros((b.n_fields, L.STATS_WORDS), dtype=torch.int64, device='cuda')
    b.ctx.bind_stream()
    L.check(b.lib.fcpp_batch_run(b.handle, *ptrs, C.c_void_p(stats.data_ptr()), 1))
    torch.cuda.synchronize()
    assert bool((stats == keep[5]).all())
    got = torch.empty_like(keep[0])
    L.check(b.lib.fcpp_memcpy_d2h(b.ctx.handle, C.c_void_p(0), C.c_void_p(0), 0))
    host = np.empty(b.total_points, dtype=np.float64)
    L.check(b.lib.fcpp_memcpy_d2h(b.ctx.handle, C.c_void_p(host.ctypes.data), ptrs[3], 8 * b.total_points))
    assert np.array_equal(host, keep[3].cpu().numpy())
    del got
    L.check(b.lib.fcpp_outputs_free(b.ctx.handle, ptrs[0]))
    b.close()


def test_output_arena_shared_by_live_batches():
    """Context.reserve_outputs(): ONE allocation, five lanes a pitch apart; every live batch's arrays come out of it (array k in lane k,
    first fit), give the plan of any other layout bit for bit, and go back when the last tensor is gone.  Nothing is reserved unless
    asked for: before the reservation 'auto' is 'plain' whatever the size."""
    import gc
    import torch
    ctx = E.get_context()
    assert ctx.outputs_info() == (0, 0, 0)
    veh = _veh(DEFAULT_VP)
    batches = [E.Batch(E.FieldTable.from_rectangles(np.random.default_rng(k).uniform(100, 600, (20, 2))), veh, E.make_options(1, 0.05 * (k + 1))) for k in range(3)]
    refs = []
    for b in batches:
        bufs = b.alloc()
        assert b.layout['layout'] == 'plain'
        r = b.run(bufs)
        refs.append([t.clone() for t in (r.x, r.y, r.kappa, r.v, r.flagseg, r.stats_raw)])
    del bufs, r
    ctx.reserve_outputs(lane_gib=1.0, pitch_gib=3.0)
    try:
        lane, pitch, live = ctx.outputs_info()
        assert (lane, pitch, live) == (1 << 30, 3 << 30, 0)
        held = []
        for b in batches:
            bufs = b.alloc(layout='arena')
            assert b.layout == {'layout': 'arena', 'pitch_GiB': 3.0, 'lane_GiB': 1.0}
            assert all(bufs[k + 1].data_ptr() - bufs[k].data_ptr() == pitch for k in range(4))
            held.append(bufs)
        # three live batches side by side in the lanes, no overlap
        spans = sorted((h[0].data_ptr(), h[0].data_ptr() + 8 * b.total_points) for h, b in zip(held, batches))
        assert all(spans[k][1] <= spans[k + 1][0] for k in range(2))
        assert ctx.outputs_info()[2] >= sum(8 * b.total_points for b in batches)
        for b, bufs, ref in zip(batches, held, refs):
            r = b.run(bufs)
            torch.cuda.synchronize()
            for a, c in zip(ref, (r.x, r.y, r.kappa, r.v, r.flagseg, r.stats_raw)):
                assert torch.equal(a, c)
        # the middle batch's arrays go back; the next allocation of that size or less takes their place (first fit)
        gone = held[1][0].data_ptr()
        del r, bufs
        held[1] = None
        gc.collect()
        again = batches[1].alloc(layout='arena')
        assert again[0].data_ptr() == gone
        del again, held
        gc.collect()
        assert ctx.outputs_info()[2] == 0
        # an array larger than a lane: a plain allocation of its own, the arena untouched
        big = E.Batch(E.FieldTable.from_rectangles([[5000.0, 2000.0]]), veh, E.make_options(1, 0.02))
        assert 8 * big.total_points > lane
        with pytest.raises(RuntimeError):
            big.alloc(layout='arena')
        bufs = big.alloc()
        assert big.layout['layout'] == 'plain'
        big.close()
    finally:
        gc.collect()
        L.check(ctx.lib.fcpp_ctx_reserve_outputs(ctx.handle, 4096, 4096))       # (back to next to nothing for the tests that follow)
    for b in batches:
        b.close()


@pytest.mark.parametrize('opt', [dict(), dict(sample_spacing=0.5)])
def test_statistics_rows_are_written_whatever_the_buffer_held(opt):
    """fcpp_batch_run writes every field's statistics row -- rows of fields that raise included (zeros) -- in both pipelines: the caller's
    buffer need not be cleared."""
    import torch
    specs = [E.FieldSpec(field_length=500.0, field_width=200.0), E.FieldSpec(field_length=15.0, field_width=200.0),
             E.FieldSpec(field_length=100.0, field_width=80.0), E.FieldSpec(field_length=10.0, field_width=10.0)]
    b = E.Batch(specs, _veh(DEFAULT_VP), E.make_options(**opt))
    for mode in (1, 0):
        bufs = list(b.alloc())
        clean = _np(b.run(tuple(bufs), mode=mode).stats_raw).copy()
        bufs[5] = torch.full_like(bufs[5], 0x5a5a5a5a5a5a5a5a)
        dirty = _np(b.run(tuple(bufs), mode=mode).stats_raw).copy()
        assert np.array_equal(clean, dirty)
        assert not clean[1].any() and not clean[3].any() and clean[0].any()
    b.close()


def test_pinned_field_tables_are_read_where_they_lie():
    """A FieldTable in pinned host memory (pin()) is not copied before the device plans it; slices of it included.  Same batch as from
    pageable records."""
    rng = np.random.default_rng(7)
    LH = rng.uniform(100.0, 600.0, size=(300, 2))
    plain = E.FieldTable.from_rectangles(LH)
    pinned = E.FieldTable.from_rectangles(LH).pin()
    assert pinned._pinned is not None and np.array_equal(plain.rec, pinned.rec)
    veh = _veh(DEFAULT_VP)
    for sl in (slice(0, 300), slice(37, 201)):
        a, c = E.Batch(plain[sl], veh), E.Batch(pinned[sl], veh)
        assert a.total_points == c.total_points and np.array_equal(E.plan_points(plain[sl], veh, E.make_options()), E.plan_points(pinned[sl], veh, E.make_options()))
        ra, rc_ = a.run(), c.run()
        for u, w in ((ra.x, rc_.x), (ra.y, rc_.y), (ra.v, rc_.v), (ra.kappa, rc_.kappa), (ra.flagseg, rc_.flagseg), (ra.stats_raw, rc_.stats_raw)):
            assert np.array_equal(_np(u), _np(w))
        assert np.array_equal(a.info.array, c.info.array)
        a.close(); c.close()


def test_obstacle_aware_swaths_on_the_half_metre_grid_the_fragile_case():
    """The FRAGILE twin of _clip_fields' tilted parallelogram (round 5; the main case keeps its obstacles off the grid): obstacle sides on multiples
    of 0.1 m, so that some detour leg is a multiple of 0.5 m long up to the rounding of the rotation into the frame of layer 1 and back, and
    its sample count int(len / 0.5) + 1 hinges on the last bit of a sine -- the library's (csrc/fcpp_math.h) or the oracle's (libm), as the
    reference's own int((max_y - min_y) / W) does (tests/test_oracle_vs_golden.py: the fields the reference decides).  What must hold whichever
    way the bits fall: the same sequence of path segments (kind, layer, swath / corner index), every segment of the same length except detour
    legs, which may differ by one sample; no point inside an obstacle; and where the counts do agree, the whole path to the usual tolerances."""
    rot = 0.3
    c, s = np.cos(rot), np.sin(rot)
    tilt = lambda pts: [(float(x * c - y * s), float(x * s + y * c)) for x, y in pts]
    verts = tilt([(0.0, 0.0), (500.0, 0.0), (560.0, 260.0), (60.0, 260.0)])
    para_obs = [tilt([(200.0, 100.0), (230.0, 100.0), (230.0, 125.0), (200.0, 125.0)]), tilt([(340.0, 150.0), (365.0, 160.0), (350.0, 185.0)])]
    spec, of = E.FieldSpec(field_vertices=verts, obstacles=para_obs), orc.make_field(verts=verts, obstacles=para_obs)
    o = E.make_options(avoid_obstacles=True)
    b = E.Batch([spec], _veh(DEFAULT_VP), o)
    res = b.run()
    fs = _np(res.flagseg).view(np.uint32)
    rc, p = orc.plan_field(of, orc.Vehicle.make(DEFAULT_VP), orc.Options.make(o.turn_model, o.clothoid_fit, o.sample_spacing, o.clothoid_frac, o.geofence_tol, o.obstacle_mode, o.ring_order))
    assert rc == 0 and b.info[0].status == 0 and b.info[0].n_swaths == p.n_swaths
    assert int(res.stats()['n_in_obstacle'][0]) == 0 and p.n_in_obstacle == 0

    def segments(words):
        key = words & ~np.uint32(L.FLAG_ALAT | L.FLAG_OUTSIDE | L.FLAG_OBSTACLE)
        cut = np.flatnonzero(np.diff(key.astype(np.int64)) != 0) + 1
        starts = np.concatenate([[0], cut])
        return key[starts], np.diff(np.concatenate([starts, [len(key)]]))

    ka, na = segments(fs)
    kb, nb = segments(p.flagseg)
    if np.array_equal(ka, kb) and np.array_equal(na, nb):
        _compare_with_oracle([spec], [of], DEFAULT_VP, dict(avoid_obstacles=True), xy_tol=1e-9, k_tol=max(K_TOL, 4e-12 / 0.25), v_tol=max(V_TOL, 200 * max(K_TOL, 4e-12 / 0.25)))
    else:
        assert np.array_equal(ka, kb)
        detour = (ka & L.KIND_MASK) == L.KIND_DETOUR
        assert np.array_equal(na[~detour], nb[~detour]) and int(np.abs(na[detour] - nb[detour]).max()) <= 1
    b.close()
