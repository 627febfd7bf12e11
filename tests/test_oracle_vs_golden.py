"""Pins the CPU oracle (oracle/fcpp_oracle.c) to the reference.

Golden vectors in tests/golden/*.npz were produced by tools/gen_golden.py, which runs the
reference's own Python code (multi_layer_planner_v3.py / genetic_algorithm_solver.py).
Tolerances: numpy's libm/SIMD transcendental kernels and glibc's differ by <= 1-2 ulp, so
coordinates are compared at 1e-11 m (tier A kernels mostly come out bit-equal); integer
quantities (point counts, swath counts, violation counts) must be exactly equal.
"""
import os

import numpy as np
import pytest

import oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

XY_TOL = 1e-11
V_TOL = 1e-10


def test_curvature(golden_kernels):
    g = golden_kernels
    k = np.array([orc.curvature(t[0], t[1], t[2]) for t in g['curv_tri']])
    np.testing.assert_allclose(k, g['curv_kappa'], rtol=1e-13, atol=1e-15)
    assert k[0] == 0 and k[1] == 0 and k[4] == 0      # degenerate stencils -> exactly 0


def test_speed_planner(golden_kernels):
    g = golden_kernels
    veh = orc.Vehicle.make(g['vp_default'])
    veh2 = orc.Vehicle.make(g['vp2'])
    offs = g['sp_offsets']
    for k in range(len(offs) - 1):
        a, b = offs[k], offs[k + 1]
        xy, v = g['sp_path'][a:b], g['sp_v_in'][a:b]
        out, _ = orc.speed_limit(xy, v, veh)
        np.testing.assert_allclose(out, g['sp_v_out'][a:b], rtol=0, atol=V_TOL)
        out2, _ = orc.speed_limit(xy, v, veh2)
        np.testing.assert_allclose(out2, g['sp2_v_out'][a:b], rtol=0, atol=V_TOL)
        sm = orc.smooth_speed_profile(xy, v, veh.max_longitudinal_accel)
        np.testing.assert_allclose(sm, g['sp_v_smooth_only'][a:b], rtol=0, atol=V_TOL)


def test_verifier_and_metrics(golden_kernels):
    g = golden_kernels
    veh = orc.Vehicle.make(g['vp_default'])
    offs = g['sp_offsets']
    for k in range(len(offs) - 1):
        a, b = offs[k], offs[k + 1]
        xy, v = g['sp_path'][a:b], g['sp_v_out'][a:b]
        st = orc.verify(xy, v, veh)
        ref = g['ver_stats'][k]
        np.testing.assert_allclose(st[[0, 1, 3, 4]], ref[[0, 1, 3, 4]], rtol=1e-12, atol=1e-13)
        assert st[2] == ref[2] and st[5] == ref[5]
        assert orc.path_length(xy) == g['len_m'][k]            # numpy's pairwise summation restated: bit for bit
        assert orc.work_time(xy, v) == g['time_s'][k]
        if k >= 1:
            st15 = orc.verify(xy, np.full(len(xy), 15.0), veh)
            ref15 = g['ver15_stats'][k - 1]
            assert st15[2] == ref15[2] and st15[5] == ref15[5]
            np.testing.assert_allclose(st15[[0, 1, 3, 4]], ref15[[0, 1, 3, 4]], rtol=1e-12, atol=1e-13)
    assert g['ver15_stats'][:, 2].max() > 0  # the violating case is really exercised


def test_samplers(golden_kernels):
    g = golden_kernels
    R = 8.0
    for args, pts in zip(g['uturn_args'], g['uturn_pts']):
        got = orc.safe_arc_turn(args[1], bool(args[2]), args[3], args[4], R)
        np.testing.assert_allclose(got, pts, rtol=0, atol=XY_TOL)
    for ci in range(4):
        got = orc.corner_arc(100.25 + ci, 50.5 - ci, ci, R)
        np.testing.assert_allclose(got, g['corner_arc_pts'][ci], rtol=0, atol=XY_TOL)
    assert np.array_equal(orc.straight(1.6, 1.6, 498.4, 1.6, 20), g['straight_pts'])      # linspace: bit-equal
    assert np.array_equal(orc.straight(498.4, 1.6, 498.4, 198.4, 20), g['straight2_pts'])
    assert np.array_equal(orc.straight(10.0, 10.0, 1.6, 1.6, 50), g['approach_pts'])
    for r_in, r_out in zip(g['rot_in'], g['rot_out']):
        np.testing.assert_allclose(orc.rotate_point(*r_in), r_out, rtol=0, atol=1e-12)


def test_reverse_fill(golden_kernels):
    g = golden_kernels
    W, R = 3.2, 8.0
    cs = [(W / 2, W / 2), (500 - W / 2, W / 2), (500 - W / 2, 200 - W / 2), (W / 2, 200 - W / 2)]
    offs = g['rev_offsets']
    for ci in range(4):
        arc = orc.corner_arc(cs[ci][0], cs[ci][1], ci, R)
        pts, ln = orc.reverse_path(arc[-1], arc[-2], 500.0, 200.0, R)
        ref = g['rev_pts'][offs[ci]:offs[ci + 1]]
        assert len(pts) == len(ref)                       # int(len/0.5): exact
        np.testing.assert_allclose(pts, ref, rtol=0, atol=XY_TOL)
        np.testing.assert_allclose(ln, g['rev_len'][ci], rtol=1e-13)


def test_u_pattern(golden_kernels):
    g = golden_kernels
    veh = orc.Vehicle.make(g['vp_default'])
    offs = g['upat_offsets']
    for k, a in enumerate(g['upat_args']):
        xy, v = orc.u_pattern(a[:4], bool(a[4]), bool(a[5]), veh)
        ref = g['upat_pts'][offs[k]:offs[k + 1]]
        assert len(xy) == len(ref)
        np.testing.assert_allclose(xy, ref, rtol=0, atol=XY_TOL)
        assert np.array_equal(v, g['upat_v'][offs[k]:offs[k + 1]])


def test_ga_tour_length(golden_ga):
    g = golden_ga
    for tag in ('n10', 'n128', 'n129', 'n33'):
        D, routes = g[f'{tag}_D'], g[f'{tag}_routes']
        assert np.array_equal(orc.ga_distance(routes, D), g[f'{tag}_dist'])   # sequential sum: bit-equal
        assert np.array_equal(orc.ga_fitness(routes, D), g[f'{tag}_fit'])


def _field_from_golden(g, name):
    verts = g[f'{name}/verts']
    start = g[f'{name}/start']
    end = g[f'{name}/end']
    obstacles = None
    if f'{name}/obs_offsets' in g:
        o, xy = g[f'{name}/obs_offsets'], g[f'{name}/obs_xy']
        obstacles = [xy[o[i]:o[i + 1]] for i in range(len(o) - 1)]
    kw = dict(start=None if np.isnan(start[0]) else start, end=None if np.isnan(end[0]) else end,
              obstacles=obstacles)
    if int(g[f'{name}/is_verts_input']):
        return orc.make_field(verts=verts, **kw)
    return orc.make_field(L=float(verts[1][0]), H=float(verts[2][1]), **kw)


def test_full_plans(golden_plans):
    """Tier B: the reference's own plan_complete_coverage(), run with tools/_shapely_standin.py in place of Shapely (not installed
    here).  What these vectors pin is the reference's numpy code given three facts the stand-in supplies: the inset polygon's corner
    coordinates, the bounds, and `gap.area > 0.1`.  The ORDER in which `buffer(-d).exterior.coords` lists the inset corners (ring
    orientation and start vertex, MLP:972) is the stand-in's assumption -- the documented intent LL, LR, UR, UL of MLP:957 and
    1049-1058, which reproduces every number the reference publishes -- not an observation of GEOS: headland loop direction and
    corner indices are 'documented intent', everything computed from them is reference-pinned.
    The `cw_*` scenarios are the same plans with the stand-in listing the inset corners the OTHER way round from the same first vertex
    (what a clockwise GEOS shell would list): options.ring_order = 1 in the oracle and in the library -- so both candidate orders are
    pinned to the reference's code, and a user with Shapely selects the one their GEOS produces.
    The `other_*` scenarios are convex quadrilaterals that are neither rectangles nor parallelograms (MLP:137-163 -> 'other')."""
    g = golden_plans
    shapes = {'rectangle': 0, 'parallelogram': 1, 'other': 2}
    assert sum(str(n).startswith('cw_') for n in g['names']) >= 8 and sum(str(n).startswith('other_') for n in g['names']) >= 3
    for name in g['names']:
        f = _field_from_golden(g, name)
        veh = orc.Vehicle.make(g[f'{name}/vp'])
        ring = int(g[f'{name}/ring_order'])
        assert ring == int(str(name).startswith('cw_'))
        rc, p = orc.plan_field(f, veh, orc.Options.make(ring_order=ring))
        assert rc == 0, name
        mp, hp = g[f'{name}/main_path'], g[f'{name}/head_path']
        assert p.n_main == len(mp) and p.n_head == len(hp), name             # counts: exact
        assert p.shape == shapes[str(g[f'{name}/shape'])], name
        assert p.start_kept == int(g[f'{name}/start_kept']) and p.end_kept == int(g[f'{name}/end_kept'])
        np.testing.assert_allclose(p.corner_angles, g[f'{name}/corner_angles'], rtol=1e-13, err_msg=name)
        np.testing.assert_allclose([p.field_length, p.field_width], g[f'{name}/field_LH'], rtol=1e-15)
        np.testing.assert_allclose(p.xy[:p.n_main], mp, rtol=0, atol=2e-10, err_msg=name)
        np.testing.assert_allclose(p.xy[p.n_main:], hp, rtol=0, atol=2e-10, err_msg=name)
        np.testing.assert_allclose(p.v[:p.n_main], g[f'{name}/main_v'], rtol=0, atol=1e-9, err_msg=name)
        np.testing.assert_allclose(p.v[p.n_main:], g[f'{name}/head_v'], rtol=0, atol=1e-9, err_msg=name)
        ms, hs = g[f'{name}/main_stats'], g[f'{name}/head_stats']
        np.testing.assert_allclose(p.main_len_m / 1000, ms[0], rtol=1e-12)
        np.testing.assert_allclose(p.main_time_s / 3600, ms[1], rtol=1e-11)
        np.testing.assert_allclose((p.main_len_m / 1000) / (p.main_time_pre_s / 3600), ms[2], rtol=1e-11)
        np.testing.assert_allclose(p.head_len_m / 1000, hs[0], rtol=1e-12)
        np.testing.assert_allclose(p.head_time_s / 3600, hs[1], rtol=1e-11)
        ver = g[f'{name}/ver']
        np.testing.assert_allclose([p.max_kappa, p.max_alat, p.viol_rate, p.max_jump], ver[[0, 1, 3, 4]],
                                   rtol=1e-9, atol=1e-12, err_msg=name)
        assert p.n_viol == ver[2] and p.passed == bool(ver[5]), name
        ap, dp = g[f'{name}/approach'], g[f'{name}/departure']
        assert (p.approach is None) == (len(ap) == 0) and (p.departure is None) == (len(dp) == 0)
        if p.approach is not None:
            np.testing.assert_allclose(p.approach, ap, rtol=0, atol=2e-10)
        if p.departure is not None:
            np.testing.assert_allclose(p.departure, dp, rtol=0, atol=2e-10)


def test_published_pins(golden_plans):
    """Numbers the reference's docs publish (README_en.md:199-215)."""
    rc, p = orc.plan_field(orc.make_field(L=500.0, H=200.0))
    assert rc == 0
    assert (p.n_swaths, p.n_main, p.n_head, p.n_loops) == (58, 1256, 435, 3)
    assert p.headland_width == 8.0 and p.n_viol == 0 and p.viol_rate == 0.0 and p.n_outside == 0
    assert abs(p.main_len_m / 1000 - 29.504996) < 1e-6 and abs(p.head_len_m / 1000 - 4.326508) < 1e-6
    assert abs(p.main_time_s / 3600 - 3.516820) < 1e-6 and abs(p.head_time_s / 3600 - 0.327350) < 1e-6
    assert abs(p.max_kappa - 0.42278406271172153) < 1e-12
    assert len(np.unique(p.v)) == 13 and len(np.unique(np.round(p.v, 9))) == 11
    # fp-fragile swath counts (SURVEY.md 7 "hard parts"): (H-2R)/W = 8.999...98 -> 9 ; 3.000...04 -> 4
    assert orc.plan_field(orc.make_field(L=500.0, H=44.8))[1].n_swaths == 9
    assert orc.plan_field(orc.make_field(L=500.0, H=25.6))[1].n_swaths == 4


def test_error_cases():
    assert orc.plan_field(orc.make_field(L=15.0, H=200.0))[0] == -1     # MLP:597-598 ValueError
    assert orc.plan_field(orc.make_field(L=500.0, H=16.002))[0] == -1   # inset area 484*0.002 < 1


# ---- corner grid verification (MLP:1426-1578), SURVEY.md 8f-1 -------------------------------------------------------
def test_corner_cover_grids_match_the_reference(golden_cover):
    """orc_cover_grid on the polylines the reference generated reproduces the reference's own grids cell for cell
    (the reference ran verify_all_corners_coverage with the stand-in's exact `contains`, tools/gen_golden.py)."""
    g = golden_cover
    for name in g['names']:
        vp = g[f'{name}/vp']
        W, R = vp[0], vp[1]
        gs = int(2 * R / 0.1)
        befores, afters = [], []
        for ci in range(4):
            k = f'{name}/c{ci}'
            assert tuple(g[k + '/grid_shape']) == (gs, gs)
            want = np.unpackbits(g[k + '/grid_bits'])[:gs * gs].reshape(gs, gs).astype(bool)
            ox, oy = g[k + '/origin']
            counts, grid = orc.cover_grid(ox, oy, 0.1, 0.0, W / 2, gs, gs, g[k + '/turn'], g[k + '/rev'], strict=True)
            assert np.array_equal(grid != 0, want), (name, ci, int(((grid != 0) != want).sum()))
            before, after = counts[1] / (gs * gs) * 100, counts[2] / (gs * gs) * 100
            assert before == g[k + '/cov'][0] and after == g[k + '/cov'][1]
            assert counts[0] == gs * gs and np.array_equal(grid == 1, (grid != 0) & (grid != 2))
            befores.append(before); afters.append(after)
        np.testing.assert_allclose([np.mean(befores), np.mean(afters)], g[f'{name}/avg'][:2], rtol=1e-14)


def test_cover_grid_area_of_a_capsule():
    """sampled area of one buffered segment -> L * 2r + pi r^2 (what Shapely's buffer(...).area tends to)"""
    counts, _ = orc.cover_grid(-2.0, -2.0, 0.01, 0.5, 1.0, 1400, 400, [[0.0, 0.0], [10.0, 0.0]], strict=False, want_grid=False)
    assert abs(counts[1] * 1e-4 - (20 + np.pi)) < 5e-3
    # a region (ring) restricts what counts: outer 14 x 4 box, inner 10 x 2 box around the segment
    def box(x0, y0, x1, y1):
        return [1, 0, -x0, 0, 1, -y0, -1, 0, x1, 0, -1, y1]
    counts, _ = orc.cover_grid(-2.0, -2.0, 0.01, 0.5, 1.0, 1400, 400, [[0.0, 0.0], [10.0, 0.0]], strict=False,
                               region=box(-2, -2, 12, 2) + box(0, -1, 10, 1), want_grid=False)
    assert abs(counts[0] * 1e-4 - (56 - 20)) < 1e-9 and abs(counts[1] * 1e-4 - np.pi) < 5e-3


# ---- GA evolution operators (GA:183-268), SURVEY.md 8f-2 ------------------------------------------------------------
def test_philox_known_answers():
    """Random123's published vectors for philox4x32-10"""
    assert orc.philox4x32([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert orc.philox4x32([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert orc.philox4x32([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


@pytest.mark.parametrize('tag', ['op_a', 'op_b', 'op_c'])
def test_ga_operators_match_the_reference(golden_ga, tag):
    """The reference's _selection / _crossover / _mutation / _elitism, replayed with the decisions they drew."""
    g = golden_ga
    n, pop, k, e = (int(v) for v in g[f'{tag}_cfg'])
    cx_rate, mu_rate = g[f'{tag}_rates']
    population, fitness = g[f'{tag}_population'], g[f'{tag}_fitness']
    sel = orc.ga_selection(population, fitness, g[f'{tag}_cand'])
    assert np.array_equal(sel, g[f'{tag}_selected'])
    off = np.empty_like(sel)
    for p in range(pop // 2):                                   # GA:198-210
        p1, p2 = sel[2 * p], sel[2 * p + 1]
        if g[f'{tag}_u_cx'][p] < cx_rate:
            a, b = sorted(int(v) for v in g[f'{tag}_cuts'][p])
            off[2 * p], off[2 * p + 1] = orc.ga_ox(p1, p2, a, b)
        else:
            off[2 * p], off[2 * p + 1] = p1, p2
    assert np.array_equal(off, g[f'{tag}_offspring'])
    mut = off.copy()
    for r in range(pop):                                        # GA:244-252
        if g[f'{tag}_u_mu'][r] < mu_rate:
            i, j = (int(v) for v in g[f'{tag}_swaps'][r])
            mut[r, i], mut[r, j] = mut[r, j], mut[r, i]
    assert np.array_equal(mut, g[f'{tag}_mutated'])
    assert np.array_equal(orc.ga_elitism(population, fitness, mut, e), g[f'{tag}_combined'])


def test_ga_evolve_oracle_is_a_valid_ga():
    """orc_ga_evolve (Philox decisions): permutations stay permutations, the best never gets worse, elites survive,
    histories are consistent, convergence stops the loop."""
    rng = np.random.default_rng(5)
    n, pop = 24, 40
    pts = rng.uniform(0, 100, size=(n, 2))
    D = np.sqrt(((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1))
    routes = np.array([rng.permutation(n) for _ in range(pop)], dtype=np.int32)
    final, best, hb, ha, res = orc.ga_evolve(D, routes, max_generations=300, elite_size=4, convergence_threshold=40, seed=99)
    assert all(sorted(r) == list(range(n)) for r in final) and sorted(best) == list(range(n))
    assert res.generations == len(hb) <= 300 and (np.diff(hb) >= 0).all() and (ha <= hb + 1e-15).all()
    d0 = orc.ga_distance(routes, D).min()
    assert res.best_distance < d0 and abs(res.best_fitness - 1 / (res.best_distance + 1e-6)) < 1e-18
    assert abs(orc.ga_distance(best[None, :], D)[0] - res.best_distance) == 0
    if res.generations < 300:
        assert res.generations - 1 - res.convergence_gen == 40
    # another seed gives another run; the same seed the same run
    f2 = orc.ga_evolve(D, routes, max_generations=300, elite_size=4, convergence_threshold=40, seed=99)[0]
    f3 = orc.ga_evolve(D, routes, max_generations=300, elite_size=4, convergence_threshold=40, seed=100)[0]
    assert np.array_equal(final, f2) and not np.array_equal(final, f3)


# ---- scheduler inputs (MVP:229-259, MFP:290-320), SURVEY.md 8f-3 ------------------------------------------------------
def test_distance_matrix_matches_the_reference(golden_ga):
    D = orc.distance_matrix(golden_ga['dm_nodes'])
    np.testing.assert_allclose(D, golden_ga['dm_D'], rtol=4.5e-16, atol=0)     # 1 ulp: numpy's norm sums the two squares through its dot kernel
    assert (np.diag(D) == 0).all() and np.array_equal(D, D.T)


def test_best_connection_picks_the_first_shortest_pair():
    f = [[0, 0], [10, 0], [10, 0]]
    t = [[20, 0], [13, 4], [13, -4], [7, 4]]          # distance 5 from f[1] to t[1], t[2], t[3] and from f[2] as well
    assert orc.best_connection(f, t) == (1, 1, 5.0)
    assert orc.best_connection([], t) == (-1, -1, float('inf'))


def _mfp_candidates(g, node):
    """exit / entry candidates of a route node as the reference builds them (MFP:113-140, 292-300): the depot itself, or the field's vertices"""
    return g['depot'][None, :] if node == 0 else g['vertices'][node - 1]


def test_scheduler_inputs_vs_the_reference_multi_field_planner(golden_mfp):
    """_calculate_distance_matrix (MFP:263-288) and _find_best_connection (MFP:290-320) as the reference's MultiFieldPlannerV38 computed
    them for 16 fields (14 parallelograms + two squares with a tie), against the oracle."""
    g = golden_mfp
    nodes = np.vstack([g['depot'][None, :], g['centroids']])
    np.testing.assert_allclose(orc.distance_matrix(nodes), g['D'], rtol=4.5e-16, atol=0)
    route = g['route']
    for k in range(len(route) - 1):
        f, t = _mfp_candidates(g, route[k]), _mfp_candidates(g, route[k + 1])
        bf, bt, bd = orc.best_connection(f, t)
        assert np.array_equal(f[bf], g['conn_from'][k]) and np.array_equal(t[bt], g['conn_to'][k]), k
        np.testing.assert_allclose(bd, g['conn_dist'][k], rtol=4.5e-16, atol=0)
    k = len(route) - 3                                           # tieA -> tieB: two pairs 300 m apart, the first one wins (MFP:308 `<`)
    assert g['conn_from'][k].tolist() == [4200.0, 0.0] and g['conn_to'][k].tolist() == [4500.0, 0.0]


def test_counts_that_hinge_on_the_last_bit_of_a_sine_the_reference_decides():
    """tests/golden/golden_fragile.npz (tools/gen_golden.py --fragile-only; fields picked by tools/fragile_tally.sh, profiles/r05_fragile_tally.txt):
    rectangles whose inset height is an exact multiple of the working width, rotated -- the reference's int((max_y - min_y) / W) + 1 (MLP:739)
    then hinges on the last bit of the rotation's angle, sine and cosine and of the inset's edge lengths.  The library computes those
    correctly rounded since round 5 (csrc/fcpp_math.h: double-double, identical on host and device), as the platform libm -- the oracle's,
    and numpy's for sine and cosine -- does in 99.9 % of arguments: over 1.2 million random fields library and oracle disagree on NO
    rotated parallelogram or quadrilateral (0 of 900 000) and on 0.06 % of such exact-multiple rectangles (169 of 300 000; 1.9 % with the
    1-ulp functions of rounds 1-4), always by one swath.  Here the REFERENCE's own counts decide 90 of those fields (60 of them fields on
    which round 4's library disagreed with the oracle): the oracle has the reference's count on 86 of the 90, the library on 85 (34 in
    round 4); the headland's point count is never in question."""
    from field_coverage_path_planning_amd import engine as E
    g = np.load(os.path.join(GOLDEN, 'golden_fragile.npz'))
    V, st, ref_main, ref_head, mism = g['verts'], g['start'], g['n_main'], g['n_head'], g['mismatch']
    o_main, o_head = [], []
    for k in range(len(V)):
        rc, p = orc.plan_field(orc.make_field(verts=[tuple(x) for x in V[k]], start=None if np.isnan(st[k, 0]) else tuple(st[k])))
        assert rc == 0
        o_main.append(p.n_main)
        o_head.append(p.n_head)
    o_main, o_head = np.array(o_main), np.array(o_head)
    info = E.plan_count(E.FieldTable.from_vertices(V, start_points=st), E.make_vehicle(), E.make_options())
    l_main, l_head = np.array([i.n_main for i in info]), np.array([i.n_head for i in info])
    assert np.array_equal(o_head, ref_head) and np.array_equal(l_head, ref_head)
    # a swath more or less = 22 points of layer 1 (2 of its line, 20 of its turn)
    assert set(np.abs(o_main - ref_main).tolist()) <= {0, 22} and set(np.abs(l_main - ref_main).tolist()) <= {0, 22}
    assert int((o_main != ref_main).sum()) <= 4                       # the oracle: the reference's count on 86 of 90
    assert int((l_main != ref_main).sum()) <= 6                       # the library: on 85 of 90 (round 4: 34)
    assert int((l_main != o_main).sum()) <= 2                         # ... and the oracle's count on 89


def test_a_corner_of_exactly_sixty_degrees_the_reverse_fill_decision():
    """tests/golden/golden_fragile.npz c60_* (tools/gen_golden.py --fragile-only; 25 fields of tools/fragile_tally.sh's class 4,
    profiles/r05_fragile_tally_corner60.txt): parallelograms DRAWN with a corner of 60 degrees, rotated.  MLP:1043 fills a corner in reverse
    when its angle is `>= 60`, the angle being degrees(arccos(c)) with c = 0.5 to the last bit or two (MLP:165-192) -- so the decision is the
    last bit of an arccos.  The library's fc_acos is the correctly rounded arccos in that window since round 5 (csrc/fcpp_math.h; 21 % of such
    fields differed from the oracle's libm acos before, none of 240 000 now).  The reference run HERE is not a fixed point to aim at on
    these fields: numpy 2.2's float64 arccos loop on this CPU differs from libm on 9 % of arguments (arccos(0.5) comes out one ulp low:
    59.99999999999999 degrees, libm and the exact value give 60.00000000000001) and its BLAS dot product fuses the multiply-add, so the
    fixture's counts are those of ONE platform; the library and the oracle agree with each other on all 25 and with that run on 15, and
    where they differ the library has the reverse fill (more headland points), never fewer."""
    from field_coverage_path_planning_amd import engine as E
    g = np.load(os.path.join(GOLDEN, 'golden_fragile.npz'))
    V, st, ref_main, ref_head = g['c60_verts'], g['c60_start'], g['c60_n_main'], g['c60_n_head']
    assert len(V) == 25
    o_main, o_head = [], []
    for k in range(len(V)):
        rc, p = orc.plan_field(orc.make_field(verts=[tuple(x) for x in V[k]], start=None if np.isnan(st[k, 0]) else tuple(st[k])))
        assert rc == 0
        o_main.append(p.n_main)
        o_head.append(p.n_head)
    info = E.plan_count(E.FieldTable.from_vertices(V, start_points=st), E.make_vehicle(), E.make_options())
    l_main, l_head = np.array([i.n_main for i in info]), np.array([i.n_head for i in info])
    assert np.array_equal(l_main, np.array(o_main)) and np.array_equal(l_head, np.array(o_head))
    assert np.array_equal(l_main, ref_main)                           # layer 1 is not in question
    assert int((l_head == ref_head).sum()) >= 15 and (l_head >= ref_head).all()
