#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN CODE in this container.

Run once here (the reference does not travel to the GPU box):

    python tools/gen_golden.py            # writes tests/golden/*.npz

Two tiers, kept in separate files so that tests can say which they rely on:

  tier A  (golden_kernels.npz, golden_ga.npz)
      direct calls of the reference's pure-numpy methods -- curvature, speed
      clamp, forward/backward sweeps, arc / corner / straight / reverse
      samplers, length / time metrics, curvature verifier, GA tour length.
      These never touch Shapely: outputs are 100 % reference-determined.

  tier B  (golden_plans.npz)
      the reference's own plan_complete_coverage() end to end.  Shapely is not
      installed here, so the import is satisfied by tools/_shapely_standin.py
      (convex-quad inset / bounds / centroid, analytic corner-gap bound).  Values
      that would depend on real GEOS clipping (coverage_rate, corner grids) are
      NOT recorded.  Counts reproduce every number the reference's docs publish
      (1256 / 435 points, 3 loops, ...): asserted at the bottom of this script.
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get('FCPP_REFERENCE', '/root/reference')
OUT = os.path.join(REPO, 'tests', 'golden')

sys.path.insert(0, HERE)
import _shapely_standin  # noqa: E402

_shapely_standin.install()
sys.path.insert(0, REF)
with contextlib.redirect_stdout(io.StringIO()):
    import matplotlib
    matplotlib.use('Agg')
    import multi_layer_planner_v3 as mlp  # the reference  # noqa: E402
    import genetic_algorithm_solver as gas  # the reference  # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


VP_FIELDS = ['working_width', 'min_turn_radius', 'max_work_speed_kmh', 'max_headland_speed_kmh',
             'headland_turn_speed_kmh', 'max_lateral_accel', 'max_longitudinal_accel', 'safety_factor']


def vp_array(vp):
    return np.array([getattr(vp, f) for f in VP_FIELDS], dtype=np.float64)


def bare_planner(vp, L=500.0, H=200.0):
    """A planner object without running __init__ (which needs Shapely)."""
    p = object.__new__(mlp.TwoLayerPathPlannerV37)
    p.vehicle = vp
    p.field_length = L
    p.field_width = H
    return p


class _Bounds:
    def __init__(self, b):
        self.bounds = tuple(b)


# ----------------------------------------------------------------------------
# tier A
# ----------------------------------------------------------------------------
def tier_a():
    rng = np.random.default_rng(20251024)
    out = {}
    vp = mlp.VehicleParams()
    pl = bare_planner(vp)
    out['vp_default'] = vp_array(vp)

    # -- curvature (MLP:513-536) on random + degenerate triples
    tri = rng.uniform(-50, 50, size=(256, 3, 2))
    tri[0, 1] = tri[0, 0]                       # ds1 == 0
    tri[1, 2] = tri[1, 1]                       # ds2 == 0
    tri[2] = np.array([[0, 0], [1, 0], [2, 0]])  # straight
    tri[3] = np.array([[0, 0], [1, 0], [0, 0]])  # reversal (dtheta = pi)
    tri[4] = np.array([[0, 0], [1, 0], [1, 1e-7]])  # ds2 < 1e-6
    tri[5] = np.array([[0, 0], [1, 0], [1 + 1e-3, 1e-9]])
    tri[6:40] *= 1e-2
    tri[40:80] *= 1e2
    out['curv_tri'] = tri
    out['curv_kappa'] = np.array([pl._calculate_curvature(t[0], t[1], t[2]) for t in tri])

    # -- speed planner (MLP:467-589) on synthetic paths with duplicates / jumps
    paths, speeds_in, speeds_out, smooth_out = [], [], [], []
    for k in range(12):
        n = int(rng.integers(3, 400))
        step = rng.uniform(0.02, 3.0)
        th = np.cumsum(rng.normal(0, 0.15, n))
        seg = rng.uniform(0.2, 1.8, n) * step
        xy = np.cumsum(np.column_stack([seg * np.cos(th), seg * np.sin(th)]), axis=0)
        # duplicates and jumps like the reference's own junctions
        for j in rng.integers(1, n, size=max(1, n // 25)):
            xy[j] = xy[j - 1]
        if k % 3 == 0 and n > 20:
            xy[n // 2:] += rng.uniform(-30, 30, 2)
        v = rng.choice([2.5, 4.0, 9.0, 14.0, 15.0], size=n)
        if k == 0:
            xy, v = xy[:3], v[:3]
        paths.append(xy)
        speeds_in.append(v)
        speeds_out.append(quiet(pl._apply_curvature_based_speed_limit, xy, v))
        smooth_out.append(pl._smooth_speed_profile(xy, v))
    out['sp_offsets'] = np.cumsum([0] + [len(p) for p in paths]).astype(np.int64)
    out['sp_path'] = np.vstack(paths)
    out['sp_v_in'] = np.concatenate(speeds_in)
    out['sp_v_out'] = np.concatenate(speeds_out)      # clamp + fwd + bwd
    out['sp_v_smooth_only'] = np.concatenate(smooth_out)  # fwd + bwd only

    # other vehicle parameters
    vp2 = mlp.VehicleParams(working_width=2.5, min_turn_radius=6.0, max_lateral_accel=1.2,
                            max_longitudinal_accel=0.7, safety_factor=0.9)
    pl2 = bare_planner(vp2)
    out['vp2'] = vp_array(vp2)
    out['sp2_v_out'] = np.concatenate([quiet(pl2._apply_curvature_based_speed_limit, p, v)
                                       for p, v in zip(paths, speeds_in)])

    # -- verifier (MLP:1373-1424) on those paths with their planned speeds
    ver = []
    for p, v in zip(paths, speeds_out):
        d = quiet(pl.verify_curvature_constraints, p, v)
        if 'max_lateral_accel' not in d:  # n < 3 early return
            ver.append([d['max_curvature'], 0, 0, 0, 0, 1])
        else:
            ver.append([d['max_curvature'], d['max_lateral_accel'], d['accel_violations'],
                        d['accel_violation_rate'], d['max_jump'], float(d['pass'])])
    out['ver_stats'] = np.array(ver, dtype=np.float64)
    # a path that does violate: planned speeds replaced by a constant 15 km/h
    vv = [quiet(pl.verify_curvature_constraints, p, np.full(len(p), 15.0)) for p in paths[1:]]
    out['ver15_stats'] = np.array([[d['max_curvature'], d['max_lateral_accel'], d['accel_violations'],
                                    d['accel_violation_rate'], d['max_jump'], float(d['pass'])]
                                   for d in vv])

    # -- metrics (MLP:1290-1311)
    out['len_m'] = np.array([pl._calculate_path_length(p) for p in paths])
    out['time_s'] = np.array([pl._calculate_work_time(p, v) for p, v in zip(paths, speeds_out)])

    # -- samplers
    arcs = []
    for (sx, sy, right, mnx, mxx) in [(484.0, 8.0, True, 8.0, 492.0), (16.0, 11.2, False, 8.0, 492.0),
                                      (120.5, 33.3, True, 7.25, 131.75)]:
        a, s = pl._generate_safe_arc_turn(np.array([sx, sy]), sy + 3.2, right, mnx, mxx)
        assert s == [4.0] * 20
        arcs.append(a)
    out['uturn_args'] = np.array([[484.0, 8.0, 1, 8.0, 492.0], [16.0, 11.2, 0, 8.0, 492.0],
                                  [120.5, 33.3, 1, 7.25, 131.75]])
    out['uturn_pts'] = np.array(arcs)
    corners = []
    for ci in range(4):
        a, s = pl._generate_corner_turn_arc((100.25 + ci, 50.5 - ci), ci)
        corners.append(a)
    out['corner_arc_pts'] = np.array(corners)       # corner (100.25+ci, 50.5-ci), index ci
    out['straight_pts'] = pl._generate_straight_segment((1.6, 1.6), (498.4, 1.6), 20)
    out['straight2_pts'] = pl._generate_straight_segment((498.4, 1.6), (498.4, 198.4), 20)
    out['approach_pts'] = pl._generate_approach_path((10.0, 10.0), (1.6, 1.6))

    # reverse fill (MLP:1154-1288) for the four corners of the outer loop of 500x200
    W, R = vp.working_width, vp.min_turn_radius
    cs = [(W / 2, W / 2), (500 - W / 2, W / 2), (500 - W / 2, 200 - W / 2), (W / 2, 200 - W / 2)]
    rev_pts, rev_len = [], []
    for ci in range(4):
        arc, _ = pl._generate_corner_turn_arc(cs[ci], ci)
        rp, rl = quiet(pl._generate_optimal_reverse_path, None, arc[-1], arc[-2], W, ci)
        rev_pts.append(rp)
        rev_len.append(rl)
    out['rev_offsets'] = np.cumsum([0] + [len(r) for r in rev_pts]).astype(np.int64)
    out['rev_pts'] = np.vstack(rev_pts)
    out['rev_len'] = np.array(rev_len)

    # u-pattern in rotated space (MLP:720-789): all four order/side combinations
    ups, upv, upo = [], [], [0]
    for (b, ro, sr) in [((8, 8, 492, 192), False, False), ((8, 8, 492, 192), True, False),
                        ((8, 8, 492, 192), False, True), ((8, 8, 492, 192), True, True),
                        ((8, 8, 92, 36.8), False, False), ((8, 8, 92, 17.6), True, True),
                        ((-3.7, 12.2, 140.3, 25.0), False, True)]:
        p, s = pl._generate_u_pattern_in_rotated_space(_Bounds(b), ro, sr)
        ups.append(p)
        upv.append(s)
        upo.append(upo[-1] + len(p))
    out['upat_args'] = np.array([[8, 8, 492, 192, 0, 0], [8, 8, 492, 192, 1, 0], [8, 8, 492, 192, 0, 1],
                                 [8, 8, 492, 192, 1, 1], [8, 8, 92, 36.8, 0, 0], [8, 8, 92, 17.6, 1, 1],
                                 [-3.7, 12.2, 140.3, 25.0, 0, 1]], dtype=np.float64)
    out['upat_offsets'] = np.array(upo, dtype=np.int64)
    out['upat_pts'] = np.vstack(ups)
    out['upat_v'] = np.concatenate(upv)

    # rotate point (MLP:265-284)
    rp_in = rng.uniform(-100, 100, size=(32, 5))  # x, y, angle, cx, cy
    out['rot_in'] = rp_in
    out['rot_out'] = np.array([pl._rotate_point((r[0], r[1]), r[2], (r[3], r[4])) for r in rp_in])
    return out


def tier_ga():
    rng = np.random.default_rng(128)
    out = {}
    solver = gas.GeneticAlgorithmSolver(gas.GAConfig())
    for tag, n, pop in [('n10', 10, 32), ('n128', 128, 64), ('n129', 129, 16), ('n33', 33, 48)]:
        pts = rng.uniform(0, 1000, size=(n, 2))
        D = np.sqrt(((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1))
        if tag == 'n33':  # asymmetric matrix: direction of the look-up matters
            D = D + rng.uniform(0, 5, size=D.shape)
        routes = np.array([rng.permutation(n) for _ in range(pop)], dtype=np.int32)
        dist = np.array([solver._calculate_distance(list(map(int, r)), D) for r in routes])
        fit = np.array([solver._calculate_fitness(list(map(int, r)), D) for r in routes])
        out[f'{tag}_D'] = D
        out[f'{tag}_routes'] = routes
        out[f'{tag}_dist'] = dist
        out[f'{tag}_fit'] = fit
    out.update(tier_ga_operators(rng))
    # MultiVehiclePlanner._build_distance_matrix (MVP:229-259), the matrix the GA consumes: depot + 60 field centroids
    import multi_vehicle_planner as mvp  # the reference
    cent = rng.uniform(-2000, 2000, size=(60, 2))
    depot = (13.5, -7.25)
    ids = [f'f{i}' for i in range(60)]
    pl = object.__new__(mvp.MultiVehiclePlanner)
    out['dm_nodes'] = np.vstack([np.array(depot)[None, :], cent])
    out['dm_D'] = quiet(pl._build_distance_matrix, ids, {k: {'centroid': tuple(c)} for k, c in zip(ids, cent)}, depot)
    return out


class _ScriptedRandom:
    """Stands in for the stdlib `random` module inside the reference's GA module: every call returns the next scripted decision,
    so that the operators' outputs can be stored next to the decisions that produced them."""

    def __init__(self, samples, uniforms):
        self.samples, self.uniforms = list(samples), list(uniforms)

    def sample(self, population, k):
        s = self.samples.pop(0)
        assert len(s) == k and all(0 <= v < len(population) for v in s) and len(set(s)) == k
        return list(s)

    def random(self):
        return self.uniforms.pop(0)


def tier_ga_operators(rng):
    """Reference `_selection`, `_crossover`, `_mutation`, `_elitism` (GA:183-268) on seeded populations with scripted draws."""
    out = {}
    real_random = gas.random
    try:
        for tag, n, pop, k, e in (('op_a', 12, 10, 3, 2), ('op_b', 129, 64, 5, 20), ('op_c', 40, 32, 5, 6)):
            cfg = gas.GAConfig(population_size=pop, tournament_size=k, elite_size=e, crossover_rate=0.85, mutation_rate=0.3)
            solver = gas.GeneticAlgorithmSolver(cfg)
            pts = rng.uniform(0, 1000, size=(n, 2))
            D = np.sqrt(((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1))
            population = [list(map(int, rng.permutation(n))) for _ in range(pop)]
            fitness = [solver._calculate_fitness(r, D) for r in population]
            cand = np.array([rng.choice(pop, size=k, replace=False) for _ in range(pop)], dtype=np.int32)
            gas.random = _ScriptedRandom([list(map(int, c)) for c in cand], [])
            selected = solver._selection(population, fitness)
            u_cx = rng.uniform(0, 1, size=pop // 2)
            cuts = np.array([rng.choice(n, size=2, replace=False) for _ in range(pop // 2)], dtype=np.int32)
            gas.random = _ScriptedRandom([list(map(int, c)) for c, u in zip(cuts, u_cx) if u < cfg.crossover_rate], list(u_cx))
            offspring = solver._crossover([r.copy() for r in selected])
            u_mu = rng.uniform(0, 1, size=pop)
            swaps = np.array([rng.choice(n, size=2, replace=False) for _ in range(pop)], dtype=np.int32)
            gas.random = _ScriptedRandom([list(map(int, c)) for c, u in zip(swaps, u_mu) if u < cfg.mutation_rate], list(u_mu))
            mutated = solver._mutation([r.copy() for r in offspring])
            combined = solver._elitism(population, mutated, fitness)
            assert len(combined) == pop
            out.update({f'{tag}_cfg': np.array([n, pop, k, e]), f'{tag}_rates': np.array([cfg.crossover_rate, cfg.mutation_rate]),
                        f'{tag}_D': D, f'{tag}_population': np.array(population, dtype=np.int32), f'{tag}_fitness': np.array(fitness),
                        f'{tag}_cand': cand, f'{tag}_selected': np.array(selected, dtype=np.int32), f'{tag}_u_cx': u_cx,
                        f'{tag}_cuts': cuts, f'{tag}_offspring': np.array(offspring, dtype=np.int32), f'{tag}_u_mu': u_mu,
                        f'{tag}_swaps': swaps, f'{tag}_mutated': np.array(mutated, dtype=np.int32),
                        f'{tag}_combined': np.array(combined, dtype=np.int32)})
    finally:
        gas.random = real_random
    return out


# ----------------------------------------------------------------------------
# tier B
# ----------------------------------------------------------------------------
def parallelogram(base, height, ang_deg, rot, ox=0.0, oy=0.0):
    a = np.radians(ang_deg)
    sx = height / np.tan(a)
    v = np.array([[0, 0], [base, 0], [base + sx, height], [sx, height]], dtype=np.float64)
    c, s = np.cos(rot), np.sin(rot)
    v = v @ np.array([[c, s], [-s, c]])
    v[:, 0] += ox
    v[:, 1] += oy
    return [(float(x), float(y)) for x, y in v]


def scenarios():
    sc = []
    d = dict
    sc.append(d(name='cfg1_500x200', L=500.0, H=200.0))
    sc.append(d(name='v351_start_end', L=500.0, H=200.0, start=(10, 10), end=(490, 190),
                vp=d(max_headland_speed_kmh=14.0)))
    sc.append(d(name='v37_small', L=100.0, H=80.0, start=(90, 70)))
    sc.append(d(name='v37_medium', L=500.0, H=200.0, start=(50, 180)))
    sc.append(d(name='v37_large', L=3500.0, H=320.0, start=(3400, 300)))
    sc.append(d(name='start_lr', L=300.0, H=120.0, start=(290, 5)))       # lower-right
    sc.append(d(name='exact_mult_44p8', L=500.0, H=44.8))                  # (H-2R)/W = 8.999..98
    sc.append(d(name='exact_mult_25p6', L=500.0, H=25.6))                  # (H-2R)/W = 3.000..04
    sc.append(d(name='exact_mult_48', L=200.0, H=48.0))                    # (H-2R)/W = 10 exactly?
    sc.append(d(name='one_pass', L=60.0, H=18.5))
    sc.append(d(name='obstacle_ignored', L=500.0, H=200.0,
                obstacles=[[(150, 80), (180, 80), (180, 110), (150, 110)]]))
    sc.append(d(name='veh_w2p5_r6', L=240.0, H=130.0, vp=d(working_width=2.5, min_turn_radius=6.0)))
    sc.append(d(name='veh_w4_r5', L=333.0, H=77.0, start=(300, 70), end=(5, 5),
                vp=d(working_width=4.0, min_turn_radius=5.0, max_work_speed_kmh=12.0,
                     max_lateral_accel=1.5, max_longitudinal_accel=1.0, safety_factor=0.8)))
    sc.append(d(name='veh_w3p5_r4', L=150.0, H=90.0, vp=d(working_width=3.5, min_turn_radius=4.0)))
    rng = np.random.default_rng(1024)
    for i in range(6):
        L, H = rng.uniform(100, 1000, 2)
        s = d(name=f'rand_rect_{i}', L=float(L), H=float(H))
        if i % 2:
            s['start'] = (float(rng.uniform(0, L)), float(rng.uniform(0, H)))
        sc.append(s)
    # vertex input: axis-aligned rectangle given as vertices, tilted rectangle, parallelograms
    sc.append(d(name='verts_rect', verts=[(0, 0), (400, 0), (400, 150), (0, 150)]))
    sc.append(d(name='verts_tilted_rect', verts=parallelogram(300, 120, 90, 0.35)))
    sc.append(d(name='verts_para_75', verts=parallelogram(400, 160, 75, 0.0)))
    sc.append(d(name='verts_para_110_rot', verts=parallelogram(350, 140, 110, -0.5, 0, 0), start=(120, 30)))
    sc.append(d(name='verts_para_65_rot', verts=parallelogram(500, 200, 65, 0.6)))
    sc.append(d(name='verts_para_55_rot', verts=parallelogram(420, 180, 55, 0.2)))   # corner < 60 deg
    rng = np.random.default_rng(65536)
    for i in range(4):
        b, h = rng.uniform(100, 1000, 2)
        ang = rng.uniform(60, 120)
        rot = rng.uniform(-np.pi / 4, np.pi / 4)
        sc.append(d(name=f'rand_para_{i}', verts=parallelogram(float(b), float(h), float(ang), float(rot))))
    # convex quadrilaterals that are neither rectangles nor parallelograms (`_detect_field_shape` -> 'other', MLP:137-163): the
    # reference plans them like any other field
    sc.append(d(name='other_trapezoid', verts=[(0.0, 0.0), (400.0, 0.0), (350.0, 150.0), (30.0, 150.0)]))
    sc.append(d(name='other_trapezoid_rot', verts=rotate([(0.0, 0.0), (520.0, 0.0), (440.0, 210.0), (90.0, 190.0)], 0.4), start=(150.0, 60.0)))
    sc.append(d(name='other_kite', verts=rotate([(0.0, 0.0), (300.0, -20.0), (360.0, 170.0), (-10.0, 140.0)], -0.25)))
    # the same plans with the inset corners listed the other way round (the stand-in's RING_ORDER = 1, fcpp_options.ring_order = 1)
    for base in ('cfg1_500x200', 'v351_start_end', 'v37_small', 'veh_w4_r5', 'verts_para_75', 'verts_para_110_rot', 'other_trapezoid',
                 'other_trapezoid_rot'):
        src = next(x for x in sc if x['name'] == base)
        sc.append(dict(src, name='cw_' + base, ring_order=1))
    return sc


def rotate(vs, ang):
    c, s_ = np.cos(ang), np.sin(ang)
    return [(float(x * c - y * s_), float(x * s_ + y * c)) for x, y in vs]


def run_scenario(s):
    vp = mlp.VehicleParams(**s.get('vp', {}))
    # the stand-in answers the corner-gap area (MLP:1070) with a lower bound: only keep
    # scenarios where that bound already decides `gap.area > 0.1`
    R_, W_ = vp.min_turn_radius, vp.working_width
    assert 4 * R_ * R_ - (np.pi * R_ / 2 * W_ + np.pi * W_ * W_ / 4) > 0.1, s['name']
    kw = dict(vehicle_params=vp, obstacles=s.get('obstacles'), start_point=s.get('start'),
              end_point=s.get('end'))
    if 'verts' in s:
        kw['field_vertices'] = s['verts']
    else:
        kw['field_length'], kw['field_width'] = s['L'], s['H']
    _shapely_standin.RING_ORDER = int(s.get('ring_order', 0))
    try:
        pl = quiet(mlp.TwoLayerPathPlannerV37, **kw)
        res = quiet(pl.plan_complete_coverage)
    finally:
        _shapely_standin.RING_ORDER = 0
    mp, hp = res['main_work']['path'], res['headland']['path']
    ms, hs = res['main_work']['speeds'], res['headland']['speeds']
    allp, alls = np.vstack([mp, hp]), np.concatenate([ms, hs])
    ver = quiet(pl.verify_curvature_constraints, allp, alls)
    o = {}
    o['vp'] = vp_array(vp)
    o['ring_order'] = np.array(int(s.get('ring_order', 0)))
    o['verts'] = np.array(pl.field_vertices, dtype=np.float64)
    o['is_verts_input'] = np.array(int('verts' in s))
    o['start'] = np.array(s.get('start', (np.nan, np.nan)), dtype=np.float64)
    o['end'] = np.array(s.get('end', (np.nan, np.nan)), dtype=np.float64)
    o['start_kept'] = np.array(int(pl.start_point is not None))
    o['end_kept'] = np.array(int(pl.end_point is not None))
    o['shape'] = np.array(pl.field_shape)
    o['corner_angles'] = np.array(pl.corner_angles, dtype=np.float64)
    o['field_LH'] = np.array([pl.field_length, pl.field_width], dtype=np.float64)
    o['headland_width'] = np.array(pl.headland_width)
    o['main_path'], o['main_v'] = mp, ms
    o['head_path'], o['head_v'] = hp, hs
    o['main_stats'] = np.array([res['main_work']['stats'][k] for k in
                                ('path_length_km', 'time_hours', 'avg_speed_kmh')])
    o['head_stats'] = np.array([res['headland']['stats'][k] for k in
                                ('path_length_km', 'time_hours', 'avg_speed_kmh')])
    o['approach'] = res['approach_path'] if res['approach_path'] is not None else np.zeros((0, 2))
    o['departure'] = res['departure_path'] if res['departure_path'] is not None else np.zeros((0, 2))
    o['ver'] = np.array([ver['max_curvature'], ver['max_lateral_accel'], ver['accel_violations'],
                         ver['accel_violation_rate'], ver['max_jump'], float(ver['pass'])])
    if s.get('obstacles'):
        flat = [np.array(ob, dtype=np.float64) for ob in s['obstacles']]
        o['obs_offsets'] = np.cumsum([0] + [len(f) for f in flat]).astype(np.int64)
        o['obs_xy'] = np.vstack(flat)
    return o


def tier_b():
    out = {}
    names = []
    for s in scenarios():
        o = run_scenario(s)
        names.append(s['name'])
        for k, v in o.items():
            out[f"{s['name']}/{k}"] = v
    out['names'] = np.array(names)
    return out


def tier_cover():
    """Corner grid verification (MLP:1426-1578) run by the reference itself; `contains` comes from the stand-in."""
    out, names = {}, []
    for name, L, H, kw in (('500x200', 500.0, 200.0, {}), ('300x120_w2.4_r6', 300.0, 120.0, dict(working_width=2.4, min_turn_radius=6.0)),
                           ('200x100_w5_r4.5', 200.0, 100.0, dict(working_width=5.0, min_turn_radius=4.5))):
        vp = mlp.VehicleParams(**kw)
        pl = quiet(mlp.TwoLayerPathPlannerV37, vp, field_length=L, field_width=H)
        res = quiet(pl.verify_all_corners_coverage, None)
        names.append(name)
        out[f'{name}/LH'] = np.array([L, H])
        out[f'{name}/vp'] = vp_array(vp)
        out[f'{name}/avg'] = np.array([res['avg_coverage_before'], res['avg_coverage_after'], res['avg_improvement']])
        hw = pl.headland_width
        corners = [(hw, hw, 0), (L - hw, hw, 1), (L - hw, H - hw, 2), (hw, H - hw, 3)]
        for (cx, cy, ci), r in zip(corners, res['corners']):
            # the same generator calls verify_all_corners_coverage makes (MLP:1546-1558), to record the polylines
            turn, _ = pl._generate_corner_turn_arc((cx, cy), ci)
            gap = pl._calculate_corner_gap_precise((cx, cy), ci, vp.min_turn_radius, vp.working_width)
            rev = np.zeros((0, 2))
            if gap is not None and gap.area > 0.1:
                rev, _ = quiet(pl._generate_optimal_reverse_path, gap, turn[-1], turn[-2], vp.working_width, ci)
            k = f'{name}/c{ci}'
            out[k + '/corner'] = np.array([cx, cy])
            out[k + '/turn'] = np.asarray(turn, dtype=np.float64)
            out[k + '/rev'] = np.asarray(rev, dtype=np.float64)
            out[k + '/origin'] = np.array(r['grid_origin'], dtype=np.float64)
            out[k + '/grid_shape'] = np.array(r['grid'].shape)
            out[k + '/grid_bits'] = np.packbits(r['grid'])
            out[k + '/cov'] = np.array([r['coverage_before'], r['coverage_after'], r['improvement']])
    out['names'] = np.array(names)
    return out


def tier_mfp():
    """MultiFieldPlannerV38 (MFP:63-320) run for real: its constructor (_prepare_fields: centroid, entry / exit candidates = the
    field's vertices), _calculate_distance_matrix (MFP:263-288) and _find_best_connection (MFP:290-320) for every consecutive pair
    of a route over the fields.  multi_field_planner.py imports a class name the reference never defines (MFP:24,
    TwoLayerPathPlannerV36): the alias below supplies it, as SURVEY.md 0 prescribes; nothing else is touched."""
    mlp.TwoLayerPathPlannerV36 = mlp.TwoLayerPathPlannerV37
    with contextlib.redirect_stdout(io.StringIO()):
        import multi_field_planner as mfp  # the reference
    rng = np.random.default_rng(38)
    defs = []
    for k in range(14):
        cx, cy = rng.uniform(-1500, 1500, 2)
        base, height = rng.uniform(120, 600, 2)
        v = parallelogram(base, height, rng.uniform(65, 115), rng.uniform(-0.6, 0.6), cx, cy)
        defs.append({'id': f'f{k}', 'vertices': [tuple(map(float, q)) for q in v]})
    # two squares whose facing sides give TWO equally short connections: the reference keeps the first one (`<`, MFP:308)
    defs.append({'id': 'tieA', 'vertices': [(4000.0, 0.0), (4200.0, 0.0), (4200.0, 200.0), (4000.0, 200.0)]})
    defs.append({'id': 'tieB', 'vertices': [(4500.0, 0.0), (4700.0, 0.0), (4700.0, 200.0), (4500.0, 200.0)]})
    depot = (25.0, -40.0)
    pl = quiet(mfp.MultiFieldPlannerV38, defs, depot, mlp.VehicleParams(), 1, '2opt')
    D, node_ids = quiet(pl._calculate_distance_matrix)
    out = {'depot': np.array(depot), 'D': D,
           'centroids': np.array([pl.fields[i].centroid for i in node_ids[1:]], dtype=np.float64),
           'vertices': np.array([np.asarray(pl.fields[i].vertices, dtype=np.float64) for i in node_ids[1:]])}
    order = [int(v) for v in rng.permutation(14)] + [14, 15]            # ... tieA -> tieB last
    route = ['depot'] + [node_ids[1 + k] for k in order] + ['depot']
    fr, to, dist = [], [], []
    for a, b in zip(route[:-1], route[1:]):
        c = quiet(pl._find_best_connection, a, b)
        fr.append(np.asarray(c.from_point, dtype=np.float64)); to.append(np.asarray(c.to_point, dtype=np.float64)); dist.append(c.distance)
    out['route'] = np.array([0] + [1 + k for k in order] + [0], dtype=np.int64)        # node indices (0 = depot)
    out['conn_from'] = np.array(fr); out['conn_to'] = np.array(to); out['conn_dist'] = np.array(dist)
    return out


def tier_fragile():
    """The fields on which a count hinges on the last bit of a sine (profiles/r05_fragile_tally.txt: rectangles whose inset height is an exact
    multiple of the working width, under rotation -- the library's fcpp_math.h and the oracle's libm disagree on 2 % of them): the REFERENCE
    decides.  Per field: its vertices and start point, and the reference's own len(main path), len(headland path)."""
    rows = []
    for line in open(os.path.join(os.path.dirname(OUT.rstrip('/')), '..', 'profiles', 'r05_fragile_tally.txt')):
        if not (line.startswith('MISMATCH') or line.startswith('FRAGILE-AGREE')):
            continue
        w = line.split()
        k = w.index('verts')
        v = [float.fromhex(x) for x in w[k + 1:k + 9]]
        start = None
        if 'start' in w:
            j = w.index('start')
            if w[j + 1] == '1':
                start = (float.fromhex(w[j + 2]), float.fromhex(w[j + 3]))
        rows.append((line.startswith('MISMATCH'), [(v[0], v[1]), (v[2], v[3]), (v[4], v[5]), (v[6], v[7])], start))
    out = {'verts': [], 'start': [], 'mismatch': [], 'n_main': [], 'n_head': []}
    for mism, verts, start in rows:
        pl = quiet(mlp.TwoLayerPathPlannerV37, vehicle_params=mlp.VehicleParams(), field_vertices=verts, start_point=start)
        res = quiet(pl.plan_complete_coverage)
        out['verts'].append(verts)
        out['start'].append(start if start is not None else (np.nan, np.nan))
        out['mismatch'].append(int(mism))
        out['n_main'].append(len(res['main_work']['path']))
        out['n_head'].append(len(res['headland']['path']))
    # a corner of exactly 60 degrees (MLP:1043 `>= 60` on arccos's last bit): tools/fragile_tally.sh's class 4, a sample of its fields
    c60 = {'c60_verts': [], 'c60_start': [], 'c60_n_main': [], 'c60_n_head': []}
    for line in open(os.path.join(os.path.dirname(OUT.rstrip('/')), '..', 'profiles', 'r05_fragile_tally_corner60.txt')):
        if not line.startswith('CORNER60'):
            continue
        w = line.split()
        k = w.index('verts')
        v = [float.fromhex(x) for x in w[k + 1:k + 9]]
        j = w.index('start')
        start = (float.fromhex(w[j + 2]), float.fromhex(w[j + 3])) if w[j + 1] == '1' else None
        verts = [(v[0], v[1]), (v[2], v[3]), (v[4], v[5]), (v[6], v[7])]
        pl = quiet(mlp.TwoLayerPathPlannerV37, vehicle_params=mlp.VehicleParams(), field_vertices=verts, start_point=start)
        res = quiet(pl.plan_complete_coverage)
        c60['c60_verts'].append(verts)
        c60['c60_start'].append(start if start is not None else (np.nan, np.nan))
        c60['c60_n_main'].append(len(res['main_work']['path']))
        c60['c60_n_head'].append(len(res['headland']['path']))
    out.update(c60)
    return {k: np.array(v) for k, v in out.items()}


def main():
    os.makedirs(OUT, exist_ok=True)
    if '--fragile-only' in sys.argv:
        f = tier_fragile()
        np.savez_compressed(os.path.join(OUT, 'golden_fragile.npz'), **f)
        print('golden_fragile.npz:', len(f['n_main']), 'fields,', int(f['mismatch'].sum()), 'of them fields the library and the oracle disagree on')
        return
    if '--mfp-only' in sys.argv:
        m = tier_mfp()
        np.savez_compressed(os.path.join(OUT, 'golden_mfp.npz'), **m)
        print('golden_mfp.npz:', {k: v.shape for k, v in m.items()})
        return
    if '--ga-only' in sys.argv:
        g = tier_ga()
        np.savez_compressed(os.path.join(OUT, 'golden_ga.npz'), **g)
        print('golden_ga.npz:', len(g), 'arrays')
        return
    if '--cover-only' in sys.argv:
        c = tier_cover()
        np.savez_compressed(os.path.join(OUT, 'golden_cover.npz'), **c)
        for n in c['names']:
            print(n, c[n + '/avg'], [c[f'{n}/c{k}/cov'].round(2).tolist() for k in range(4)])
        return
    a = tier_a()
    np.savez_compressed(os.path.join(OUT, 'golden_kernels.npz'), **a)
    g = tier_ga()
    np.savez_compressed(os.path.join(OUT, 'golden_ga.npz'), **g)
    b = tier_b()
    np.savez_compressed(os.path.join(OUT, 'golden_plans.npz'), **b)
    np.savez_compressed(os.path.join(OUT, 'golden_cover.npz'), **tier_cover())
    np.savez_compressed(os.path.join(OUT, 'golden_mfp.npz'), **tier_mfp())
    np.savez_compressed(os.path.join(OUT, 'golden_fragile.npz'), **tier_fragile())

    # --- the reference's own published pins (README_en.md:199-215, doc/V3.5.1:109-111)
    assert len(b['cfg1_500x200/main_path']) == 1256, len(b['cfg1_500x200/main_path'])
    assert len(b['cfg1_500x200/head_path']) == 435
    assert b['cfg1_500x200/ver'][2] == 0 and b['cfg1_500x200/ver'][3] == 0.0
    ap = b['v351_start_end/approach']
    dp = b['v351_start_end/departure']
    la = np.sqrt((np.diff(ap, axis=0) ** 2).sum(1)).sum()
    ld = np.sqrt((np.diff(dp, axis=0) ** 2).sum(1)).sum()
    assert abs(la - 11.9) < 0.05 and abs(ld - 515.2) < 0.05, (la, ld)
    assert len(b['v37_small/main_path']) == 442 and len(b['v37_large/main_path']) == 2092
    tot = sum(v.nbytes for d in (a, g, b) for v in d.values())
    print(f'golden written to {OUT}: {len(a)}+{len(g)}+{len(b)} arrays, {tot/1e6:.2f} MB raw')
    for n in b['names']:
        print(f"  {n:24s} main={len(b[n + '/main_path']):5d} head={len(b[n + '/head_path']):4d} "
              f"shape={b[n + '/shape']}")


if __name__ == '__main__':
    main()
