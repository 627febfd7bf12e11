"""ctypes bindings of the C oracle (oracle/fcpp_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libfcpp_oracle.so')

c_double_p = C.POINTER(C.c_double)
c_i64_p = C.POINTER(C.c_int64)
c_i32_p = C.POINTER(C.c_int32)
c_u32_p = C.POINTER(C.c_uint32)

KIND_SWATH, KIND_UTURN, KIND_HEAD_START, KIND_HEAD_STRAIGHT, KIND_CORNER, KIND_REVERSE, KIND_DETOUR = range(7)
KIND_MASK, FLAG_HEADLAND, FLAG_ALAT, FLAG_OUTSIDE, FLAG_OBSTACLE, INDEX_SHIFT = 7, 8, 16, 32, 64, 8


def build(force=False):
    src = os.path.join(_HERE, 'fcpp_oracle.c')
    hdr = os.path.join(_HERE, 'fcpp_oracle.h')
    stale = (not os.path.exists(_SO)) or (os.path.exists(src) and
                                          os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    if force or stale:
        subprocess.check_call(['make', '-C', _HERE, '-s'])
    return _SO


class Vehicle(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        'working_width', 'min_turn_radius', 'max_work_speed_kmh', 'max_headland_speed_kmh',
        'headland_turn_speed_kmh', 'max_lateral_accel', 'max_longitudinal_accel', 'safety_factor')]

    @classmethod
    def make(cls, arr=None, **kw):
        d = dict(working_width=3.2, min_turn_radius=8.0, max_work_speed_kmh=9.0,
                 max_headland_speed_kmh=15.0, headland_turn_speed_kmh=4.0, max_lateral_accel=2.0,
                 max_longitudinal_accel=1.5, safety_factor=0.85)
        if arr is not None:
            d = {n: float(x) for (n, _), x in zip(cls._fields_, arr)}
        d.update(kw)
        return cls(**d)


class GaConfig(C.Structure):
    _fields_ = [('population_size', C.c_int32), ('max_generations', C.c_int32), ('crossover_rate', C.c_double),
                ('mutation_rate', C.c_double), ('elite_size', C.c_int32), ('tournament_size', C.c_int32),
                ('convergence_threshold', C.c_int32), ('_pad', C.c_int32), ('seed', C.c_uint64)]


class GaResult(C.Structure):
    _fields_ = [('generations', C.c_int32), ('convergence_gen', C.c_int32), ('best_distance', C.c_double), ('best_fitness', C.c_double)]


class Options(C.Structure):
    _fields_ = [('turn_model', C.c_int32), ('clothoid_fit', C.c_int32), ('sample_spacing', C.c_double),
                ('clothoid_frac', C.c_double), ('geofence_tol', C.c_double), ('obstacle_mode', C.c_int32), ('ring_order', C.c_int32)]

    @classmethod
    def make(cls, turn_model=0, clothoid_fit=1, sample_spacing=0.0, clothoid_frac=0.5, geofence_tol=1e-6, obstacle_mode=0, ring_order=0):
        return cls(turn_model, clothoid_fit, sample_spacing, clothoid_frac, geofence_tol, obstacle_mode, ring_order)


class Field(C.Structure):
    _fields_ = [('vx', C.c_double * 4), ('vy', C.c_double * 4), ('from_vertices', C.c_int32),
                ('has_start', C.c_int32), ('has_end', C.c_int32),
                ('start_x', C.c_double), ('start_y', C.c_double), ('end_x', C.c_double), ('end_y', C.c_double),
                ('n_obstacles', C.c_int32), ('obs_offsets', c_i64_p), ('obs_xy', c_double_p)]


class Plan(C.Structure):
    _fields_ = [('n_main', C.c_int64), ('n_head', C.c_int64),
                ('n_swaths', C.c_int32), ('n_loops', C.c_int32), ('start_corner', C.c_int32),
                ('reverse_order', C.c_int32), ('start_from_right', C.c_int32), ('rotated', C.c_int32),
                ('start_kept', C.c_int32), ('end_kept', C.c_int32), ('shape', C.c_int32),
                ('n_reverse', C.c_int32 * 4),
                ('corner_angles', C.c_double * 4), ('field_length', C.c_double), ('field_width', C.c_double),
                ('headland_width', C.c_double), ('rotation_angle', C.c_double),
                ('xy', c_double_p), ('v', c_double_p), ('kappa', c_double_p), ('flagseg', c_u32_p),
                ('main_len_m', C.c_double), ('main_time_pre_s', C.c_double), ('main_time_s', C.c_double),
                ('head_len_m', C.c_double), ('head_time_pre_s', C.c_double), ('head_time_s', C.c_double),
                ('max_kappa', C.c_double), ('max_alat', C.c_double), ('viol_rate', C.c_double),
                ('max_jump', C.c_double),
                ('n_viol', C.c_int64), ('n_outside', C.c_int64), ('n_in_obstacle', C.c_int64),
                ('n_adjusted', C.c_int64), ('pass_', C.c_int32),
                ('has_approach', C.c_int32), ('has_departure', C.c_int32),
                ('approach', C.c_double * 100), ('departure', C.c_double * 100)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_curvature.restype = C.c_double
        L.orc_curvature.argtypes = [c_double_p] * 3
        L.orc_smooth_speed_profile.argtypes = [c_double_p, c_double_p, C.c_int64, C.c_double]
        L.orc_speed_limit.restype = C.c_int64
        L.orc_speed_limit.argtypes = [c_double_p, c_double_p, c_double_p, C.c_int64, C.POINTER(Vehicle)]
        L.orc_verify.argtypes = [c_double_p, c_double_p, C.c_int64, C.POINTER(Vehicle), c_double_p]
        L.orc_path_length.restype = C.c_double
        L.orc_path_length.argtypes = [c_double_p, C.c_int64]
        L.orc_work_time.restype = C.c_double
        L.orc_work_time.argtypes = [c_double_p, c_double_p, C.c_int64]
        L.orc_linspace.argtypes = [C.c_double, C.c_double, C.c_int64, c_double_p]
        L.orc_safe_arc_turn.argtypes = [C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, c_double_p]
        L.orc_corner_arc.argtypes = [C.c_double, C.c_double, C.c_int, C.c_double, C.c_int, c_double_p]
        L.orc_straight.argtypes = [C.c_double] * 4 + [C.c_int64, c_double_p]
        L.orc_rotate_point.argtypes = [C.c_double] * 5 + [c_double_p]
        L.orc_distance_to_boundary.restype = C.c_double
        L.orc_distance_to_boundary.argtypes = [C.c_double] * 7
        L.orc_reverse_path.restype = C.c_int64
        L.orc_reverse_path.argtypes = [c_double_p, c_double_p, C.c_double, C.c_double, C.c_double, C.c_double,
                                       c_double_p, c_double_p]
        L.orc_u_pattern.restype = C.c_int64
        L.orc_u_pattern.argtypes = [C.c_double] * 4 + [C.c_int, C.c_int, C.POINTER(Vehicle), c_double_p,
                                                       c_double_p, C.c_int64]
        L.orc_ga_distance.restype = C.c_double
        L.orc_ga_distance.argtypes = [c_i32_p, C.c_int32, c_double_p]
        L.orc_ga_fitness.restype = C.c_double
        L.orc_ga_fitness.argtypes = [c_i32_p, C.c_int32, c_double_p]
        L.orc_plan_field.restype = C.c_int
        L.orc_plan_field.argtypes = [C.POINTER(Field), C.POINTER(Vehicle), C.POINTER(Options), C.POINTER(Plan)]
        L.orc_plan_free.argtypes = [C.POINTER(Plan)]
        L.orc_fresnel.argtypes = [C.c_double, c_double_p, c_double_p]
        L.orc_cac_length.restype = C.c_double
        L.orc_cac_length.argtypes = [C.c_double] * 3
        L.orc_cac_point.argtypes = [C.c_double] * 7 + [c_double_p]
        L.orc_cac_fit_radius.restype = C.c_double
        L.orc_cac_fit_radius.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int]
        L.orc_outside_polygon.restype = C.c_int
        L.orc_outside_polygon.argtypes = [C.c_double, C.c_double, c_double_p, C.c_int64, C.c_double]
        L.orc_point_in_polygon.restype = C.c_int
        L.orc_point_in_polygon.argtypes = [C.c_double, C.c_double, c_double_p, C.c_int64]
        c_u32_p = C.POINTER(C.c_uint32)
        L.orc_philox4x32.restype = None
        L.orc_philox4x32.argtypes = [c_u32_p, c_u32_p, c_u32_p]
        L.orc_ga_selection.restype = None
        L.orc_ga_selection.argtypes = [c_i32_p, c_double_p, C.c_int32, C.c_int32, c_i32_p, C.c_int32, c_i32_p]
        L.orc_ga_ox.restype = None
        L.orc_ga_ox.argtypes = [c_i32_p, c_i32_p, C.c_int32, C.c_int32, C.c_int32, c_i32_p, c_i32_p]
        L.orc_ga_elitism.restype = None
        L.orc_ga_elitism.argtypes = [c_i32_p, c_double_p, C.c_int32, C.c_int32, C.c_int32, c_i32_p]
        L.orc_ga_evolve.restype = None
        L.orc_ga_evolve.argtypes = [C.c_int32, C.POINTER(GaConfig), c_double_p, c_i32_p, c_i32_p, c_double_p, C.POINTER(GaResult)]
        L.orc_distance_matrix.restype = None
        L.orc_distance_matrix.argtypes = [C.c_int32, c_double_p, c_double_p, c_double_p]
        L.orc_best_connection.restype = C.c_double
        L.orc_best_connection.argtypes = [c_double_p, c_double_p, C.c_int64, c_double_p, c_double_p, C.c_int64, C.POINTER(C.c_int64),
                                          C.POINTER(C.c_int64)]
        L.orc_cover_grid.restype = None
        L.orc_cover_grid.argtypes = [C.c_double] * 5 + [C.c_int32, C.c_int32, c_double_p, c_double_p, C.c_int32, c_double_p, c_double_p,
                                     C.c_int32, C.c_int, c_double_p, C.POINTER(C.c_uint8), C.POINTER(C.c_int64)]
        L.orc_outside_convex.restype = C.c_int
        L.orc_outside_convex.argtypes = [C.c_double, C.c_double, c_double_p, c_double_p, C.c_int, C.c_double]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---- thin numpy wrappers -------------------------------------------------------
def curvature(p1, p2, p3):
    p1, p2, p3 = _f64(p1), _f64(p2), _f64(p3)
    return lib().orc_curvature(_dp(p1), _dp(p2), _dp(p3))


def smooth_speed_profile(xy, v, a_lon):
    xy, v = _f64(xy), _f64(v).copy()
    lib().orc_smooth_speed_profile(_dp(xy), _dp(v), len(v), a_lon)
    return v


def speed_limit(xy, v, veh):
    xy, v = _f64(xy), _f64(v)
    out = np.empty_like(v)
    adj = lib().orc_speed_limit(_dp(xy), _dp(v), _dp(out), len(v), C.byref(veh))
    return out, adj


def verify(xy, v, veh):
    xy, v = _f64(xy), _f64(v)
    out = np.zeros(6)
    lib().orc_verify(_dp(xy), _dp(v), len(v), C.byref(veh), _dp(out))
    return out


def path_length(xy):
    xy = _f64(xy)
    return lib().orc_path_length(_dp(xy), len(xy))


def work_time(xy, v):
    xy, v = _f64(xy), _f64(v)
    return lib().orc_work_time(_dp(xy), _dp(v), len(v))


def linspace(a, b, n):
    out = np.empty(n)
    lib().orc_linspace(a, b, n, _dp(out))
    return out


def safe_arc_turn(y, turn_right, min_x, max_x, R):
    out = np.empty((20, 2))
    lib().orc_safe_arc_turn(y, int(turn_right), min_x, max_x, R, _dp(out))
    return out


def corner_arc(cx, cy, ci, R, n=15):
    out = np.empty((n, 2))
    lib().orc_corner_arc(cx, cy, ci, R, n, _dp(out))
    return out


def straight(x0, y0, x1, y1, n):
    out = np.empty((n, 2))
    lib().orc_straight(x0, y0, x1, y1, n, _dp(out))
    return out


def rotate_point(x, y, ang, cx, cy):
    out = np.empty(2)
    lib().orc_rotate_point(x, y, ang, cx, cy, _dp(out))
    return out


def reverse_path(end, second_last, L, H, R, spacing=0.0):
    end, sl = _f64(end), _f64(second_last)
    ln = C.c_double()
    n = lib().orc_reverse_path(_dp(end), _dp(sl), L, H, R, spacing, C.byref(ln), None)
    out = np.empty((n, 2))
    lib().orc_reverse_path(_dp(end), _dp(sl), L, H, R, spacing, C.byref(ln), _dp(out))
    return out, ln.value


def u_pattern(bounds, reverse_order, start_from_right, veh):
    cap = (int((bounds[3] - bounds[1]) / veh.working_width) + 2) * 22
    xy, v = np.empty((cap, 2)), np.empty(cap)
    n = lib().orc_u_pattern(bounds[0], bounds[1], bounds[2], bounds[3], int(reverse_order),
                            int(start_from_right), C.byref(veh), _dp(xy), _dp(v), cap)
    assert n >= 0
    return xy[:n].copy(), v[:n].copy()


def ga_distance(routes, D):
    routes = np.ascontiguousarray(routes, dtype=np.int32)
    D = _f64(D)
    n = D.shape[0]
    return np.array([lib().orc_ga_distance(r.ctypes.data_as(c_i32_p), n, _dp(D)) for r in routes])


def ga_fitness(routes, D):
    routes = np.ascontiguousarray(routes, dtype=np.int32)
    D = _f64(D)
    n = D.shape[0]
    return np.array([lib().orc_ga_fitness(r.ctypes.data_as(c_i32_p), n, _dp(D)) for r in routes])


def fresnel(t):
    t = np.atleast_1d(_f64(t))
    c, s = np.empty_like(t), np.empty_like(t)
    cc, ss = C.c_double(), C.c_double()
    for i, x in enumerate(t):
        lib().orc_fresnel(float(x), C.byref(cc), C.byref(ss))
        c[i], s[i] = cc.value, ss.value
    return c, s


def cac_points(x0, y0, th0, dth, R, f, fit, n):
    L = lib()
    Re = L.orc_cac_fit_radius(dth, R, f, fit)
    T = L.orc_cac_length(dth, Re, f)
    s = linspace(0.0, T, n)
    out = np.empty((n, 2))
    tmp = np.empty(2)
    for i in range(n):
        L.orc_cac_point(x0, y0, th0, dth, Re, f, s[i], _dp(tmp))
        out[i] = tmp
    return out, Re, T


def outside_polygon(px, py, poly, tol):
    """fcpp_validate's geofence rule for an arbitrary simple polygon (build-defined): signed distance to the boundary < -tol"""
    poly = _f64(poly)
    return bool(lib().orc_outside_polygon(px, py, _dp(poly), len(poly), float(tol)))


def difference_centroid(main_quad, field, r):
    """MLP:599-609, 288: centroid of the main boundary minus the obstacles' W/2 buffers (oracle restatement of GEOS' polygonal buffer);
    -> (covered, cx, cy)"""
    q = _f64(main_quad)
    mx, my = _f64(q[:, 0]), _f64(q[:, 1])
    cx, cy = C.c_double(0.0), C.c_double(0.0)
    fn = lib().orc_difference_centroid
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    ok = fn(_dp(mx), _dp(my), C.byref(field), float(r), C.byref(cx), C.byref(cy))
    return bool(ok), cx.value, cy.value


def point_in_polygon(px, py, poly):
    poly = _f64(poly)
    return bool(lib().orc_point_in_polygon(px, py, _dp(poly), len(poly)))


def _ip(a):
    return a.ctypes.data_as(c_i32_p)


def philox4x32(ctr, key):
    c = (C.c_uint32 * 4)(*[int(v) & 0xffffffff for v in ctr])
    k = (C.c_uint32 * 2)(*[int(v) & 0xffffffff for v in key])
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32(c, k, o)
    return [int(v) for v in o]


def ga_selection(population, fitness, cand):
    population = np.ascontiguousarray(population, dtype=np.int32)
    cand = np.ascontiguousarray(cand, dtype=np.int32)
    fitness = _f64(fitness)
    out = np.empty_like(population)
    lib().orc_ga_selection(_ip(population), _dp(fitness), population.shape[0], population.shape[1], _ip(cand), cand.shape[1], _ip(out))
    return out


def ga_ox(p1, p2, a, b):
    p1, p2 = np.ascontiguousarray(p1, dtype=np.int32), np.ascontiguousarray(p2, dtype=np.int32)
    c1, c2 = np.empty_like(p1), np.empty_like(p2)
    lib().orc_ga_ox(_ip(p1), _ip(p2), len(p1), int(a), int(b), _ip(c1), _ip(c2))
    return c1, c2


def ga_elitism(old_population, old_fitness, new_population, e):
    old_population = np.ascontiguousarray(old_population, dtype=np.int32)
    out = np.ascontiguousarray(new_population, dtype=np.int32).copy()
    lib().orc_ga_elitism(_ip(old_population), _dp(_f64(old_fitness)), old_population.shape[0], old_population.shape[1], int(e), _ip(out))
    return out


def ga_evolve(D, routes, population_size=None, max_generations=500, crossover_rate=0.85, mutation_rate=0.02, elite_size=20,
              tournament_size=5, convergence_threshold=50, seed=0):
    """-> (final population, best_route, best_fitness_history, avg_fitness_history, GaResult)"""
    D = _f64(D)
    routes = np.ascontiguousarray(routes, dtype=np.int32).copy()
    pop, n = routes.shape
    cfg = GaConfig(pop, max_generations, crossover_rate, mutation_rate, elite_size, tournament_size, convergence_threshold, 0, seed)
    best = np.empty(n, dtype=np.int32)
    hist = np.zeros(2 * max_generations, dtype=np.float64)
    res = GaResult()
    lib().orc_ga_evolve(n, C.byref(cfg), _dp(D), _ip(routes), _ip(best), _dp(hist), C.byref(res))
    g = res.generations
    return routes, best, hist[:g].copy(), hist[max_generations:max_generations + g].copy(), res


def distance_matrix(xy):
    xy = np.asarray(xy, dtype=np.float64).reshape(-1, 2)
    x, y = _f64(xy[:, 0].copy()), _f64(xy[:, 1].copy())
    D = np.empty((len(x), len(x)), dtype=np.float64)
    lib().orc_distance_matrix(len(x), _dp(x), _dp(y), _dp(D))
    return D


def best_connection(from_xy, to_xy):
    """-> (index into from_xy, index into to_xy, distance)"""
    f, t = np.asarray(from_xy, dtype=np.float64).reshape(-1, 2), np.asarray(to_xy, dtype=np.float64).reshape(-1, 2)
    fx, fy, tx, ty = (_f64(a.copy()) for a in (f[:, 0], f[:, 1], t[:, 0], t[:, 1]))
    bf, bt = C.c_int64(), C.c_int64()
    d = lib().orc_best_connection(_dp(fx), _dp(fy), len(fx), _dp(tx), _dp(ty), len(tx), C.byref(bf), C.byref(bt))
    return bf.value, bt.value, d


def cover_grid(ox, oy, res, shift, radius, nx, ny, a_xy, b_xy=None, strict=True, region=None, want_grid=True):
    """-> (counts[3], grid (ny, nx) uint8 or None); region = 24 doubles (4 outer + 4 inner half-planes) or None."""
    a = _f64(np.asarray(a_xy, dtype=np.float64).reshape(-1, 2))
    b = _f64(np.asarray(b_xy if b_xy is not None else np.zeros((0, 2)), dtype=np.float64).reshape(-1, 2))
    ax, ay = _f64(a[:, 0].copy()), _f64(a[:, 1].copy())
    bx, by = _f64(b[:, 0].copy()), _f64(b[:, 1].copy())
    reg = _f64(np.asarray(region, dtype=np.float64)) if region is not None else None
    grid = np.zeros((ny, nx), dtype=np.uint8) if want_grid else None
    counts = np.zeros(3, dtype=np.int64)
    lib().orc_cover_grid(ox, oy, res, shift, radius, nx, ny, _dp(ax), _dp(ay), len(ax), _dp(bx), _dp(by), len(bx), int(bool(strict)),
                         _dp(reg) if reg is not None else None, grid.ctypes.data_as(C.POINTER(C.c_uint8)) if want_grid else None,
                         counts.ctypes.data_as(C.POINTER(C.c_int64)))
    return counts, grid


class PlanResult:
    """numpy view of one orc_plan (arrays copied, C memory released)."""

    def __init__(self, p):
        n = p.n_main + p.n_head
        self.n_main, self.n_head, self.n = p.n_main, p.n_head, n
        for k in ('n_swaths', 'n_loops', 'start_corner', 'reverse_order', 'start_from_right', 'rotated',
                  'start_kept', 'end_kept', 'shape', 'field_length', 'field_width', 'headland_width',
                  'rotation_angle', 'main_len_m', 'main_time_pre_s', 'main_time_s', 'head_len_m',
                  'head_time_pre_s', 'head_time_s', 'max_kappa', 'max_alat', 'viol_rate', 'max_jump',
                  'n_viol', 'n_outside', 'n_in_obstacle', 'n_adjusted'):
            setattr(self, k, getattr(p, k))
        self.passed = bool(p.pass_)
        self.n_reverse = list(p.n_reverse)
        self.corner_angles = np.array(list(p.corner_angles))
        self.xy = np.ctypeslib.as_array(p.xy, shape=(n, 2)).copy()
        self.v = np.ctypeslib.as_array(p.v, shape=(n,)).copy()
        self.kappa = np.ctypeslib.as_array(p.kappa, shape=(n,)).copy()
        self.flagseg = np.ctypeslib.as_array(p.flagseg, shape=(n,)).copy()
        self.approach = np.array(list(p.approach)).reshape(50, 2) if p.has_approach else None
        self.departure = np.array(list(p.departure)).reshape(50, 2) if p.has_departure else None


def make_field(verts=None, L=None, H=None, start=None, end=None, obstacles=None):
    f = Field()
    if verts is None:
        verts = [(0.0, 0.0), (L, 0.0), (L, H), (0.0, H)]
        f.from_vertices = 0
    else:
        f.from_vertices = 1
    for i, (x, y) in enumerate(verts):
        f.vx[i], f.vy[i] = float(x), float(y)
    if start is not None:
        f.has_start, f.start_x, f.start_y = 1, float(start[0]), float(start[1])
    if end is not None:
        f.has_end, f.end_x, f.end_y = 1, float(end[0]), float(end[1])
    keep = []
    if obstacles is not None and len(obstacles):
        offs = np.cumsum([0] + [len(o) for o in obstacles]).astype(np.int64)
        xy = _f64(np.vstack([np.asarray(o, dtype=np.float64) for o in obstacles]))
        f.n_obstacles = len(obstacles)
        f.obs_offsets = offs.ctypes.data_as(c_i64_p)
        f.obs_xy = _dp(xy)
        keep = [offs, xy]
    f._keep = keep
    return f


def plan_field(field, veh=None, opt=None):
    """Returns (rc, PlanResult|None); rc<0 mirrors the reference's ValueError cases."""
    veh = veh or Vehicle.make()
    opt = opt or Options.make()
    p = Plan()
    rc = lib().orc_plan_field(C.byref(field), C.byref(veh), C.byref(opt), C.byref(p))
    if rc != 0:
        return rc, None
    res = PlanResult(p)
    lib().orc_plan_free(C.byref(p))
    return 0, res
