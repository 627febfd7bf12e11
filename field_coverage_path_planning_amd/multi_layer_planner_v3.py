"""Drop-in mirror of the reference's `multi_layer_planner_v3` surface over libfcpp.so (MI355X / HIP).

Same names, arguments, result keys and error behaviour as the reference module
(/root/reference/multi_layer_planner_v3.py = "MLP"):

    VehicleParams                                       MLP:29-39
    TwoLayerPathPlannerV37(vehicle_params, field_length=, field_width=, field_vertices=,
                           obstacles=, start_point=, end_point=)            MLP:63-72
        .plan_complete_coverage() -> dict                                   MLP:387-465
        .verify_curvature_constraints(path, speeds) -> dict                 MLP:1373-1424
        ._calculate_path_length / ._calculate_work_time / ._calculate_curvature
        ._apply_curvature_based_speed_limit / ._smooth_speed_profile

Every number is computed by the HIP library through its C ABI (include/fcpp.h); this file only
converts arguments and assembles the result dictionary.  The names the reference's own tests and
README import but the reference never defines (TwoLayerPathPlannerV35/V36, TwoLayerPlannerV35/V36,
`vehicle=`, `.plan()`) are provided as aliases so that those scripts resolve.

        .verify_all_corners_coverage(result['headland']) / .verify_corner_coverage_grid_based(...)   MLP:1426-1578
        ._calculate_coverage_rate(path, area)                               MLP:1357-1371

Not reproduced: matplotlib plotting helpers and the Shapely objects under result['...']['area'] (a plain
vertex-list polygon is returned instead).  `coverage_rate` is sampled on a 0.1 m grid (`coverage_resolution=`)
instead of GEOS polygon clipping.
"""
import math
import time
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

from . import _lib as L
from . import engine as E

__all__ = ['VehicleParams', 'TwoLayerPathPlannerV37', 'TwoLayerPathPlannerV35', 'TwoLayerPathPlannerV36',
           'TwoLayerPlannerV35', 'TwoLayerPlannerV36', 'TwoLayerPlannerV37', 'QuadPolygon']


@dataclass
class VehicleParams:
    """车辆参数 (MLP:29-39)"""
    working_width: float = 3.2
    min_turn_radius: float = 8.0
    max_work_speed_kmh: float = 9.0
    max_headland_speed_kmh: float = 15.0
    headland_turn_speed_kmh: float = 4.0
    max_lateral_accel: float = 2.0
    max_longitudinal_accel: float = 1.5
    safety_factor: float = 0.85


class QuadPolygon:
    """Minimal stand-in for the shapely Polygon attributes callers of the reference read
    (`.area`, `.bounds`, `.centroid.x/.y`, `.exterior.coords`), e.g. multi_field_planner.py:117-118."""

    class _Pt:
        def __init__(self, x, y):
            self.x, self.y = x, y
            self.coords = [(x, y)]

    class _Ring:
        def __init__(self, vs):
            self.coords = list(vs) + [vs[0]]

    def __init__(self, vertices, area=None, hole=None):
        self.vertices = [(float(x), float(y)) for x, y in vertices]
        self._area = area
        self.hole = [(float(x), float(y)) for x, y in hole] if hole else None   # headland ring = field minus main-work polygon

    @property
    def area(self):
        if self._area is not None:
            return self._area
        v = self.vertices
        return abs(sum(v[i][0] * v[(i + 1) % len(v)][1] - v[(i + 1) % len(v)][0] * v[i][1]
                       for i in range(len(v)))) / 2

    @property
    def bounds(self):
        xs, ys = [p[0] for p in self.vertices], [p[1] for p in self.vertices]
        return (min(xs), min(ys), max(xs), max(ys))

    @property
    def centroid(self):
        v = self.vertices
        a = cx = cy = 0.0
        for i in range(len(v)):
            x0, y0 = v[i]
            x1, y1 = v[(i + 1) % len(v)]
            cr = x0 * y1 - x1 * y0
            a += cr
            cx += (x0 + x1) * cr
            cy += (y0 + y1) * cr
        a *= 0.5
        return QuadPolygon._Pt(cx / (6 * a), cy / (6 * a))

    @property
    def exterior(self):
        return QuadPolygon._Ring(self.vertices)

    @property
    def is_empty(self):
        return len(self.vertices) == 0


_SHAPES = {0: 'rectangle', 1: 'parallelogram', 2: 'other'}

_RING_ORDER_WARNED = False
RING_ORDER_CHECK = '''from shapely.geometry import Polygon
v = [(0, 0), (500, 0), (500, 200), (0, 200)]
ring = list(Polygon(v).buffer(-1.6).exterior.coords)[:-1]
# ring[0] ~ (1.6, 1.6) and ring[1] ~ (498.4, 1.6): ring_order=0   (as the vertices: LL, LR, UR, UL)
# ring[0] ~ (1.6, 1.6) and ring[1] ~ (1.6, 198.4): ring_order=1   (the other way round: LL, UL, UR, LR)'''


def _warn_ring_order_once():
    """The order in which Shapely lists the inset corners of a headland loop (MLP:964-972) is a GEOS fact the reference neither documents nor
    tests, and its corner formulas index that list (MLP:1049-1060).  Both orders are pinned to the reference's own code (tests/golden:
    cw_*); which one a given Shapely / GEOS emits cannot be observed here (no Shapely in this build's environment), so for fields given by
    vertices the caller should say -- once per process this reminds them how to find out."""
    global _RING_ORDER_WARNED
    if _RING_ORDER_WARNED:
        return
    _RING_ORDER_WARNED = True
    import warnings
    warnings.warn("TwoLayerPathPlannerV37(field_vertices=...) without ring_order=: the headland loops are built in the order of the field's "
                  "vertices (ring_order=0, the intent the reference documents at MLP:957); the reference itself takes the order from "
                  "Shapely's buffer(-d).exterior.coords, which this build cannot observe.  With Shapely at hand, check once:\n" + RING_ORDER_CHECK +
                  "\nand pass ring_order=0 or 1 explicitly (INTEGRATION.md, 'Ring order of the inset corners').", stacklevel=3)


class TwoLayerPathPlannerV37:
    """两层路径规划器 V3.7 (MLP:42-61) -- HIP-backed."""

    def __init__(self, vehicle_params: VehicleParams = None, field_length: float = None,
                 field_width: float = None, field_vertices: List[Tuple[float, float]] = None,
                 obstacles: List[List[Tuple[float, float]]] = None, start_point: Tuple[float, float] = None,
                 end_point: Tuple[float, float] = None, *, vehicle: VehicleParams = None, verbose: bool = False,
                 turn_model: str = 'arc', sample_spacing: float = 0.0, clothoid_frac: float = 0.5,
                 clothoid_fit: int = 1, geofence_tol: float = 1e-6, coverage_resolution: float = 0.1, device: int = None,
                 avoid_obstacles: bool = False, ring_order: int = None):
        if vehicle_params is None:
            vehicle_params = vehicle if vehicle is not None else VehicleParams()   # README_en.md:274-302 uses vehicle=
        self.vehicle = vehicle_params
        self.obstacles = obstacles or []
        self.verbose = verbose
        self._device = device
        self.coverage_resolution = float(coverage_resolution)   # sample spacing of coverage_rate [m] (build-defined)
        self._spec = E.FieldSpec(field_length, field_width, field_vertices, self.obstacles, start_point, end_point)
        self._batch = self._bufs = None
        self._veh = E.make_vehicle(self.vehicle)
        self._ring_order_given = ring_order is not None
        ring_order = 0 if ring_order is None else int(ring_order)
        self._opt = E.make_options(L.TURN_CLOTHOID if str(turn_model).lower().startswith('cloth') else L.TURN_ARC,
                                   sample_spacing, clothoid_frac, clothoid_fit, geofence_tol, avoid_obstacles, ring_order)
        # _process_field_input (MLP:109-135): ValueError when no field is given
        if field_vertices is not None:
            self.field_vertices = field_vertices
        elif field_length is not None and field_width is not None:
            self.field_vertices = [(0, 0), (field_length, 0), (field_length, field_width), (0, field_width)]
        else:
            raise ValueError("必须提供 field_vertices 或 (field_length, field_width)")
        self.field_polygon = QuadPolygon(self.field_vertices)
        self._table = E.FieldTable.from_specs([self._spec])
        info = E.plan_count(self._table, self._veh, self._opt)[0]       # host-side setup in libfcpp
        self._info = info
        self.field_length = info.field_length if field_vertices is not None else field_length
        self.field_width = info.field_width if field_vertices is not None else field_width
        self.field_shape = _SHAPES[info.shape]                          # MLP:137-163
        self.corner_angles = [info.corner_angles[i] for i in range(4)]  # MLP:165-192
        self.headland_width = info.headland_width                       # MLP:310
        aspect = self.field_length / self.field_width                   # MLP:312-320
        self.main_work_pattern = "U型往复" if (aspect > 3.0 or aspect >= 1.5) else "Ω型跨行"
        self.start_point = tuple(map(float, start_point)) if (start_point is not None and info.start_kept) else None
        self.end_point = tuple(map(float, end_point)) if (end_point is not None and info.end_kept) else None
        if field_vertices is not None and info.shape != 0 and not self._ring_order_given:
            _warn_ring_order_once()
        if verbose:
            print(f"[V3.7.0/fcpp] 初始化完成: 形状={self.field_shape}, 田头宽度={self.headland_width:.1f}m, "
                  f"障碍物={len(self.obstacles)}")

    # ------------------------------------------------------------------------------------------
    def plan_complete_coverage(self) -> Dict:
        """完整的两层路径规划 (MLP:387-465)."""
        t0 = time.time()
        info = self._info
        if info.status == L.EINVAL:
            raise ValueError(f"田头宽度{self.headland_width}m过大，无法定义主作业区域")   # MLP:598
        if info.status == L.EHEADLAND:
            raise ValueError("all the input array dimensions except for the concatenation axis must match "
                             "exactly (headland loop inset is empty, MLP:967-969 -> :939)")
        if info.status != L.OK:
            raise ValueError(f"unsupported field (libfcpp status {info.status})")
        # the planner keeps its batch (descriptors on the device, tiling, output arrays): a second plan_complete_coverage() of the same
        # planner only runs the kernels again.  Results are copied out, so the caller owns fresh arrays every time (as in the reference).
        if self._batch is None:
            self._batch = E.Batch(self._table, self._veh, self._opt, device=self._device)
        if self._bufs is None:
            self._bufs = self._batch.alloc()
        batch = self._batch
        res = batch.run(self._bufs)
        ap, dp = batch.connectors()
        n_main, n_head = info.n_main, info.n_head
        # coverage of the headland ring by the headland path, straight from the device arrays (MLP:884, 1357-1371)
        coverage_rate = self._coverage_rate_dev(res.x[n_main:], res.y[n_main:], self._headland_area())
        # (one copy for the four float64 arrays, one for the two connectors: every copy to the host is a synchronisation of its own)
        import torch
        x, y, kappa, v = torch.stack((res.x, res.y, res.kappa, res.v)).cpu().numpy()
        fs = res.flagseg.cpu().numpy().view(np.uint32)
        st = {k: a[0] for k, a in res.stats().items()}
        ap, dp = torch.cat((ap, dp)).cpu().numpy()
        path = np.column_stack([x, y])
        main_len, head_len = st['main_len_m'], st['head_len_m']
        main_pre, head_pre = st['main_time_pre_s'], st['head_time_pre_s']
        W, R = self.vehicle.working_width, self.vehicle.min_turn_radius
        main_area = QuadPolygon(_inset_for_area(self.field_vertices, R))
        head_area = self._headland_area()
        main_work = {
            'path': path[:n_main], 'speeds': v[:n_main], 'pattern': self.main_work_pattern, 'area': main_area,
            'stats': {
                'path_length_km': main_len / 1000,
                'time_hours': st['main_time_s'] / 3600,                                      # MLP:423-426
                'avg_speed_kmh': (main_len / 1000) / (main_pre / 3600) if main_pre > 0 else 0,  # MLP:627 (pre-clamp time)
            },
            'kappa': kappa[:n_main], 'flagseg': fs[:n_main],
        }
        headland = {
            'path': path[n_main:], 'speeds': v[n_main:], 'area': head_area,
            'stats': {
                'path_length_km': head_len / 1000,
                'time_hours': st['head_time_s'] / 3600,                                      # MLP:428-431
                'avg_speed_kmh': (head_len / 1000) / (head_pre / 3600) if head_pre > 0 else 0,
                'coverage_rate': coverage_rate,   # MLP:884 (0..1), sampled at self.coverage_resolution
            },
            'kappa': kappa[n_main:], 'flagseg': fs[n_main:],
        }
        result = {
            'main_work': main_work, 'headland': headland,
            'approach_path': ap if self.start_point else None,      # MLP:437-441
            'departure_path': dp if self.end_point else None,       # MLP:443-447
            'total_time': time.time() - t0,
            'version': 'V3.5.1',
            'features': ['真正两层', '切线倒车', '网格验证', '强制降速', '智能起点'],
            # extras (not in the reference)
            'validation': {k: st[k] for k in ('max_kappa', 'max_alat', 'max_jump', 'n_viol', 'n_outside',
                                              'n_in_obstacle', 'n_adjusted')},
            'num_passes': info.n_swaths, 'num_loops': info.n_loops, 'start_corner_index': info.start_corner,
        }
        if self.verbose:
            print(f"路径规划完成! 总耗时: {result['total_time']:.3f}秒  "
                  f"(main {n_main} pts, headland {n_head} pts)")
        # the results are host arrays now: a planner keeps device output arrays only while they are small (a second plan of the same
        # planner then only runs the kernels); the arrays of a densely sampled field -- 36 bytes per point -- go back at once
        if 36 * batch.total_points > (64 << 20):
            del res
            self._bufs = None
        return result

    plan = plan_complete_coverage   # README_en.md:274-302

    def close(self):
        """release the planner's device batch (also done when the planner is collected)"""
        if getattr(self, '_batch', None) is not None:
            self._batch.close()
            self._batch = self._bufs = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------
    def verify_curvature_constraints(self, path: np.ndarray, speeds: np.ndarray) -> Dict:
        """验证曲率约束 (MLP:1373-1424)."""
        path = np.asarray(path, dtype=np.float64)
        if len(path) < 3:
            return {'max_curvature': 0, 'violations': 0, 'pass': True}     # MLP:1377-1378
        st = E.verify(path[:, 0], path[:, 1], speeds, self._veh, device=self._device)
        m = len(path) - 2
        viol = int(st['n_viol'][0])
        rate = viol / m * 100 if m > 0 else 0
        return {
            'max_curvature': float(st['max_kappa'][0]),
            'max_lateral_accel': float(st['max_alat'][0]),
            'max_allowed_accel': self.vehicle.max_lateral_accel,
            'accel_violations': viol,
            'accel_violation_rate': rate,
            'max_jump': float(st['max_jump'][0]) if m > 1 else 0,
            'pass': rate < 5,
        }

    def _calculate_path_length(self, path: np.ndarray) -> float:
        """MLP:1290-1296 (called by test/test_v351_start_end_points.py:133)."""
        path = np.asarray(path, dtype=np.float64)
        if len(path) < 2:
            return 0.0
        st = E.verify(path[:, 0], path[:, 1], np.ones(len(path)), self._veh, device=self._device)
        return float(st['main_len_m'][0])

    def _calculate_work_time(self, path: np.ndarray, speeds: np.ndarray) -> float:
        """MLP:1298-1311."""
        path = np.asarray(path, dtype=np.float64)
        if len(path) < 2 or len(speeds) == 0:
            return 0.0
        st = E.verify(path[:, 0], path[:, 1], speeds, self._veh, device=self._device)
        return float(st['main_time_s'][0])

    def _calculate_curvature(self, p1, p2, p3) -> float:
        """MLP:513-536."""
        pts = np.array([p1, p2, p3], dtype=np.float64)
        return float(E.curvature(pts[:, 0], pts[:, 1], device=self._device).cpu().numpy()[1])

    def _apply_curvature_based_speed_limit(self, path: np.ndarray, speeds: np.ndarray) -> np.ndarray:
        """MLP:467-511."""
        path = np.asarray(path, dtype=np.float64)
        if len(path) < 3:
            return speeds
        out, _ = E.speed_plan(path[:, 0], path[:, 1], speeds, self._veh, clamp=True, device=self._device)
        return out.cpu().numpy()

    def _smooth_speed_profile(self, path: np.ndarray, speeds: np.ndarray) -> np.ndarray:
        """MLP:538-589."""
        path = np.asarray(path, dtype=np.float64)
        if len(path) < 2:
            return speeds
        out, _ = E.speed_plan(path[:, 0], path[:, 1], speeds, self._veh, clamp=False, device=self._device)
        return out.cpu().numpy()

    def _generate_approach_path(self, start, end, num_points: int = 50) -> np.ndarray:
        """MLP:1313-1333."""
        seg = np.array([[start[0], start[1], end[0], end[1]]], dtype=np.float64)
        return E.straight_segments(seg, num_points, device=self._device).cpu().numpy()[0]

    _generate_departure_path = _generate_approach_path            # MLP:1335-1355

    def _generate_straight_segment(self, start, end, num_points: int = 20) -> np.ndarray:
        """MLP:1013-1022."""
        return self._generate_approach_path(start, end, num_points)

    # ---- coverage (MLP:1357-1371, 1426-1578): sampled on the GPU by fcpp_cover_grid ----------------------------------
    def _headland_area(self) -> QuadPolygon:
        """field minus the polygon inset by headland_width (MLP:867-877); the whole field if the inset is empty."""
        inner = _inset_for_area(self.field_vertices, self.headland_width)
        main = QuadPolygon(inner)
        if not _inset_is_valid(self.field_vertices, inner) or main.area < 1.0:
            return QuadPolygon(self.field_vertices)
        return QuadPolygon(self.field_vertices, area=self.field_polygon.area - main.area, hole=inner)

    def _coverage_rate_dev(self, px, py, area) -> float:
        n = int(px.shape[0])
        if n < 2:
            return 0.0                                                          # MLP:1359-1360
        x0, y0, x1, y1 = area.bounds
        res = self.coverage_resolution
        nx, ny = max(1, int(math.ceil((x1 - x0) / res))), max(1, int(math.ceil((y1 - y0) / res)))
        job = E.make_cover_job(x0, y0, res, nx, ny, self.vehicle.working_width / 2, n, shift=0.5, strict=False,
                               outer=E.half_planes(area.vertices), inner=E.half_planes(area.hole) if area.hole else None)
        counts, _ = E.cover_grid([job], px, py, device=self._device)
        tot, cov = (int(c) for c in counts.cpu().numpy()[0, :2])
        return cov / tot if tot > 0 else 0.0                                    # MLP:1368-1371 (0..1, not percent)

    def _calculate_coverage_rate(self, path: np.ndarray, area) -> float:
        """计算覆盖率 (MLP:1357-1371): share of `area` within working_width/2 of the path, sampled on a regular grid of
        `coverage_resolution` metres (cell centres) instead of Shapely's buffer / intersection."""
        path = np.asarray(path, dtype=np.float64)
        if len(path) < 2:
            return 0.0
        if not isinstance(area, QuadPolygon):
            area = QuadPolygon(list(area.exterior.coords)[:4])
        return self._coverage_rate_dev(np.ascontiguousarray(path[:, 0]), np.ascontiguousarray(path[:, 1]), area)

    def _corner_turns(self, corners, with_reverse):
        """[(turn (15, 2), reverse (n, 2) | None)] for (corner, corner_index) pairs: fcpp_corner_turns, the library's own
        generator of MLP:1580-1608 / 1024-1084 / 1154-1288 (no host restatement)."""
        return E.corner_turns([c for c, _ in corners], [ci for _, ci in corners], with_reverse, self._veh,
                              self.field_length, self.field_width, device=self._device)

    def _generate_corner_turn_arc(self, corner, corner_index: int):
        """15-point quarter arc of radius R leaving `corner` (MLP:1580-1608; quadrant formulas MLP:1049-1060)."""
        turn, _ = self._corner_turns([(corner, corner_index)], [0])[0]
        return turn, [self.vehicle.headland_turn_speed_kmh] * len(turn)

    def verify_corner_coverage_grid_based(self, corner, corner_index: int, turn_path: np.ndarray,
                                          reverse_path: np.ndarray = None) -> Dict:
        """网格化验证转角覆盖率 (MLP:1426-1509): 0.1 m grid over the 2R x 2R corner square; a cell is covered if its
        corner point lies strictly within W/2 of the turn polyline, then (cells still open) of the reverse polyline."""
        return self._corner_grids([(corner, corner_index, turn_path, reverse_path)])[0]

    def _corner_grids(self, items):
        R, W, res = self.vehicle.min_turn_radius, self.vehicle.working_width, 0.1
        gs = int(2 * R / res)                                                   # MLP:1456
        jobs, pts = [], []
        first = 0
        origins = []
        for (cx, cy), ci, turn, rev in items:
            origin = [(cx, cy), (cx - 2 * R, cy), (cx - 2 * R, cy - 2 * R), (cx, cy - 2 * R)][ci if ci in (0, 1, 2) else 3]   # MLP:1460-1467
            turn = np.asarray(turn, dtype=np.float64).reshape(-1, 2)
            rev = np.asarray(rev, dtype=np.float64).reshape(-1, 2) if rev is not None and len(rev) > 0 else np.zeros((0, 2))
            jobs.append(E.make_cover_job(origin[0], origin[1], res, gs, gs, W / 2, len(turn), len(rev), pts_first=first))
            pts += [turn, rev]
            first += len(turn) + len(rev)
            origins.append(origin)
        xy = np.vstack(pts)
        counts, grid = E.cover_grid(jobs, np.ascontiguousarray(xy[:, 0]), np.ascontiguousarray(xy[:, 1]), want_grid=True,
                                    device=self._device)
        grid = grid.cpu().numpy().reshape(len(items), gs, gs)
        counts = counts.cpu().numpy()
        out = []
        for k in range(len(items)):
            before, after = counts[k, 1] / (gs * gs) * 100, counts[k, 2] / (gs * gs) * 100     # MLP:1485, 1499
            out.append({'coverage_before': before, 'coverage_after': after, 'improvement': after - before,
                        'grid': grid[k] != 0, 'grid_origin': origins[k], 'grid_resolution': res})
        return out

    def verify_all_corners_coverage(self, headland_result: Dict = None) -> Dict:
        """验证所有4个角落的覆盖率 (MLP:1511-1578): the four corners of the rectangle inset by headland_width, each with its
        quarter-arc turn and, where the corner gap is large enough, the reverse fill; one GPU call for all four."""
        hw, Lf, Hf = self.headland_width, self.field_length, self.field_width
        R, W = self.vehicle.min_turn_radius, self.vehicle.working_width
        # gap.area > 0.1 (MLP:1557) by the same analytic bound libfcpp's host setup uses (fcpp_host.cpp)
        reverse = 4 * R * R - (math.pi * R * W / 2 + math.pi * W * W / 4) > 0.1
        cs = [((hw, hw), 0), ((Lf - hw, hw), 1), ((Lf - hw, Hf - hw), 2), ((hw, Hf - hw), 3)]
        items = [(c, ci, turn, rev) for (c, ci), (turn, rev) in zip(cs, self._corner_turns(cs, [int(reverse)] * 4))]
        corners = self._corner_grids(items)
        if self.verbose:
            for ci, r in enumerate(corners):
                print(f"  角落{ci}: 填充前={r['coverage_before']:.1f}%, 填充后={r['coverage_after']:.1f}%, "
                      f"改进=+{r['improvement']:.1f}%")
        b = float(np.mean([r['coverage_before'] for r in corners]))
        a = float(np.mean([r['coverage_after'] for r in corners]))
        return {'corners': corners, 'avg_coverage_before': b, 'avg_coverage_after': a, 'avg_improvement': a - b}


def _inset_is_valid(vertices, inset):
    """every inset edge keeps the direction of its source edge (otherwise the inset polygon is empty)"""
    n = len(vertices)
    for i in range(n):
        ex, ey = vertices[(i + 1) % n][0] - vertices[i][0], vertices[(i + 1) % n][1] - vertices[i][1]
        fx, fy = inset[(i + 1) % n][0] - inset[i][0], inset[(i + 1) % n][1] - inset[i][1]
        if fx * ex + fy * ey <= 0:
            return False
    return True


def _inset_for_area(vertices, d):
    """vertices of the main-work polygon, for the informational 'area' entry only."""
    v = [(float(x), float(y)) for x, y in vertices]
    n = len(v)
    a2 = sum(v[i][0] * v[(i + 1) % n][1] - v[(i + 1) % n][0] * v[i][1] for i in range(n))
    sg = 1.0 if a2 > 0 else -1.0
    nrm = []
    for i in range(n):
        ex, ey = v[(i + 1) % n][0] - v[i][0], v[(i + 1) % n][1] - v[i][1]
        ln = math.hypot(ex, ey)
        nrm.append((-ey / ln * sg, ex / ln * sg))
    out = []
    for i in range(n):
        ax, ay = nrm[i - 1]
        bx, by = nrm[i]
        den = 1.0 + ax * bx + ay * by
        out.append((v[i][0] + d * (ax + bx) / den, v[i][1] + d * (ay + by) / den))
    return out


# names the reference's README / tests / sibling modules import: all the same class
TwoLayerPathPlannerV35 = TwoLayerPathPlannerV37
TwoLayerPathPlannerV36 = TwoLayerPathPlannerV37
TwoLayerPlannerV35 = TwoLayerPathPlannerV37
TwoLayerPlannerV36 = TwoLayerPathPlannerV37
TwoLayerPlannerV37 = TwoLayerPathPlannerV37
