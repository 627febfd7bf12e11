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

Not reproduced: matplotlib plotting helpers, the Shapely objects under result['...']['area'] (a plain
vertex-list polygon is returned instead) and `coverage_rate` (needs polygon clipping, SURVEY.md 8f -> NaN).
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

    def __init__(self, vertices, area=None):
        self.vertices = [(float(x), float(y)) for x, y in vertices]
        self._area = area

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


class TwoLayerPathPlannerV37:
    """两层路径规划器 V3.7 (MLP:42-61) -- HIP-backed."""

    def __init__(self, vehicle_params: VehicleParams = None, field_length: float = None,
                 field_width: float = None, field_vertices: List[Tuple[float, float]] = None,
                 obstacles: List[List[Tuple[float, float]]] = None, start_point: Tuple[float, float] = None,
                 end_point: Tuple[float, float] = None, *, vehicle: VehicleParams = None, verbose: bool = False,
                 turn_model: str = 'arc', sample_spacing: float = 0.0, clothoid_frac: float = 0.5,
                 clothoid_fit: int = 1, geofence_tol: float = 1e-6, device: int = None):
        if vehicle_params is None:
            vehicle_params = vehicle if vehicle is not None else VehicleParams()   # README_en.md:274-302 uses vehicle=
        self.vehicle = vehicle_params
        self.obstacles = obstacles or []
        self.verbose = verbose
        self._device = device
        self._spec = E.FieldSpec(field_length, field_width, field_vertices, self.obstacles, start_point, end_point)
        self._veh = E.make_vehicle(self.vehicle)
        self._opt = E.make_options(L.TURN_CLOTHOID if str(turn_model).lower().startswith('cloth') else L.TURN_ARC,
                                   sample_spacing, clothoid_frac, clothoid_fit, geofence_tol)
        # _process_field_input (MLP:109-135): ValueError when no field is given
        if field_vertices is not None:
            self.field_vertices = field_vertices
        elif field_length is not None and field_width is not None:
            self.field_vertices = [(0, 0), (field_length, 0), (field_length, field_width), (0, field_width)]
        else:
            raise ValueError("必须提供 field_vertices 或 (field_length, field_width)")
        self.field_polygon = QuadPolygon(self.field_vertices)
        info = E.plan_count([self._spec], self._veh, self._opt)[0]      # host-side setup in libfcpp
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
        batch = E.Batch([self._spec], self._veh, self._opt, device=self._device)
        try:
            res = batch.run()
            ap, dp = batch.connectors()
            n_main, n_head = info.n_main, info.n_head
            x, y = res.x.cpu().numpy(), res.y.cpu().numpy()
            v, kappa, fs = res.v.cpu().numpy(), res.kappa.cpu().numpy(), res.flagseg.cpu().numpy().view(np.uint32)
            st = {k: a[0] for k, a in res.stats().items()}
            ap, dp = ap.cpu().numpy()[0], dp.cpu().numpy()[0]
        finally:
            batch.close()
        path = np.column_stack([x, y])
        main_len, head_len = st['main_len_m'], st['head_len_m']
        main_pre, head_pre = st['main_time_pre_s'], st['head_time_pre_s']
        W, R = self.vehicle.working_width, self.vehicle.min_turn_radius
        main_area = QuadPolygon(_inset_for_area(self.field_vertices, R))
        head_area = QuadPolygon(self.field_vertices, area=self.field_polygon.area - main_area.area)
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
                'coverage_rate': float('nan'),   # needs polygon clipping (MLP:1357-1371), SURVEY.md 8f
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
        return result

    plan = plan_complete_coverage   # README_en.md:274-302

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
