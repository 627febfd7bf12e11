"""TEST-ONLY stand-in for the `shapely` package, used ONLY by tools/gen_golden.py.

Shapely/GEOS is a third-party dependency of the reference that is absent from
this image (no wheel, no network).  The reference module imports it at the top
(multi_layer_planner_v3.py:20-21), so *any* call into the reference needs the
names `Polygon`, `LineString`, `Point`, `unary_union` to resolve.

This file is NOT product code and is NOT part of the oracle.  It implements,
for convex quadrilaterals only, the handful of geometric facts the reference's
path generator reads from Shapely objects:

  * Polygon.bounds / .area / .is_empty / .centroid / .exterior.coords
  * Polygon.buffer(d<0)  -> sharp (mitre) inset of a convex polygon, vertices in
                            the SAME order as the input ring (the documented
                            intent order 0=LL,1=LR,2=UR,3=UL of MLP:957)
  * Polygon.buffer(d>0), unary_union, difference -> objects that keep the
    minuend's bounds/centroid (the reference's generator reads only `.bounds`,
    MLP:731-732, so obstacle differences never change the path)
  * LineString.buffer(w).area for the corner-gap decision (MLP:1070,1143-1148):
    answered by the analytic bound  area(square) - (L*W + pi*(W/2)^2)  which is
    a LOWER bound of the true gap area; the decision `gap.area > 0.1` is
    therefore exact whenever the bound itself exceeds 0.1.

  * LineString.buffer(w).contains(Point) for the corner grid verification
    (MLP:1471-1497): the exact distance test (GEOS would use a 32-gon per circle).

What would need real polygon clipping (coverage_rate) returns NaN and is
excluded from the golden vectors.
"""
import math
import sys
import types


def _area_centroid(vs):
    a = 0.0
    cx = 0.0
    cy = 0.0
    n = len(vs)
    for i in range(n):
        x0, y0 = vs[i]
        x1, y1 = vs[(i + 1) % n]
        cr = x0 * y1 - x1 * y0
        a += cr
        cx += (x0 + x1) * cr
        cy += (y0 + y1) * cr
    a *= 0.5
    if abs(a) < 1e-300:
        return 0.0, (vs[0][0], vs[0][1])
    return a, (cx / (6.0 * a), cy / (6.0 * a))


class _Coords(list):
    pass


class _Centroid:
    def __init__(self, x, y):
        self.x = x
        self.y = y
        self.coords = [(x, y)]


class _Ring:
    def __init__(self, vs):
        self.coords = _Coords(list(vs) + [vs[0]])


class Polygon:
    def __init__(self, vertices=None, _empty=False, _area_override=None):
        self._empty = _empty or not vertices
        self._vs = [(float(x), float(y)) for x, y in (vertices or [])]
        self._area_override = _area_override

    # --- queries the reference performs -------------------------------------
    @property
    def is_empty(self):
        return self._empty

    @property
    def bounds(self):
        xs = [v[0] for v in self._vs]
        ys = [v[1] for v in self._vs]
        return (min(xs), min(ys), max(xs), max(ys))

    @property
    def area(self):
        if self._area_override is not None:
            return self._area_override
        if self._empty:
            return 0.0
        return abs(_area_centroid(self._vs)[0])

    @property
    def centroid(self):
        c = _area_centroid(self._vs)[1]
        return _Centroid(c[0], c[1])

    @property
    def exterior(self):
        return _Ring(self._vs)

    def buffer(self, d):
        if d >= 0:
            # only reached for obstacles (MLP:605); result only enters a
            # difference() whose bounds we keep.
            return Polygon(self._vs)
        return _inset_convex(self._vs, -d)

    def difference(self, other):
        if isinstance(other, _BufferedLine):
            # corner gap (MLP:1148): analytic lower bound of the area
            return Polygon(self._vs, _area_override=self.area - other.area)
        # obstacle / headland differences: generator reads .bounds only
        return Polygon(self._vs, _area_override=float('nan'))

    def intersection(self, other):
        return Polygon(self._vs, _area_override=float('nan'))

    def contains(self, pt):
        return False


# Order in which buffer(-d).exterior.coords lists the inset corners (MLP:964-972): the one fact about GEOS's output ring the
# reference's headland generator depends on and never states.  0: the order of the input ring (the documented intent of MLP:957);
# 1: the opposite direction from the same first vertex (0, 3, 2, 1) -- what a clockwise shell starting there lists.
# tools/gen_golden.py generates fixtures for both; fcpp_options.ring_order selects it in the library.
RING_ORDER = 0


def _inset_convex(vs, d):
    """Sharp inset of a convex polygon by distance d (same vertex order).

    Vertex i moves along the mitre of its two edges:
        v_i + d * (n_a + n_b) / (1 + n_a . n_b)
    with n_a, n_b the inward unit normals of edge (i-1) and edge i.  For an
    axis-aligned rectangle this is exactly (x +- d, y +- d).
    """
    n = len(vs)
    a, _ = _area_centroid(vs)
    sgn = 1.0 if a > 0 else -1.0  # CCW -> inward normal is left of edge
    normals = []
    for i in range(n):
        x0, y0 = vs[i]
        x1, y1 = vs[(i + 1) % n]
        ex, ey = x1 - x0, y1 - y0
        ln = math.hypot(ex, ey)
        normals.append((-ey / ln * sgn, ex / ln * sgn))
    out = []
    for i in range(n):
        ax, ay = normals[(i - 1) % n]
        bx, by = normals[i]
        den = 1.0 + (ax * bx + ay * by)
        out.append((vs[i][0] + d * (ax + bx) / den, vs[i][1] + d * (ay + by) / den))
    # validity: every inset edge must keep the direction of its source edge
    for i in range(n):
        x0, y0 = out[i]
        x1, y1 = out[(i + 1) % n]
        ex, ey = vs[(i + 1) % n][0] - vs[i][0], vs[(i + 1) % n][1] - vs[i][1]
        if (x1 - x0) * ex + (y1 - y0) * ey <= 0:
            return Polygon(None, _empty=True)
    if RING_ORDER == 1 and n == 4:
        out = [out[0], out[3], out[2], out[1]]
    return Polygon(out)


class _NaNArea:
    area = float('nan')


class _BufferedLine:
    def __init__(self, length, w, pts=()):
        self.area = length * 2.0 * w + math.pi * w * w
        self._pts, self._w = list(pts), w

    def intersection(self, other):   # coverage_rate (MLP:1365): needs real clipping
        return _NaNArea()

    def contains(self, pt):
        """Interior of the exact buffer (GEOS uses a polygonal approximation of the round parts: up to
        w * (1 - cos(pi/32)) smaller there).  Point-to-segment test without division, the same one the oracle
        and the HIP kernel use (include/fcpp.h, fcpp_cover_grid)."""
        X, Y, r2 = pt.x, pt.y, self._w * self._w
        for (ax, ay), (bx, by) in zip(self._pts[:-1], self._pts[1:]):
            abx, aby, apx, apy = bx - ax, by - ay, X - ax, Y - ay
            len2 = abx * abx + aby * aby
            dot = apx * abx + apy * aby
            if dot <= 0.0:
                lhs, rhs = apx * apx + apy * apy, r2
            elif dot >= len2:
                bpx, bpy = X - bx, Y - by
                lhs, rhs = bpx * bpx + bpy * bpy, r2
            else:
                cr = abx * apy - aby * apx
                lhs, rhs = cr * cr, r2 * len2
            if lhs < rhs:
                return True
        return False


class LineString:
    def __init__(self, pts):
        self._pts = [(float(p[0]), float(p[1])) for p in pts]

    def buffer(self, w):
        length = 0.0
        for i in range(1, len(self._pts)):
            length += math.hypot(self._pts[i][0] - self._pts[i - 1][0],
                                 self._pts[i][1] - self._pts[i - 1][1])
        return _BufferedLine(length, w, self._pts)


class Point:
    def __init__(self, x, y=None):
        if y is None:
            x, y = x
        self.x = float(x)
        self.y = float(y)


def unary_union(polys):
    return polys[0] if polys else Polygon(None, _empty=True)


def install():
    """Register the stand-in under the names the reference imports."""
    if 'shapely' in sys.modules and not getattr(sys.modules['shapely'], '_FCPP_STANDIN', False):
        raise RuntimeError('a real shapely is importable; do not use the stand-in')
    shp = types.ModuleType('shapely')
    shp._FCPP_STANDIN = True
    geo = types.ModuleType('shapely.geometry')
    ops = types.ModuleType('shapely.ops')
    geo.Polygon, geo.LineString, geo.Point = Polygon, LineString, Point
    ops.unary_union = unary_union
    shp.geometry, shp.ops = geo, ops
    sys.modules['shapely'] = shp
    sys.modules['shapely.geometry'] = geo
    sys.modules['shapely.ops'] = ops
