// fcpp_host.cpp -- host-side mirror of TwoLayerPathPlannerV37.__init__ and of the O(1)-per-field
// decisions of plan_complete_coverage() (reference: multi_layer_planner_v3.py = "MLP").
//
// Nothing here touches path points: it turns each field into (a) the integer facts the reference
// derives with Python floats (swath count, loop count, start corner, pass order, reverse-fill
// counts) using the same float64 operation order, and (b) a closed-form device descriptor
// (DevField + a few DevPrim) from which the HIP kernels compute any path point from its index.
//
// Shapely is replaced by exact formulas for convex quadrilaterals (the only shapes the reference's
// generator handles meaningfully): mitre inset, area centroid, bounds.  GEOS-specific values are
// not reproducible here (DESIGN.md "parity unpinned").
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "fcpp_geom.h"
#include "fcpp_internal.h"
#include "fcpp_parallel.h"

namespace fcpp {
namespace {

struct Quad { double x[4], y[4]; };

double area_centroid(const Quad &q, double &cx, double &cy)
{
    double a = 0, sx = 0, sy = 0;
    for (int i = 0; i < 4; ++i) {
        int j = (i + 1) & 3;
        double cr = q.x[i] * q.y[j] - q.x[j] * q.y[i];
        a += cr;
        sx += (q.x[i] + q.x[j]) * cr;
        sy += (q.y[i] + q.y[j]) * cr;
    }
    a *= 0.5;
    if (fabs(a) < 1e-300) { cx = q.x[0]; cy = q.y[0]; return 0.0; }
    cx = sx / (6.0 * a); cy = sy / (6.0 * a);
    return a;
}

// Polygon.buffer(-d) for a convex quadrilateral: mitre inset, vertex order kept.  The directions along which the vertices move do
// not depend on d: a field computes them once (four insets per field: the work area and the headland loops).
struct Mitre { double sx[4], sy[4], den[4]; };
void mitre_of(const Quad &q, Mitre &m)
{
    double cx, cy;
    const double sgn = area_centroid(q, cx, cy) > 0 ? 1.0 : -1.0;
    double nx[4], ny[4];
    for (int i = 0; i < 4; ++i) {
        int j = (i + 1) & 3;
        double ex = q.x[j] - q.x[i], ey = q.y[j] - q.y[i];
        double ln = hypot(ex, ey);
        nx[i] = -ey / ln * sgn; ny[i] = ex / ln * sgn;
    }
    for (int i = 0; i < 4; ++i) {
        int p = (i + 3) & 3;
        m.den[i] = 1.0 + (nx[p] * nx[i] + ny[p] * ny[i]);
        m.sx[i] = nx[p] + nx[i]; m.sy[i] = ny[p] + ny[i];
    }
}
// false = empty
bool inset(const Quad &q, const Mitre &m, double d, Quad &o)
{
    for (int i = 0; i < 4; ++i) {
        o.x[i] = q.x[i] + d * m.sx[i] / m.den[i];
        o.y[i] = q.y[i] + d * m.sy[i] / m.den[i];
    }
    for (int i = 0; i < 4; ++i) {
        int j = (i + 1) & 3;
        double ex = q.x[j] - q.x[i], ey = q.y[j] - q.y[i];
        if ((o.x[j] - o.x[i]) * ex + (o.y[j] - o.y[i]) * ey <= 0) return false;
    }
    return true;
}

double abs_area(const Quad &q) { double cx, cy; return fabs(area_centroid(q, cx, cy)); }

bool is_convex(const Quad &q)
{
    int pos = 0, neg = 0;
    for (int i = 0; i < 4; ++i) {
        int j = (i + 1) & 3, k = (i + 2) & 3;
        double cr = (q.x[j] - q.x[i]) * (q.y[k] - q.y[j]) - (q.y[j] - q.y[i]) * (q.x[k] - q.x[j]);
        if (cr > 0) ++pos; else if (cr < 0) ++neg;
    }
    return (pos == 0 || neg == 0) && (pos + neg) > 0;
}

// MLP:165-192
double corner_angle(const Quad &q, int i)
{
    int p = (i + 3) & 3, n = (i + 1) & 3;
    double v1x = q.x[p] - q.x[i], v1y = q.y[p] - q.y[i];
    double v2x = q.x[n] - q.x[i], v2y = q.y[n] - q.y[i];
    double c = (v1x * v2x + v1y * v2y) / (sqrt(v1x * v1x + v1y * v1y) * sqrt(v2x * v2x + v2y * v2y));
    c = std::min(1.0, std::max(-1.0, c));
    return acos(c) * (180.0 / kPi);
}

// MLP:194-222
bool is_parallelogram(const Quad &q)
{
    double ex[4], ey[4];
    for (int i = 0; i < 4; ++i) { int j = (i + 1) & 3; ex[i] = q.x[j] - q.x[i]; ey[i] = q.y[j] - q.y[i]; }
    for (int k = 0; k < 2; ++k) {
        double cross = fabs(ex[k] * ey[k + 2] - ey[k] * ex[k + 2]);
        double na = sqrt(ex[k] * ex[k] + ey[k] * ey[k]), nb = sqrt(ex[k + 2] * ex[k + 2] + ey[k + 2] * ey[k + 2]);
        if (!(cross < 0.01 * (na * nb))) return false;
    }
    return true;
}

// MLP:265-284
void rotate_point(double x, double y, double ca, double sa, double cx, double cy, double &ox, double &oy)
{
    x -= cx; y -= cy;
    double xn = x * ca - y * sa;
    double yn = x * sa + y * ca;
    ox = xn + cx; oy = yn + cy;
}

// (saturates at 2^40 points -- far beyond any size check of the callers -- instead of converting an out-of-range double)
constexpr int64_t kCountCap = (int64_t)1 << 40;
int64_t n_for_length(double len, double ds)
{
    const double c = ceil(len / ds);
    if (!(c < (double)kCountCap)) return kCountCap;
    int64_t n = (int64_t)c + 1;
    return n < 2 ? 2 : n;
}

double lin_step(double a, double b, int64_t n) { return n > 1 ? (b - a) / (double)(n - 1) : 0.0; }

// MLP:1220-1288: distance along (dx,dy) to the bbox-at-origin boundary, capped at 3R, default 2R
double distance_to_boundary(double x, double y, double dx, double dy, double L, double H, double R)
{
    double best = 0; bool have = false;
    auto take = [&](double t) { if (t > 0 && (!have || t < best)) { best = t; have = true; } };
    if (fabs(dx) > 1e-6) { take((0 - x) / dx); take((L - x) / dx); }
    if (fabs(dy) > 1e-6) { take((0 - y) / dy); take((H - y) / dy); }
    if (!have) return 2.0 * R;
    return std::min(best, 3.0 * R);
}

// a straight primitive whose numpy.linspace step underflowed to 0 although its ends differ (no real field has one): form bit 3 sends the
// one-point-per-lane kernel through the general evaluation (fcpp_pointfn.h: eval_prim_lanes)
void flag_degenerate(DevPrim &p)
{
    if (p.kind == PRIM_LINSPACE && ((p.a[4] == 0.0 && p.a[2] != p.a[0]) || (p.a[5] == 0.0 && p.a[3] != p.a[1]))) p.form |= 8;
    if (p.kind == PRIM_RAY && p.a[5] == 0.0 && p.a[4] != 0.0) p.form |= 8;
}

const int kCornerQuadrant[4] = { 1, 2, 3, 0 };  // start heading of the corner arcs = q * pi/2 (MLP:1049-1060)

// everything about a batch that does not depend on the field: validated parameters, turn shapes, sample counts
struct PlanConsts {
    const fcpp_vehicle *veh; const fcpp_options *opt;
    double W, R, ds;
    bool clip, cloth;
    CacShape sh_pi, sh_half;
    double Re_pi, Re_half, len_uturn, len_corner, gap_lb;
    // the last two samples of a corner turn (the reverse fill leaves along their chord): angle / arc length and, for arcs, cos and sin
    int64_t nt_corner;
    double arc_step, arc_c1, arc_s1, arc_c2, arc_s2;
};

int plan_prepare(const fcpp_vehicle &veh, const fcpp_options &opt, PlanConsts &c, TurnTemplates &tt, std::string &err)
{
    const double W = veh.working_width, R = veh.min_turn_radius, ds = opt.sample_spacing;
    if (!(W > 0) || !(R > 0) || !(ds >= 0) || !(opt.clothoid_frac >= 0 && opt.clothoid_frac <= 1) ||
        (opt.turn_model != FCPP_TURN_ARC && opt.turn_model != FCPP_TURN_CLOTHOID) ||
        (opt.obstacle_mode != FCPP_OBSTACLES_FLAG && opt.obstacle_mode != FCPP_OBSTACLES_AVOID) ||
        (opt.ring_order != FCPP_RING_AS_VERTICES && opt.ring_order != FCPP_RING_REVERSED)) {
        err = "invalid vehicle parameters or options";
        return FCPP_EINVAL;
    }
    if (!(veh.max_longitudinal_accel > 0) || !(veh.max_lateral_accel > 0)) {
        err = "accelerations must be positive";
        return FCPP_EINVAL;
    }
    {   // every parameter a finite number (a NaN passes none of the comparisons above, an infinity or a denormal width all of them)
        const double vals[] = { veh.working_width, veh.min_turn_radius, veh.max_work_speed_kmh, veh.max_headland_speed_kmh,
                                veh.headland_turn_speed_kmh, veh.max_lateral_accel, veh.max_longitudinal_accel, veh.safety_factor,
                                opt.sample_spacing, opt.clothoid_frac, opt.geofence_tol };
        for (double v : vals)
            if (!isfinite(v)) { err = "vehicle parameters and options must be finite"; return FCPP_EINVAL; }
        if (W < 1e-6 || R < 1e-6 || (ds > 0 && ds < 1e-9)) { err = "working width, turn radius or sample spacing too small"; return FCPP_EINVAL; }
    }
    c.veh = &veh; c.opt = &opt; c.W = W; c.R = R; c.ds = ds;
    c.clip = opt.obstacle_mode == FCPP_OBSTACLES_AVOID;
    c.cloth = opt.turn_model == FCPP_TURN_CLOTHOID;
    c.sh_pi = make_cac_shape(kPi, opt.clothoid_frac); c.sh_half = make_cac_shape(kHalfPi, opt.clothoid_frac);
    auto fit_radius = [&](const CacShape &sh) {
        if (!opt.clothoid_fit) return R;
        return R * (2 * sin(sh.D / 2)) / sqrt(sh.ex * sh.ex + sh.ey * sh.ey);
    };
    c.Re_pi = c.cloth ? fit_radius(c.sh_pi) : R; c.Re_half = c.cloth ? fit_radius(c.sh_half) : R;
    c.len_uturn = c.cloth ? c.sh_pi.T * c.Re_pi : kPi * R;
    c.len_corner = c.cloth ? c.sh_half.T * c.Re_half : kHalfPi * R;
    // gap.area > 0.1 (MLP:1070): 2R x 2R square minus the arc buffered by W/2.  The buffer's area is at
    // most (pi R/2) W + pi W^2/4, so the decision is certain when this lower bound exceeds 0.1.
    c.gap_lb = 4 * R * R - (kPi * R / 2 * W + kPi * W * W / 4);
    c.nt_corner = ds > 0 ? n_for_length(c.len_corner, ds) : 15;
    c.arc_step = lin_step(0.0, kHalfPi, c.nt_corner);
    {
        const double th1 = linspace_at(0.0, kHalfPi, c.arc_step, c.nt_corner, c.nt_corner - 1);
        const double th2 = linspace_at(0.0, kHalfPi, c.arc_step, c.nt_corner, c.nt_corner - 2);
        c.arc_c1 = cos(th1); c.arc_s1 = sin(th1); c.arc_c2 = cos(th2); c.arc_s2 = sin(th2);
    }
    // turn templates (same sample counts as every field computes)
    memset(&tt, 0, sizeof(tt));
    tt.turn_model = opt.turn_model; tt.R = R;
    tt.nu = (int32_t)std::min<int64_t>(ds > 0 ? n_for_length(c.len_uturn, ds) : 20, INT32_MAX);
    tt.nc = (int32_t)std::min<int64_t>(ds > 0 ? n_for_length(c.len_corner, ds) : 15, INT32_MAX);
    tt.u_end = c.cloth ? c.sh_pi.T * c.Re_pi : kPi; tt.u_step = lin_step(0.0, tt.u_end, tt.nu); tt.u_Re = c.Re_pi;
    tt.c_end = c.cloth ? c.sh_half.T * c.Re_half : kHalfPi; tt.c_step = lin_step(0.0, tt.c_end, tt.nc); tt.c_Re = c.Re_half;
    return FCPP_OK;
}

// One field: __init__ + the O(1) decisions of plan_complete_coverage.  Fills `in` (point_offset stays 0) and, with want_device, `df`
// (pt_off 0, prim_first = index into `prims`, to which the field's primitives are appended).  A field that raises gets in.status < 0
// and no points.  -> points of the field.
int64_t plan_field(const PlanConsts &pc, const fcpp_field &f, const fcpp_polys *polys, bool want_device, fcpp_field_info &in, DevField &df,
                   std::vector<DevPrim> &prims)
{
    const fcpp_vehicle &veh = *pc.veh;
    const fcpp_options &opt = *pc.opt;
    const double W = pc.W, R = pc.R, ds = pc.ds;
    const bool clip = pc.clip, cloth = pc.cloth;
    const CacShape &sh_pi = pc.sh_pi, &sh_half = pc.sh_half;
    const double Re_pi = pc.Re_pi, Re_half = pc.Re_half, len_uturn = pc.len_uturn, gap_lb = pc.gap_lb;
    memset(&in, 0, sizeof(in));
    memset(&df, 0, sizeof(df));
    const size_t prim_mark = prims.size();
    auto fail = [&](int code) -> int64_t {
        in.status = code; in.n_main = in.n_head = 0;
        memset(in.n_reverse, 0, sizeof(in.n_reverse));
        df.n_main = df.n_total = 0; df.gen_main = 0; df.prim_first = (int32_t)prim_mark; df.prim_count = 0;
        prims.resize(prim_mark);
        return 0;
    };
    {
        Quad q;
        bool finite = true;
        for (int i = 0; i < 4; ++i) { q.x[i] = f.vx[i]; q.y[i] = f.vy[i]; finite = finite && isfinite(q.x[i]) && isfinite(q.y[i]); }
        if (!finite || !is_convex(q)) { return fail(FCPP_EUNSUPPORTED); }

        // ---- __init__ (MLP:109-135, 137-163, 310, 322-343)
        double bminx = q.x[0], bmaxx = q.x[0], bminy = q.y[0], bmaxy = q.y[0];
        for (int i = 1; i < 4; ++i) {
            bminx = std::min(bminx, q.x[i]); bmaxx = std::max(bmaxx, q.x[i]);
            bminy = std::min(bminy, q.y[i]); bmaxy = std::max(bmaxy, q.y[i]);
        }
        const double L = f.from_vertices ? (bmaxx - bminx) : q.x[1];
        const double H = f.from_vertices ? (bmaxy - bminy) : q.y[2];
        in.field_length = L; in.field_width = H;
        bool all90 = true;
        for (int i = 0; i < 4; ++i) {
            in.corner_angles[i] = corner_angle(q, i);
            if (!(fabs(in.corner_angles[i] - 90) < 1.0)) all90 = false;
        }
        in.shape = all90 ? 0 : (is_parallelogram(q) ? 1 : 2);
        const double hw = R;
        in.headland_width = hw;
        const bool has_start = f.has_start && (0 <= f.start_x && f.start_x <= L && 0 <= f.start_y && f.start_y <= H);
        const bool has_end = f.has_end && (0 <= f.end_x && f.end_x <= L && 0 <= f.end_y && f.end_y <= H);
        in.start_kept = has_start; in.end_kept = has_end;

        // ---- start corner (MLP:345-385)
        int sci = 0;
        if (has_start) {
            const double cxs[4] = { hw / 2, L - hw / 2, L - hw / 2, hw / 2 };
            const double cys[4] = { hw / 2, hw / 2, H - hw / 2, H - hw / 2 };
            double best = 0;
            for (int i = 0; i < 4; ++i) {
                double dx = cxs[i] - f.start_x, dy = cys[i] - f.start_y;
                double d = sqrt(dx * dx + dy * dy);
                if (i == 0 || d < best) { best = d; sci = i; }
            }
        }
        in.start_corner = sci;

        // ---- layer 1 frame (MLP:591-611, 670-718)
        Quad mq;
        Mitre mit;
        mitre_of(q, mit);
        if (!inset(q, mit, hw, mq) || abs_area(mq) < 1.0) { return fail(FCPP_EINVAL); }
        const double rot = atan2(q.y[1] - q.y[0], q.x[1] - q.x[0]);
        in.rotation_angle = rot;
        const bool rotated = fabs(rot) > 0.01;
        in.rotated = rotated;
        double ccx = 0, ccy = 0, sx = f.start_x, sy = f.start_y;
        Quad rq = mq;
        if (rotated) {
            area_centroid(mq, ccx, ccy);
            const double ca = cos(-rot), sa = sin(-rot);
            for (int i = 0; i < 4; ++i) rotate_point(mq.x[i], mq.y[i], ca, sa, ccx, ccy, rq.x[i], rq.y[i]);
            if (has_start) rotate_point(sx, sy, ca, sa, ccx, ccy, sx, sy);
        }
        double min_x = rq.x[0], max_x = rq.x[0], min_y = rq.y[0], max_y = rq.y[0];
        for (int i = 1; i < 4; ++i) {
            min_x = std::min(min_x, rq.x[i]); max_x = std::max(max_x, rq.x[i]);
            min_y = std::min(min_y, rq.y[i]); max_y = std::max(max_y, rq.y[i]);
        }
        int reverse_order = 0, start_from_right = 0;   // MLP:631-668
        if (has_start) {
            if (sy > (min_y + max_y) / 2) reverse_order = 1;
            if (sx > (min_x + max_x) / 2) start_from_right = 1;
        }
        in.reverse_order = reverse_order; in.start_from_right = start_from_right;

        // ---- layer 1 sizes (MLP:736-739)
        const double lsx = min_x + R, lex = max_x - R;
        const double Pd = (max_y - min_y) / W;          // int(height / W) + 1 (MLP:739); refused below when beyond 32 bits
        const int64_t P = Pd < (double)INT32_MAX ? (int64_t)Pd + 1 : (int64_t)INT32_MAX + 1;
        const int64_t n_line = ds > 0 ? n_for_length(fabs(lex - lsx), ds) : 2;
        const int64_t n_turn = ds > 0 ? n_for_length(len_uturn, ds) : 20;
        // (the swath index lives in bits 8..31 of the flag / segment word)
        if (P >= ((int64_t)1 << (32 - FCPP_INDEX_SHIFT)) || n_line + n_turn > INT32_MAX - 2 * TILE_POINTS) { return fail(FCPP_ESIZE); }
        in.n_swaths = (int32_t)P;
        int64_t n_main = P * n_line + (P - 1) * n_turn;
        df.gen_main = n_main;
        df.prim_first = (int32_t)prims.size();
        if (clip) {
            // ---- obstacle-aware swaths (include/fcpp.h): layer 1 as a list of primitives -- sub-swaths, detour legs, U-turns
            struct Box { double x0, y0, x1, y1; };
            std::vector<Box> boxes;
            const double ca = cos(-rot), sa = sin(-rot);
            bool bad_obs = f.n_obstacles < 0 || (f.n_obstacles > 0 && (!polys || f.obstacle_first < 0 || f.obstacle_first + f.n_obstacles > polys->n_polys));
            for (int k = 0; k < f.n_obstacles && !bad_obs; ++k) {
                const int64_t a0 = polys->offsets[f.obstacle_first + k], a1 = polys->offsets[f.obstacle_first + k + 1];
                if (a1 <= a0) continue;
                Box b = { HUGE_VAL, HUGE_VAL, -HUGE_VAL, -HUGE_VAL };
                for (int64_t q2 = a0; q2 < a1; ++q2) {
                    double ox = polys->x[q2], oy = polys->y[q2];
                    if (!(isfinite(ox) && isfinite(oy))) { bad_obs = true; break; }
                    if (rotated) rotate_point(ox, oy, ca, sa, ccx, ccy, ox, oy);
                    b.x0 = std::min(b.x0, ox); b.x1 = std::max(b.x1, ox); b.y0 = std::min(b.y0, oy); b.y1 = std::max(b.y1, oy);
                }
                b.x0 -= W / 2; b.y0 -= W / 2; b.x1 += W / 2; b.y1 += W / 2;
                boxes.push_back(b);
            }
            if (bad_obs) { return fail(FCPP_ESIZE); }
            // grown boxes that overlap or touch become ONE box (their bounding box), until no two do: the boxes are then disjoint,
            // so a detour leg -- which runs on the boundary of its own box -- cannot enter another one
            for (bool merged = true; merged;) {
                merged = false;
                for (size_t i = 0; i < boxes.size(); ++i)
                    for (size_t j = i + 1; j < boxes.size();) {
                        Box &a = boxes[i];
                        const Box &b = boxes[j];
                        if (a.x0 <= b.x1 + 1e-9 && b.x0 <= a.x1 + 1e-9 && a.y0 <= b.y1 + 1e-9 && b.y0 <= a.y1 + 1e-9) {
                            a.x0 = std::min(a.x0, b.x0); a.y0 = std::min(a.y0, b.y0); a.x1 = std::max(a.x1, b.x1); a.y1 = std::max(a.y1, b.y1);
                            boxes.erase(boxes.begin() + (long)j);
                            merged = true;
                        } else ++j;
                    }
            }
            const double rc = cos(rot), rs = sin(rot);
            const double lo = std::min(lsx, lex), hi = std::max(lsx, lex);
            int64_t pos1 = 0;
            bool unsupported = false;
            auto push1 = [&](DevPrim &pr) { pr.start = pos1; pos1 += pr.n; flag_degenerate(pr); if (want_device) prims.push_back(pr); };
            auto world = [&](double &x, double &y) { if (rotated) rotate_point(x, y, rc, rs, ccx, ccy, x, y); };
            auto push_line = [&](double ax, double ay, double bx, double by, uint32_t kind, int64_t pi, double vnom, bool detour) {
                const double len = sqrt((bx - ax) * (bx - ax) + (by - ay) * (by - ay));
                int64_t np = detour ? (ds > 0 ? n_for_length(len, ds) : std::max<int64_t>(2, (int64_t)(len / 0.5) + 1))
                                    : (ds > 0 ? n_for_length(len, ds) : 2);
                if (np > INT32_MAX) { unsupported = true; return; }
                world(ax, ay); world(bx, by);
                DevPrim pr;
                memset(&pr, 0, sizeof(pr));
                pr.kind = PRIM_LINSPACE; pr.n = (int32_t)np; pr.v_nom = vnom;
                pr.fs = kind | ((uint32_t)pi << FCPP_INDEX_SHIFT);
                pr.a[0] = ax; pr.a[1] = ay; pr.a[2] = bx; pr.a[3] = by;
                pr.a[4] = lin_step(ax, bx, np); pr.a[5] = lin_step(ay, by, np);
                push1(pr);
            };
            std::vector<int> blk;
            for (int64_t idx = 0; idx < P && !unsupported; ++idx) {
                const int64_t pi = reverse_order ? (P - 1 - idx) : idx;
                const double y = min_y + (double)pi * W;
                const bool go_left = start_from_right ? ((idx & 1) == 0) : ((idx & 1) == 1);
                const double xs = go_left ? lex : lsx, xe = go_left ? lsx : lex;
                blk.clear();
                for (size_t k = 0; k < boxes.size(); ++k)
                    if (boxes[k].y0 < y && y < boxes[k].y1 && boxes[k].x1 > lo && boxes[k].x0 < hi) blk.push_back((int)k);
                std::sort(blk.begin(), blk.end(), [&](int a, int b) { return go_left ? boxes[a].x1 > boxes[b].x1 : boxes[a].x0 < boxes[b].x0; });
                double cur = xs;
                for (size_t k = 0; k < blk.size() && !unsupported; ++k) {
                    const Box &b = boxes[blk[k]];
                    const double nearx = go_left ? b.x1 : b.x0, farx = go_left ? b.x0 : b.x1;
                    // the box must lie strictly inside the line, beyond the previous box
                    if (!(b.x0 > lo + 1e-9 && b.x1 < hi - 1e-9) || !(go_left ? nearx < cur - 1e-9 : nearx > cur + 1e-9)) { unsupported = true; break; }
                    // over the nearer side (top or bottom) if that keeps the detour inside the work area's y-range, else over the
                    // other one; a box that leaves room on neither side cannot be driven around
                    const bool top_ok = b.y1 <= max_y + 1e-9, bot_ok = b.y0 >= min_y - 1e-9;
                    const bool want_top = b.y1 - y <= y - b.y0;
                    if (!top_ok && !bot_ok) { unsupported = true; break; }
                    const double ys = (want_top ? top_ok : !bot_ok) ? b.y1 : b.y0;
                    push_line(cur, y, nearx, y, FCPP_KIND_SWATH, pi, veh.max_work_speed_kmh, false);
                    push_line(nearx, y, nearx, ys, FCPP_KIND_DETOUR, pi, veh.headland_turn_speed_kmh, true);
                    push_line(nearx, ys, farx, ys, FCPP_KIND_DETOUR, pi, veh.headland_turn_speed_kmh, true);
                    push_line(farx, ys, farx, y, FCPP_KIND_DETOUR, pi, veh.headland_turn_speed_kmh, true);
                    cur = farx;
                }
                if (unsupported) break;
                push_line(cur, y, xe, y, FCPP_KIND_SWATH, pi, veh.max_work_speed_kmh, false);
                if (idx < P - 1) {
                    const bool turn_right = !go_left;
                    DevPrim pr;
                    memset(&pr, 0, sizeof(pr));
                    pr.kind = PRIM_UTURN; pr.n = (int32_t)n_turn; pr.v_nom = veh.headland_turn_speed_kmh;
                    pr.fs = FCPP_KIND_UTURN | ((uint32_t)pi << FCPP_INDEX_SHIFT);
                    pr.form = (turn_right ? 1 : 0) | (rotated ? 2 : 0) | (cloth ? 4 : 0);
                    pr.a[0] = cloth ? (turn_right ? (max_x - R) : (min_x + R)) : (turn_right ? max_x : min_x);
                    pr.a[1] = y;
                    pr.a[2] = rc; pr.a[3] = rs; pr.a[4] = ccx; pr.a[5] = ccy;
                    push1(pr);
                }
            }
            if (unsupported) {
                return fail(FCPP_EUNSUPPORTED);
            }
            n_main = pos1;
            df.gen_main = 0;
        }
        in.n_main = n_main;

        df.n_main = n_main;
        df.lsx = lsx; df.lex = lex; df.line_step = lin_step(lsx, lex, n_line);
        df.min_x = min_x; df.max_x = max_x; df.min_y = min_y; df.W = W; df.R = R;
        df.turn_end = cloth ? sh_pi.T * Re_pi : kPi;
        df.turn_step = lin_step(0.0, df.turn_end, n_turn);
        df.turn_Re = Re_pi;
        df.rot_cos = cos(rot); df.rot_sin = sin(rot); df.rot_cx = ccx; df.rot_cy = ccy;
        df.v_work = veh.max_work_speed_kmh; df.v_turn = veh.headland_turn_speed_kmh;
        df.P = (int32_t)P; df.n_line = (int32_t)n_line; df.n_turn = (int32_t)n_turn;
        df.reverse_order = reverse_order; df.start_from_right = start_from_right; df.rotated = rotated;
        df.turn_model = opt.turn_model;

        // ---- layer 2 (MLP:898-1084)
        const int num_loops = (int)ceil(hw / W);
        in.n_loops = num_loops;
        int64_t pos = n_main;
        bool bad = false;
        double first_head[2] = { 0, 0 }, last_head[2] = { 0, 0 };
        auto push = [&](DevPrim &p) { p.start = pos; pos += p.n; flag_degenerate(p); if (want_device) prims.push_back(p); };
        for (int loop = 0; loop < num_loops && !bad; ++loop) {
            const double offset = W / 2 + loop * W;
            Quad c;
            if (!inset(q, mit, offset, c) || abs_area(c) < 1.0) { bad = true; break; }
            if (opt.ring_order == FCPP_RING_REVERSED) { std::swap(c.x[1], c.x[3]); std::swap(c.y[1], c.y[3]); }      // ring lists 0, 3, 2, 1
            const uint32_t lp = FCPP_FLAG_HEADLAND | ((uint32_t)(loop * 8) << FCPP_INDEX_SHIFT);
            DevPrim p;
            memset(&p, 0, sizeof(p));
            p.kind = PRIM_POINT; p.n = 1; p.v_nom = veh.max_headland_speed_kmh;
            p.fs = FCPP_KIND_HEAD_START | lp | ((uint32_t)sci << FCPP_INDEX_SHIFT);
            p.a[0] = c.x[sci]; p.a[1] = c.y[sci];
            push(p);
            if (loop == 0) { first_head[0] = c.x[sci]; first_head[1] = c.y[sci]; }
            for (int i = 0; i < 4; ++i) {
                const int cur = (sci + i) & 3, nxt = (sci + i + 1) & 3;
                const double seg_len = hypot(c.x[nxt] - c.x[cur], c.y[nxt] - c.y[cur]);
                const int64_t ns = ds > 0 ? n_for_length(seg_len, ds) : 20;
                const int64_t nt = pc.nt_corner;
                if (ns > INT32_MAX || nt > INT32_MAX) { bad = true; break; }
                memset(&p, 0, sizeof(p));
                p.kind = PRIM_LINSPACE; p.n = (int32_t)ns; p.v_nom = veh.max_headland_speed_kmh;
                p.fs = FCPP_KIND_HEAD_STRAIGHT | lp | ((uint32_t)cur << FCPP_INDEX_SHIFT);
                p.a[0] = c.x[cur]; p.a[1] = c.y[cur]; p.a[2] = c.x[nxt]; p.a[3] = c.y[nxt];
                p.a[4] = lin_step(c.x[cur], c.x[nxt], ns); p.a[5] = lin_step(c.y[cur], c.y[nxt], ns);
                push(p);
                last_head[0] = c.x[nxt]; last_head[1] = c.y[nxt];
                if (i == 3) break;
                // corner turn at `nxt` (MLP:1024-1063 / 1580-1608)
                double e1[2], e2[2];  // last and second-to-last point of the turn (for the reverse direction)
                memset(&p, 0, sizeof(p));
                p.n = (int32_t)nt; p.v_nom = veh.headland_turn_speed_kmh;
                p.fs = FCPP_KIND_CORNER | lp | ((uint32_t)nxt << FCPP_INDEX_SHIFT);
                if (!cloth) {
                    p.kind = PRIM_ARC; p.form = nxt;
                    p.a[0] = c.x[nxt]; p.a[1] = c.y[nxt]; p.a[2] = R; p.a[3] = kHalfPi;
                    p.a[4] = pc.arc_step;
                    corner_arc_point(nxt, c.x[nxt], c.y[nxt], R, pc.arc_c1, pc.arc_s1, e1[0], e1[1]);
                    corner_arc_point(nxt, c.x[nxt], c.y[nxt], R, pc.arc_c2, pc.arc_s2, e2[0], e2[1]);
                } else {
                    p.kind = PRIM_CAC; p.form = kCornerQuadrant[nxt];
                    const double T = sh_half.T * Re_half;
                    p.a[0] = c.x[nxt]; p.a[1] = c.y[nxt]; p.a[2] = kCornerQuadrant[nxt] * kHalfPi; p.a[3] = -kHalfPi;
                    p.a[4] = Re_half; p.a[5] = lin_step(0.0, T, nt); p.a[6] = T;
                    const double s1 = linspace_at(0.0, T, p.a[5], nt, nt - 1), s2 = linspace_at(0.0, T, p.a[5], nt, nt - 2);
                    cac_world_point(sh_half, c.x[nxt], c.y[nxt], p.form, -1.0, Re_half, s1, e1[0], e1[1]);
                    cac_world_point(sh_half, c.x[nxt], c.y[nxt], p.form, -1.0, Re_half, s2, e2[0], e2[1]);
                }
                push(p);
                // reverse fill (MLP:1043, 224-242, 1066-1082, 1154-1218)
                const bool add_rev = (loop == 0) && (in.corner_angles[nxt] >= 60);
                if (add_rev) {
                    if (!(gap_lb > 0.1)) { bad = true; in.status = FCPP_EUNSUPPORTED; break; }
                    const double tx = e1[0] - e2[0], ty = e1[1] - e2[1];
                    const double nrm = sqrt(tx * tx + ty * ty);
                    double dx = -1.0, dy = 0.0;
                    if (nrm > 1e-6) { dx = -tx / nrm; dy = -ty / nrm; }
                    const double len = distance_to_boundary(e1[0], e1[1], dx, dy, L, H, R);
                    int64_t nr;
                    if (ds > 0) nr = n_for_length(len, ds);
                    else { nr = (int64_t)(len / 0.5); if (nr < 10) nr = 10; }
                    memset(&p, 0, sizeof(p));
                    p.kind = PRIM_RAY; p.n = (int32_t)nr; p.v_nom = 2.5;   // MLP:1080
                    p.fs = FCPP_KIND_REVERSE | lp | ((uint32_t)nxt << FCPP_INDEX_SHIFT);
                    p.a[0] = e1[0]; p.a[1] = e1[1]; p.a[2] = dx; p.a[3] = dy; p.a[4] = len;
                    p.a[5] = lin_step(0.0, len, nr);
                    push(p);
                    in.n_reverse[nxt] = (int32_t)nr;
                }
            }
        }
        if (bad) {
            return fail(in.status ? in.status : FCPP_EHEADLAND);
        }
        in.n_head = pos - n_main;
        if (has_start) {   // MLP:437-441
            in.approach_from[0] = f.start_x; in.approach_from[1] = f.start_y;
            in.approach_to[0] = first_head[0]; in.approach_to[1] = first_head[1];
        }
        if (has_end) {     // MLP:443-447
            in.departure_from[0] = last_head[0]; in.departure_from[1] = last_head[1];
            in.departure_to[0] = f.end_x; in.departure_to[1] = f.end_y;
        }
        if (want_device) {
            df.n_total = pos;
            df.prim_count = (int32_t)prims.size() - df.prim_first;
            df.obs_first = (int32_t)f.obstacle_first; df.obs_count = f.n_obstacles;
            // geofence half-planes: inside <=> ex*px + ey*py + eo >= -tol
            double cx, cy;
            const double sgn = area_centroid(q, cx, cy) > 0 ? 1.0 : -1.0;
            for (int i = 0; i < 4; ++i) {
                int j = (i + 1) & 3;
                double ex = q.x[j] - q.x[i], ey = q.y[j] - q.y[i];
                double ln = sqrt(ex * ex + ey * ey);
                df.ex[i] = -ey / ln * sgn; df.ey[i] = ex / ln * sgn;
                df.eo[i] = -(df.ex[i] * q.x[i] + df.ey[i] * q.y[i]);
            }
        }
        return pos;
    }
}

// what decides a field's plan besides the vehicle and the options: its constructor arguments (the obstacle list only matters for
// obstacle-aware swaths; such fields are never shared)
struct FieldKey {
    double v[12];
    int32_t from_vertices, has_start, has_end;
    bool operator==(const FieldKey &o) const { return memcmp(this, &o, sizeof(FieldKey)) == 0; }
};
FieldKey key_of(const fcpp_field &f)
{
    FieldKey k;
    memset(&k, 0, sizeof(k));
    for (int i = 0; i < 4; ++i) { k.v[i] = f.vx[i]; k.v[4 + i] = f.vy[i]; }
    k.has_start = f.has_start != 0; k.has_end = f.has_end != 0;
    if (k.has_start) { k.v[8] = f.start_x; k.v[9] = f.start_y; }
    if (k.has_end) { k.v[10] = f.end_x; k.v[11] = f.end_y; }
    k.from_vertices = f.from_vertices != 0;
    return k;
}
uint64_t hash_of(const FieldKey &k)
{
    uint64_t h = 1469598103934665603ull;
    const unsigned char *p = reinterpret_cast<const unsigned char *>(&k);
    uint64_t w;
    for (size_t i = 0; i + 8 <= sizeof(FieldKey); i += 8) { memcpy(&w, p + i, 8); h = (h ^ w) * 1099511628211ull; h ^= h >> 29; }
    return h;
}

}  // namespace

int plan_templates(const fcpp_vehicle &veh, const fcpp_options &opt, TurnTemplates &tt, std::string &err)
{
    PlanConsts pc;
    return plan_prepare(veh, opt, pc, tt, err);
}

int validate_polys(const fcpp_polys *polys, std::string &err)
{
    if (!polys) return FCPP_OK;
    const int64_t n = polys->n_polys;
    if (n < 0 || n > INT32_MAX) { err = "bad polygon count"; return FCPP_ESIZE; }
    if (n == 0) return FCPP_OK;
    if (!polys->offsets || polys->offsets[0] != 0) { err = "polygon offsets must start at 0"; return FCPP_ESIZE; }
    for (int64_t k = 0; k < n; ++k)
        if (polys->offsets[k + 1] < polys->offsets[k]) { err = "polygon offsets must be non-decreasing"; return FCPP_ESIZE; }
    if (polys->offsets[n] > 0 && (!polys->x || !polys->y)) { err = "polygon coordinates are NULL"; return FCPP_EINVAL; }
    return FCPP_OK;
}

int build_host_plan(const fcpp_vehicle &veh, const fcpp_options &opt, int64_t n, const fcpp_field *fields, const fcpp_polys *polys,
                    bool want_device, HostPlan &out, std::string &err)
{
    PlanConsts pc;
    int rc = plan_prepare(veh, opt, pc, out.tt, err);
    if (rc != FCPP_OK) return rc;
    // the polygon table is checked before any field dereferences it (obstacle-aware swaths read it on the host)
    if ((rc = validate_polys(polys, err)) != FCPP_OK) return rc;
    const int64_t n_polys = polys ? polys->n_polys : 0;
    for (int64_t i = 0; i < n; ++i)
        if (fields[i].n_obstacles < 0 || fields[i].obstacle_first < 0 ||
            (fields[i].n_obstacles > 0 && fields[i].obstacle_first + fields[i].n_obstacles > n_polys)) {
            // (without a table -- fcpp_plan_count(..., NULL, ...) in the reference's mode -- the ranges are not used by anything)
            if (polys || pc.clip) { err = "field obstacle range outside the polygon table"; return FCPP_ESIZE; }
        }
    out.info.assign((size_t)n, fcpp_field_info());
    out.fields.clear(); out.blocks.clear();
    if (want_device) out.fields.resize((size_t)n);
    out.same_as.clear();
    if (want_device) out.same_as.assign((size_t)n, -1);
    const int64_t nb = (n + PLAN_BLOCK_FIELDS - 1) / PLAN_BLOCK_FIELDS;
    out.blocks.resize((size_t)nb);
    std::vector<int64_t> block_points((size_t)nb, 0);
    const bool share = getenv("FCPP_NO_SHARE") == nullptr;      // (diagnostic: plan every field on its own; tests/native/tiler_check_driver.cpp)

    // ---- the blocks, side by side: every block plans its fields into its own primitive list; point offsets and primitive indices are
    // relative to the block until the bases are known.  (Side by side from 16 384 fields on: a field takes 0.1-1 us to plan, waking
    // the pool's threads twice 0.5 ms -- the headline's 4096 equal fields 0.13 ms in the calling thread, 0.71 ms on sixteen.)
    const auto for_blocks = [&](const std::function<void(int64_t)> &fn) {
        if (n >= 16384) WorkerPool::parallel_for(nb, fn);
        else for (int64_t b = 0; b < nb; ++b) fn(b);
    };
    for_blocks([&](int64_t b) {
        PlanBlock &blk = out.blocks[(size_t)b];
        blk.f0 = b * PLAN_BLOCK_FIELDS; blk.f1 = std::min(n, blk.f0 + PLAN_BLOCK_FIELDS);
        // fields with the same constructor arguments (a batch of equal fields is the headline workload) are planned once per block:
        // the later ones copy the first one's decisions and share its primitives
        FieldKey keys[PLAN_BLOCK_FIELDS];
        uint64_t hashes[PLAN_BLOCK_FIELDS];
        int n_proto = 0, proto_field[PLAN_BLOCK_FIELDS];
        int64_t pt = 0;
        DevField scratch_df;
        for (int64_t fi = blk.f0; fi < blk.f1; ++fi) {
            const fcpp_field &f = fields[fi];
            fcpp_field_info &in = out.info[(size_t)fi];
            DevField &df = want_device ? out.fields[(size_t)fi] : scratch_df;
            const bool shareable = share && !(pc.clip && f.n_obstacles > 0);
            int64_t npts = -1;
            if (shareable) {
                const FieldKey k = key_of(f);
                const uint64_t h = hash_of(k);
                int hit = -1;
                for (int j = 0; j < n_proto; ++j)
                    if (hashes[j] == h && keys[j] == k) { hit = j; break; }
                if (hit >= 0) {
                    const int64_t pf = proto_field[hit];
                    in = out.info[(size_t)pf];
                    if (want_device) { df = out.fields[(size_t)pf]; out.same_as[(size_t)fi] = (int32_t)pf; }
                    npts = in.n_main + in.n_head;
                } else { keys[n_proto] = k; hashes[n_proto] = h; proto_field[n_proto] = (int)fi; ++n_proto; }
            }
            if (npts < 0) npts = plan_field(pc, f, polys, want_device, in, df, blk.prims);
            in.point_offset = pt;
            if (want_device) {
                df.pt_off = pt;
                df.obs_first = (int32_t)f.obstacle_first; df.obs_count = f.n_obstacles;
            }
            pt += npts;
        }
        blk.points = pt;
        block_points[(size_t)b] = pt;
    });

    // ---- bases: a serial prefix over the blocks, then the fields' offsets made batch-wide (side by side again)
    int64_t pt_off = 0, prim_off = 0;
    for (int64_t b = 0; b < nb; ++b) {
        PlanBlock &blk = out.blocks[(size_t)b];
        blk.point_base = pt_off; blk.prim_base = prim_off;
        pt_off += blk.points; prim_off += (int64_t)blk.prims.size();
        if (pt_off > kCountCap) { err = "batch too large"; return FCPP_ESIZE; }          // (2^40 points = 40 TB of output: no batch gets there)
        if (prim_off > ((int64_t)1 << 26)) {     // (primitive indices are 32-bit; 2^26 records are 7 GB of host memory)
            err = "too many path primitives in one batch (obstacle-aware swaths of very large fields): split the batch";
            return FCPP_ESIZE;
        }
    }
    out.total_points = pt_off; out.total_prims = prim_off;
    for_blocks([&](int64_t b) {
        const PlanBlock &blk = out.blocks[(size_t)b];
        for (int64_t fi = blk.f0; fi < blk.f1; ++fi) {
            out.info[(size_t)fi].point_offset += blk.point_base;
            if (want_device) { out.fields[(size_t)fi].pt_off += blk.point_base; out.fields[(size_t)fi].prim_first += (int32_t)blk.prim_base; }
        }
    });
    if (!want_device) out.blocks.clear();
    return FCPP_OK;
}

}  // namespace fcpp
