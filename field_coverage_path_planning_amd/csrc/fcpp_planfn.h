// fcpp_planfn.h -- ONE field's setup as a host+device function: TwoLayerPathPlannerV37.__init__ and the O(1)-per-field decisions
// of plan_complete_coverage() (reference: multi_layer_planner_v3.py = "MLP").
//
// Nothing here touches path points: a field becomes (a) the integer facts the reference derives with Python floats (swath count,
// loop count, start corner, pass order, reverse-fill counts), with the same float64 operation order, and (b) a closed-form device
// descriptor (DevField + a few DevPrim) from which the HIP kernels compute any path point from its index.
//
// The same source runs
//   * on the GPU, one thread per field (fcpp_devplan.hip: k_plan_fields) -- the setup of a batch at the reference's sampling, and
//   * on the host (fcpp_host.cpp): fcpp_plan_count, batches the device planner does not take (obstacle-aware swaths, dense
//     sampling), and the checker of the device-built tables (tests/test_gpu_devplan.py compares them table by table, byte for byte).
// Transcendentals come from fcpp_math.h (plain IEEE operations, bit-identical on both sides).
//
// Shapely is replaced by exact formulas for convex quadrilaterals (the only shapes the reference's generator handles meaningfully):
// mitre inset, area centroid, bounds.  GEOS-specific values are not reproducible here (DESIGN.md "parity unpinned").
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "fcpp_geom.h"
#include "fcpp_internal.h"
#include "fcpp_math.h"

namespace fcpp {

template <class T> FCPP_HD T hmin(T a, T b) { return b < a ? b : a; }
template <class T> FCPP_HD T hmax(T a, T b) { return a < b ? b : a; }

struct Quad { double x[4], y[4]; };

FCPP_HD double area_centroid(const Quad &q, double &cx, double &cy)
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
FCPP_HD void mitre_of(const Quad &q, Mitre &m)
{
    double cx, cy;
    const double sgn = area_centroid(q, cx, cy) > 0 ? 1.0 : -1.0;
    double nx[4], ny[4];
    for (int i = 0; i < 4; ++i) {
        int j = (i + 1) & 3;
        double ex = q.x[j] - q.x[i], ey = q.y[j] - q.y[i];
        double ln = fc_hypot(ex, ey);
        nx[i] = -ey / ln * sgn; ny[i] = ex / ln * sgn;
    }
    for (int i = 0; i < 4; ++i) {
        int p = (i + 3) & 3;
        m.den[i] = 1.0 + (nx[p] * nx[i] + ny[p] * ny[i]);
        m.sx[i] = nx[p] + nx[i]; m.sy[i] = ny[p] + ny[i];
    }
}
// false = empty
FCPP_HD bool inset(const Quad &q, const Mitre &m, double d, Quad &o)
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

FCPP_HD double abs_area(const Quad &q) { double cx, cy; return fabs(area_centroid(q, cx, cy)); }

FCPP_HD bool is_convex(const Quad &q)
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
FCPP_HD double corner_angle(const Quad &q, int i)
{
    int p = (i + 3) & 3, n = (i + 1) & 3;
    double v1x = q.x[p] - q.x[i], v1y = q.y[p] - q.y[i];
    double v2x = q.x[n] - q.x[i], v2y = q.y[n] - q.y[i];
    double c = (v1x * v2x + v1y * v2y) / (sqrt(v1x * v1x + v1y * v1y) * sqrt(v2x * v2x + v2y * v2y));
    c = hmin(1.0, hmax(-1.0, c));
    return fc_acos(c) * (180.0 / kPi);
}

// MLP:194-222
FCPP_HD bool is_parallelogram(const Quad &q)
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
FCPP_HD void rotate_point(double x, double y, double ca, double sa, double cx, double cy, double &ox, double &oy)
{
    x -= cx; y -= cy;
    double xn = x * ca - y * sa;
    double yn = x * sa + y * ca;
    ox = xn + cx; oy = yn + cy;
}

// (saturates at 2^40 points -- far beyond any size check of the callers -- instead of converting an out-of-range double)
constexpr int64_t kCountCap = (int64_t)1 << 40;
FCPP_HD int64_t n_for_length(double len, double ds)
{
    const double c = ceil(len / ds);
    if (!(c < (double)kCountCap)) return kCountCap;
    int64_t n = (int64_t)c + 1;
    return n < 2 ? 2 : n;
}

FCPP_HD double lin_step(double a, double b, int64_t n) { return n > 1 ? (b - a) / (double)(n - 1) : 0.0; }

// MLP:1220-1288: distance along (dx,dy) to the bbox-at-origin boundary, capped at 3R, default 2R
FCPP_HD double distance_to_boundary(double x, double y, double dx, double dy, double L, double H, double R)
{
    double best = 0; bool have = false;
#define FCPP_TAKE(expr) do { const double t_ = (expr); if (t_ > 0 && (!have || t_ < best)) { best = t_; have = true; } } while (0)
    if (fabs(dx) > 1e-6) { FCPP_TAKE((0 - x) / dx); FCPP_TAKE((L - x) / dx); }
    if (fabs(dy) > 1e-6) { FCPP_TAKE((0 - y) / dy); FCPP_TAKE((H - y) / dy); }
#undef FCPP_TAKE
    if (!have) return 2.0 * R;
    return hmin(best, 3.0 * R);
}

// a straight primitive whose numpy.linspace step underflowed to 0 although its ends differ (no real field has one): form bit 3 sends the
// one-point-per-lane kernel through the general evaluation (fcpp_pointfn.h: eval_prim_lanes)
FCPP_HD void flag_degenerate(DevPrim &p)
{
    if (p.kind == PRIM_LINSPACE && ((p.a[4] == 0.0 && p.a[2] != p.a[0]) || (p.a[5] == 0.0 && p.a[3] != p.a[1]))) p.form |= 8;
    if (p.kind == PRIM_RAY && p.a[5] == 0.0 && p.a[4] != 0.0) p.form |= 8;
}

FCPP_HD int corner_quadrant(int ci) { return (ci + 1) & 3; }   // start heading of the corner arcs = q * pi/2 (MLP:1049-1060): 1, 2, 3, 0

// everything about a batch that does not depend on the field: validated parameters, turn shapes, sample counts.  Plain data: the
// device planner takes it as a kernel argument.
struct PlanConsts {
    fcpp_vehicle veh;
    fcpp_options opt;
    double W, R, ds;
    int32_t clip, cloth;
    double Re_pi, Re_half, len_uturn, len_corner, gap_lb;
    double uturn_dx, uturn_h;     // extent of a U-turn beyond its line end along the line / above the line (obstacle-aware swaths: its zone)
    double gap_area;                  // area of the corner gap (MLP:1086-1152) where the lower bound does not decide; else gap_lb
    int32_t gap_decision, _pad0;      // `gap.area > 0.1` (MLP:1070): 1 yes, 0 no, -1 within what GEOS' polygonal buffer leaves open
    double turn_end_pi;               // U-turn: arcs pi, clothoid total length
    double half_T;                    // corner turn, clothoid: total length
    // the last two samples of a corner turn (the reverse fill leaves along their chord): for arcs cos and sin of their angles, for the
    // clothoid model the unit-shape points (already mirrored: turn sign -1) that cac_world_point would evaluate
    int64_t nt_corner;
    double arc_step, arc_c1, arc_s1, arc_c2, arc_s2;
    double cac_step, cac_u1x, cac_u1y, cac_u2x, cac_u2y;
    int32_t max_prims;                // primitives a field can have without obstacle-aware swaths: 8 per loop + 3 reverse fills
    int32_t _pad;
};

// a field's layer-1 frame, handed to the sink of obstacle-aware swaths (host only)
struct Layer1Frame {
    double rot, ccx, ccy, lsx, lex, min_x, max_x, min_y, max_y;
    int64_t P, n_turn;
    int32_t rotated, reverse_order, start_from_right;
};

// One field: __init__ + the O(1) decisions of plan_complete_coverage.  Fills `in` (point_offset stays 0) and `df` (pt_off 0,
// prim_first = sink.size() at entry); the field's primitives go to `sink`:
//     int64_t size() ; void push(const DevPrim &) ; void truncate(int64_t) ;
//     int clipped_layer1(const PlanConsts &, const fcpp_field &, const Layer1Frame &, int64_t &n_main)   (obstacle-aware swaths: FCPP_OK or the field's error)
//     int headland_straight(const PlanConsts &, const Quad &, const DevPrim &, int64_t &pos), bool box_meets_square(x, y, half),
//     bool box_meets_segment(ax, ay, bx, by)                                                                (obstacle-aware swaths, layer 2)
// A field that raises gets in.status < 0 and no points.  -> points of the field.
template <class Sink>
FCPP_HD int64_t plan_field_t(const PlanConsts &pc, const fcpp_field &f, fcpp_field_info &in, DevField &df, Sink &sink)
{
    const fcpp_vehicle &veh = pc.veh;
    const fcpp_options &opt = pc.opt;
    const double W = pc.W, R = pc.R, ds = pc.ds;
    const bool clip = pc.clip != 0, cloth = pc.cloth != 0;
    memset(&in, 0, sizeof(in));
    memset(&df, 0, sizeof(df));
    const int64_t prim_mark = sink.size();
#define FCPP_FAIL(code) do { in.status = (code); in.n_main = in.n_head = 0; memset(in.n_reverse, 0, sizeof(in.n_reverse)); \
                             df.n_main = df.n_total = 0; df.gen_main = 0; df.prim_first = (int32_t)prim_mark; df.prim_count = 0; \
                             sink.truncate(prim_mark); return 0; } while (0)
    Quad q;
    bool finite = true;
    for (int i = 0; i < 4; ++i) { q.x[i] = f.vx[i]; q.y[i] = f.vy[i]; finite = finite && isfinite(q.x[i]) && isfinite(q.y[i]); }
    if (!finite || !is_convex(q)) FCPP_FAIL(FCPP_EUNSUPPORTED);

    // ---- __init__ (MLP:109-135, 137-163, 310, 322-343)
    double bminx = q.x[0], bmaxx = q.x[0], bminy = q.y[0], bmaxy = q.y[0];
    for (int i = 1; i < 4; ++i) {
        bminx = hmin(bminx, q.x[i]); bmaxx = hmax(bmaxx, q.x[i]);
        bminy = hmin(bminy, q.y[i]); bmaxy = hmax(bmaxy, q.y[i]);
    }
    const double L = f.from_vertices ? (bmaxx - bminx) : q.x[1];
    const double H = f.from_vertices ? (bmaxy - bminy) : q.y[2];
    in.field_length = L; in.field_width = H;
    // (what the function reads again further down -- corner angles, the geofence edges, the field's status -- is kept in locals: read back
    // through `in` / `df` on the device every such value is a round trip to memory behind all the stores issued before it)
    bool all90 = true;
    double cang[4];
    for (int i = 0; i < 4; ++i) {
        cang[i] = corner_angle(q, i);
        in.corner_angles[i] = cang[i];
        if (!(fabs(cang[i] - 90) < 1.0)) all90 = false;
    }
    int fail_status = 0;
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
    if (!inset(q, mit, hw, mq) || abs_area(mq) < 1.0) FCPP_FAIL(FCPP_EINVAL);
    const double e0x = q.x[1] - q.x[0], e0y = q.y[1] - q.y[0];
    const double rot = (e0x == 0.0 && e0y == 0.0) ? 0.0 : fc_atan2_cr(e0y, e0x);          // (correctly rounded: fcpp_math.h, round 5)
    in.rotation_angle = rot;
    const bool rotated = fabs(rot) > 0.01;
    in.rotated = rotated;
    double rc, rs;                       // cos / sin of +rot; the frame of layer 1 is reached with -rot: (rc, -rs)
    fc_sincos_cr(rot, rs, rc);
    double ccx = 0, ccy = 0, sx = f.start_x, sy = f.start_y;
    Quad rq = mq;
    if (rotated) {
        area_centroid(mq, ccx, ccy);
        for (int i = 0; i < 4; ++i) rotate_point(mq.x[i], mq.y[i], rc, -rs, ccx, ccy, rq.x[i], rq.y[i]);
        if (has_start) rotate_point(sx, sy, rc, -rs, ccx, ccy, sx, sy);
    }
    double min_x = rq.x[0], max_x = rq.x[0], min_y = rq.y[0], max_y = rq.y[0];
    for (int i = 1; i < 4; ++i) {
        min_x = hmin(min_x, rq.x[i]); max_x = hmax(max_x, rq.x[i]);
        min_y = hmin(min_y, rq.y[i]); max_y = hmax(max_y, rq.y[i]);
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
    const int64_t n_turn = ds > 0 ? n_for_length(pc.len_uturn, ds) : 20;
    // (the swath index lives in bits 8..31 of the flag / segment word)
    if (P >= ((int64_t)1 << (32 - FCPP_INDEX_SHIFT)) || n_line + n_turn > INT32_MAX - 2 * TILE_POINTS) FCPP_FAIL(FCPP_ESIZE);
    in.n_swaths = (int32_t)P;
    int64_t n_main = P * n_line + (P - 1) * n_turn;
    df.gen_main = n_main;
    const int32_t prim_first = (int32_t)sink.size();
    df.prim_first = prim_first;
    if (clip) {
        // ---- obstacle-aware swaths (include/fcpp.h): layer 1 as a list of primitives -- sub-swaths, detour legs, U-turns
        Layer1Frame fr;
        fr.rot = rot; fr.ccx = ccx; fr.ccy = ccy; fr.lsx = lsx; fr.lex = lex; fr.min_x = min_x; fr.max_x = max_x; fr.min_y = min_y; fr.max_y = max_y;
        fr.P = P; fr.n_turn = n_turn; fr.rotated = rotated; fr.reverse_order = reverse_order; fr.start_from_right = start_from_right;
        const int rcode = sink.clipped_layer1(pc, f, fr, n_main);
        if (rcode != FCPP_OK) FCPP_FAIL(rcode);
        df.gen_main = 0;
    }
    in.n_main = n_main;

    df.n_main = n_main;
    df.lsx = lsx; df.lex = lex; df.line_step = lin_step(lsx, lex, n_line);
    df.min_x = min_x; df.max_x = max_x; df.min_y = min_y; df.W = W; df.R = R;
    df.turn_end = pc.turn_end_pi;
    df.turn_step = lin_step(0.0, df.turn_end, n_turn);
    df.turn_Re = pc.Re_pi;
    df.rot_cos = rc; df.rot_sin = rs; df.rot_cx = ccx; df.rot_cy = ccy;
    df.v_work = veh.max_work_speed_kmh; df.v_turn = veh.headland_turn_speed_kmh;
    df.P = (int32_t)P; df.n_line = (int32_t)n_line; df.n_turn = (int32_t)n_turn;
    df.reverse_order = reverse_order; df.start_from_right = start_from_right; df.rotated = rotated;
    df.turn_model = opt.turn_model;

    double gex[4], gey[4], geo[4];
    {   // geofence half-planes: inside <=> ex*px + ey*py + eo >= -tol   (before layer 2: the sink of the device planner tests the primitives it
        // is handed against them, fcpp_cutfn.h)
        double cx, cy;
        const double sgn = area_centroid(q, cx, cy) > 0 ? 1.0 : -1.0;
        for (int i = 0; i < 4; ++i) {
            int j = (i + 1) & 3;
            double ex = q.x[j] - q.x[i], ey = q.y[j] - q.y[i];
            double ln = sqrt(ex * ex + ey * ey);
            gex[i] = -ey / ln * sgn; gey[i] = ex / ln * sgn;
            geo[i] = -(gex[i] * q.x[i] + gey[i] * q.y[i]);
            df.ex[i] = gex[i]; df.ey[i] = gey[i]; df.eo[i] = geo[i];
        }
    }

    // ---- layer 2 (MLP:898-1084)
    const int num_loops = (int)ceil(hw / W);
    in.n_loops = num_loops;
    int64_t pos = n_main;
    bool bad = false;
    double first_head[2] = { 0, 0 }, last_head[2] = { 0, 0 };
#define FCPP_PUSH(p) do { (p).start = pos; pos += (p).n; flag_degenerate(p); sink.push(p); } while (0)
    for (int loop = 0; loop < num_loops && !bad; ++loop) {
        const double offset = W / 2 + loop * W;
        Quad c;
        if (!inset(q, mit, offset, c) || abs_area(c) < 1.0) { bad = true; break; }
        if (opt.ring_order == FCPP_RING_REVERSED) {      // ring lists 0, 3, 2, 1
            double t = c.x[1]; c.x[1] = c.x[3]; c.x[3] = t;
            t = c.y[1]; c.y[1] = c.y[3]; c.y[3] = t;
        }
        const uint32_t lp = FCPP_FLAG_HEADLAND | ((uint32_t)(loop * 8) << FCPP_INDEX_SHIFT);
        DevPrim p;
        memset(&p, 0, sizeof(p));
        p.kind = PRIM_POINT; p.n = 1; p.v_nom = veh.max_headland_speed_kmh;
        p.fs = FCPP_KIND_HEAD_START | lp | ((uint32_t)sci << FCPP_INDEX_SHIFT);
        p.a[0] = c.x[sci]; p.a[1] = c.y[sci];
        FCPP_PUSH(p);
        if (loop == 0) { first_head[0] = c.x[sci]; first_head[1] = c.y[sci]; }
        for (int i = 0; i < 4; ++i) {
            const int cur = (sci + i) & 3, nxt = (sci + i + 1) & 3;
            int64_t ns = 20;
            if (ds > 0) ns = n_for_length(fc_hypot(c.x[nxt] - c.x[cur], c.y[nxt] - c.y[cur]), ds);
            const int64_t nt = pc.nt_corner;
            if (ns > INT32_MAX || nt > INT32_MAX) { bad = true; break; }
            memset(&p, 0, sizeof(p));
            p.kind = PRIM_LINSPACE; p.n = (int32_t)ns; p.v_nom = veh.max_headland_speed_kmh;
            p.fs = FCPP_KIND_HEAD_STRAIGHT | lp | ((uint32_t)cur << FCPP_INDEX_SHIFT);
            p.a[0] = c.x[cur]; p.a[1] = c.y[cur]; p.a[2] = c.x[nxt]; p.a[3] = c.y[nxt];
            p.a[4] = lin_step(c.x[cur], c.x[nxt], ns); p.a[5] = lin_step(c.y[cur], c.y[nxt], ns);
            if (clip) {       // obstacle-aware: the straight is led around the boxes it crosses (include/fcpp.h); the turn at its end must be free
                const int rcode = sink.headland_straight(pc, q, p, pos);
                if (rcode == FCPP_OK && i < 3 && sink.box_meets_square(c.x[nxt], c.y[nxt], 2.0 * R)) { bad = true; fail_status = FCPP_EUNSUPPORTED; break; }
                if (rcode != FCPP_OK) { bad = true; fail_status = rcode; break; }
            } else FCPP_PUSH(p);
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
                const int qd = corner_quadrant(nxt);
                p.kind = PRIM_CAC; p.form = qd;
                p.a[0] = c.x[nxt]; p.a[1] = c.y[nxt]; p.a[2] = qd * kHalfPi; p.a[3] = -kHalfPi;
                p.a[4] = pc.Re_half; p.a[5] = pc.cac_step; p.a[6] = pc.half_T;
                cac_world_from_unit(pc.cac_u1x, pc.cac_u1y, c.x[nxt], c.y[nxt], qd, pc.Re_half, e1[0], e1[1]);
                cac_world_from_unit(pc.cac_u2x, pc.cac_u2y, c.x[nxt], c.y[nxt], qd, pc.Re_half, e2[0], e2[1]);
            }
            FCPP_PUSH(p);
            // reverse fill (MLP:1043, 224-242, 1066-1082, 1154-1218)
            const double ang_nxt = nxt == 0 ? cang[0] : (nxt == 1 ? cang[1] : (nxt == 2 ? cang[2] : cang[3]));
            const bool want_rev = (loop == 0) && (ang_nxt >= 60);       // MLP:1043
            if (want_rev && pc.gap_decision < 0) { bad = true; fail_status = FCPP_EUNSUPPORTED; break; }
            const bool add_rev = want_rev && pc.gap_decision > 0;                     // MLP:1070: gap.area > 0.1
            if (add_rev) {
                const double tx = e1[0] - e2[0], ty = e1[1] - e2[1];
                const double nrm = sqrt(tx * tx + ty * ty);
                double dx = -1.0, dy = 0.0;
                if (nrm > 1e-6) { dx = -tx / nrm; dy = -ty / nrm; }
                const double len = distance_to_boundary(e1[0], e1[1], dx, dy, L, H, R);
                if (clip && sink.box_meets_segment(e1[0], e1[1], e1[0] + len * dx, e1[1] + len * dy)) { bad = true; fail_status = FCPP_EUNSUPPORTED; break; }
                int64_t nr;
                if (ds > 0) nr = n_for_length(len, ds);
                else { nr = (int64_t)(len / 0.5); if (nr < 10) nr = 10; }
                memset(&p, 0, sizeof(p));
                p.kind = PRIM_RAY; p.n = (int32_t)nr; p.v_nom = 2.5;   // MLP:1080
                p.fs = FCPP_KIND_REVERSE | lp | ((uint32_t)nxt << FCPP_INDEX_SHIFT);
                p.a[0] = e1[0]; p.a[1] = e1[1]; p.a[2] = dx; p.a[3] = dy; p.a[4] = len;
                p.a[5] = lin_step(0.0, len, nr);
                FCPP_PUSH(p);
                in.n_reverse[nxt] = (int32_t)nr;
            }
        }
    }
#undef FCPP_PUSH
    if (bad) FCPP_FAIL(fail_status ? fail_status : FCPP_EHEADLAND);
    in.n_head = pos - n_main;
    if (has_start) {   // MLP:437-441
        in.approach_from[0] = f.start_x; in.approach_from[1] = f.start_y;
        in.approach_to[0] = first_head[0]; in.approach_to[1] = first_head[1];
    }
    if (has_end) {     // MLP:443-447
        in.departure_from[0] = last_head[0]; in.departure_from[1] = last_head[1];
        in.departure_to[0] = f.end_x; in.departure_to[1] = f.end_y;
    }
    df.n_total = pos;
    df.prim_count = (int32_t)(sink.size() - prim_first);
    df.obs_first = (int32_t)f.obstacle_first; df.obs_count = f.n_obstacles;
    if (!clip && lsx < lex) {
        // Do layer 1's lines and U-turns all lie inside the geofence?  Their bounding box in the frame -- the lines' ends plus a U-turn's
        // extent beyond them on either side, the passes' heights plus a U-turn's height (clothoid turns: 0.1 % of R for what the extents'
        // sampling may have missed) -- is convex like the field: inside iff its four corners are, with the tiler's margin (fcpp_tilefn.h: tiler_inside).
        const double slack = cloth ? 1e-3 * R : 0.0;       // (the reference's half circle: 2 R and R exactly -- its far end lies ON the field's edge when the headland is R wide)
        const double ext = pc.uturn_dx + slack, hgt = pc.uturn_h + slack;
        const double bx[2] = { lsx - ext, lex + ext }, by[2] = { min_y, min_y + (double)(P - 1) * W + hgt };
        const double margin = 1e-7 - opt.geofence_tol;
        bool in = true;
        for (int cxi = 0; cxi < 2; ++cxi)
            for (int cyi = 0; cyi < 2; ++cyi) {
                double px = bx[cxi], py = by[cyi];
                if (rotated) { const double tx = px - ccx, ty = py - ccy; px = (tx * rc - ty * rs) + ccx; py = (tx * rs + ty * rc) + ccy; }
                const double m = margin + 5.684341886080802e-14 * (fabs(px) + fabs(py));      // (256 ulps of the coordinates)
                for (int e = 0; e < 4; ++e)
                    if (!(gex[e] * px + gey[e] * py + geo[e] >= m)) in = false;
            }
        df.span_inside = in ? 1 : 0;
    }
    return pos;
#undef FCPP_FAIL
}

}  // namespace fcpp

#include <string>
namespace fcpp {
// validates vehicle and options, fills the batch constants and the turn templates' description (fcpp_host.cpp)
int plan_prepare(const fcpp_vehicle &veh, const fcpp_options &opt, PlanConsts &c, TurnTemplates &tt, std::string &err);
}  // namespace fcpp
