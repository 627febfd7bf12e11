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
#include <mutex>
#include <utility>
#include <new>
#include <vector>

#include "fcpp_geom.h"
#include "fcpp_internal.h"
#include "fcpp_parallel.h"
#include "fcpp_planfn.h"

namespace fcpp {

namespace {
// Area of { p in [0, 2R]^2 : dist(p, polyline) <= W/2 } for the 30-point quarter arc (0,0) -> (R, R) of MLP:1124-1141 (corner 0; the other
// corners are its mirror images): column by column -- a vertical line meets every capsule (segment + radius) in one interval, the
// intervals of a column are merged and clipped to the square -- with the midpoint rule over 32768 columns (error far below 1e-6 m^2).
double corner_cover_area(double R, double W)
{
    static std::mutex mu;
    static double last_R[2] = { -1.0, -1.0 }, last_W[2] = { -1.0, -1.0 }, last_area[2] = { 0.0, 0.0 };
    static int next = 0;
    {
        std::lock_guard<std::mutex> lk(mu);
        for (int q = 0; q < 2; ++q) if (R == last_R[q] && W == last_W[q]) return last_area[q];
    }
    const int NP = 30, NCOL = 32768;
    const double r = W / 2, side = 2 * R, dx = side / NCOL;
    double px[NP], py[NP];
    for (int k = 0; k < NP; ++k) { const double th = kHalfPi * k / (NP - 1); px[k] = R * (1 - cos(th)); py[k] = R * sin(th); }
    double area = 0.0;
    std::pair<double, double> iv[NP];
    for (int j = 0; j < NCOL; ++j) {
        const double x = (j + 0.5) * dx;
        int ni = 0;
        for (int k = 0; k + 1 < NP; ++k) {
            double lo = HUGE_VAL, hi = -HUGE_VAL;
            const double ax = px[k], ay = py[k], bx = px[k + 1], by = py[k + 1];
            for (int e = 0; e < 2; ++e) {            // the two end discs
                const double cx = e ? bx : ax, cy = e ? by : ay, h2 = r * r - (x - cx) * (x - cx);
                if (h2 >= 0) { const double h = sqrt(h2); lo = std::min(lo, cy - h); hi = std::max(hi, cy + h); }
            }
            // the rectangle between them: 0 <= t <= 1 and |n| <= r, both linear in y
            const double ex = bx - ax, ey = by - ay, len = sqrt(ex * ex + ey * ey);
            if (len > 0) {
                double l2 = -HUGE_VAL, h2 = HUGE_VAL;
                auto clip = [&](double a0, double a1, double lo_v, double hi_v) {      // lo_v <= a0 + a1 y <= hi_v
                    if (a1 == 0) { if (a0 < lo_v || a0 > hi_v) { l2 = HUGE_VAL; h2 = -HUGE_VAL; } return; }
                    double y0 = (lo_v - a0) / a1, y1 = (hi_v - a0) / a1;
                    if (y0 > y1) std::swap(y0, y1);
                    l2 = std::max(l2, y0); h2 = std::min(h2, y1);
                };
                clip(((x - ax) * ex - ay * ey) / (len * len), ey / (len * len), 0.0, 1.0);      // t(y)
                clip((-(x - ax) * ey - ay * ex) / len, ex / len, -r, r);                         // n(y)
                if (l2 <= h2) { lo = std::min(lo, l2); hi = std::max(hi, h2); }
            }
            lo = std::max(lo, 0.0); hi = std::min(hi, side);
            if (lo < hi) iv[ni++] = { lo, hi };
        }
        std::sort(iv, iv + ni);
        double covered = 0.0, cur_lo = 0.0, cur_hi = -1.0;
        for (int q = 0; q < ni; ++q) {
            if (cur_hi < cur_lo) { cur_lo = iv[q].first; cur_hi = iv[q].second; }
            else if (iv[q].first <= cur_hi) cur_hi = std::max(cur_hi, iv[q].second);
            else { covered += cur_hi - cur_lo; cur_lo = iv[q].first; cur_hi = iv[q].second; }
        }
        if (cur_hi >= cur_lo) covered += cur_hi - cur_lo;
        area += covered * dx;
    }
    std::lock_guard<std::mutex> lk(mu);
    last_R[next] = R; last_W[next] = W; last_area[next] = area;
    next ^= 1;
    return area;
}
}  // namespace

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
    // (the same vehicle and options as the last call of this thread -- a caller that plans batch after batch: the constants below, with
    // the clothoid's 257-sample extents and its Fresnel evaluations, cost 5-10 us, and a plan call asks for them twice)
    struct Prepared { fcpp_vehicle veh; fcpp_options opt; PlanConsts c; TurnTemplates tt; bool valid = false; };
    static thread_local Prepared last;
    if (last.valid && memcmp(&last.veh, &veh, sizeof veh) == 0 && memcmp(&last.opt, &opt, sizeof opt) == 0) { c = last.c; tt = last.tt; return FCPP_OK; }
    last.valid = false;
    memset(&c, 0, sizeof c);
    c.veh = veh; c.opt = opt; c.W = W; c.R = R; c.ds = ds;
    c.clip = opt.obstacle_mode == FCPP_OBSTACLES_AVOID;
    c.cloth = opt.turn_model == FCPP_TURN_CLOTHOID;
    const CacShape sh_pi = make_cac_shape(kPi, opt.clothoid_frac), sh_half = make_cac_shape(kHalfPi, opt.clothoid_frac);
    auto fit_radius = [&](const CacShape &sh) {
        if (!opt.clothoid_fit) return R;
        return R * (2 * sin(sh.D / 2)) / sqrt(sh.ex * sh.ex + sh.ey * sh.ey);
    };
    c.Re_pi = c.cloth ? fit_radius(sh_pi) : R; c.Re_half = c.cloth ? fit_radius(sh_half) : R;
    c.len_uturn = c.cloth ? sh_pi.T * c.Re_pi : kPi * R;
    c.len_corner = c.cloth ? sh_half.T * c.Re_half : kHalfPi * R;
    c.turn_end_pi = c.cloth ? sh_pi.T * c.Re_pi : kPi;
    c.half_T = sh_half.T * c.Re_half;
    // gap.area > 0.1 (MLP:1070): 2R x 2R square minus the arc buffered by W/2.  The buffer's area is at
    // most (pi R/2) W + pi W^2/4, so the decision is certain when this lower bound exceeds 0.1.
    c.gap_lb = 4 * R * R - (kPi * R / 2 * W + kPi * W * W / 4);
    // Where the bound does not decide (wide implements on a tight radius: W >~ 1.6 R) the area itself does: the gap is the square minus
    // the 30-point arc buffered by W/2, the same for every corner of every field (the formulas are axis-aligned, MLP:1101-1148) -- its
    // exact area by integration, once per (R, W).  GEOS buffers with polygonal round parts, inscribed in the exact ones: with 8 segments
    // per quarter circle or more (Shapely's default is 16) its buffer contains the exact buffer of radius (W/2) cos(pi/32), so
    //     gap(W/2)  <=  GEOS' gap  <=  gap((W/2) cos(pi/32))
    // and `gap.area > 0.1` (MLP:1070) is certain unless 0.1 lies between the two (FCPP_EUNSUPPORTED then, as before for the whole regime).
    c.gap_area = c.gap_lb;
    c.gap_decision = 1;
    if (!(c.gap_lb > 0.1)) {
        const double gap_lo = 4 * R * R - corner_cover_area(R, W), gap_hi = 4 * R * R - corner_cover_area(R, W * 0.99518472667219693);
        c.gap_area = gap_lo;
        c.gap_decision = gap_lo > 0.1 + 1e-6 ? 1 : (gap_hi < 0.1 - 1e-6 ? 0 : -1);
    }
    c.nt_corner = ds > 0 ? n_for_length(c.len_corner, ds) : 15;
    c.arc_step = lin_step(0.0, kHalfPi, c.nt_corner);
    {
        const double th1 = linspace_at(0.0, kHalfPi, c.arc_step, c.nt_corner, c.nt_corner - 1);
        const double th2 = linspace_at(0.0, kHalfPi, c.arc_step, c.nt_corner, c.nt_corner - 2);
        c.arc_c1 = cos(th1); c.arc_s1 = sin(th1); c.arc_c2 = cos(th2); c.arc_s2 = sin(th2);
    }
    {   // the clothoid corner turn's last two samples as unit-shape points (turn sign -1): the same for every field
        c.cac_step = lin_step(0.0, c.half_T, c.nt_corner);
        const double s1 = linspace_at(0.0, c.half_T, c.cac_step, c.nt_corner, c.nt_corner - 1);
        const double s2 = linspace_at(0.0, c.half_T, c.cac_step, c.nt_corner, c.nt_corner - 2);
        cac_unit_point(sh_half, s1 / c.Re_half, c.cac_u1x, c.cac_u1y);
        cac_unit_point(sh_half, s2 / c.Re_half, c.cac_u2x, c.cac_u2y);
        c.cac_u1y *= -1.0; c.cac_u2y *= -1.0;
    }
    {
        const double nl = ceil(R / W);
        c.max_prims = nl < 1e6 ? (int32_t)(8 * (int64_t)nl + 3) : INT32_MAX;
    }
    // turn templates (same sample counts as every field computes)
    memset(&tt, 0, sizeof(tt));
    tt.turn_model = opt.turn_model; tt.R = R;
    tt.nu = (int32_t)std::min<int64_t>(ds > 0 ? n_for_length(c.len_uturn, ds) : 20, INT32_MAX);
    tt.nc = (int32_t)std::min<int64_t>(ds > 0 ? n_for_length(c.len_corner, ds) : 15, INT32_MAX);
    tt.u_end = c.cloth ? sh_pi.T * c.Re_pi : kPi; tt.u_step = lin_step(0.0, tt.u_end, tt.nu); tt.u_Re = c.Re_pi;
    // the zone a U-turn occupies beyond the end of its line (the reference's half circle about (max_x, y): 2 R along the line, R above
    // it; the clothoid turn: its shape's extents, sampled at 257 parameter values as the oracle samples its own)
    c.uturn_dx = 2.0 * R; c.uturn_h = R;
    if (c.cloth) {
        double mx = 0.0, my = 0.0;
        for (int k = 0; k <= 256; ++k) {
            double X, Y;
            cac_unit_point(sh_pi, sh_pi.T * (double)k / 256.0, X, Y);
            mx = std::max(mx, X); my = std::max(my, Y);
        }
        c.uturn_dx = c.Re_pi * my; c.uturn_h = c.Re_pi * mx;
    }
    tt.c_end = c.cloth ? sh_half.T * c.Re_half : kHalfPi; tt.c_step = lin_step(0.0, tt.c_end, tt.nc); tt.c_Re = c.Re_half;
    last.veh = veh; last.opt = opt; last.c = c; last.tt = tt; last.valid = true;
    return FCPP_OK;
}

namespace {

// ---- obstacle-aware swaths, round 4: the W/2-grown POLYGON of an obstacle (frame of layer 1) -----------------------------------------
// Convex hull of the vertices (counter-clockwise), every edge moved W/2 outwards, neighbours joined at their mitre point -- a convex
// polygon that contains every point within W/2 of the hull -- and clipped to the obstacle's grown bounding box (so that it stays inside
// the box the merging rule reasons about).  Empty: no polygon (fewer than three hull vertices), the box is used.
struct GPt { double x, y; };
static std::vector<GPt> grown_polygon(std::vector<GPt> pts, double half, double x0, double y0, double x1, double y1)
{
    std::vector<GPt> none;
    std::sort(pts.begin(), pts.end(), [](const GPt &a, const GPt &b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
    const size_t n = pts.size();
    if (n < 3) return none;
    auto cross = [](const GPt &o, const GPt &a, const GPt &b) { return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x); };
    std::vector<GPt> h(2 * n);
    size_t k = 0;
    for (size_t i = 0; i < n; ++i) { while (k >= 2 && cross(h[k - 2], h[k - 1], pts[i]) <= 0) --k; h[k++] = pts[i]; }
    for (size_t i = n - 1, t = k + 1; i > 0; --i) { while (k >= t && cross(h[k - 2], h[k - 1], pts[i - 1]) <= 0) --k; h[k++] = pts[i - 1]; }
    h.resize(k - 1);
    const size_t m = h.size();
    if (m < 3) return none;
    std::vector<GPt> nrm(m), g;
    for (size_t i = 0; i < m; ++i) {
        const GPt &a = h[i], &b = h[(i + 1) % m];
        const double dx = b.x - a.x, dy = b.y - a.y, ln = sqrt(dx * dx + dy * dy);
        if (!(ln > 0)) return none;
        nrm[i] = { dy / ln, -dx / ln };                       // outward of a counter-clockwise polygon
    }
    for (size_t i = 0; i < m; ++i) {
        const GPt &n0 = nrm[(i + m - 1) % m], &n1 = nrm[i];
        const double den = 1.0 + (n0.x * n1.x + n0.y * n1.y);
        if (den >= 0.5) g.push_back({ h[i].x + half * (n0.x + n1.x) / den, h[i].y + half * (n0.y + n1.y) / den });     // mitre (turn <= 120 degrees)
        else {
            // a sharp vertex (a needle's tip: the mitre point runs away and loses its digits): a square cap -- the incoming offset
            // line carried `half` beyond the vertex, the outgoing one begun `half` before it; the chord between them stays >= half
            // away from the vertex for every turn up to 180 degrees.  (edge direction = outward normal turned by +90 degrees)
            g.push_back({ h[i].x + half * (n0.x - n0.y), h[i].y + half * (n0.y + n0.x) });
            g.push_back({ h[i].x + half * (n1.x + n1.y), h[i].y + half * (n1.y - n1.x) });
        }
    }
    // Sutherland-Hodgman against x >= x0, x <= x1, y >= y0, y <= y1
    for (int side = 0; side < 4; ++side) {
        std::vector<GPt> out;
        // (a vertex ON the box -- every vertex of an axis-parallel rectangle's polygon -- is inside whichever way its last bit fell)
        auto inside = [&](const GPt &p) { return side == 0 ? p.x >= x0 - 1e-9 : (side == 1 ? p.x <= x1 + 1e-9 : (side == 2 ? p.y >= y0 - 1e-9 : p.y <= y1 + 1e-9)); };
        auto cut = [&](const GPt &a, const GPt &b) {
            GPt r;
            if (side < 2) { const double xc = side == 0 ? x0 : x1; r.x = xc; r.y = a.y + (b.y - a.y) * ((xc - a.x) / (b.x - a.x)); }
            else { const double yc = side == 2 ? y0 : y1; r.y = yc; r.x = a.x + (b.x - a.x) * ((yc - a.y) / (b.y - a.y)); }
            return r;
        };
        for (size_t i = 0; i < g.size(); ++i) {
            const GPt &a = g[i], &b = g[(i + 1) % g.size()];
            const bool ia = inside(a), ib = inside(b);
            if (ia) out.push_back(a);
            if (ia != ib) out.push_back(cut(a, b));
        }
        g.swap(out);
        if (g.size() < 3) return none;
    }
    // vertices closer than 1e-7 to their predecessor are one vertex
    std::vector<GPt> u;
    for (const GPt &p : g)
        if (u.empty() || fabs(p.x - u.back().x) + fabs(p.y - u.back().y) > 1e-7) u.push_back(p);
    while (u.size() > 1 && fabs(u.front().x - u.back().x) + fabs(u.front().y - u.back().y) <= 1e-7) u.pop_back();
    if (u.size() < 3) return none;
    return u;
}
// The way around a grown polygon g (counter-clockwise, convex) for a line at height y travelling left or right: where the line meets
// the polygon (false: it passes clear of it) and the vertices of the upper / lower chain from the near point to the far point.
static bool polygon_chains(const std::vector<GPt> &g, double y, bool go_left, double &nearx, double &farx, std::vector<GPt> &upper, std::vector<GPt> &lower)
{
    const size_t m = g.size();
    int eu = -1, ed = -1;            // the edge that crosses the line going up (on the right of a ccw polygon) / going down (on the left)
    double xu = 0, xd = 0;
    for (size_t j = 0; j < m; ++j) {
        const GPt &a = g[j], &b = g[(j + 1) % m];
        if ((a.y < y) == (b.y < y)) continue;
        const double x = a.x + (b.x - a.x) * ((y - a.y) / (b.y - a.y));
        if (a.y < y) { eu = (int)j; xu = x; } else { ed = (int)j; xd = x; }
    }
    if (eu < 0 || ed < 0) return false;
    // counter-clockwise from the up-crossing (right side, x = xu): the upper vertices g[eu + 1 .. ed], then from the down-crossing (left
    // side, x = xd) the lower vertices g[ed + 1 .. eu]
    // (scratch of the thread, kept between calls: a 5000 x 2000 m field with 32 obstacles asks ~1000 times, and four vectors growing an element
    // at a time were ~6000 allocations of its plan)
    static thread_local std::vector<GPt> up_ccw, lo_ccw;
    up_ccw.clear(); lo_ccw.clear();
    for (size_t j = ((size_t)eu + 1) % m;; j = (j + 1) % m) { up_ccw.push_back(g[j]); if (j == (size_t)ed) break; }
    for (size_t j = ((size_t)ed + 1) % m;; j = (j + 1) % m) { lo_ccw.push_back(g[j]); if (j == (size_t)eu) break; }
    upper.clear(); lower.clear();
    if (go_left) {                   // from the right (xu) to the left (xd): upper chain counter-clockwise, lower chain clockwise
        nearx = xu; farx = xd;
        upper = up_ccw;
        lower.assign(lo_ccw.rbegin(), lo_ccw.rend());
    } else {
        nearx = xd; farx = xu;
        upper.assign(up_ccw.rbegin(), up_ccw.rend());
        lower = lo_ccw;
    }
    return true;
}

// the host's primitive sink: a block's list; with obstacle-aware swaths also the field's obstacle polygons
struct HostSink {
    std::vector<DevPrim> *prims;      // NULL: sizes only (fcpp_plan_count)
    const fcpp_polys *polys;
    int64_t n = 0;                    // primitives so far (counted also without a list)
    int64_t size() const { return n; }
    void push(const DevPrim &p) { if (prims) prims->push_back(p); ++n; }
    void truncate(int64_t m) { if (prims) prims->resize((size_t)m); n = m; }

    // obstacle-aware swaths: the field's merged grown boxes in the frame of layer 1 and that frame (set by clipped_layer1, used by layer 2)
    struct Box { double x0, y0, x1, y1; };
    std::vector<Box> boxes;
    std::vector<std::vector<GPt>> gpoly;      // per box: the grown polygon of its obstacle (empty: merged boxes, or no polygon)
    bool fr_rotated = false;
    double fr_c = 1.0, fr_s = 0.0, fr_cx = 0.0, fr_cy = 0.0;
    void to_frame(double &x, double &y) const { if (fr_rotated) rotate_point(x, y, fr_c, -fr_s, fr_cx, fr_cy, x, y); }
    void to_world(double &x, double &y) const { if (fr_rotated) rotate_point(x, y, fr_c, fr_s, fr_cx, fr_cy, x, y); }

    // parameter range [t0, t1] of the frame segment a + t d, t in [0, 1], inside box b, and the faces it enters / leaves through
    // (0: x0, 1: x1, 2: y0, 3: y1); false: no proper crossing (touching does not count)
    static bool seg_box(const Box &b, double ax, double ay, double dx, double dy, double &t0, double &t1, int &f0, int &f1)
    {
        t0 = 0.0; t1 = 1.0; f0 = f1 = -1;
        const double a[2] = { ax, ay }, d[2] = { dx, dy }, lo[2] = { b.x0, b.y0 }, hi[2] = { b.x1, b.y1 };
        for (int k = 0; k < 2; ++k) {
            if (fabs(d[k]) < 1e-300) { if (!(a[k] > lo[k] && a[k] < hi[k])) return false; continue; }
            double ta = (lo[k] - a[k]) / d[k], tb = (hi[k] - a[k]) / d[k];
            int fa = 2 * k, fb = 2 * k + 1;
            if (ta > tb) { std::swap(ta, tb); std::swap(fa, fb); }
            if (ta > t0) { t0 = ta; f0 = fa; }
            if (tb < t1) { t1 = tb; f1 = fb; }
        }
        return t1 - t0 > 1e-12;
    }
    bool box_meets_square(double wx, double wy, double half) const      // a world point's square of half-size `half` in the frame
    {
        to_frame(wx, wy);
        for (const Box &b : boxes)
            if (b.x0 < wx + half && b.x1 > wx - half && b.y0 < wy + half && b.y1 > wy - half) return true;
        return false;
    }
    bool box_meets_segment(double ax, double ay, double bx, double by) const
    {
        to_frame(ax, ay); to_frame(bx, by);
        double t0, t1; int f0, f1;
        for (const Box &b : boxes)
            if (seg_box(b, ax, ay, bx - ax, by - ay, t0, t1, f0, f1)) return true;
        return false;
    }

    // Round 4: a headland straight (proto: the straight as the plain mode pushes it, a[0..3] = its end points) that crosses grown boxes
    // is cut at every box and led around it along the box's boundary -- the shorter way whose corners stay at least W/2 inside the
    // field -- as detour legs; the pieces of the straight keep its sample density.  A box over an end of the straight (where the corner
    // turns are) or one with no way around inside the field refuses the field.
    int headland_straight(const PlanConsts &pc, const Quad &q, const DevPrim &proto, int64_t &pos)
    {
        auto push1 = [&](DevPrim &pr) { pr.start = pos; pos += pr.n; flag_degenerate(pr); push(pr); };
        if (boxes.empty()) { DevPrim pr = proto; push1(pr); return FCPP_OK; }
        const double W = pc.W, ds = pc.ds;
        double ax = proto.a[0], ay = proto.a[1], bx = proto.a[2], by = proto.a[3];
        const double len_total = sqrt((bx - ax) * (bx - ax) + (by - ay) * (by - ay));
        to_frame(ax, ay); to_frame(bx, by);
        const double dx = bx - ax, dy = by - ay;
        struct Hit { double t0, t1; int f0, f1, box; };
        std::vector<Hit> hits;
        for (size_t k = 0; k < boxes.size(); ++k) {
            Hit h;
            if (seg_box(boxes[k], ax, ay, dx, dy, h.t0, h.t1, h.f0, h.f1)) { h.box = (int)k; hits.push_back(h); }
        }
        if (hits.empty()) { DevPrim pr = proto; push1(pr); return FCPP_OK; }
        std::sort(hits.begin(), hits.end(), [](const Hit &a, const Hit &b) { return a.t0 < b.t0; });
        // the field polygon's inward edge normals (world), for "at least W/2 inside"
        double cx, cy;
        const double sgn = area_centroid(q, cx, cy) > 0 ? 1.0 : -1.0;
        auto inside_margin = [&](double wx, double wy) {
            for (int i = 0; i < 4; ++i) {
                const int j = (i + 1) & 3;
                const double ex = q.x[j] - q.x[i], ey = q.y[j] - q.y[i], ln = sqrt(ex * ex + ey * ey);
                if (((-ey / ln * sgn) * (wx - q.x[i]) + (ex / ln * sgn) * (wy - q.y[i])) < W / 2 - 1e-6) return false;
            }
            return true;
        };
        bool fail = false;
        const double step0 = len_total / 19.0;
        auto push_line = [&](double x0, double y0, double x1, double y1, bool detour) {
            const double len = sqrt((x1 - x0) * (x1 - x0) + (y1 - y0) * (y1 - y0));
            int64_t np;
            if (ds > 0) np = n_for_length(len, ds);
            else np = std::max<int64_t>(2, (int64_t)(len / (detour ? 0.5 : step0)) + 1);
            if (np > INT32_MAX) { fail = true; return; }
            to_world(x0, y0); to_world(x1, y1);
            DevPrim pr = proto;
            pr.kind = PRIM_LINSPACE; pr.n = (int32_t)np;
            if (detour) { pr.v_nom = pc.veh.headland_turn_speed_kmh; pr.fs = (proto.fs & ~(uint32_t)FCPP_KIND_MASK) | FCPP_KIND_DETOUR; }
            pr.a[0] = x0; pr.a[1] = y0; pr.a[2] = x1; pr.a[3] = y1;
            pr.a[4] = lin_step(x0, x1, np); pr.a[5] = lin_step(y0, y1, np);
            push1(pr);
        };
        double px = ax, py = ay;
        for (const Hit &h : hits) {
            if (!(h.t0 > 1e-9 && h.t1 < 1.0 - 1e-9) || h.f0 < 0 || h.f1 < 0) return FCPP_EUNSUPPORTED;
            const Box &b = boxes[(size_t)h.box];
            double e0x = ax + h.t0 * dx, e0y = ay + h.t0 * dy, e1x = ax + h.t1 * dx, e1y = ay + h.t1 * dy;
            auto snap = [&](int face, double &x, double &y) {
                if (face == 0) x = b.x0; else if (face == 1) x = b.x1; else if (face == 2) y = b.y0; else y = b.y1;
                x = std::min(std::max(x, b.x0), b.x1); y = std::min(std::max(y, b.y0), b.y1);
            };
            snap(h.f0, e0x, e0y); snap(h.f1, e1x, e1y);
            // perimeter coordinate, counter-clockwise from (x0, y0): faces in that order are y0 (2), x1 (1), y1 (3), x0 (0)
            const double w = b.x1 - b.x0, hh = b.y1 - b.y0, per = 2 * (w + hh);
            auto peri = [&](int face, double x, double y) {
                return face == 2 ? x - b.x0 : (face == 1 ? w + (y - b.y0) : (face == 3 ? w + hh + (b.x1 - x) : 2 * w + hh + (b.y1 - y)));
            };
            const double s0 = peri(h.f0, e0x, e0y), s1 = peri(h.f1, e1x, e1y);
            const double cs[4] = { 0.0, w, w + hh, 2 * w + hh };                       // corners (x0,y0), (x1,y0), (x1,y1), (x0,y1)
            const double cxs[4] = { b.x0, b.x1, b.x1, b.x0 }, cys[4] = { b.y0, b.y0, b.y1, b.y1 };
            double best_len = HUGE_VAL;
            std::vector<int> best;
            bool best_ccw = true;
            for (int dir = 0; dir < 2; ++dir) {
                // corners strictly between s0 and s1 going counter-clockwise (dir 0) / clockwise (dir 1)
                const double span = dir == 0 ? fmod(s1 - s0 + per, per) : fmod(s0 - s1 + per, per);
                std::vector<std::pair<double, int>> cc;
                for (int c = 0; c < 4; ++c) {
                    const double off = dir == 0 ? fmod(cs[c] - s0 + per, per) : fmod(s0 - cs[c] + per, per);
                    if (off > 1e-9 && off < span - 1e-9) cc.push_back({ off, c });
                }
                std::sort(cc.begin(), cc.end());
                bool ok = true;
                for (auto &pc2 : cc) { double wx = cxs[pc2.second], wy = cys[pc2.second]; to_world(wx, wy); if (!inside_margin(wx, wy)) ok = false; }
                if (!ok) continue;
                if (span < best_len - 1e-9) {
                    best_len = span; best.clear();
                    for (auto &pc2 : cc) best.push_back(pc2.second);
                    best_ccw = dir == 0;
                }
            }
            (void)best_ccw;
            if (best_len == HUGE_VAL) return FCPP_EUNSUPPORTED;
            push_line(px, py, e0x, e0y, false);
            double lx = e0x, ly = e0y;
            for (int c : best) { push_line(lx, ly, cxs[c], cys[c], true); lx = cxs[c]; ly = cys[c]; }
            push_line(lx, ly, e1x, e1y, true);
            px = e1x; py = e1y;
            if (fail) return FCPP_ESIZE;
        }
        push_line(px, py, bx, by, false);
        return fail ? FCPP_ESIZE : FCPP_OK;
    }

    // obstacle-aware swaths (include/fcpp.h): layer 1 as a list of primitives -- sub-swaths, detour legs, U-turns
    int clipped_layer1(const PlanConsts &pc, const fcpp_field &f, const Layer1Frame &fr, int64_t &n_main)
    {
        const fcpp_vehicle &veh = pc.veh;
        const double W = pc.W, R = pc.R, ds = pc.ds;
        const bool cloth = pc.cloth != 0, rotated = fr.rotated != 0;
        const double rot = fr.rot, ccx = fr.ccx, ccy = fr.ccy, lsx = fr.lsx, lex = fr.lex, min_y = fr.min_y, max_y = fr.max_y;
        const int64_t P = fr.P, n_turn = fr.n_turn;
        boxes.clear(); gpoly.clear();
        double rc, rs;
        fc_sincos_cr(rot, rs, rc);
        fr_rotated = rotated; fr_c = rc; fr_s = rs; fr_cx = ccx; fr_cy = ccy;
        const double ca = rc, sa = -rs;
        bool bad_obs = f.n_obstacles < 0 || (f.n_obstacles > 0 && (!polys || f.obstacle_first < 0 || f.obstacle_first + f.n_obstacles > polys->n_polys));
        for (int k = 0; k < f.n_obstacles && !bad_obs; ++k) {
            const int64_t a0 = polys->offsets[f.obstacle_first + k], a1 = polys->offsets[f.obstacle_first + k + 1];
            if (a1 <= a0) continue;
            Box b = { HUGE_VAL, HUGE_VAL, -HUGE_VAL, -HUGE_VAL };
            std::vector<GPt> pts;
            for (int64_t q2 = a0; q2 < a1; ++q2) {
                double ox = polys->x[q2], oy = polys->y[q2];
                if (!(isfinite(ox) && isfinite(oy))) { bad_obs = true; break; }
                if (rotated) rotate_point(ox, oy, ca, sa, ccx, ccy, ox, oy);
                b.x0 = std::min(b.x0, ox); b.x1 = std::max(b.x1, ox); b.y0 = std::min(b.y0, oy); b.y1 = std::max(b.y1, oy);
                pts.push_back({ ox, oy });
            }
            b.x0 -= W / 2; b.y0 -= W / 2; b.x1 += W / 2; b.y1 += W / 2;
            boxes.push_back(b);
            gpoly.push_back(bad_obs ? std::vector<GPt>() : grown_polygon(pts, W / 2, b.x0, b.y0, b.x1, b.y1));
        }
        if (bad_obs) return FCPP_ESIZE;
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
                        gpoly[i].clear(); gpoly.erase(gpoly.begin() + (long)j);      // (merged boxes are passed as a box)
                        merged = true;
                    } else ++j;
                }
        }
        const double lo = std::min(lsx, lex), hi = std::max(lsx, lex);
        int64_t pos1 = 0;
        bool unsupported = false;
        auto push1 = [&](DevPrim &pr) { pr.start = pos1; pos1 += pr.n; flag_degenerate(pr); push(pr); };
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
        // End zones (round 4).  The turn after a pass starts where its line ends and occupies uturn_dx beyond that end and uturn_h above
        // the line.  A box that meets the turn's zone, or either line within it, moves the turn inwards until the zone is free (again
        // if the moved zone meets another box); both passes end / start there, and the strip beyond stays unworked instead of the field
        // being refused.  clip_lo / clip_hi: every pass's line ends (frame x).  The free ends -- start of the first pass, end of the
        // last -- are not moved.
        auto pass_y = [&](int64_t idx) { return min_y + (double)(fr.reverse_order ? (P - 1 - idx) : idx) * W; };
        auto pass_left = [&](int64_t idx) { return fr.start_from_right ? ((idx & 1) == 0) : ((idx & 1) == 1); };
        std::vector<double> clip_lo((size_t)P, lo), clip_hi((size_t)P, hi);
        if (lsx < lex && !boxes.empty()) {
            const double DX = pc.uturn_dx, H = pc.uturn_h;
            for (int64_t idx = 0; idx + 1 < P; ++idx) {
                const double ya = pass_y(idx), yb = pass_y(idx + 1), ylo = std::min(ya, yb), yhi = std::max(ya + H, yb);
                const bool right = !pass_left(idx);
                double x = right ? hi : lo;
                for (bool moved = true; moved;) {
                    moved = false;
                    for (const Box &b : boxes) {
                        if (!(b.y0 < yhi - 1e-9 && b.y1 > ylo + 1e-9)) continue;
                        if (right ? (b.x1 > x - 1e-9 && b.x0 < x + DX + 1e-9 && b.x0 - DX - 1e-6 < x)
                                  : (b.x0 < x + 1e-9 && b.x1 > x - DX - 1e-9 && b.x1 + DX + 1e-6 > x)) {
                            x = right ? b.x0 - DX - 1e-6 : b.x1 + DX + 1e-6;
                            moved = true;
                        }
                    }
                }
                if (right) clip_hi[(size_t)idx] = clip_hi[(size_t)idx + 1] = x;
                else clip_lo[(size_t)idx] = clip_lo[(size_t)idx + 1] = x;
            }
        }
        std::vector<int> blk;
        std::vector<GPt> upper, lower;           // (the chains of polygon_chains: their capacity is kept from pass to pass)
        for (int64_t idx = 0; idx < P && !unsupported; ++idx) {
            const int64_t pi = fr.reverse_order ? (P - 1 - idx) : idx;
            const double y = min_y + (double)pi * W;
            const bool go_left = fr.start_from_right ? ((idx & 1) == 0) : ((idx & 1) == 1);
            const double lo = clip_lo[(size_t)idx], hi = clip_hi[(size_t)idx];       // (this pass's own line)
            const bool ordered = lsx < lex;                                           // (else: a work area narrower than 2 R, lines as before)
            if (ordered && !(hi - lo > 1e-9)) { unsupported = true; break; }
            const double xs = ordered ? (go_left ? hi : lo) : (go_left ? lex : lsx), xe = ordered ? (go_left ? lo : hi) : (go_left ? lsx : lex);
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
                const std::vector<GPt> &g = gpoly[(size_t)blk[k]];
                if (!g.empty()) {
                    // Round 4: along the obstacle's W/2-grown POLYGON -- the swath is worked up to the polygon, not to its box, and the way
                    // around is the shorter of its upper and lower chain that stays inside the work area's y-range (never longer than
                    // the box's three legs; a line that passes clear of the polygon is not interrupted at all)
                    double pnear, pfar;
                    if (!polygon_chains(g, y, go_left, pnear, pfar, upper, lower)) continue;
                    auto chain_len = [&](const std::vector<GPt> &c) {
                        double l = 0, qx = pnear, qy = y;
                        for (const GPt &p : c) { l += sqrt((p.x - qx) * (p.x - qx) + (p.y - qy) * (p.y - qy)); qx = p.x; qy = p.y; }
                        return l + sqrt((pfar - qx) * (pfar - qx) + (y - qy) * (y - qy));
                    };
                    bool top_ok = true, bot_ok = true;
                    for (const GPt &p : upper) top_ok = top_ok && p.y <= max_y + 1e-9;
                    for (const GPt &p : lower) bot_ok = bot_ok && p.y >= min_y - 1e-9;
                    if (!top_ok && !bot_ok) { unsupported = true; break; }
                    const bool take_top = top_ok && (!bot_ok || chain_len(upper) <= chain_len(lower));
                    const std::vector<GPt> &c = take_top ? upper : lower;
                    push_line(cur, y, pnear, y, FCPP_KIND_SWATH, pi, veh.max_work_speed_kmh, false);
                    double qx = pnear, qy = y;
                    for (const GPt &p : c) {
                        if (fabs(p.x - qx) + fabs(p.y - qy) > 1e-9) push_line(qx, qy, p.x, p.y, FCPP_KIND_DETOUR, pi, veh.headland_turn_speed_kmh, true);
                        qx = p.x; qy = p.y;
                    }
                    push_line(qx, qy, pfar, y, FCPP_KIND_DETOUR, pi, veh.headland_turn_speed_kmh, true);
                    cur = pfar;
                    continue;
                }
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
                // (the turn starts where the line ends: the field's own end zone, or one moved inwards by a box)
                const double xt = ordered ? (turn_right ? hi : lo) : (turn_right ? lex : lsx);
                pr.a[0] = cloth ? xt : (turn_right ? xt + R : xt - R);
                pr.a[1] = y;
                pr.a[2] = rc; pr.a[3] = rs; pr.a[4] = ccx; pr.a[5] = ccy;
                push1(pr);
            }
        }
        if (unsupported) return FCPP_EUNSUPPORTED;
        n_main = pos1;
        return FCPP_OK;
    }
};

// One field on the host (fcpp_planfn.h).  `prims`: the block's list (NULL: sizes only).  -> points of the field.
int64_t plan_field(const PlanConsts &pc, const fcpp_field &f, const fcpp_polys *polys, bool want_device, fcpp_field_info &in, DevField &df,
                   std::vector<DevPrim> &prims)
{
    HostSink sink{ want_device ? &prims : nullptr, polys, want_device ? (int64_t)prims.size() : 0 };
    return plan_field_t(pc, f, in, df, sink);
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

static int build_host_plan_impl(const fcpp_vehicle &veh, const fcpp_options &opt, int64_t n, const fcpp_field *fields, const fcpp_polys *polys,
                                bool want_device, HostPlan &out, std::string &err);

int build_host_plan(const fcpp_vehicle &veh, const fcpp_options &opt, int64_t n, const fcpp_field *fields, const fcpp_polys *polys,
                    bool want_device, HostPlan &out, std::string &err)
{
    // (the per-block work grows vectors on worker threads: fcpp_parallel.h hands the first std::bad_alloc back to this thread)
    try { return build_host_plan_impl(veh, opt, n, fields, polys, want_device, out, err); }
    catch (const std::bad_alloc &) { err = "out of host memory"; return FCPP_ENOMEM; }
}

static int build_host_plan_impl(const fcpp_vehicle &veh, const fcpp_options &opt, int64_t n, const fcpp_field *fields, const fcpp_polys *polys,
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
    // (Fields that DIFFER take 0.8 us each -- cfg2's 1024 rectangles 0.85 ms in the calling thread: side by side from 512 fields on when the
    // batch does not look like copies of one field.)
    const bool differ = n >= 512 && memcmp(&fields[0], &fields[n / 2], sizeof(fcpp_field)) != 0 && memcmp(&fields[0], &fields[n - 1], sizeof(fcpp_field)) != 0;
    const auto for_blocks = [&](const std::function<void(int64_t)> &fn) {
        if (n >= 16384 || differ) WorkerPool::parallel_for(nb, fn);
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
