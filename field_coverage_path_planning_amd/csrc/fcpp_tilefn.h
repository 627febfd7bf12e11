// fcpp_tilefn.h -- the pieces of the tiler that the host tiler (fcpp_tiler.cpp) and the device tiler (fcpp_devplan.hip) must compute
// bit for bit alike: a path point as the tiler sees it (only distances between consecutive points and the margin to the geofence are
// taken from it), the halo walks of a wave tile, the margins of a straight.  One source, compiled for both sides with
// -ffp-contract=off, so the device-built tables equal the host-built ones byte for byte (tests/test_gpu_devplan.py).
#pragma once
#include <math.h>
#include <stdint.h>

#include "fcpp_geom.h"
#include "fcpp_internal.h"

namespace fcpp {

struct Pt2 { double x, y; };      // layout of HIP's double2: the turn templates as the device built them

constexpr int WAVE_HALO_MAX = 40;

// layer 1, closed form: point `off` of pass position `idx` (the formulas of eval_main, fcpp_pointfn.h, on the template copy `tu`)
FCPP_HD void tiler_point_main(const DevField &F, const Pt2 *tu, int64_t idx, int64_t off, double &px, double &py)
{
    const int64_t pi = F.reverse_order ? (F.P - 1 - idx) : idx;
    const double y = F.min_y + (double)pi * F.W;
    const bool go_left = F.start_from_right ? ((idx & 1) == 0) : ((idx & 1) == 1);
    if (off < F.n_line) {
        px = go_left ? linspace_at(F.lex, F.lsx, -F.line_step, F.n_line, off) : linspace_at(F.lsx, F.lex, F.line_step, F.n_line, off);
        py = y;
    } else {
        const Pt2 t = tu[off - F.n_line];
        const bool turn_right = !go_left;
        if (F.turn_model == FCPP_TURN_ARC) px = turn_right ? (F.max_x - t.x) : (F.min_x + t.x);
        else px = turn_right ? ((F.max_x - F.R) + t.x) : ((F.min_x + F.R) - t.x);
        py = y + t.y;
    }
    if (F.rotated) {
        const double tx = px - F.rot_cx, ty = py - F.rot_cy;
        px = (tx * F.rot_cos - ty * F.rot_sin) + F.rot_cx;
        py = (tx * F.rot_sin + ty * F.rot_cos) + F.rot_cy;
    }
}

// sample r of primitive q (the formulas of eval_prim, fcpp_pointfn.h, on the template copies).  Index: int64_t on the host, int on the
// device (r < q.n, an int32: the same values -- the conversion to double is exact either way -- with one instruction instead of a sequence)
template <class Index>
FCPP_HD void tiler_point_prim(const DevPrim &q, const Pt2 *tu, const Pt2 *tc, Index r, double &px, double &py)
{
    auto lin = [&](double a, double b, double step) -> double {
        if (sizeof(Index) == 4) return linspace_at32(a, b, step, q.n, (int)r);
        return linspace_at(a, b, step, q.n, (int64_t)r);
    };
    if (q.kind == PRIM_LINSPACE) { px = lin(q.a[0], q.a[2], q.a[4]); py = lin(q.a[1], q.a[3], q.a[5]); }
    else if (q.kind == PRIM_POINT) { px = q.a[0]; py = q.a[1]; }
    else if (q.kind == PRIM_RAY) { const double t = lin(0.0, q.a[4], q.a[5]); px = q.a[0] + t * q.a[2]; py = q.a[1] + t * q.a[3]; }
    else if (q.kind == PRIM_UTURN) {
        const Pt2 t = tu[r];
        const bool turn_right = q.form & 1;
        if (!(q.form & 4)) px = turn_right ? (q.a[0] - t.x) : (q.a[0] + t.x);
        else px = turn_right ? (q.a[0] + t.x) : (q.a[0] - t.x);
        py = q.a[1] + t.y;
        if (q.form & 2) {
            const double tx = px - q.a[4], ty = py - q.a[5];
            px = (tx * q.a[2] - ty * q.a[3]) + q.a[4];
            py = (tx * q.a[3] + ty * q.a[2]) + q.a[5];
        }
    } else {
        const Pt2 t = tc[r];
        const int ci = q.kind == PRIM_ARC ? q.form : ((q.form + 3) & 3);
        if (ci == 0)      { px = q.a[0] + t.x; py = q.a[1] + t.y; }
        else if (ci == 1) { px = q.a[0] - t.y; py = q.a[1] + t.x; }
        else if (ci == 2) { px = q.a[0] - t.x; py = q.a[1] - t.y; }
        else              { px = q.a[0] + t.y; py = q.a[1] - t.x; }
    }
}

// Is the point so far inside every edge of the field polygon that the device's geofence test of it cannot fire?  The device flags a
// point whose edge function is below -geofence_tol; tiler and kernel evaluate the point by the same formulas and differ by roundings that
// grow with the coordinates' magnitude (~1e-12 m at 1e3 m, ~1e-9 m at UTM-sized 5e6 m).  margin = 1e-7 - geofence_tol, plus 256 ulps of
// the point's own coordinates: the slack stays orders of magnitude above those roundings wherever the field lies.
FCPP_HD bool tiler_inside(const DevField &F, double px, double py, double margin)
{
    const double m = margin + 5.684341886080802e-14 * (fabs(px) + fabs(py));      // 256 x 2^-52
    for (int e = 0; e < 4; ++e)
        if (!(F.ex[e] * px + F.ey[e] * py + F.eo[e] >= m)) return false;
    return true;
}

// samples a straight of step `step_len` needs either side of a quiet zone: the farthest a slower point can pull speeds below the
// straight's nominal value c_nom = (v_nom / 3.6)^2; -1: no quiet zone (degenerate step)
FCPP_HD int64_t tiler_need_for(double c_nom, double step_len, double two_a)
{
    if (!(step_len >= 1e-6)) return -1;
    return (int64_t)(c_nom / (two_a * step_len)) + 3;
}

// Halos of a wave tile (fcpp_sparse.hip).  dist(i) = |p_i - p_(i-1)| for the indices the walks touch.  Backwards from the point before
// the first output s (whose final speed the segment metrics need) until the couplings 2a|dp| add up to `cap`, a skipped step or the
// path's start, plus one lane for the stencil of the outermost point; forwards likewise from the last output e.  -1: the halo would
// exceed WAVE_HALO_MAX lanes.  (A step within 0.1 % of the 1e-6 threshold counts neither as skipped nor as a coupling.)
template <class Dist>
FCPP_HD int tiler_back_halo(const Dist &dist, int64_t s, double two_a, double cap)
{
    if (s == 0) return 0;
    const int64_t j = s - 1;
    int64_t m = j;
    double acc = 0.0;
    for (;;) {
        if (m == 0) return (int)(j + 1);
        const double dm = dist(m);
        if (dm < 0.999e-6) return (int)(j - (m - 1) + 1);
        if (dm > 1.001e-6) acc += two_a * dm;
        --m;
        if (acc >= cap) return (int)(j - m + 1);
        if (j - m + 1 > WAVE_HALO_MAX) return -1;
    }
}
template <class Dist>
FCPP_HD int tiler_fwd_halo(const Dist &dist, int64_t e, int64_t n, double two_a, double cap)
{
    if (e == n - 1) return 0;
    int64_t m = e;
    double acc = 0.0;
    for (;;) {
        const double dm = dist(m + 1);
        if (dm < 0.999e-6) return (int)(m + 1 - e);
        if (dm > 1.001e-6) acc += two_a * dm;
        ++m;
        if (m == n - 1 || acc >= cap) return (int)(m - e);
        if (m - e > WAVE_HALO_MAX) return -1;
    }
}

// the cap of the halo walks for the batch's u_cap = (v_max / 3.6)^2
FCPP_HD double tiler_halo_cap(double u_cap) { return u_cap * (1.0 + 1e-9) + 1e-12; }

// reduction class of a path by its statistics entries: 8 lanes, a wavefront, a workgroup, 64 workgroups + join
FCPP_HD int tiler_reduce_class(int64_t ne, int64_t reduce_wg_max) { return ne <= 64 ? 0 : (ne <= 256 ? 1 : (ne <= reduce_wg_max ? 2 : 3)); }

}  // namespace fcpp
