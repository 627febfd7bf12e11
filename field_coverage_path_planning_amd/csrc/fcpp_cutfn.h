// fcpp_cutfn.h -- the cut of a field's general stretch into wave tiles IN CLOSED FORM, one source for the host tiler (fcpp_tiler.cpp) and
// the device planner (fcpp_devplan.hip: the counting pass, k_tile_fields) -- round 5.
//
// At the reference's own sampling (sample_spacing = 0: 2 points per swath line, 20 per U-turn, 15 per corner arc, 20 per headland side,
// MLP:761-767, 807, 1046, 1013) a field's path is [ span: all complete passes, closed form ] [ general stretch: the last line + layer 2 ],
// and the general stretch is a few hundred points in at most a few dozen primitives.  What the cut needs of them is the distance between
// consecutive points (the halos of a wave tile are sized from the couplings 2a|dp|, fcpp_tilefn.h) -- and inside a primitive that distance
// is the primitive's own step (a straight's numpy.linspace step, a ray's parameter step, the chord between two samples of the batch's
// turn template), between two primitives the distance of two END points that their records name.  So the step lengths are O(primitives)
// values, not O(points) evaluations: round 4's device tiler evaluated every point of the stretch into an LDS window to measure them
// (51 of the 190 us of a fresh 4096-field plan call, 0.71 of cfg5's 3.3 ms); here the field's wavefront takes the primitives a lane each and
// the tiles of a candidate cut a lane each.
//
// The rule (the same on both sides, so host-built and device-built tables stay equal byte for byte):
//   * T = the smallest number of NEAR-EQUAL tiles (sizes differ by at most one point, the larger ones first) such that every tile fits:
//     back halo + outputs + forward halo <= 128 lanes, its points in at most nine primitives, at least eight outputs per tile.
//     (Round 4 cut greedily -- every tile as many outputs as fit -- which made tile k's start depend on tile k - 1's cut: a chain of
//     dependent halo walks.  Near-equal tiles have independent halos and the same number of tiles for every field measured.)
//   * halos as in fcpp_tilefn.h (tiler_back_halo / tiler_fwd_halo), on the closed-form distances.
//   * a tile is `inside` when every primitive its outputs touch lies inside the geofence with the tiler's margin as a whole (a straight:
//     both ends; a corner turn: the corners of the box its template spans) and, for outputs of layer 1, the field's span_inside.
// A stretch the rule cannot cut (a halo beyond WAVE_HALO_MAX lanes, more than CUT_TILES_MAX tiles) stays with the general kernel, as before.
#pragma once
#include <math.h>
#include <stdint.h>

#include "fcpp_internal.h"
#include "fcpp_tilefn.h"

namespace fcpp {

constexpr int CUT_TILES_MAX = 16;        // wave tiles of one field's general stretch (fields of the reference's sizes have four to six)
constexpr int CUT_PRIMS_MAX = 32;        // primitives of a field the closed-form cut takes (8 per headland loop + 3 reverse fills: three loops = 27)
constexpr int CUT_WAVE_LANES = 128;      // points of a wave tile (two per lane, fcpp_sparse2_fn.h)

// what the cut keeps of a primitive
struct CutPrim {
    double step;             // |p_r - p_(r-1)| inside the primitive for straights and rays (turns: the batch's chord table)
    double din;              // |first point - the point before it| (the previous primitive's last point; primitive 0: the last point of layer 1)
    int16_t start_rel;       // first point, relative to n_main (a field of this cut has at most CUT_TILES_MAX x 128 general points)
    int16_t n;
    uint8_t kind;
    uint8_t inside;          // every point of the primitive lies inside the geofence with the tiler's margin
    uint8_t _pad[2];
};
static_assert(sizeof(CutPrim) == 24, "a row of the device planner's LDS table");

struct CutTile {
    int32_t s;               // first output point (index in the field's path)
    uint8_t c, hb, hf, inside;
};
enum { CUT_OK = 0, CUT_GENERAL = 1, CUT_WINDOW = 2 };
struct FieldCut {
    int32_t n_tiles;
    int32_t status;          // CUT_OK: n_tiles wave tiles; CUT_GENERAL: the stretch stays with the general kernel; CUT_WINDOW: not a field of
                             // this cut (no closed-form span, too many primitives): round 4's window cut decides
    CutTile t[CUT_TILES_MAX];
};
static_assert(sizeof(CutTile) == 8 && sizeof(FieldCut) == 8 + 8 * CUT_TILES_MAX, "plain records");

// the batch's constants of the cut
struct CutConsts {
    const Pt2 *tu, *tc;      // the turn templates (host copies on the host, device arrays on the device: the same values)
    const Pt2 *dk_u, *dk_c;  // per sample k: x = |t_k - t_(k-1)| (k_build_template_metrics; entry 0 unused)
    int32_t nu, nc;
    int32_t turn_quiet, wave_factor;
    double two_a, u_cap, c_line, fence_margin;
    double jump[2];          // length of the jump from a U-turn's last sample to the next line's first point; [1]: passes descending in y
    double tc_lo[2], tc_hi[2];   // the corner template's box: min / max of its samples' x and y
    double u_step_min, c_step_min;   // the shortest chord of the U-turn / of the corner template (the step a turn's samples count with in the halo walks)
};

// The span of a field at sparse sampling: all complete passes (line + turn) in closed form, or 0 (fcpp_tiler.cpp: tile_field).
FCPP_HD int64_t cut_span_points(const DevField &F, const CutConsts &cc)
{
    const int64_t per = (int64_t)F.n_line + F.n_turn, P = F.P;
    const bool turn_quiet = cc.turn_quiet && F.n_turn == cc.nu && F.line_step > 0.0;
    const int64_t need1 = tiler_need_for(cc.c_line, fabs(F.line_step), cc.two_a);
    if (!(need1 >= 0 && per > 0 && F.gen_main > 0)) return 0;
    const bool span = turn_quiet && P >= 2 && (int64_t)F.n_line - need1 < 64 && (P - 1) * per < (int64_t)0x7fffffff;
    return span ? (P - 1) * per : 0;
}

// Is this a field of the closed-form cut?  (reference sampling, a span, wave tiles allowed, few primitives)
FCPP_HD bool cut_applies(const DevField &F, const CutConsts &cc, int64_t S)
{
    if (S <= 0 || F.n_line != 2 || F.prim_count > CUT_PRIMS_MAX || F.gen_main != F.n_main || F.n_total - S > (int64_t)CUT_TILES_MAX * CUT_WAVE_LANES) return false;
    return F.n_turn == cc.nu && (double)cc.wave_factor * cc.two_a * fabs(F.line_step) >= cc.u_cap;
}

// first / last point of a primitive: the values tiler_point_prim gives for samples 0 and n - 1, kind by kind without its sample
// arithmetic -- a straight's ends are in its record, a ray's last point is origin + length x direction (numpy.linspace ends on `stop`),
// a turn's are the template's first / last sample through the quadrant formulas (only they go through tiler_point_prim)
FCPP_HD void cut_prim_end(const DevPrim &q, const CutConsts &cc, bool last, double &x, double &y)
{
    const bool turn = q.kind == PRIM_ARC || q.kind == PRIM_CAC || q.kind == PRIM_UTURN;
    if (turn || (q.form & 8)) { tiler_point_prim(q, cc.tu, cc.tc, last ? (int)q.n - 1 : 0, x, y); return; }
    x = q.a[0]; y = q.a[1];
    if (last && q.n > 1) {
        if (q.kind == PRIM_LINSPACE) { x = q.a[2]; y = q.a[3]; }
        else if (q.kind == PRIM_RAY) { x = q.a[0] + q.a[4] * q.a[2]; y = q.a[1] + q.a[4] * q.a[3]; }
    }
}

// One primitive's record from its own end points (fx, fy) / (ex, ey) (cut_prim_end) and the point before its first (px, py).  F: the field's
// geofence edges.  ok is cleared for a primitive that is not this cut's.
FCPP_HD CutPrim cut_prim_rec(const DevPrim &q, const DevField &F, const CutConsts &cc, double px, double py, double fx, double fy, double ex, double ey, bool &ok)
{
    CutPrim c;
    c.start_rel = (int16_t)(q.start - F.n_main); c.n = (int16_t)q.n; c.kind = (uint8_t)q.kind; c.step = 0.0; c._pad[0] = c._pad[1] = 0;
    if (q.start - F.n_main > 0x7fff || q.n > 0x7fff) ok = false;
    { const double dx = fx - px, dy = fy - py; c.din = sqrt(dx * dx + dy * dy); }
    bool in = tiler_inside(F, fx, fy, cc.fence_margin) && tiler_inside(F, ex, ey, cc.fence_margin);
    if (q.form & 8) ok = false;                               // a degenerate straight (flag_degenerate): not this cut's
    if (q.kind == PRIM_LINSPACE) c.step = sqrt(q.a[4] * q.a[4] + q.a[5] * q.a[5]);
    else if (q.kind == PRIM_RAY) c.step = fabs(q.a[5]);
    else if (q.kind == PRIM_ARC || q.kind == PRIM_CAC) {
        if (q.n != cc.nc) ok = false;
        c.step = cc.c_step_min;
        // every sample is corner + (+-tx, +-ty) or (+-ty, +-tx) with t in the template's box: inside iff the box's four corners are
        const Pt2 corners[4] = { { cc.tc_lo[0], cc.tc_lo[1] }, { cc.tc_hi[0], cc.tc_lo[1] }, { cc.tc_lo[0], cc.tc_hi[1] }, { cc.tc_hi[0], cc.tc_hi[1] } };
        for (int k = 0; k < 4; ++k) {
            double bx, by;
            tiler_point_prim(q, cc.tu, &corners[k], 0, bx, by);
            in = in && tiler_inside(F, bx, by, cc.fence_margin);
        }
    } else if (q.kind != PRIM_POINT) ok = false;             // (U-turn primitives: obstacle-aware swaths, not this cut's)
    if (!(q.n >= 1)) ok = false;
    c.inside = in ? 1 : 0;
    return c;
}
// the same along the path: (lx, ly) in = the point before the primitive's first, out = its own last point
FCPP_HD CutPrim cut_prim_info(const DevPrim &q, const DevField &F, const CutConsts &cc, double &lx, double &ly, bool &ok)
{
    double fx, fy;
    cut_prim_end(q, cc, false, fx, fy);
    double ex = fx, ey = fy;
    if (q.n > 1) cut_prim_end(q, cc, true, ex, ey);
    const CutPrim c = cut_prim_rec(q, F, cc, lx, ly, fx, fy, ex, ey, ok);
    lx = ex; ly = ey;
    return c;
}

// the field's last point of layer 1 (the point before primitive 0's first)
FCPP_HD void cut_main_end(const DevField &F, const CutConsts &cc, double &x, double &y)
{
    tiler_point_main(F, cc.tu, (int64_t)F.P - 1, (int64_t)F.n_line - 1, x, y);
}

// |p_i - p_(i-1)| in closed form, i >= 1.  PV: prims(k) -> const CutPrim &.  k_hint: the primitive of the last call (walks move by one point).
template <class PV>
struct CutDist {
    const DevField &F;
    const CutConsts &cc;
    const PV &pv;
    int np;
    // the cursor: primitive k holds the points [k_lo, k_hi) (relative to n_main); a walk moves it a primitive at a time
    mutable int k;
    mutable int32_t k_lo, k_hi;
    int64_t per, S;
    FCPP_HD void load(int kk) const { k = kk; k_lo = pv(kk).start_rel; k_hi = k_lo + pv(kk).n; }
    FCPP_HD CutDist(const DevField &f, const CutConsts &c, const PV &p, int n) : F(f), cc(c), pv(p), np(n), k(0), k_lo(0), k_hi(0), per((int64_t)f.n_line + f.n_turn),
                                                                                  S(((int64_t)f.P - 1) * ((int64_t)f.n_line + f.n_turn))
    {
        if (np > 0) load(0);
    }
    FCPP_HD int prim_of(int64_t i) const                        // the primitive that holds point i >= n_main (the cursor moves to it)
    {
        const int32_t rel = (int32_t)(i - F.n_main);
        while (k > 0 && rel < k_lo) load(k - 1);
        while (k + 1 < np && rel >= k_hi) load(k + 1);
        return k;
    }
    FCPP_HD double operator()(int64_t i) const
    {
        if (i < F.n_main) {
            // (offset in the pass without a division: the walks stay within a few passes of the path's last line, whose first point is S)
            int64_t off = i - S;
            while (off < 0) off += per;
            while (off >= per) off -= per;
            if (off == 0) return cc.jump[F.reverse_order ? 1 : 0];
            if (off < F.n_line) return fabs(F.line_step);
            const int64_t c = off - F.n_line;
            return c == 0 ? 0.0 : cc.dk_u[c].x;                // (a turn starts on its line's last point)
        }
        const int kk = prim_of(i);
        const int32_t r = (int32_t)(i - F.n_main) - k_lo;
        const CutPrim &p = pv(kk);
        if (r == 0) return p.din;
        return (p.kind == PRIM_ARC || p.kind == PRIM_CAC) ? cc.dk_c[r].x : p.step;
    }
};

// ---- the halos, a primitive at a time -------------------------------------------------------------------------------------------------
// The walks of tiler_back_halo / tiler_fwd_halo (fcpp_tilefn.h) on the closed-form step lengths, taken a SEGMENT at a time: inside a
// segment every step has the segment's own length, so the steps a walk takes there follow from one division.  Segments: primitive k >= 0 of
// layer 2; -1 = the path's last swath line (its first step: the jump from the turn before it); -2 = the U-turn before that line (a turn
// starts on its line's last point -- a skipped step, where every backward walk ends).  A turn's steps count with the SHORTEST chord of the
// batch's template (arcs: all chords are equal; clothoid turns: within a per cent): a halo may come out a lane longer than the point-by-point
// walk's, never shorter.  Same return values as the point-by-point walks otherwise (tests/native/tiler_check_driver.cpp compares them).
struct CutSeg { int32_t n; double step, din; };
template <class PV>
FCPP_HD CutSeg cut_seg(const DevField &F, const CutConsts &cc, const PV &pv, int k)
{
    CutSeg g;
    if (k >= 0) { const CutPrim &p = pv(k); g.n = p.n; g.step = p.step; g.din = p.din; }
    else if (k == -1) { g.n = F.n_line; g.step = fabs(F.line_step); g.din = cc.jump[F.reverse_order ? 1 : 0]; }
    else { g.n = F.n_turn; g.step = cc.u_step_min; g.din = 0.0; }
    return g;
}
// the segment of path point i >= S - n_turn (S = the last line's first point) and i's offset in it
template <class PV>
FCPP_HD int cut_locate(const DevField &F, const PV &pv, int np, int64_t i, int32_t &r)
{
    const int64_t S = ((int64_t)F.P - 1) * ((int64_t)F.n_line + F.n_turn);
    if (i < S) { r = (int32_t)(i - (S - F.n_turn)); return -2; }
    if (i < F.n_main) { r = (int32_t)(i - S); return -1; }
    const int32_t rel = (int32_t)(i - F.n_main);
    int k = 0;
    for (int q = 1; q < np; ++q) k += pv(q).start_rel <= rel ? 1 : 0;
    r = rel - pv(k).start_rel;
    return k;
}
// steps of length w >= 0 until acc + n w >= cap: the smallest such n >= 1 (INT32_MAX: never)
FCPP_HD int32_t cut_steps_needed(double acc, double w, double cap)
{
    if (!(w > 0.0)) return INT32_MAX;
    const double q = (cap - acc) / w;
    if (!(q < 1e6)) return INT32_MAX;
    int32_t n = (int32_t)q;
    if (n < 1) n = 1;
    while (acc + (double)n * w < cap) ++n;
    while (n > 1 && acc + (double)(n - 1) * w >= cap) --n;
    return n;
}
// backwards from point j = s - 1 in segment k at offset r; k_end: the segment of the halo's first point
template <class PV>
FCPP_HD int cut_back_halo(const DevField &F, const CutConsts &cc, const PV &pv, int k, int32_t r, double cap, int &k_end)
{
    int32_t t = 0;                     // steps taken
    double acc = 0.0;
    for (;;) {
        const CutSeg g = cut_seg(F, cc, pv, k);
        k_end = k;
        if (r > 0) {
            if (g.step < 0.999e-6) return t + 2;                         // a skipped step
            const double w = g.step > 1.001e-6 ? cc.two_a * g.step : 0.0;
            const int32_t need = cut_steps_needed(acc, w, cap);
            if (need <= r) { t += need; return t <= WAVE_HALO_MAX ? t + 1 : -1; }
            t += r; acc += (double)r * w;
            if (t + 1 > WAVE_HALO_MAX) return -1;
        }
        // the segment's first point: the step from the segment before it
        if (k <= -2) return t + 2;                                       // (the turn starts on its line's last point)
        if (g.din < 0.999e-6) { k_end = k - 1; return t + 2; }
        if (g.din > 1.001e-6) acc += cc.two_a * g.din;
        t += 1;
        --k;
        k_end = k;
        if (acc >= cap) return t + 1;
        if (t + 1 > WAVE_HALO_MAX) return -1;
        r = cut_seg(F, cc, pv, k).n - 1;
    }
}
// forwards from point e in segment k at offset r; np primitives, the last one ends the path; k_end: the segment of the halo's last point
template <class PV>
FCPP_HD int cut_fwd_halo(const DevField &F, const CutConsts &cc, const PV &pv, int np, int k, int32_t r, double cap, int &k_end)
{
    int32_t t = 0;
    double acc = 0.0;
    for (;;) {
        const CutSeg g = cut_seg(F, cc, pv, k);
        k_end = k;
        const int32_t avail = g.n - 1 - r;
        if (avail > 0) {
            if (g.step < 0.999e-6) return t + 1;
            const double w = g.step > 1.001e-6 ? cc.two_a * g.step : 0.0;
            const int32_t need = cut_steps_needed(acc, w, cap);
            if (need <= avail) return t + need - 1 <= WAVE_HALO_MAX ? t + need : -1;
            t += avail; acc += (double)avail * w;
            if (k == np - 1) return t - 1 <= WAVE_HALO_MAX ? t : -1;           // the path's last point
            if (t > WAVE_HALO_MAX) return -1;
        } else if (k == np - 1) return t;
        const CutSeg h = cut_seg(F, cc, pv, k + 1);
        if (h.din < 0.999e-6) { k_end = k + 1; return t + 1; }
        if (h.din > 1.001e-6) acc += cc.two_a * h.din;
        t += 1;
        ++k;
        k_end = k;
        if ((k == np - 1 && h.n == 1) || acc >= cap) return t;
        if (t > WAVE_HALO_MAX) return -1;
        r = 0;
    }
}

// One tile of a candidate cut: outputs [s, s + c), e = s + c - 1; (kj, rj): the segment and offset of point s - 1 (unused when s == 0),
// (ke, re): of point e (cut_locate).  -> 0: it fits (Hb, Hf, inside set); 1: the stretch cannot be cut into wave tiles at all (a halo beyond
// WAVE_HALO_MAX lanes: more tiles do not help); 2: the tile does not fit (too many lanes or primitives: more, smaller tiles may)
template <class PV>
FCPP_HD int cut_tile_eval_at(const DevField &F, const CutConsts &cc, const PV &pv, int np, double cap, int64_t s, int64_t c, int kj, int32_t rj, int ke, int32_t re,
                             int &Hb, int &Hf, bool &in)
{
    const int64_t n = F.n_total, e = s + c - 1;
    int ka = 0, kb = ke, ko = 0;                 // the segments of the tile's first point, last point, first output
    Hb = 0;
    if (s > 0) {
        ko = rj + 1 < cut_seg(F, cc, pv, kj).n ? kj : kj + 1;          // (the first output is the next point)
        Hb = cut_back_halo(F, cc, pv, kj, rj, cap, ka);
    } else ko = F.n_main > 0 ? -1 : 0;
    if (Hb < 0) return 1;
    Hf = 0;
    if (e != n - 1) Hf = cut_fwd_halo(F, cc, pv, np, ke, re, cap, kb);
    if (Hf < 0) return 1;
    if (Hb + c + Hf > CUT_WAVE_LANES) return 2;
    if (Hb == 0) ka = ko;
    if ((kb > 0 ? kb : 0) - (ka > 0 ? ka : 0) > 8) return 2;          // the tile record names nine primitives
    in = true;
    if (ko < 0) in = F.span_inside != 0;                               // outputs of layer 1: the last line (the span's test covers every line)
    for (int q = ko > 0 ? ko : 0; q <= ke; ++q) in = in && pv(q).inside != 0;
    return 0;
}
template <class PV>
FCPP_HD int cut_tile_eval(const DevField &F, const CutConsts &cc, const PV &pv, int np, double cap, int64_t s, int64_t c, int &Hb, int &Hf, bool &in)
{
    int32_t rj = 0, re = 0;
    const int kj = s > 0 ? cut_locate(F, pv, np, s - 1, rj) : 0;
    const int ke = cut_locate(F, pv, np, s + c - 1, re);
    return cut_tile_eval_at(F, cc, pv, np, cap, s, c, kj, rj, ke, re, Hb, Hf, in);
}
// the candidate cuts: T near-equal tiles, T = cut_first_T(G), + 1, ... up to CUT_TILES_MAX; tile t of T: outputs [a + cut_tile_start, + cut_tile_count)
// (32-bit: a field of this cut has at most CUT_TILES_MAX x 128 general points)
FCPP_HD int32_t cut_first_T(int32_t G) { return (G + CUT_WAVE_LANES - 1) / CUT_WAVE_LANES; }
FCPP_HD bool cut_T_possible(int32_t G, int32_t T) { return T <= CUT_TILES_MAX && !(G / T < 8 && T > 1); }      // (fewer than eight outputs per tile: the general kernel's)
FCPP_HD int32_t cut_tile_start(int32_t G, int32_t T, int32_t t) { const int32_t base = G / T, rem = G - base * T; return t * base + (t < rem ? t : rem); }
FCPP_HD int32_t cut_tile_count(int32_t G, int32_t T, int32_t t) { const int32_t base = G / T, rem = G - base * T; return base + (t < rem ? 1 : 0); }

// The cut of the general stretch [a, n_total), tile after tile (the host; the device runs the tiles of a candidate cut side by side, a lane
// each, and takes the same decisions: fcpp_devplan.hip).  pv(k): the field's CutPrim records (cut_prim_info in path order).
template <class PV>
FCPP_HD void cut_field(const DevField &F, const CutConsts &cc, const PV &pv, int np, int64_t a, FieldCut &out)
{
    const int32_t G = (int32_t)(F.n_total - a);
    out.n_tiles = 0; out.status = CUT_GENERAL;
    for (int k = 0; k < CUT_TILES_MAX; ++k) { out.t[k].s = 0; out.t[k].c = out.t[k].hb = out.t[k].hf = out.t[k].inside = 0; }
    if (G <= 0) { out.status = CUT_OK; return; }
    const double cap = tiler_halo_cap(cc.u_cap);
    for (int32_t T = cut_first_T(G); cut_T_possible(G, T); ++T) {
        bool ok = true;
        for (int32_t t = 0; t < T && ok; ++t) {
            const int64_t s = a + cut_tile_start(G, T, t), c = cut_tile_count(G, T, t);
            int Hb = 0, Hf = 0;
            bool in = false;
            const int code = cut_tile_eval(F, cc, pv, np, cap, s, c, Hb, Hf, in);
            if (code == 1) return;
            if (code == 2) { ok = false; break; }
            CutTile &ct = out.t[t];
            ct.s = (int32_t)s; ct.c = (uint8_t)c; ct.hb = (uint8_t)Hb; ct.hf = (uint8_t)Hf; ct.inside = in ? 1 : 0;
        }
        if (ok) { out.n_tiles = (int32_t)T; out.status = CUT_OK; return; }
    }
}

}  // namespace fcpp
