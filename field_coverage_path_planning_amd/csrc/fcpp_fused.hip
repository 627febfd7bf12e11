// fcpp_fused.hip -- pipeline B: the whole hot path in ONE pass over HBM (36 B written per point, nothing read
// back): generate -> curvature -> clamp -> forward/backward sweeps -> validate -> metrics.
//
// Work decomposition (gfx950: 64-wide waves, 256 CUs): one 256-thread workgroup per 2048-point tile of one
// field's path; thread t owns the 8 CONSECUTIVE points 8t..8t+7, so
//   * the 3-point curvature stencil and the segment lengths live in registers; only the two end
//     neighbours of a thread come from the adjacent lane (DPP shuffle) or, at wave edges, from 64 bytes of LDS;
//   * the min-plus scans of the sweeps run over registers: 8 serial steps per thread, one 6-step wave scan
//     of the per-thread maps, 4 wave aggregates through LDS;
//   * results are transposed through a 4.5 KB per-wave LDS buffer so that every global store instruction
//     writes 512 contiguous bytes per wave (SoA arrays, coalesced).
// No workgroup ever waits for another one: what the sweeps need from outside the tile is RECOMPUTED.
// A constraint at point j can only bind at point i while u0_j + 2a*dist(i,j) < u_cap = (v_max/3.6)^2, i.e.
// within u_cap/(2a) metres (5.8 m for the default vehicle), so wave 0 re-generates 64-point chunks before the
// tile (and wave 3 after it) until the accumulated 2a*distance exceeds u_cap or the path ends; usually one
// chunk.  The recomputation is exact (same arithmetic as the owning tile), so results do not depend on tiling.
#include "fcpp_devfn.h"

namespace fcpp {

struct HaloInfo {
    double px, py;       // the neighbouring path point (index s-1 or s+count)
    double carry;        // value the forward (backward) sweep carries into the tile; +inf if none
    double kappa, v0, vnom, u0;
    int valid;
};

struct RedSharedF { double d[NWAVE][9]; long long i[NWAVE][4]; };

static constexpr int TR_WORDS = 64 * IPT + 64;   // padded per-wave transposition buffer (doubles)

struct FusedShared {
    HaloInfo back, fwd;
    double efx[NWAVE], efy[NWAVE], elx[NWAVE], ely[NWAVE];   // first / last point of each wave
    Agg wf[NWAVE], wb[NWAVE];
    double ev[NWAVE], ek[NWAVE];                                // last item of each wave: final v, kappa
    uint32_t efs[NWAVE];                                         // ... and its segment word
    double tr[NWAVE][TR_WORDS];
    RedSharedF R;
};

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// curvature (MLP:513-536) from the two chords and their lengths.  dtheta = atan2(sin(t2-t1), cos(t2-t1)) is the
// signed angle between the chords = atan2(cross, dot); exactly collinear chords (every interior point of an
// axis-aligned straight run) and tiny angles never reach atan2.
__device__ __forceinline__ double curv_chords(double dx1, double dy1, double ds1, double dx2, double dy2, double ds2)
{
    if (ds1 < 1e-6 || ds2 < 1e-6) return 0.0;
    const double cr = dx1 * dy2 - dy1 * dx2, dt = dx1 * dx2 + dy1 * dy2;
    if (cr == 0.0 && dt > 0.0) return 0.0;
    double dth;
    if (dt > 0.0 && fabs(cr) <= 1e-8 * dt) dth = cr / dt;     // atan(x) = x (1 - x^2/3 + ..): exact to < 1e-16 relative
    else dth = atan2_slow(cr, dt);
    return fabs(2 * dth / (ds1 + ds2));
}

// speed after the curvature clamp (MLP:496-504); nominal = the primitive's nominal speed
__device__ __forceinline__ double clamped_speed(double v_nom, double kappa, const DevConst &cst, bool &clamped)
{
    clamped = false;
    if (kappa > 1e-6) {
        const double vmax_ms = sqrt(cst.a_lat / kappa) * cst.sf;
        const double vmax_kmh = vmax_ms * 3.6;
        if (v_nom > vmax_kmh) { clamped = true; return vmax_kmh; }
    }
    return v_nom;
}

__device__ __forceinline__ double clamp_speed(double v, double kappa, const DevConst &cst, int &adj)
{
    bool cl;
    const double r = clamped_speed(v, kappa, cst, cl);
    adj += cl ? 1 : 0;
    return r;
}

// Straight-primitive lookup shared by the fast paths: is the index range [i_lo, i_hi] (inclusive) inside ONE
// straight primitive (a swath line or a headland straight)?  Returns its linspace description.
struct StraightRun {
    double ax, ay, bx, by, sx, sy;   // start, stop, step of numpy.linspace (x and y)
    int64_t r0, n;                   // position of i_lo inside the primitive, samples of the primitive
    uint32_t fs;
    bool rot;
};

__device__ __forceinline__ bool find_straight(const DevField &f, const DevPrim *__restrict__ prims, int64_t i_lo,
                                              int64_t i_hi, StraightRun &o)
{
    if (i_lo < 0 || i_hi >= f.n_total) return false;
    if (i_hi < f.n_main) {
        const int64_t per = (int64_t)f.n_line + f.n_turn;
        const int64_t idx = i_lo / per;
        o.r0 = i_lo - idx * per;
        if (o.r0 + (i_hi - i_lo) >= f.n_line) return false;
        const int64_t pi = f.reverse_order ? (f.P - 1 - idx) : idx;
        const bool go_left = f.start_from_right ? ((idx & 1) == 0) : ((idx & 1) == 1);
        o.ax = go_left ? f.lex : f.lsx; o.bx = go_left ? f.lsx : f.lex; o.sx = go_left ? -f.line_step : f.line_step;
        o.ay = o.by = f.min_y + (double)pi * f.W; o.sy = 0.0;
        o.n = f.n_line; o.rot = f.rotated != 0;
        o.fs = FCPP_KIND_SWATH | ((uint32_t)pi << FCPP_INDEX_SHIFT);
        return true;
    }
    if (i_lo < f.n_main) return false;
    int a = f.prim_first, b = f.prim_first + f.prim_count - 1;
    while (a < b) {
        const int m = (a + b + 1) >> 1;
        if (prims[m].start <= i_lo) a = m; else b = m - 1;
    }
    const DevPrim &p = prims[a];
    o.r0 = i_lo - p.start;
    if (p.kind != PRIM_LINSPACE || o.r0 + (i_hi - i_lo) >= p.n) return false;
    o.ax = p.a[0]; o.bx = p.a[2]; o.sx = p.a[4]; o.ay = p.a[1]; o.by = p.a[3]; o.sy = p.a[5];
    o.n = p.n; o.fs = p.fs; o.rot = false;
    return true;
}

__device__ __forceinline__ void straight_point(const DevField &f, const StraightRun &r, int64_t rk, double &px, double &py)
{
    px = (double)rk * r.sx + r.ax; py = (double)rk * r.sy + r.ay;     // numpy.linspace: k*step + start
    if (rk == r.n - 1) { px = r.bx; py = r.by; }                       // ... and the last sample is `stop`
    if (r.rot) {
        const double tx = px - f.rot_cx, ty = py - f.rot_cy;
        px = (tx * f.rot_cos - ty * f.rot_sin) + f.rot_cx;
        py = (tx * f.rot_sin + ty * f.rot_cos) + f.rot_cy;
    }
}

// One wave recomputes what the sweeps carry across the tile edge (see the header comment).
template <bool BACK>
__device__ void halo_wave(const DevField &f, const DevPrim *__restrict__ prims, const DevConst &cst,
                          int64_t edge, HaloInfo *out)
{
    const int lane = threadIdx.x & 63;
    const int64_t n = f.n_total;
    const double two_a = 2 * cst.a_lon;
    if (BACK ? (edge <= 0) : (edge >= n)) {
        if (lane == 0) { out->valid = 0; out->carry = FCPP_INF; out->px = out->py = 0; out->kappa = out->v0 = out->vnom = out->u0 = 0; }
        return;
    }
    // Fast case (most tiles at fine sampling): the point next to the edge lies on a straight primitive that extends at
    // least u_cap / (2a) metres away from the tile.  Every point on that stretch has curvature 0 and the same u0, and
    // nothing farther away can bind, so the carried value is that u0 and no point needs generating.
    {
        StraightRun r;
        if (BACK ? find_straight(f, prims, edge - 2, edge - 1, r) : find_straight(f, prims, edge, edge + 1, r)) {
            const double step_len = sqrt(r.sx * r.sx + r.sy * r.sy);
            if (step_len >= 1e-6) {
                const int64_t need = (int64_t)(cst.u_cap / (two_a * step_len)) + 3;
                const int64_t pos = BACK ? r.r0 + 1 : r.r0;   // position of the neighbouring point in the primitive
                if (BACK ? (pos - need >= 0) : (pos + need <= r.n - 1)) {
                    if (lane == 0) {
                        const double ms = nominal_ms(r.fs, cst);
                        double px, py;
                        straight_point(f, r, pos, px, py);
                        out->valid = 1; out->px = px; out->py = py; out->kappa = 0.0;
                        out->v0 = out->vnom = nominal_speed(r.fs, cst); out->u0 = ms * ms; out->carry = ms * ms;
                    }
                    return;
                }
            }
        }
    }
    Agg total = { FCPP_INF, 0.0 };
    bool first = true;
    int64_t b = BACK ? edge - 64 : edge;
    for (;;) {
        const int64_t i = b + lane;
        const bool act = i >= 0 && i < n;
        GenOut g; g.x = g.y = g.v = 0; g.fs = 0;
        if (act) g = gen_point_slow(&f, prims, i, &cst);
        double xm = __shfl_up(g.x, 1), ym = __shfl_up(g.y, 1), xp = __shfl_down(g.x, 1), yp = __shfl_down(g.y, 1);
        const int64_t ei = lane == 0 ? i - 1 : i + 1;
        if ((lane == 0 || lane == 63) && ei >= 0 && ei < n) {   // the chunk's two outer neighbours
            const GenOut e = gen_point_slow(&f, prims, ei, &cst);
            if (lane == 0) { xm = e.x; ym = e.y; } else { xp = e.x; yp = e.y; }
        }
        double kappa = 0, dprev = 0, dnext = 0;
        const double dx1 = g.x - xm, dy1 = g.y - ym, dx2 = xp - g.x, dy2 = yp - g.y;
        if (act && i > 0) dprev = sqrt(dx1 * dx1 + dy1 * dy1);
        if (act && i < n - 1) dnext = sqrt(dx2 * dx2 + dy2 * dy2);
        if (act && i > 0 && i < n - 1) kappa = curv_chords(dx1, dy1, dprev, dx2, dy2, dnext);
        int adj = 0;
        const double v0 = clamp_speed(g.v, kappa, cst, adj);
        const double ms = v0 / 3.6;
        Agg me;
        me.c = act ? ms * ms : FCPP_INF;
        if (BACK) me.w = !act ? 0.0 : ((i == 0 || dprev < 1e-6) ? FCPP_INF : two_a * dprev);
        else      me.w = !act ? 0.0 : ((i == n - 1 || dnext < 1e-6) ? FCPP_INF : two_a * dnext);
        Agg inc = me;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            if (BACK) {
                Agg p = { __shfl_up(inc.c, o), __shfl_up(inc.w, o) };
                if (lane >= o) inc = combine_after(p, inc);
            } else {
                Agg p = { __shfl_down(inc.c, o), __shfl_down(inc.w, o) };
                if (lane + o < 64) inc = combine_after(p, inc);
            }
        }
        const int src = BACK ? 63 : 0;
        Agg chunk = { __shfl(inc.c, src), __shfl(inc.w, src) };
        total = first ? chunk : combine_after(chunk, total);   // the farther chunk acts first
        if (first && lane == src) {
            out->valid = 1; out->px = g.x; out->py = g.y; out->kappa = kappa; out->v0 = v0; out->vnom = g.v; out->u0 = me.c;
        }
        first = false;
        const bool done = (total.w >= cst.u_cap) || (BACK ? (b <= 0) : (b + 64 >= n));
        if (done) break;
        b += BACK ? -64 : 64;
    }
    if (lane == 0) out->carry = total.c;
}

// wave-wide reductions; a ballot skips the butterfly when every lane holds the neutral element (most tiles have
// only one layer, no curvature and no flags)
__device__ __forceinline__ double wave_sum(double v)
{
    if (__ballot(v != 0.0) == 0ull) return 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_max0(double v)   // v >= 0
{
    if (__ballot(v != 0.0) == 0ull) return 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ long long wave_sum_i(int v)
{
    if (__ballot(v != 0) == 0ull) return 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int MINW>
__global__ __launch_bounds__(BLOCK, MINW) void k_plan_fused(const DevTile *__restrict__ tiles,
                                                      const DevField *__restrict__ fields,
                                                      const DevPrim *__restrict__ prims, DevConst cst, DevObstacles obs,
                                                      double *__restrict__ xo, double *__restrict__ yo,
                                                      double *__restrict__ ko, double *__restrict__ vo,
                                                      uint32_t *__restrict__ fso, TilePartial *__restrict__ partial)
{
    __shared__ FusedShared S;
    const DevTile tl = tiles[blockIdx.x];
    const DevField &f = fields[tl.field];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n = f.n_total, s = tl.start;
    const int cnt = tl.count;
    const double two_a = 2 * cst.a_lon;

    if (wave == 0) halo_wave<true>(f, prims, cst, s, &S.back);
    else if (wave == NWAVE - 1) halo_wave<false>(f, prims, cst, s + cnt, &S.fwd);

    // ---- 0. where this thread's run sits relative to the path's special indices (small ints from here on) ----
    const int j0 = tid * IPT;
    const int64_t i0 = s + j0;
    const int nvalid = min(max(cnt - j0, 0), IPT);
    const bool at_start = (i0 == 0);                                  // item 0 is the path's first point
    const int64_t rem_end = (n - 1) - i0, rem_seam = f.n_main - i0;
    const int k_end = (rem_end >= 0 && rem_end < IPT) ? (int)rem_end : 1000;     // item that is the path's last point
    const int k_seam = rem_seam < 0 ? -1 : (rem_seam >= IPT ? 1000 : (int)rem_seam);  // item with index n_main (first of layer 2)

    // ---- 1. generate this thread's 8 consecutive points -------------------------------------------
    // Fast path: the whole run lies on one straight primitive (a swath line or a headland straight), which is
    // the case for ~95 % of the threads at fine sampling: 8 x (cvt, mul, add).  Everything else -- turn points,
    // primitive boundaries, partial tiles -- goes through the out-of-line generic generator.
    double X[IPT + 2], Y[IPT + 2];
    uint32_t fs[IPT];
    bool straight = false;
    uint32_t run_fs = 0;
    if (nvalid == IPT) {
        double ax = 0, ay = 0, bx = 0, by = 0, sx = 0, sy = 0;
        int r0 = 0, nl = 0;
        bool rot = false;
        if (k_seam >= IPT) {                       // whole run in layer 1: decode from the tile's precomputed pass position
            const int per = f.n_line + f.n_turn;
            int off = tl.off0 + j0, idx = tl.idx0;
            if (off >= per) { const int q = off / per; off -= q * per; idx += q; }
            if (off + IPT <= f.n_line) {
                const int pi = f.reverse_order ? (f.P - 1 - idx) : idx;
                const bool go_left = f.start_from_right ? ((idx & 1) == 0) : ((idx & 1) == 1);
                ax = go_left ? f.lex : f.lsx; bx = go_left ? f.lsx : f.lex; sx = go_left ? -f.line_step : f.line_step;
                ay = by = f.min_y + (double)pi * f.W; sy = 0.0;
                r0 = off; nl = f.n_line; rot = f.rotated != 0;
                run_fs = FCPP_KIND_SWATH | ((uint32_t)pi << FCPP_INDEX_SHIFT);
                straight = true;
            }
        } else if (k_seam < 0 || k_seam == 0) {    // whole run in layer 2
            int a = f.prim_first, b = f.prim_first + f.prim_count - 1;
            while (a < b) {
                const int m = (a + b + 1) >> 1;
                if (prims[m].start <= i0) a = m; else b = m - 1;
            }
            const DevPrim &p = prims[a];
            const int64_t rr = i0 - p.start;
            if (p.kind == PRIM_LINSPACE && rr + IPT <= p.n) {
                ax = p.a[0]; bx = p.a[2]; sx = p.a[4]; ay = p.a[1]; by = p.a[3]; sy = p.a[5];
                r0 = (int)rr; nl = p.n; run_fs = p.fs;
                straight = true;
            }
        }
        if (straight) {
#pragma unroll
            for (int k = 0; k < IPT; ++k) {
                const int rk = r0 + k;
                double px = (double)rk * sx + ax, py = (double)rk * sy + ay;   // numpy.linspace: k*step + start
                if (rk == nl - 1) { px = bx; py = by; }                        // ... and the last sample is `stop`
                if (rot) {
                    const double tx = px - f.rot_cx, ty = py - f.rot_cy;
                    px = (tx * f.rot_cos - ty * f.rot_sin) + f.rot_cx;
                    py = (tx * f.rot_sin + ty * f.rot_cos) + f.rot_cy;
                }
                X[k + 1] = px; Y[k + 1] = py; fs[k] = run_fs;
            }
        }
    }
    if (!straight) {
#pragma unroll 1
        for (int k = 0; k < IPT; ++k) {
            GenOut g; g.x = g.y = 0; g.fs = 0;
            if (k < nvalid) g = gen_point_slow(&f, prims, i0 + k, &cst);
            // compile-time indices only (runtime-indexed register arrays would go to scratch)
#pragma unroll
            for (int q = 0; q < IPT; ++q) if (q == k) { X[q + 1] = g.x; Y[q + 1] = g.y; fs[q] = g.fs; }
        }
    }
    // end neighbours: previous thread's last point, next thread's first point
    X[0] = __shfl_up(X[IPT], 1); Y[0] = __shfl_up(Y[IPT], 1);
    X[IPT + 1] = __shfl_down(X[1], 1); Y[IPT + 1] = __shfl_down(Y[1], 1);
    if (lane == 0) { S.efx[wave] = X[1]; S.efy[wave] = Y[1]; }
    if (lane == 63) { S.elx[wave] = X[IPT]; S.ely[wave] = Y[IPT]; }
    __syncthreads();
    if (lane == 0) {
        if (wave > 0) { X[0] = S.elx[wave - 1]; Y[0] = S.ely[wave - 1]; }
        else { X[0] = S.back.px; Y[0] = S.back.py; }
    }
    if (lane == 63) {
        if (wave < NWAVE - 1) { X[IPT + 1] = S.efx[wave + 1]; Y[IPT + 1] = S.efy[wave + 1]; }
        else { X[IPT + 1] = S.fwd.px; Y[IPT + 1] = S.fwd.py; }
    }

    // ---- 2. segment lengths, couplings, curvature, clamp ---------------------------------------------
    // d[k] = |P(item k) - P(item k-1)|, w[k] = coupling of the sweeps across that segment; item -1 / item IPT are
    // the end neighbours.  Skipped steps (d < 1e-6, MLP:560-561 / 576-577) and the path ends cut the propagation.
    double d[IPT + 1], w[IPT + 1];
#pragma unroll
    for (int k = 0; k <= IPT; ++k) {
        const double dx = X[k + 1] - X[k], dy = Y[k + 1] - Y[k];
        // sqrt(fl(t*t)) == |t| exactly in IEEE arithmetic: axis-aligned steps need no square root
        d[k] = (dy == 0.0) ? fabs(dx) : ((dx == 0.0) ? fabs(dy) : sqrt(dx * dx + dy * dy));
        const bool cut = (d[k] < 1e-6) || (k == 0 && at_start) || (k == k_end + 1);
        w[k] = cut ? FCPP_INF : two_a * d[k];
    }
    double kap[IPT], c[IPT];
    int adj = 0;
    unsigned clmask = 0;    // items slowed by the curvature clamp
    const double ms_run = nominal_ms(run_fs, cst);
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        double kk = 0.0;
        if (k < nvalid && !(k == 0 && at_start) && k != k_end)
            kk = curv_chords(X[k + 1] - X[k], Y[k + 1] - Y[k], d[k], X[k + 2] - X[k + 1], Y[k + 2] - Y[k + 1], d[k + 1]);
        kap[k] = kk;
        double ms = straight ? ms_run : nominal_ms(fs[k], cst);
        if (kk > 1e-6) {
            bool cl;
            const double vc = clamped_speed(nominal_speed(fs[k], cst), kk, cst, cl);
            if (cl) { ms = vc / 3.6; clmask |= 1u << k; ++adj; }
        }
        c[k] = (k < nvalid) ? ms * ms : FCPP_INF;
    }

    // ---- 3. forward / backward sweeps as min-plus scans over registers ----------------------------
    // (items beyond a partial tile have c = +inf; they sit at the path end, where nothing propagates)
    Agg fa = { FCPP_INF, 0.0 }, ba = { FCPP_INF, 0.0 };
#pragma unroll
    for (int k = 0; k < IPT; ++k) { fa.c = fmin(c[k], fa.c + w[k]); fa.w += w[k]; }
#pragma unroll
    for (int k = IPT - 1; k >= 0; --k) { ba.c = fmin(c[k], ba.c + w[k + 1]); ba.w += w[k + 1]; }
    Agg fi = fa, bi = ba;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        Agg pf = { __shfl_up(fi.c, o), __shfl_up(fi.w, o) };
        Agg pb = { __shfl_down(bi.c, o), __shfl_down(bi.w, o) };
        if (lane >= o) fi = combine_after(pf, fi);
        if (lane + o < 64) bi = combine_after(pb, bi);
    }
    if (lane == 63) S.wf[wave] = fi;
    if (lane == 0) S.wb[wave] = bi;
    __syncthreads();
    Agg ef = { __shfl_up(fi.c, 1), __shfl_up(fi.w, 1) };
    if (lane == 0) ef = { FCPP_INF, 0.0 };
    Agg eb = { __shfl_down(bi.c, 1), __shfl_down(bi.w, 1) };
    if (lane == 63) eb = { FCPP_INF, 0.0 };
    Agg pre = { FCPP_INF, 0.0 }, suf = { FCPP_INF, 0.0 };
    for (int q = 0; q < wave; ++q) pre = combine_after(pre, S.wf[q]);
    for (int q = NWAVE - 1; q > wave; --q) suf = combine_after(suf, S.wb[q]);
    ef = combine_after(pre, ef);
    eb = combine_after(suf, eb);
    const double carry_f = S.back.carry, carry_b = S.fwd.carry;
    double uf = fmin(ef.c, carry_f + ef.w), ub = fmin(eb.c, carry_b + eb.w);
    double vf[IPT];   // first the swept u = (v/3.6)^2, then the final speed in km/h
#pragma unroll
    for (int k = 0; k < IPT; ++k) { uf = fmin(c[k], uf + w[k]); vf[k] = uf; }
#pragma unroll
    for (int k = IPT - 1; k >= 0; --k) { ub = fmin(c[k], ub + w[k + 1]); vf[k] = fmin(vf[k], ub); }
    const double b_first = ub;   // backward value at this thread's first item
    const double vn_run = nominal_speed(run_fs, cst);
    bool uniform = straight;     // every item (and the previous point) still runs at the run's nominal speed
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        if (vf[k] < c[k]) { vf[k] = sqrt(vf[k]) * 3.6; uniform = false; }     // slowed by a sweep
        else if ((clmask >> k) & 1u) {                                         // untouched: exactly the clamped value
            bool cl;
            vf[k] = clamped_speed(nominal_speed(fs[k], cst), kap[k], cst, cl);
            uniform = false;
        } else vf[k] = straight ? vn_run : nominal_speed(fs[k], cst);          // untouched: exactly the nominal value
    }

    // ---- 4. previous point's final v / kappa / nominal v (for the segment metrics) ------------------
    double vprev = __shfl_up(vf[IPT - 1], 1), kprev = __shfl_up(kap[IPT - 1], 1);
    uint32_t fsprev = __shfl_up(fs[IPT - 1], 1);
    if (lane == 63) { S.ev[wave] = vf[IPT - 1]; S.ek[wave] = kap[IPT - 1]; S.efs[wave] = fs[IPT - 1]; }
    __syncthreads();
    double vnprev = nominal_speed(fsprev, cst);
    if (lane == 0) {
        if (wave > 0) { vprev = S.ev[wave - 1]; kprev = S.ek[wave - 1]; vnprev = nominal_speed(S.efs[wave - 1], cst); }
        else if (S.back.valid) {
            // final value at s-1: forward part = carry_f, backward part = B(s) + w(s-1,s)
            const double up = fmin(carry_f, b_first + w[0]);
            vprev = (up < S.back.u0) ? sqrt(up) * 3.6 : S.back.v0;
            kprev = S.back.kappa; vnprev = S.back.vnom;
        }
    }

    // ---- 5. validator + metrics (MLP:1290-1311, 1373-1424; geofence / obstacles) --------------------
    double s_len[2] = { 0, 0 }, s_tpre[2] = { 0, 0 }, s_t[2] = { 0, 0 }, mk = 0, ma = 0, mj = 0;
    int nv = 0, nout = 0, nobs = 0;
    uniform = uniform && (vprev == vn_run) && (vnprev == vn_run);
    if (uniform) {
        // one layer, one speed, before and after the speed plan: sum the lengths, divide once
        // (item 0's segment does not count if it starts the path or the headland layer)
        double sd = (at_start || k_seam == 0) ? 0.0 : d[0];
#pragma unroll
        for (int k = 1; k < IPT; ++k) sd += d[k];
        const int layer = k_seam <= 0 ? 1 : 0;
        const double t = sd / fmax(ms_run, 0.1);
        s_len[layer] = sd; s_tpre[layer] = t; s_t[layer] = t;
    } else {
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            if (k < nvalid && !(k == 0 && at_start) && k != k_seam) {      // the seam main|headland belongs to neither layer
                const int layer = k > k_seam ? 1 : 0;
                const double vn_k = nominal_speed(fs[k], cst);
                const double vp = k == 0 ? vprev : vf[k - 1];
                const double vnp = k == 0 ? vnprev : nominal_speed(fs[k - 1], cst);
                s_len[layer] += d[k];
                const double tpre = d[k] / fmax(((vnp + vn_k) / 2) / 3.6, 0.1);
                s_tpre[layer] += tpre;
                s_t[layer] += (vp == vnp && vf[k] == vn_k) ? tpre : d[k] / fmax(((vp + vf[k]) / 2) / 3.6, 0.1);
            }
        }
    }
    const int ob0 = f.obs_first, ob1 = f.obs_first + f.obs_count;
    // convexity: a straight run whose two end points pass the geofence lies inside as a whole
    bool run_inside = false;
    if (straight) {
        bool out = false;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            out = out || (f.ex[e] * X[1] + f.ey[e] * Y[1] + f.eo[e] < -cst.geofence_tol)
                      || (f.ex[e] * X[IPT] + f.ey[e] * Y[IPT] + f.eo[e] < -cst.geofence_tol);
        run_inside = !out;
    }
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        if (k < nvalid) {
            const double px = X[k + 1], py = Y[k + 1];
            const double kp = k == 0 ? kprev : kap[k - 1];
            if (!(k == 0 && at_start) && k != k_end) {           // interior points of the path, MLP:1383-1391
                if (kap[k] > 0.0) {            // kappa == 0 contributes a_lat = 0: neither a maximum nor a violation
                    const double ms = vf[k] / 3.6, alat = ms * ms * kap[k];
                    mk = fmax(mk, kap[k]); ma = fmax(ma, alat);
                    if (alat > cst.a_lat) { ++nv; fs[k] |= FCPP_FLAG_ALAT; }
                }
                // |kappa_i - kappa_(i-1)| for i >= 2 (MLP:1404-1406)
                if (kap[k] != kp && !(i0 + k == 1)) mj = fmax(mj, fabs(kap[k] - kp));
            }
            if (!run_inside) {
                bool out = false;
#pragma unroll
                for (int e = 0; e < 4; ++e) out = out || (f.ex[e] * px + f.ey[e] * py + f.eo[e] < -cst.geofence_tol);
                if (out) { ++nout; fs[k] |= FCPP_FLAG_OUTSIDE; }
            }
            if (ob1 > ob0) {
                bool inside_any = false;
                for (int b = ob0; b < ob1 && !inside_any; ++b) {
                    const int64_t a0 = obs.offsets[b], a1 = obs.offsets[b + 1];
                    bool in = false;
                    for (int64_t q = a0, r = a1 - 1; q < a1; r = q++) {
                        const double xi = obs.x[q], yi = obs.y[q], xj = obs.x[r], yj = obs.y[r];
                        if (((yi > py) != (yj > py)) && (px < (xj - xi) * (py - yi) / (yj - yi) + xi)) in = !in;
                    }
                    inside_any = in;
                }
                if (inside_any) { ++nobs; fs[k] |= FCPP_FLAG_OBSTACLE; }
            }
        }
    }

    // ---- 6. coalesced SoA stores through the per-wave transposition buffer --------------------------
    {
        double *buf = S.tr[wave];
        const int64_t g0 = f.pt_off + s + wave * (64 * IPT);
        const int cw = min(max(cnt - wave * (64 * IPT), 0), 64 * IPT);
        auto put = [&](double *__restrict__ dst, const double *vals) {
#pragma unroll
            for (int k = 0; k < IPT; ++k) buf[lidx(lane * IPT + k)] = vals[k];
            wave_sync();
#pragma unroll
            for (int m = 0; m < IPT; ++m) {
                const int p = lane + 64 * m;
                if (p < cw) dst[g0 + p] = buf[lidx(p)];
            }
            wave_sync();
        };
        put(xo, &X[1]);
        put(yo, &Y[1]);
        put(ko, kap);
        put(vo, vf);
        uint32_t *b32 = reinterpret_cast<uint32_t *>(buf);
#pragma unroll
        for (int k = 0; k < IPT; ++k) b32[2 * lidx(lane * IPT + k)] = fs[k];
        wave_sync();
#pragma unroll
        for (int m = 0; m < IPT; ++m) {
            const int p = lane + 64 * m;
            if (p < cw) fso[g0 + p] = b32[2 * lidx(p)];
        }
    }

    // ---- 7. fixed-shape block reduction of the metrics ---------------------------------------------
    double dv[9] = { s_len[0], s_tpre[0], s_t[0], s_len[1], s_tpre[1], s_t[1], mk, ma, mj };
    long long iv[4];
#pragma unroll
    for (int k = 0; k < 6; ++k) dv[k] = wave_sum(dv[k]);
#pragma unroll
    for (int k = 6; k < 9; ++k) dv[k] = wave_max0(dv[k]);
    iv[0] = wave_sum_i(nv); iv[1] = wave_sum_i(nout); iv[2] = wave_sum_i(nobs); iv[3] = wave_sum_i(adj);
    if (lane == 0) {
        for (int k = 0; k < 9; ++k) S.R.d[wave][k] = dv[k];
        for (int k = 0; k < 4; ++k) S.R.i[wave][k] = iv[k];
    }
    __syncthreads();
    if (tid == 0) {
        double a[9]; long long b[4];
        for (int k = 0; k < 9; ++k) a[k] = S.R.d[0][k];
        for (int k = 0; k < 4; ++k) b[k] = S.R.i[0][k];
        for (int wv = 1; wv < NWAVE; ++wv) {
            for (int k = 0; k < 6; ++k) a[k] += S.R.d[wv][k];
            for (int k = 6; k < 9; ++k) a[k] = fmax(a[k], S.R.d[wv][k]);
            for (int k = 0; k < 4; ++k) b[k] += S.R.i[wv][k];
        }
        TilePartial tp;
        tp.main_len = a[0]; tp.main_time_pre = a[1]; tp.main_time = a[2];
        tp.head_len = a[3]; tp.head_time_pre = a[4]; tp.head_time = a[5];
        tp.max_kappa = a[6]; tp.max_alat = a[7]; tp.max_jump = a[8];
        tp.n_viol = b[0]; tp.n_outside = b[1]; tp.n_in_obstacle = b[2]; tp.n_adjusted = b[3];
        partial[blockIdx.x] = tp;
    }
}

int launch_plan_fused(hipStream_t st, int variant, int64_t n_tiles, const DevTile *tiles, const DevField *fields,
                      const DevPrim *prims, const DevConst &cst, const DevObstacles &obs, double *x, double *y,
                      double *kappa, double *v, uint32_t *fs, TilePartial *partial)
{
    if (n_tiles <= 0) return 0;
    // variant = register budget: minimum waves per SIMD the compiler must allow (3 -> <=168 VGPRs, 4 -> <=128, 2 -> <=256)
    if (variant == 4)
        hipLaunchKernelGGL(k_plan_fused<4>, dim3((unsigned)n_tiles), dim3(BLOCK), 0, st, tiles, fields, prims, cst, obs, x, y,
                           kappa, v, fs, partial);
    else if (variant == 2)
        hipLaunchKernelGGL(k_plan_fused<2>, dim3((unsigned)n_tiles), dim3(BLOCK), 0, st, tiles, fields, prims, cst, obs, x, y,
                           kappa, v, fs, partial);
    else
        hipLaunchKernelGGL(k_plan_fused<3>, dim3((unsigned)n_tiles), dim3(BLOCK), 0, st, tiles, fields, prims, cst, obs, x, y,
                           kappa, v, fs, partial);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

}  // namespace fcpp
