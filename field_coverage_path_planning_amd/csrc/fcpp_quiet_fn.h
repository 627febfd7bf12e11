// fcpp_quiet_fn.h -- closed-form ("quiet") runs as device functions: the chunk writer of k_plan_quiet and the run statistics, shared
// by the streaming kernel k_plan_quiet (fcpp_fused.hip) and the statistics reduction (fcpp_kernels.hip).
#pragma once
#include "fcpp_pointfn.h"

namespace fcpp {

// Stores of the streaming kernel, 16 bytes per lane.  NT: non-temporal -- measured per kernel (one library per variant, fresh processes
// in turn on one box): the span instance (KINDS 16) at the reference's sampling loses 15 % with it (headline 31.6 -> 36.4 us), the dense
// instance (KINDS 14) neither gains nor loses (cfg2 at 0.1 m 5.62 vs 5.59 ms, cfg3 0.355 vs 0.358 ms): plain stores in both.
#ifndef FCPP_DENSE_NT
#define FCPP_DENSE_NT 0      // (1: a build for the A/B)
#endif
typedef double st_pair_f64 __attribute__((ext_vector_type(2)));
typedef uint32_t st_pair_u32 __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ void st2(double *ptr, double a, double b)
{
    st_pair_f64 v = { a, b };
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<st_pair_f64 *>(ptr));
    else *reinterpret_cast<st_pair_f64 *>(ptr) = v;
}
template <bool NT>
__device__ __forceinline__ void st2(uint32_t *ptr, uint32_t a, uint32_t b)
{
    st_pair_u32 v = { a, b };
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<st_pair_u32 *>(ptr));
    else *reinterpret_cast<st_pair_u32 *>(ptr) = v;
}
// ---- quiet runs (found by the host tiler): stretches of a straight primitive whose points, and everything within reach of
// the sweeps, lie on that primitive.  Then kappa = 0, nobody is clamped and u = u_nominal everywhere: closed-form results, no
// neighbours, no scan, no LDS.  The kernel is pure HBM streaming, so what matters is the shape of its stores: a run is cut into
// chunks on 512-point boundaries of the batch arrays, one chunk per wavefront, every store instruction of a full chunk writes
// one ALIGNED KiB (16 bytes per lane) -- measured 15-25 % faster than the same bytes through tiles that start anywhere in a
// cache line (tools/micro/stream_probe.hip), and far less sensitive to where the arrays happen to live in device memory.
// aligned pair store of one lane: points (g, g + 1) of the batch arrays, g even
template <bool NT>
__device__ __forceinline__ void store_pair(bool has0, bool has1, int64_t g, double px0, double px1, double py0, double py1, double k0,
                                           double k1, double v, uint32_t f0, uint32_t f1, double *__restrict__ xo,
                                           double *__restrict__ yo, double *__restrict__ ko, double *__restrict__ vo,
                                           uint32_t *__restrict__ fso)
{
    if (has0 && has1) {
        st2<NT>(xo + g, px0, px1);
        st2<NT>(yo + g, py0, py1);
        st2<NT>(ko + g, k0, k1);
        st2<NT>(vo + g, v, v);
        st2<NT>(fso + g, f0, f1);
    } else if (has0) {
        xo[g] = px0; yo[g] = py0; ko[g] = k0; vo[g] = v; fso[g] = f0;
    } else if (has1) {
        xo[g + 1] = px1; yo[g + 1] = py1; ko[g + 1] = k1; vo[g + 1] = v; fso[g + 1] = f1;
    }
}

// Curvature of the first point of a swath line that follows a quiet U-turn: its stencil spans the jump back from the turn's
// last sample (MLP:776-780: the reference's turn ends on the field edge, the next line starts R inside).  Same arithmetic
// as the general kernel's stencil.  Also returns the length of the jump.
__device__ __forceinline__ double line_start_curvature(const DevField &q, const DevConst &cst, int idx, double &jump_len)
{
    const int per = q.n_line + q.n_turn;
    double tx, ty, x0, y0, x1, y1;
    uint32_t fw;
    eval_main(q, cst, idx - 1, per - 1, tx, ty, fw);
    eval_main(q, cst, idx, 0, x0, y0, fw);
    eval_main(q, cst, idx, 1, x1, y1, fw);
    const double dx1 = x0 - tx, dy1 = y0 - ty, dx2 = x1 - x0, dy2 = y1 - y0;
    jump_len = seg_len(dx1, dy1);
    return curv_chords(dx1, dy1, jump_len, dx2, dy2, seg_len(dx2, dy2));
}

// the batch's U-turn template into the wavefront's LDS copy: x[TMPL_LDS] | y[TMPL_LDS] | kappa[TMPL_LDS]
__device__ __forceinline__ void stage_turn_template(const DevConst &cst, double *__restrict__ tmpl_lds)
{
    const int lane = threadIdx.x & 63;
    if (lane < cst.tmpl_n) {
        const double2 t = cst.tmpl_u[lane];
        tmpl_lds[lane] = t.x; tmpl_lds[TMPL_LDS + lane] = t.y; tmpl_lds[2 * TMPL_LDS + lane] = cst.tmpl_u_dk[lane].y;
    }
    wave_sync();
}

// OBS: the batch has obstacle polygons (without them the polygon tests are compiled out: registers and code of a streaming kernel).
// KINDS: bit k set = chunks of kind k may occur in this instance (1 swath line, 2 headland straight, 3 U-turn, 4 layer-1 span).
// The kinds are compiled into separate kernels where that saves registers (occupancy of a pure streaming kernel).
template <int KINDS, bool STAGED, bool OBS>
__device__ __forceinline__ void quiet_tile(const DevTile &tl, const DevField *__restrict__ fg, const DevPrim *__restrict__ prims,
                                           const DevConst &cst, const DevObstacles &obs, double *my_lds /* 2*OBS_LDS_VERTS doubles of this wave */,
                                           double *__restrict__ tmpl_lds /* STAGED: 3*TMPL_LDS doubles of this wave */,
                                           double *__restrict__ xo, double *__restrict__ yo, double *__restrict__ ko,
                                           double *__restrict__ vo, uint32_t *__restrict__ fso,
                                           unsigned long long *sink_outside, unsigned long long *sink_obstacle)
{
    // sink_*: where the chunk's flag counts are added (integer atomics; global memory or LDS)
    const int lane = threadIdx.x & 63;
    const DevField &q = *fg;
    const int cnt = tl.count;    // <= TILE_POINTS; a chunk that starts on an odd index ends on a 512 boundary (<= 511 points)
    const int64_t g0 = q.pt_off + tl.start;
    // Aligned pairs: pair m covers the chunk-local points (2m - odd, 2m - odd + 1), whose global index is even whatever the
    // parity of the chunk's first global index (odd = 1: the chunk's first point is the second half of pair 0).
    const int odd = (int)(g0 & 1);
    constexpr bool NT = FCPP_DENSE_NT && KINDS != 16;                         // non-temporal stores in the dense instance only (st2)
    const double ntol = -cst.geofence_tol;
    auto outside = [&](double px, double py) -> bool {
        bool out = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) out = out | (q.ex[e] * px + q.ey[e] * py + q.eo[e] < ntol);
        return out;
    };
    int nout = 0, nobs = 0;
    if ((KINDS & 16) && (KINDS == 16 || tl.quiet == 4)) {
        // ---- a span of layer 1 whose passes (swath line + U-turn) are all closed form.  Point i of the span is (pass i / per,
        // offset i % per): a line sample k * step + start (kappa 0; the first point of a line after a turn has the curvature of
        // the jump stencil), or a turn sample = template + translation (mirror), curvature = the shape's own, nominal speeds.
        // The field's constants are fetched ONCE, into scalar registers the compiler must treat as opaque (FCPP_PIN): it would
        // otherwise re-load them from the descriptor after every group of stores (it cannot hold ~40 scalar values and prefers
        // re-loading to spilling), and every such re-load is a scalar-cache round trip the wave waits out -- eight samples times
        // four or five waits a wave.  Pinned values that do not fit are spilled to lanes of a vector register, which costs an
        // instruction, not a memory access.
        int per = q.n_line + q.n_turn, nl = q.n_line, last = q.n_turn - 1, n_pass = q.P, idx_base = tl.idx0;
        int rev = q.reverse_order, sfr = q.start_from_right, rotated = q.rotated;
        const bool arc = q.turn_model == FCPP_TURN_ARC;
        double xr = arc ? q.max_x : (q.max_x - q.R), xl = arc ? q.min_x : (q.min_x + q.R);
        double k_last = cst.turn_kappa_last[q.reverse_order ? 1 : 0];
        double k_start = cst.field_junc[tl.field].x;     // the same for every line of the field (mirror images)
        double min_y = q.min_y, Wd = q.W, lstep = q.line_step, lex = q.lex, lsx = q.lsx;
        double rc = q.rot_cos, rs = q.rot_sin, rcx = q.rot_cx, rcy = q.rot_cy;
        double v_work = cst.v_work, v_turn = cst.v_turn;
        double e0x = q.ex[0], e0y = q.ey[0], e0o = q.eo[0], e1x = q.ex[1], e1y = q.ey[1], e1o = q.eo[1];
        double e2x = q.ex[2], e2y = q.ey[2], e2o = q.eo[2], e3x = q.ex[3], e3y = q.ey[3], e3o = q.eo[3];
        double ntl = ntol;
        int n_obs = q.obs_count;
        const bool fence = q.span_inside == 0;      // (wave-uniform: a span that lies inside the geofence as a whole is not tested point by point)
        FCPP_PIN(per); FCPP_PIN(nl); FCPP_PIN(last); FCPP_PIN(n_pass); FCPP_PIN(idx_base); FCPP_PIN(rev); FCPP_PIN(sfr); FCPP_PIN(rotated);
        FCPP_PIN(xr); FCPP_PIN(xl); FCPP_PIN(k_last); FCPP_PIN(k_start); FCPP_PIN(min_y); FCPP_PIN(Wd); FCPP_PIN(lstep); FCPP_PIN(lex); FCPP_PIN(lsx);
        FCPP_PIN(v_work); FCPP_PIN(v_turn); FCPP_PIN(n_obs);
        // ... and the rotation and the geofence edges (17 values that only ever meet vector operands) in VECTOR registers: the kernel
        // runs four waves per SIMD (launch_plan_quiet), which leaves each wave more than 100 of them
        FCPP_PIN_V(rc); FCPP_PIN_V(rs); FCPP_PIN_V(rcx); FCPP_PIN_V(rcy);
        FCPP_PIN_V(e0x); FCPP_PIN_V(e0y); FCPP_PIN_V(e0o); FCPP_PIN_V(e1x); FCPP_PIN_V(e1y); FCPP_PIN_V(e1o);
        FCPP_PIN_V(e2x); FCPP_PIN_V(e2y); FCPP_PIN_V(e2o); FCPP_PIN_V(e3x); FCPP_PIN_V(e3y); FCPP_PIN_V(e3o); FCPP_PIN_V(ntl);
        auto outside_s = [&](double px, double py) -> bool {
            return (e0x * px + e0y * py + e0o < ntl) | (e1x * px + e1y * py + e1o < ntl) | (e2x * px + e2y * py + e2o < ntl) |
                   (e3x * px + e3y * py + e3o < ntl);
        };
        // The turn template goes through LDS (x, y, kappa of its <= TMPL_LDS samples, one copy per wavefront): vector loads inside
        // the store loop would share the wave's memory counter with its stores -- waiting for a template sample then means waiting
        // for every store issued before it to be acknowledged by the memory system -- while LDS reads have a counter of their own.
        // STAGED is chosen per launch: the template is the batch's (cst.tmpl_n samples), spans exist only for fields that use it.
        // (staged by the kernel before it fetches its chunk descriptor: stage_turn_template)
        auto sample = [&](int dq, int off, double &px, double &py, double &kp, double &v, uint32_t &fw) {
            const int idx = idx_base + dq;
            const int pi = rev ? (n_pass - 1 - idx) : idx;
            const double y = min_y + (double)pi * Wd;
            const bool go_left = sfr ? ((idx & 1) == 0) : ((idx & 1) == 1);
            if (off < nl) {
                px = go_left ? ((double)off * -lstep + lex) : ((double)off * lstep + lsx);
                if (off == nl - 1) px = go_left ? lsx : lex;
                py = y;
                kp = (off == 0 && idx > 0) ? k_start : 0.0;
                v = v_work;
                fw = FCPP_KIND_SWATH | ((uint32_t)pi << FCPP_INDEX_SHIFT);
            } else {
                const int c = off - nl;
                double tx, ty, tk;
                if (STAGED) { tx = tmpl_lds[c]; ty = tmpl_lds[TMPL_LDS + c]; tk = tmpl_lds[2 * TMPL_LDS + c]; }
                else { const double2 t = cst.tmpl_u[c]; tx = t.x; ty = t.y; tk = cst.tmpl_u_dk[c].y; }
                const bool turn_right = !go_left;
                px = arc ? (turn_right ? (xr - tx) : (xl + tx)) : (turn_right ? (xr + tx) : (xl - tx));
                py = y + ty;
                kp = c == last ? k_last : tk;
                v = v_turn;
                fw = FCPP_KIND_UTURN | ((uint32_t)pi << FCPP_INDEX_SHIFT);
            }
            if (rotated) {                                     // rotate_back (fcpp_pointfn.h) on the pinned copies
                const double tx = px - rcx, ty = py - rcy;
                px = (tx * rc - ty * rs) + rcx;
                py = (tx * rs + ty * rc) + rcy;
            }
        };
        // (pass, offset) of the pair's second point by one division; its first point is the point before it, and the pairs of the
        // following rounds lie 128 points further on: both by carry, not by division
        const unsigned uper = (unsigned)per, step_q = 128u / uper, step_r = 128u - step_q * uper;      // wave-uniform
        const unsigned a1 = (unsigned)(tl.off0 + 2 * lane - odd + 1);
        int dq1 = (int)(a1 / uper), off1 = (int)(a1 - (unsigned)dq1 * uper);
#pragma unroll
        for (int k = 0; k < TILE_POINTS / 128; ++k) {
            // (a round = 128 points; a chunk that ends before this round -- the first and the last chunk of a span, a third of the headline's
            // -- skips it: wave-uniform.  Rounds 2-4 ran all four rounds of every chunk, storing nothing in the ones beyond its end.)
            if (k > 0 && 128 * k - odd >= cnt) break;
            const int j = 2 * (lane + 64 * k) - odd;
            const bool has0 = j >= 0 && j < cnt, has1 = j + 1 < cnt;
            double px0, py0, k0, v0, px1, py1, k1, v1;
            uint32_t f0, f1;
            const int dq0 = off1 == 0 ? dq1 - 1 : dq1, off0 = off1 == 0 ? per - 1 : off1 - 1;
            sample(dq0, off0, px0, py0, k0, v0, f0);     // (lanes beyond the chunk decode points beyond it: arithmetic only, never stored)
            sample(dq1, off1, px1, py1, k1, v1, f1);
            off1 += (int)step_r; dq1 += (int)step_q;
            if (off1 >= per) { off1 -= per; ++dq1; }
            if (fence) {
                const bool o0 = has0 && outside_s(px0, py0), o1 = has1 && outside_s(px1, py1);  // turns may leave the field: every point is tested
                nout += (o0 ? 1 : 0) + (o1 ? 1 : 0);
                f0 |= o0 ? FCPP_FLAG_OUTSIDE : 0u;
                f1 |= o1 ? FCPP_FLAG_OUTSIDE : 0u;
            }
            if (OBS && n_obs > 0) {
                // bounding box of the wave's points of this pass, then the culled polygon tests
                double mnx = has0 ? px0 : (has1 ? px1 : FCPP_INF), mxx = has0 ? px0 : (has1 ? px1 : -FCPP_INF);
                double mny = has0 ? py0 : (has1 ? py1 : FCPP_INF), mxy = has0 ? py0 : (has1 ? py1 : -FCPP_INF);
                if (has1) { mnx = fmin(mnx, px1); mxx = fmax(mxx, px1); mny = fmin(mny, py1); mxy = fmax(mxy, py1); }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    mnx = fmin(mnx, __shfl_xor(mnx, o)); mny = fmin(mny, __shfl_xor(mny, o));
                    mxx = fmax(mxx, __shfl_xor(mxx, o)); mxy = fmax(mxy, __shfl_xor(mxy, o));
                }
                const double ox[2] = { has0 ? px0 : px1, px1 }, oy[2] = { has0 ? py0 : py1, py1 };
                const unsigned m = obstacle_mask<2>(obs, q.obs_first, q.obs_first + q.obs_count, my_lds, mnx, mny, mxx, mxy, ox, oy,
                                                    has1 ? 2 : (has0 ? 1 : 0));
                const bool b0 = has0 && (m & 1u), b1 = has1 && (m & 2u);
                nobs += (b0 ? 1 : 0) + (b1 ? 1 : 0);
                f0 |= b0 ? FCPP_FLAG_OBSTACLE : 0u;
                f1 |= b1 ? FCPP_FLAG_OBSTACLE : 0u;
            }
            // (a pair may straddle a line / turn boundary: each point carries its own speed)
            const int64_t g = g0 + j;
            if (has0 && has1) {
                st2<NT>(xo + g, px0, px1);
                st2<NT>(yo + g, py0, py1);
                st2<NT>(ko + g, k0, k1);
                st2<NT>(vo + g, v0, v1);
                st2<NT>(fso + g, f0, f1);
            } else if (has0) {
                xo[g] = px0; yo[g] = py0; ko[g] = k0; vo[g] = v0; fso[g] = f0;
            } else if (has1) {
                xo[g + 1] = px1; yo[g + 1] = py1; ko[g + 1] = k1; vo[g + 1] = v1; fso[g + 1] = f1;
            }
        }
    } else if ((KINDS & 8) && (KINDS == 8 || tl.quiet == 3)) {
        // ---- U-turn in closed form: sample = template + translation (mirror), curvature = the shape's own, nominal turn speed ----
        const int idx = tl.idx0;
        const int pi = q.reverse_order ? (q.P - 1 - idx) : idx;
        const double y = q.min_y + (double)pi * q.W;
        const bool go_left = q.start_from_right ? ((idx & 1) == 0) : ((idx & 1) == 1);
        const bool turn_right = !go_left, arc = q.turn_model == FCPP_TURN_ARC;
        const double xr = arc ? q.max_x : (q.max_x - q.R), xl = arc ? q.min_x : (q.min_x + q.R);
        const uint32_t fw = FCPP_KIND_UTURN | ((uint32_t)pi << FCPP_INDEX_SHIFT);
        const double k_last = cst.turn_kappa_last[q.reverse_order ? 1 : 0];
        const int last = q.n_turn - 1;
        auto sample = [&](int tk, double &px, double &py, double &kp) {
            const int c = min(max(tk, 0), last);
            const double2 t = cst.tmpl_u[c];
            px = arc ? (turn_right ? (xr - t.x) : (xl + t.x)) : (turn_right ? (xr + t.x) : (xl - t.x));
            py = y + t.y;
            if (q.rotated) rotate_back(q, px, py);
            kp = c == last ? k_last : cst.tmpl_u_dk[c].y;
        };
#pragma unroll
        for (int k = 0; k < TILE_POINTS / 128; ++k) {
            if (k > 0 && 128 * k - odd >= cnt) break;          // (the chunk ends before this round)
            const int j = 2 * (lane + 64 * k) - odd;
            const bool has0 = j >= 0 && j < cnt, has1 = j + 1 < cnt;
            double px0, py0, k0, px1, py1, k1;
            sample(tl.off0 + j, px0, py0, k0);
            sample(tl.off0 + j + 1, px1, py1, k1);
            uint32_t f0 = fw, f1 = fw;
            const bool o0 = has0 && outside(px0, py0), o1 = has1 && outside(px1, py1);      // a turn may leave the field: every point is tested
            nout += (o0 ? 1 : 0) + (o1 ? 1 : 0);
            f0 |= o0 ? FCPP_FLAG_OUTSIDE : 0u;
            f1 |= o1 ? FCPP_FLAG_OUTSIDE : 0u;
            if (OBS && q.obs_count > 0) {
                // bounding box of the wave's points of this pass, then the culled polygon tests
                double mnx = has0 ? px0 : (has1 ? px1 : FCPP_INF), mxx = has0 ? px0 : (has1 ? px1 : -FCPP_INF);
                double mny = has0 ? py0 : (has1 ? py1 : FCPP_INF), mxy = has0 ? py0 : (has1 ? py1 : -FCPP_INF);
                if (has1) { mnx = fmin(mnx, px1); mxx = fmax(mxx, px1); mny = fmin(mny, py1); mxy = fmax(mxy, py1); }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    mnx = fmin(mnx, __shfl_xor(mnx, o)); mny = fmin(mny, __shfl_xor(mny, o));
                    mxx = fmax(mxx, __shfl_xor(mxx, o)); mxy = fmax(mxy, __shfl_xor(mxy, o));
                }
                const double ox[2] = { has0 ? px0 : px1, px1 }, oy[2] = { has0 ? py0 : py1, py1 };
                const unsigned m = obstacle_mask<2>(obs, q.obs_first, q.obs_first + q.obs_count, my_lds, mnx, mny, mxx, mxy, ox, oy,
                                                    has1 ? 2 : (has0 ? 1 : 0));
                const bool b0 = has0 && (m & 1u), b1 = has1 && (m & 2u);
                nobs += (b0 ? 1 : 0) + (b1 ? 1 : 0);
                f0 |= b0 ? FCPP_FLAG_OBSTACLE : 0u;
                f1 |= b1 ? FCPP_FLAG_OBSTACLE : 0u;
            }
            store_pair<NT>(has0, has1, g0 + j, px0, px1, py0, py1, k0, k1, cst.v_turn, f0, f1, xo, yo, ko, vo, fso);
        }
    } else if (KINDS & 6) {
        double ax, ay, sx, sy, bx, by, vnom;      // numpy.linspace(a, b, n): sample k = k * step + a, the last one is b itself
        int n_lin;
        uint32_t fw;
        bool rot;
        if (tl.quiet == 1) {        // swath line of layer 1: idx0 = pass position, off0 = offset in the pass
            const int idx = tl.idx0;
            const int pi = q.reverse_order ? (q.P - 1 - idx) : idx;
            const bool go_left = q.start_from_right ? ((idx & 1) == 0) : ((idx & 1) == 1);
            ax = go_left ? q.lex : q.lsx; bx = go_left ? q.lsx : q.lex; sx = go_left ? -q.line_step : q.line_step;
            ay = by = q.min_y + (double)pi * q.W; sy = 0.0;
            n_lin = q.n_line;
            rot = q.rotated != 0;
            fw = FCPP_KIND_SWATH | ((uint32_t)pi << FCPP_INDEX_SHIFT);
            vnom = cst.v_work;
        } else {                    // headland straight: idx0 = primitive, off0 = offset in it
            const DevPrim &p = prims[tl.idx0];
            ax = p.a[0]; ay = p.a[1]; bx = p.a[2]; by = p.a[3]; sx = p.a[4]; sy = p.a[5];
            n_lin = p.n;
            rot = false;
            fw = p.fs; vnom = p.v_nom;
        }
        auto lin = [&](int r, double &px, double &py) {
            px = (double)r * sx + ax; py = (double)r * sy + ay;
            if (r == n_lin - 1) { px = bx; py = by; }
        };
        // a line that starts right after a quiet U-turn: its first point's curvature stencil spans the jump from the turn's end
        const bool has_start = tl.quiet == 1 && tl.off0 == 0 && tl.idx0 > 0;     // wave-uniform
        const double k_start = has_start ? cst.field_junc[tl.field].x : 0.0;
        // geofence by convexity: both end points inside => the whole segment is inside
        double ex0, ey0, ex1, ey1;
        lin(tl.off0, ex0, ey0);
        lin(tl.off0 + cnt - 1, ex1, ey1);
        if (rot) { rotate_back(q, ex0, ey0); rotate_back(q, ex1, ey1); }
        const bool ends_out = outside(ex0, ey0) || outside(ex1, ey1);
        const double bminx = fmin(ex0, ex1), bmaxx = fmax(ex0, ex1), bminy = fmin(ey0, ey1), bmaxy = fmax(ey0, ey1);
#pragma unroll
        for (int k = 0; k < TILE_POINTS / 128; ++k) {
            if (k > 0 && 128 * k - odd >= cnt) break;          // (the chunk ends before this round: wave-uniform)
            const int j = 2 * (lane + 64 * k) - odd;
            const bool has0 = j >= 0 && j < cnt, has1 = j + 1 < cnt;     // (no exit of single lanes: the obstacle test below is wave-wide)
            double px0, py0, px1, py1;
            lin(tl.off0 + j, px0, py0);
            lin(tl.off0 + j + 1, px1, py1);
            if (rot) { rotate_back(q, px0, py0); rotate_back(q, px1, py1); }
            uint32_t f0 = fw, f1 = fw;
            if (ends_out) {      // wave-uniform
                const bool o0 = has0 && outside(px0, py0), o1 = has1 && outside(px1, py1);
                nout += (o0 ? 1 : 0) + (o1 ? 1 : 0);
                f0 |= o0 ? FCPP_FLAG_OUTSIDE : 0u;
                f1 |= o1 ? FCPP_FLAG_OUTSIDE : 0u;
            }
            if (OBS && q.obs_count > 0) {
                // lanes without a valid first point test only their second one (or nothing)
                const double ox[2] = { has0 ? px0 : px1, px1 }, oy[2] = { has0 ? py0 : py1, py1 };
                const unsigned m = obstacle_mask<2>(obs, q.obs_first, q.obs_first + q.obs_count, my_lds, bminx, bminy, bmaxx, bmaxy,
                                                    ox, oy, has1 ? 2 : (has0 ? 1 : 0));
                const bool b0 = has0 && (m & 1u), b1 = has1 && (m & 2u);
                nobs += (b0 ? 1 : 0) + (b1 ? 1 : 0);
                f0 |= b0 ? FCPP_FLAG_OBSTACLE : 0u;
                f1 |= b1 ? FCPP_FLAG_OBSTACLE : 0u;
            }
            // chunk-local point 0 is the line's first point only in the run's first chunk (has_start)
            const double k0 = (has_start && j == 0) ? k_start : 0.0, k1 = (has_start && j + 1 == 0) ? k_start : 0.0;
            store_pair<NT>(has0, has1, g0 + j, px0, px1, py0, py1, k0, k1, vnom, f0, f1, xo, yo, ko, vo, fso);
        }
    }
    if (__ballot(nout | nobs)) {   // integer counts: the order of the additions does not matter
        const long long io = wave_sum_i(nout), ib = wave_sum_i(nobs);
        if (lane == 0 && io) atomicAdd(sink_outside, (unsigned long long)io);
        if (lane == 0 && ib) atomicAdd(sink_obstacle, (unsigned long long)ib);
    }
}

// length / time statistics of one quiet run in closed form (count x step; a span: per pass the line's steps and the turn shape's own
// totals).  The flag counts of the run's points are not part of it (k_plan_quiet counts them while storing).
// what the closed-form run statistics need of a field: fetched by the reduction as soon as it knows its path (the field IS the path),
// beside the entry lists -- through the tile record it would be one more dependent round trip per entry
struct FieldStatView {
    int n_line, n_turn, reverse_order;
    double line_step;
    int64_t n_main;
    double2 junc;
    __device__ __forceinline__ void load(const DevField *__restrict__ fields, const DevConst &cst, int64_t field)
    {
        const DevField &q = fields[field];
        n_line = q.n_line; n_turn = q.n_turn; reverse_order = q.reverse_order; line_step = q.line_step; n_main = q.n_main;
        junc = cst.field_junc ? cst.field_junc[field] : make_double2(0.0, 0.0);
    }
};

__device__ __forceinline__ TilePartial quiet_run_partial(const DevRun &run, const DevTile &tl, const FieldStatView &q,
                                                         const DevPrim *__restrict__ prims, const DevConst &cst)
{
    TilePartial tp;
    tp.main_len = tp.main_time_pre = tp.main_time = tp.head_len = tp.head_time_pre = tp.head_time = 0.0;
    tp.max_kappa = tp.max_alat = tp.max_jump = 0.0;
    tp.n_viol = 0; tp.n_outside = 0; tp.n_in_obstacle = 0; tp.n_adjusted = 0;
    if (tl.quiet == 4) {
        // a span of whole passes: per pass the line's n_line - 1 steps and the turn shape's own totals (the turn's first segment has
        // length 0: it starts on the line's end); every pass but the path's first starts with the jump from the previous turn
        // (per-lane selects: a per-lane index into the kernel-argument arrays would copy them to scratch memory)
        const bool rv = q.reverse_order != 0;
        const double tmk = rv ? cst.turn_max_kappa[1] : cst.turn_max_kappa[0], tmj = rv ? cst.turn_max_jump[1] : cst.turn_max_jump[0];
        const double tkl = rv ? cst.turn_kappa_last[1] : cst.turn_kappa_last[0];
        const int per = q.n_line + q.n_turn;
        const double n_pass = (double)(run.count / per), n_jump = n_pass - (tl.idx0 == 0 ? 1.0 : 0.0);
        const double line_len = (double)(q.n_line - 1) * fabs(q.line_step);
        const double2 junc = q.junc;
        const double jl = n_jump > 0.0 ? junc.y : 0.0, k0 = n_jump > 0.0 ? junc.x : 0.0;
        tp.main_len = n_pass * (line_len + cst.turn_len) + n_jump * jl;
        tp.main_time_pre = tp.main_time = n_pass * (line_len / fmax(cst.ms_work, 0.1) + cst.turn_time) +
                                          n_jump * (jl / fmax(((cst.v_turn + cst.v_work) / 2) / 3.6, 0.1));     // MLP:1305-1309
        tp.max_kappa = fmax(tmk, k0);
        tp.max_alat = fmax(cst.ms_turn * cst.ms_turn * tmk, cst.ms_work * cst.ms_work * k0);
        tp.max_jump = fmax(tmj, n_jump > 0.0 ? fmax(fabs(k0 - tkl), k0) : 0.0);
    } else     if (tl.quiet == 3) {            // a whole U-turn: the shape's own totals (its first segment has length 0: the turn starts on the line's end)
        const bool rv = q.reverse_order != 0;
        const double tmk = rv ? cst.turn_max_kappa[1] : cst.turn_max_kappa[0];
        tp.main_len = cst.turn_len; tp.main_time_pre = tp.main_time = cst.turn_time;
        tp.max_kappa = tmk; tp.max_alat = cst.ms_turn * cst.ms_turn * tmk;
        tp.max_jump = rv ? cst.turn_max_jump[1] : cst.turn_max_jump[0];
    } else {
        double step_len, msnom;
        int layer;
        if (tl.quiet == 1) { step_len = fabs(q.line_step); msnom = cst.ms_work; layer = 0; }
        else {
            const DevPrim &p = prims[tl.idx0];
            const double sx = p.a[4], sy = p.a[5];
            step_len = (sy == 0.0) ? fabs(sx) : ((sx == 0.0) ? fabs(sy) : sqrt(sx * sx + sy * sy));
            msnom = nominal_ms(p.fs, cst); layer = tl.start >= q.n_main ? 1 : 0;      // (a straight of layer 1: obstacle-aware swaths)
        }
        // one segment of one step per point: a run's first segment comes from its left neighbour on the same straight -- unless the
        // run starts the line (off0 = 0, swath lines between quiet U-turns): then it is the jump from the previous turn's end, or
        // nothing at all for the path's first point
        const bool at_line_start = tl.quiet == 1 && tl.off0 == 0;
        double len = (double)(at_line_start ? run.count - 1 : run.count) * step_len, t = len / fmax(msnom, 0.1);
        if (at_line_start && tl.idx0 > 0) {
            const double k0 = q.junc.x, jl = q.junc.y;
            len += jl;
            t += jl / fmax(((cst.v_turn + cst.v_work) / 2) / 3.6, 0.1);        // MLP:1305-1309: mean of the two end speeds
            tp.max_kappa = k0; tp.max_alat = cst.ms_work * cst.ms_work * k0;
            tp.max_jump = fmax(fabs(k0 - (q.reverse_order ? cst.turn_kappa_last[1] : cst.turn_kappa_last[0])), k0);
        }
        tp.main_len = layer ? 0.0 : len; tp.main_time_pre = layer ? 0.0 : t; tp.main_time = layer ? 0.0 : t;
        tp.head_len = layer ? len : 0.0; tp.head_time_pre = layer ? t : 0.0; tp.head_time = layer ? t : 0.0;
    }
    return tp;
}

}  // namespace fcpp
