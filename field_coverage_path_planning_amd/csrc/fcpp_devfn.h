// fcpp_devfn.h -- device functions shared by the staged kernels (fcpp_kernels.hip) and the fused
// single-pass kernel (fcpp_fused.hip): point generator, curvature, min-plus tile scan.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "fcpp_device.h"
#include "fcpp_geom.h"
#include "fcpp_internal.h"

#include <hip/hip_ext.h>

// kernel launch of the planner pipelines: plain, or with the dispatch's start / stop events when profiling is armed (fcpp_device.h)
#define FCPP_LAUNCH(kernel, grid, block, shmem, st, ...)                                                                         \
    do {                                                                                                                         \
        if (::fcpp::g_launch_prof.start) {                                                                                       \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, st, ::fcpp::g_launch_prof.start, ::fcpp::g_launch_prof.stop, 0, __VA_ARGS__); \
            ::fcpp::g_launch_prof = ::fcpp::LaunchProf();                                                                        \
        } else hipLaunchKernelGGL(kernel, grid, block, shmem, st, __VA_ARGS__);                                                  \
    } while (0)

namespace fcpp {

static constexpr int BLOCK = 256;
static constexpr int IPT = TILE_POINTS / BLOCK;  // 8
static constexpr int NWAVE = BLOCK / 64;
#define FCPP_INF __builtin_huge_val()

// --------------------------------------------------------------------------------------------
// point generator
// --------------------------------------------------------------------------------------------
struct GenOut { double x, y, v; uint32_t fs; };

__device__ __forceinline__ void gen_headland(const DevField &f, const DevPrim *__restrict__ prims, int lo, int cnt, int64_t i,
                                             const DevConst &cst, GenOut &o)
{
    // binary search: last primitive with start <= i
    int a = lo, b = lo + cnt - 1;
    while (a < b) {
        int m = (a + b + 1) >> 1;
        if (prims[m].start <= i) a = m; else b = m - 1;
    }
    const DevPrim &p = prims[a];
    const int64_t k = i - p.start;
    o.v = p.v_nom; o.fs = p.fs;
    switch (p.kind) {
        case PRIM_POINT: o.x = p.a[0]; o.y = p.a[1]; break;
        case PRIM_LINSPACE:
            o.x = linspace_at(p.a[0], p.a[2], p.a[4], p.n, k);
            o.y = linspace_at(p.a[1], p.a[3], p.a[5], p.n, k);
            break;
        case PRIM_ARC: {
            const double th = linspace_at(0.0, p.a[3], p.a[4], p.n, k);
            double s, c;
            sincos(th, &s, &c);
            corner_arc_point(p.form, p.a[0], p.a[1], p.a[2], c, s, o.x, o.y);
        } break;
        case PRIM_RAY: {
            const double t = linspace_at(0.0, p.a[4], p.a[5], p.n, k);
            o.x = p.a[0] + t * p.a[2];
            o.y = p.a[1] + t * p.a[3];
        } break;
        case PRIM_UTURN: {      // a U-turn of layer 1 as a primitive (obstacle-aware swaths): the formulas of gen_point below
            const bool turn_right = p.form & 1;
            const double s = linspace_at(0.0, f.turn_end, f.turn_step, f.n_turn, k);
            double px, py;
            if (!(p.form & 4)) {
                double sn, cs;
                sincos(s, &sn, &cs);
                px = turn_right ? (p.a[0] - f.R * cs) : (p.a[0] + f.R * cs);
                py = p.a[1] + f.R * sn;
            } else cac_world_point(cst.shapes[0], p.a[0], p.a[1], 1, turn_right ? -1.0 : 1.0, f.turn_Re, s, px, py);
            if (p.form & 2) {
                const double tx = px - p.a[4], ty = py - p.a[5];
                const double xn = tx * p.a[2] - ty * p.a[3], yn = tx * p.a[3] + ty * p.a[2];
                px = xn + p.a[4]; py = yn + p.a[5];
            }
            o.x = px; o.y = py;
        } break;
        default: {  // PRIM_CAC
            const double s = linspace_at(0.0, p.a[6], p.a[5], p.n, k);
            cac_world_point(cst.shapes[1], p.a[0], p.a[1], p.form, p.a[3] < 0 ? -1.0 : 1.0, p.a[4], s, o.x, o.y);
        } break;
    }
}

__device__ __forceinline__ void gen_point(const DevField &f, const DevPrim *__restrict__ prims, int64_t i,
                                          const DevConst &cst, GenOut &o)
{
    if (i >= f.gen_main) { gen_headland(f, prims, f.prim_first, f.prim_count, i, cst, o); return; }
    // layer 1, MLP:750-780: pass idx = i / (n_line + n_turn)
    const int64_t per = (int64_t)f.n_line + f.n_turn;
    const int64_t idx = i / per;
    const int64_t r = i - idx * per;
    const int64_t pi = f.reverse_order ? (f.P - 1 - idx) : idx;       // MLP:745-748
    const double y = f.min_y + (double)pi * f.W;                      // MLP:751
    const bool go_left = f.start_from_right ? ((idx & 1) == 0) : ((idx & 1) == 1);  // MLP:754-759
    double px, py;
    if (r < f.n_line) {
        px = go_left ? linspace_at(f.lex, f.lsx, -f.line_step, f.n_line, r)
                     : linspace_at(f.lsx, f.lex, f.line_step, f.n_line, r);
        py = y;
        o.v = f.v_work; o.fs = FCPP_KIND_SWATH | ((uint32_t)pi << FCPP_INDEX_SHIFT);
    } else {
        const int64_t k = r - f.n_line;
        const bool turn_right = !go_left;                             // MLP:776
        const double s = linspace_at(0.0, f.turn_end, f.turn_step, f.n_turn, k);
        if (f.turn_model == FCPP_TURN_ARC) {                          // MLP:807-825
            double sn, cs;
            sincos(s, &sn, &cs);
            px = turn_right ? (f.max_x - f.R * cs) : (f.min_x + f.R * cs);
            py = y + f.R * sn;
        } else {
            // same start pose and heading change as the reference semicircle, clothoid-arc-clothoid shape
            cac_world_point(cst.shapes[0], turn_right ? (f.max_x - f.R) : (f.min_x + f.R), y, 1,
                            turn_right ? -1.0 : 1.0, f.turn_Re, s, px, py);
        }
        o.v = f.v_turn; o.fs = FCPP_KIND_UTURN | ((uint32_t)pi << FCPP_INDEX_SHIFT);
    }
    if (f.rotated) {                                                  // MLP:271-282 with angle = +rotation
        const double tx = px - f.rot_cx, ty = py - f.rot_cy;
        const double xn = tx * f.rot_cos - ty * f.rot_sin;
        const double yn = tx * f.rot_sin + ty * f.rot_cos;
        px = xn + f.rot_cx; py = yn + f.rot_cy;
    }
    o.x = px; o.y = py;
}

// --------------------------------------------------------------------------------------------
// curvature (MLP:513-536) and clamp (MLP:490-504)
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ double curvature3(double x1, double y1, double x2, double y2, double x3, double y3)
{
    const double dx1 = x2 - x1, dy1 = y2 - y1, dx2 = x3 - x2, dy2 = y3 - y2;
    const double ds1 = sqrt(dx1 * dx1 + dy1 * dy1), ds2 = sqrt(dx2 * dx2 + dy2 * dy2);
    if (ds1 < 1e-6 || ds2 < 1e-6) return 0.0;
    // atan2(sin(t2 - t1), cos(t2 - t1)) of the two headings == signed angle between the two chords
    const double cr = dx1 * dy2 - dy1 * dx2, dt = dx1 * dx2 + dy1 * dy2;
    const double dth = atan2_fd(cr, dt);
    return fabs(2 * dth / (ds1 + ds2));
}

// --------------------------------------------------------------------------------------------
// min-plus scan of one tile held in LDS
// --------------------------------------------------------------------------------------------
// LDS index with one pad slot per 8 items: thread-blocked ds_read_b64 access (stride 9 doubles) is
// conflict-free on the 64-bank LDS.
__device__ __forceinline__ int lidx(int j) { return j + (j >> 3); }
static constexpr int LDS_TILE = TILE_POINTS + 1 + ((TILE_POINTS + 1) >> 3) + 1;

struct Agg { double c, w; };
__device__ __forceinline__ Agg combine_after(Agg prev, Agg me)  // apply prev first, then me
{
    Agg r;
    r.c = fmin(me.c, prev.c + me.w);
    r.w = prev.w + me.w;
    return r;
}

struct TileScanShared {
    double c[LDS_TILE];
    double w[LDS_TILE];   // w[j] for j in [0, count]: w[count] couples the tile's last point to the next one
    Agg wf[NWAVE], wb[NWAVE];
};

// On entry sc/sw hold c_j (j < count) and w_j (j <= count).  carry_f / carry_b are the values arriving
// from the left / right neighbour tiles (+inf if none).  On exit sc[j] = min(fwd_j, bwd_j); the tile's
// own aggregates (carry-independent) are returned for the spine.
__device__ __forceinline__ void tile_scan(TileScanShared &S, int count, double carry_f, double carry_b,
                                          Agg &tile_f, Agg &tile_b)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = tid * IPT;
    // c[k], wf[k]: item j = base + k and its coupling to j-1; wb[k]: coupling of item j to j+1.
    // Items beyond the tile are identity maps (c = +inf, w = 0).
    double c[IPT], wf[IPT], wb[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int j = base + k;
        const bool in = j < count;
        c[k] = in ? S.c[lidx(j)] : FCPP_INF;
        wf[k] = in ? S.w[lidx(j)] : 0.0;
        wb[k] = in ? S.w[lidx(j + 1)] : 0.0;
    }

    // thread aggregates
    Agg f = { FCPP_INF, 0.0 }, b = { FCPP_INF, 0.0 };
#pragma unroll
    for (int k = 0; k < IPT; ++k) { f.c = fmin(c[k], f.c + wf[k]); f.w += wf[k]; }
#pragma unroll
    for (int k = IPT - 1; k >= 0; --k) { b.c = fmin(c[k], b.c + wb[k]); b.w += wb[k]; }

    // inclusive wave scans: forward over lanes 0..63, backward over lanes 63..0
    Agg fi = f, bi = b;
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
    // exclusive prefix of this thread = (waves before) o (lanes before)
    Agg ef = { __shfl_up(fi.c, 1), __shfl_up(fi.w, 1) };
    if (lane == 0) ef = { FCPP_INF, 0.0 };
    Agg eb = { __shfl_down(bi.c, 1), __shfl_down(bi.w, 1) };
    if (lane == 63) eb = { FCPP_INF, 0.0 };
    Agg pre = { FCPP_INF, 0.0 }, suf = { FCPP_INF, 0.0 };
    for (int q = 0; q < wave; ++q) pre = combine_after(pre, S.wf[q]);
    for (int q = NWAVE - 1; q > wave; --q) suf = combine_after(suf, S.wb[q]);
    ef = combine_after(pre, ef);
    eb = combine_after(suf, eb);
    Agg tf = { FCPP_INF, 0.0 }, tb = { FCPP_INF, 0.0 };
    for (int q = 0; q < NWAVE; ++q) tf = combine_after(tf, S.wf[q]);
    for (int q = NWAVE - 1; q >= 0; --q) tb = combine_after(tb, S.wb[q]);
    tile_f = tf; tile_b = tb;

    // second pass with the carried-in values
    double uf = fmin(ef.c, carry_f + ef.w);
    double ub = fmin(eb.c, carry_b + eb.w);
    double rf[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) { uf = fmin(c[k], uf + wf[k]); rf[k] = uf; }
#pragma unroll
    for (int k = IPT - 1; k >= 0; --k) {
        ub = fmin(c[k], ub + wb[k]);
        const int j = base + k;
        if (j < count) S.c[lidx(j)] = fmin(rf[k], ub);
    }
    __syncthreads();
}

__device__ __forceinline__ double nominal_speed(uint32_t fs, const DevConst &c)
{
    const uint32_t k = fs & FCPP_KIND_MASK;         // (selects: see nominal_ms)
    double r = c.v_head;
    r = (k == FCPP_KIND_REVERSE) ? 2.5 : r;
    r = (k == FCPP_KIND_UTURN || k == FCPP_KIND_CORNER || k == FCPP_KIND_DETOUR) ? c.v_turn : r;
    return (k == FCPP_KIND_SWATH) ? c.v_work : r;
}

// out-of-line copy of atan2_fd (fcpp_geom.h) for the eight-points-per-lane kernel's halo code: only turn points reach it, and keeping
// it out of that kernel's unrolled per-item loops keeps its register budget sane
__device__ __noinline__ double atan2_slow(double y, double x) { return atan2_fd(y, x); }

__device__ __forceinline__ double nominal_ms(uint32_t fs, const DevConst &c)
{
    // (selects, not a switch: the compiler turns a switch over run-time values into a five-entry table in scratch memory)
    const uint32_t k = fs & FCPP_KIND_MASK;
    double r = c.ms_head;
    r = (k == FCPP_KIND_REVERSE) ? c.ms_rev : r;
    r = (k == FCPP_KIND_UTURN || k == FCPP_KIND_CORNER || k == FCPP_KIND_DETOUR) ? c.ms_turn : r;
    return (k == FCPP_KIND_SWATH) ? c.ms_work : r;
}

}  // namespace fcpp
