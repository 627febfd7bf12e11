// fcpp_kernels.hip -- gfx950 (MI355X) kernels of the coverage-path hot path.
//
// Pipeline A ("staged", this file): one pass per operator, each a one-thread-per-point kernel over
// 2048-point workgroup tiles that never straddle a path:
//   k_generate     point index -> (x, y, nominal v, segment word)        MLP:720-830, 898-1084, 1154-1218
//   k_curv_clamp   3-point curvature + lateral-acceleration clamp         MLP:467-536
//   k_scan_*       forward/backward sweeps as min-plus scans              MLP:538-589
//   k_validate     a_lat / geofence / obstacle flags + metrics partials   MLP:1290-1311, 1373-1424
//   k_reduce_stats fixed-order reduction of the partials per path
// All arithmetic is float64 (1e-6 m on 5 km coordinates rules out fp32); there is no dense
// contraction anywhere, so no MFMA: the kernels are bounded by HBM traffic and fp64 VALU rate.
//
// Sweeps as scans.  With u = (v/3.6)^2 the reference's forward loop is u_i = min(u_i, u_{i-1} + w_i),
// w_i = 2 a |p_i - p_{i-1}|, where a step with |p_i - p_{i-1}| < 1e-6 is skipped (w_i = +inf: the
// point keeps its value and propagation restarts there).  Each element is the map u -> min(c, u + w);
// maps compose associatively as (c, w) o (c', w') = (min(c, c' + w), w' + w), which is scanned with
// wave shuffles inside a tile and a short spine across tiles.  The backward loop is the mirror image,
// and because both loops break at the same places the result of "forward then backward" equals
// min(forward(u0), backward(u0)), so the two scans are independent.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "fcpp_device.h"
#include "fcpp_geom.h"
#include "fcpp_internal.h"

namespace fcpp {

static constexpr int BLOCK = 256;
static constexpr int IPT = TILE_POINTS / BLOCK;  // 8
static constexpr int NWAVE = BLOCK / 64;
#define FCPP_INF __builtin_huge_val()

// --------------------------------------------------------------------------------------------
// point generator
// --------------------------------------------------------------------------------------------
struct GenOut { double x, y, v; uint32_t fs; };

__device__ __forceinline__ void gen_headland(const DevPrim *__restrict__ prims, int lo, int cnt, int64_t i,
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
        default: {  // PRIM_CAC
            const double s = linspace_at(0.0, p.a[6], p.a[5], p.n, k);
            cac_world_point(cst.sh_half, p.a[0], p.a[1], p.form, p.a[3] < 0 ? -1.0 : 1.0, p.a[4], s, o.x, o.y);
        } break;
    }
}

__device__ __forceinline__ void gen_point(const DevField &f, const DevPrim *__restrict__ prims, int64_t i,
                                          const DevConst &cst, GenOut &o)
{
    if (i >= f.n_main) { gen_headland(prims, f.prim_first, f.prim_count, i, cst, o); return; }
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
            if (turn_right) cac_world_point(cst.sh_pi, f.max_x - f.R, y, 1, -1.0, f.turn_Re, s, px, py);
            else            cac_world_point(cst.sh_pi, f.min_x + f.R, y, 1, 1.0, f.turn_Re, s, px, py);
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

__global__ __launch_bounds__(BLOCK) void k_generate(const DevTile *__restrict__ tiles,
                                                    const DevField *__restrict__ fields,
                                                    const DevPrim *__restrict__ prims, DevConst cst,
                                                    double *__restrict__ x, double *__restrict__ y,
                                                    double *__restrict__ v, uint32_t *__restrict__ fs)
{
    const DevTile t = tiles[blockIdx.x];
    const DevField &f = fields[t.field];
    for (int j = threadIdx.x; j < t.count; j += BLOCK) {
        GenOut o;
        gen_point(f, prims, t.start + j, cst, o);
        const int64_t g = f.pt_off + t.start + j;
        x[g] = o.x; y[g] = o.y; v[g] = o.v; fs[g] = o.fs;
    }
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
    const double dth = atan2(cr, dt);
    return fabs(2 * dth / (ds1 + ds2));
}

__global__ __launch_bounds__(BLOCK) void k_curv_clamp(const DevTile *__restrict__ tiles,
                                                      const DevPath *__restrict__ paths, DevConst cst,
                                                      int do_clamp, const double *__restrict__ x,
                                                      const double *__restrict__ y,
                                                      const double *__restrict__ v_in, double *__restrict__ v_out,
                                                      double *__restrict__ kappa,
                                                      unsigned long long *__restrict__ n_adjusted)
{
    const DevTile t = tiles[blockIdx.x];
    const DevPath p = paths[t.field];
    int adj = 0;
    for (int j = threadIdx.x; j < t.count; j += BLOCK) {
        const int64_t i = t.start + j, g = p.off + i;
        double k = 0.0;
        double vv = v_in[g];
        if (i > 0 && i < p.n - 1) {
            k = curvature3(x[g - 1], y[g - 1], x[g], y[g], x[g + 1], y[g + 1]);
            if (do_clamp && p.n >= 3 && k > 1e-6) {
                const double vmax_ms = sqrt(cst.a_lat / k) * cst.sf;
                const double vmax_kmh = vmax_ms * 3.6;
                if (vv > vmax_kmh) { vv = vmax_kmh; ++adj; }
            }
        }
        if (kappa) kappa[g] = k;
        v_out[g] = vv;
    }
    if (n_adjusted) {
        // wave reduce then one atomic per wave (integer: order independent)
        for (int o = 32; o > 0; o >>= 1) adj += __shfl_down(adj, o);
        if ((threadIdx.x & 63) == 0 && adj) atomicAdd(&n_adjusted[t.field], (unsigned long long)adj);
    }
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

// fill S.c / S.w of one tile from global x, y, v (km/h); min_n: shorter paths are left untouched
__device__ __forceinline__ void tile_load(TileScanShared &S, const DevTile &t, const DevPath &p, double two_a,
                                          const double *__restrict__ x, const double *__restrict__ y,
                                          const double *__restrict__ v)
{
    for (int j = threadIdx.x; j <= t.count; j += BLOCK) {
        const int64_t i = t.start + j, g = p.off + i;
        double w = FCPP_INF;
        if (i > 0 && i < p.n) {
            const double dx = x[g] - x[g - 1], dy = y[g] - y[g - 1];
            const double d = sqrt(dx * dx + dy * dy);
            if (!(d < 1e-6)) w = two_a * d;               // MLP:560-561 / 576-577: skipped step
        }
        S.w[lidx(j)] = w;
        if (j < t.count) {
            const double ms = v[g] / 3.6;
            S.c[lidx(j)] = ms * ms;
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(BLOCK) void k_scan_tiles(const DevTile *__restrict__ tiles,
                                                      const DevPath *__restrict__ paths, DevConst cst,
                                                      const double *__restrict__ x, const double *__restrict__ y,
                                                      const double *__restrict__ v, Agg *__restrict__ agg_f,
                                                      Agg *__restrict__ agg_b)
{
    __shared__ TileScanShared S;
    const DevTile t = tiles[blockIdx.x];
    const DevPath p = paths[t.field];
    tile_load(S, t, p, 2 * cst.a_lon, x, y, v);
    Agg tf, tb;
    tile_scan(S, t.count, FCPP_INF, FCPP_INF, tf, tb);
    if (threadIdx.x == 0) { agg_f[blockIdx.x] = tf; agg_b[blockIdx.x] = tb; }
}

// spine: carry_f[t] = value entering tile t from the left, carry_b[t] from the right.  Path boundaries
// need no special case: the first point of a path has w = +inf, which makes its tile's map constant.
__global__ __launch_bounds__(BLOCK) void k_scan_spine(int64_t n_tiles, const Agg *__restrict__ agg_f,
                                                      const Agg *__restrict__ agg_b,
                                                      double *__restrict__ carry_f, double *__restrict__ carry_b)
{
    __shared__ Agg sh[NWAVE];
    __shared__ double carry_sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int dir = 0; dir < 2; ++dir) {
        const Agg *__restrict__ agg = dir ? agg_b : agg_f;
        double *__restrict__ carry = dir ? carry_b : carry_f;
        if (tid == 0) carry_sh = FCPP_INF;
        __syncthreads();
        for (int64_t base = 0; base < n_tiles; base += BLOCK) {
            // position q in scan order; dir 1 walks the tiles from the last to the first
            const int64_t q = base + tid;
            const int64_t ti = dir ? (n_tiles - 1 - q) : q;
            Agg me = { FCPP_INF, 0.0 };
            if (q < n_tiles) me = agg[ti];
            Agg inc = me;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                Agg pv = { __shfl_up(inc.c, o), __shfl_up(inc.w, o) };
                if (lane >= o) inc = combine_after(pv, inc);
            }
            if (lane == 63) sh[wave] = inc;
            __syncthreads();
            Agg ex = { __shfl_up(inc.c, 1), __shfl_up(inc.w, 1) };
            if (lane == 0) ex = { FCPP_INF, 0.0 };
            Agg pre = { FCPP_INF, 0.0 };
            for (int k = 0; k < wave; ++k) pre = combine_after(pre, sh[k]);
            ex = combine_after(pre, ex);
            const double cin = carry_sh;
            if (q < n_tiles) carry[ti] = fmin(ex.c, cin + ex.w);
            Agg tot = { FCPP_INF, 0.0 };
            for (int k = 0; k < NWAVE; ++k) tot = combine_after(tot, sh[k]);
            __syncthreads();
            if (tid == 0) carry_sh = fmin(tot.c, cin + tot.w);
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(BLOCK) void k_scan_apply(const DevTile *__restrict__ tiles,
                                                      const DevPath *__restrict__ paths, DevConst cst, int min_n,
                                                      const double *__restrict__ x, const double *__restrict__ y,
                                                      const double *__restrict__ v_in, double *__restrict__ v_out,
                                                      const double *__restrict__ carry_f,
                                                      const double *__restrict__ carry_b)
{
    __shared__ TileScanShared S;
    const DevTile t = tiles[blockIdx.x];
    const DevPath p = paths[t.field];
    if (p.n < min_n) {   // MLP:480-481 / 551-552: too short, returned unchanged
        if (v_out != v_in)
            for (int j = threadIdx.x; j < t.count; j += BLOCK) v_out[p.off + t.start + j] = v_in[p.off + t.start + j];
        return;
    }
    tile_load(S, t, p, 2 * cst.a_lon, x, y, v_in);
    Agg tf, tb;
    tile_scan(S, t.count, carry_f[blockIdx.x], carry_b[blockIdx.x], tf, tb);
    for (int j = threadIdx.x; j < t.count; j += BLOCK) {
        const int64_t g = p.off + t.start + j;
        const double v0 = v_in[g], ms = v0 / 3.6, u0 = ms * ms, u = S.c[lidx(j)];
        v_out[g] = (u < u0) ? sqrt(u) * 3.6 : v0;   // untouched points keep their exact input value
    }
}

// --------------------------------------------------------------------------------------------
// validator + metrics (MLP:1290-1311, 1373-1424; geofence / obstacles build-defined)
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ double nominal_speed(uint32_t fs, const DevConst &c)
{
    switch (fs & FCPP_KIND_MASK) {
        case FCPP_KIND_SWATH: return c.v_work;
        case FCPP_KIND_UTURN: case FCPP_KIND_CORNER: return c.v_turn;
        case FCPP_KIND_REVERSE: return 2.5;
        default: return c.v_head;
    }
}

struct RedShared { double d[NWAVE][9]; long long i[NWAVE][3]; };

static constexpr int OBS_LDS_VERTS = 1024;

__global__ __launch_bounds__(BLOCK) void k_validate(const DevTile *__restrict__ tiles,
                                                    const DevPath *__restrict__ paths,
                                                    const DevField *__restrict__ fields /* may be NULL */,
                                                    DevConst cst, DevObstacles obs, const double *__restrict__ x,
                                                    const double *__restrict__ y, const double *__restrict__ kappa,
                                                    const double *__restrict__ v, uint32_t *__restrict__ fsw,
                                                    TilePartial *__restrict__ partial)
{
    __shared__ RedShared R;
    __shared__ double ox[OBS_LDS_VERTS], oy[OBS_LDS_VERTS];
    const DevTile t = tiles[blockIdx.x];
    const DevPath p = paths[t.field];
    const int64_t n_main = fields ? fields[t.field].n_main : p.n;
    // stage this field's obstacle vertices in LDS (falls back to global memory if they do not fit)
    int64_t ov0 = 0, ov1 = 0; int ob0 = 0, ob1 = 0; bool obs_lds = false;
    if (fields && fields[t.field].obs_count > 0) {
        ob0 = fields[t.field].obs_first; ob1 = ob0 + fields[t.field].obs_count;
        ov0 = obs.offsets[ob0]; ov1 = obs.offsets[ob1];
        obs_lds = (ov1 - ov0) <= OBS_LDS_VERTS;
        if (obs_lds)
            for (int k = threadIdx.x; k < (int)(ov1 - ov0); k += BLOCK) { ox[k] = obs.x[ov0 + k]; oy[k] = obs.y[ov0 + k]; }
        __syncthreads();
    }
    double s_len[2] = { 0, 0 }, s_tpre[2] = { 0, 0 }, s_t[2] = { 0, 0 };
    double mk = 0, ma = 0, mj = 0;
    long long nv = 0, nout = 0, nobs = 0;
    for (int j = threadIdx.x; j < t.count; j += BLOCK) {
        const int64_t i = t.start + j, g = p.off + i;
        const double px = x[g], py = y[g], vi = v[g], ki = kappa[g];
        uint32_t fs = fsw ? fsw[g] : 0u;
        // segment (i-1, i): length and time (MLP:1294-1311); the seam main|headland belongs to neither
        if (i > 0 && i != n_main) {
            const int layer = i > n_main ? 1 : 0;
            const double dx = px - x[g - 1], dy = py - y[g - 1];
            const double d = sqrt(dx * dx + dy * dy);
            s_len[layer] += d;
            double ms = ((v[g - 1] + vi) / 2) / 3.6;
            s_t[layer] += d / fmax(ms, 0.1);
            if (fsw) {
                double mp = ((nominal_speed(fsw[g - 1], cst) + nominal_speed(fs, cst)) / 2) / 3.6;
                s_tpre[layer] += d / fmax(mp, 0.1);
            }
        }
        if (i > 0 && i < p.n - 1) {     // MLP:1383-1391
            const double ms = vi / 3.6, alat = ms * ms * ki;
            mk = fmax(mk, ki); ma = fmax(ma, alat);
            if (alat > cst.a_lat) { ++nv; fs |= FCPP_FLAG_ALAT; }
            if (i > 1) mj = fmax(mj, fabs(ki - kappa[g - 1]));   // MLP:1404-1406
        }
        if (fields) {
            const DevField &f = fields[t.field];
            bool out = false;
#pragma unroll
            for (int e = 0; e < 4; ++e) out = out || (f.ex[e] * px + f.ey[e] * py + f.eo[e] < -cst.geofence_tol);
            if (out) { ++nout; fs |= FCPP_FLAG_OUTSIDE; }
            bool inside_any = false;
            for (int b = ob0; b < ob1 && !inside_any; ++b) {
                const int64_t a0 = obs.offsets[b], a1 = obs.offsets[b + 1];
                bool in = false;
                for (int64_t k = a0, q = a1 - 1; k < a1; q = k++) {
                    const double xi = obs_lds ? ox[k - ov0] : obs.x[k], yi = obs_lds ? oy[k - ov0] : obs.y[k];
                    const double xj = obs_lds ? ox[q - ov0] : obs.x[q], yj = obs_lds ? oy[q - ov0] : obs.y[q];
                    if (((yi > py) != (yj > py)) && (px < (xj - xi) * (py - yi) / (yj - yi) + xi)) in = !in;
                }
                inside_any = in;
            }
            if (inside_any) { ++nobs; fs |= FCPP_FLAG_OBSTACLE; }
        }
        if (fsw) fsw[g] = fs;
    }
    // fixed-shape block reduction: lanes (xor butterfly) -> waves (serial) : deterministic
    double dv[9] = { s_len[0], s_tpre[0], s_t[0], s_len[1], s_tpre[1], s_t[1], mk, ma, mj };
    long long iv[3] = { nv, nout, nobs };
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int k = 0; k < 6; ++k) dv[k] += __shfl_xor(dv[k], o);
#pragma unroll
        for (int k = 6; k < 9; ++k) dv[k] = fmax(dv[k], __shfl_xor(dv[k], o));
#pragma unroll
        for (int k = 0; k < 3; ++k) iv[k] += __shfl_xor(iv[k], o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        for (int k = 0; k < 9; ++k) R.d[wave][k] = dv[k];
        for (int k = 0; k < 3; ++k) R.i[wave][k] = iv[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        TilePartial tp;
        double a[9]; long long b[3];
        for (int k = 0; k < 9; ++k) a[k] = R.d[0][k];
        for (int k = 0; k < 3; ++k) b[k] = R.i[0][k];
        for (int wv = 1; wv < NWAVE; ++wv) {
            for (int k = 0; k < 6; ++k) a[k] += R.d[wv][k];
            for (int k = 6; k < 9; ++k) a[k] = fmax(a[k], R.d[wv][k]);
            for (int k = 0; k < 3; ++k) b[k] += R.i[wv][k];
        }
        tp.main_len = a[0]; tp.main_time_pre = a[1]; tp.main_time = a[2];
        tp.head_len = a[3]; tp.head_time_pre = a[4]; tp.head_time = a[5];
        tp.max_kappa = a[6]; tp.max_alat = a[7]; tp.max_jump = a[8];
        tp.n_viol = b[0]; tp.n_outside = b[1]; tp.n_in_obstacle = b[2]; tp.n_adjusted = 0;
        partial[blockIdx.x] = tp;
    }
}

// one wave per path: lanes stride over the path's tiles in a fixed assignment, then a fixed butterfly
__global__ __launch_bounds__(64) void k_reduce_stats(int64_t n_paths, const int64_t *__restrict__ tile_first,
                                                     const TilePartial *__restrict__ partial,
                                                     const unsigned long long *__restrict__ n_adjusted,
                                                     fcpp_field_stats *__restrict__ stats)
{
    const int64_t pth = blockIdx.x;
    if (pth >= n_paths) return;
    const int lane = threadIdx.x;
    double a[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    long long b[3] = { 0, 0, 0 };
    for (int64_t t = tile_first[pth] + lane; t < tile_first[pth + 1]; t += 64) {
        const TilePartial tp = partial[t];
        a[0] += tp.main_len; a[1] += tp.main_time_pre; a[2] += tp.main_time;
        a[3] += tp.head_len; a[4] += tp.head_time_pre; a[5] += tp.head_time;
        a[6] = fmax(a[6], tp.max_kappa); a[7] = fmax(a[7], tp.max_alat); a[8] = fmax(a[8], tp.max_jump);
        b[0] += tp.n_viol; b[1] += tp.n_outside; b[2] += tp.n_in_obstacle;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int k = 0; k < 6; ++k) a[k] += __shfl_xor(a[k], o);
#pragma unroll
        for (int k = 6; k < 9; ++k) a[k] = fmax(a[k], __shfl_xor(a[k], o));
#pragma unroll
        for (int k = 0; k < 3; ++k) b[k] += __shfl_xor(b[k], o);
    }
    if (lane == 0) {
        fcpp_field_stats s;
        s.main_len_m = a[0]; s.main_time_pre_s = a[1]; s.main_time_s = a[2];
        s.head_len_m = a[3]; s.head_time_pre_s = a[4]; s.head_time_s = a[5];
        s.max_kappa = a[6]; s.max_alat = a[7]; s.max_jump = a[8];
        s.n_viol = b[0]; s.n_outside = b[1]; s.n_in_obstacle = b[2];
        s.n_adjusted = n_adjusted ? (int64_t)n_adjusted[pth] : 0;
        stats[pth] = s;
    }
}

// --------------------------------------------------------------------------------------------
// small operators
// --------------------------------------------------------------------------------------------
__global__ void k_straight(int64_t n_seg, const double *__restrict__ seg, int n_pts, const int32_t *__restrict__ mask,
                           double *__restrict__ out)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_seg * n_pts) return;
    const int64_t s = g / n_pts, k = g - s * n_pts;
    if (mask && !mask[s]) return;
    const double x0 = seg[4 * s], y0 = seg[4 * s + 1], x1 = seg[4 * s + 2], y1 = seg[4 * s + 3];
    const double sx = n_pts > 1 ? (x1 - x0) / (double)(n_pts - 1) : 0.0;
    const double sy = n_pts > 1 ? (y1 - y0) / (double)(n_pts - 1) : 0.0;
    out[2 * g] = linspace_at(x0, x1, sx, n_pts, k);
    out[2 * g + 1] = linspace_at(y0, y1, sy, n_pts, k);
}

__global__ void k_fresnel(int64_t n, const double *__restrict__ t, double *__restrict__ c, double *__restrict__ s)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    double cc, ss;
    fresnel_cs(t[g], cc, ss);
    c[g] = cc; s[g] = ss;
}

// GA tour length (GA:174-181): one wavefront per chromosome; lanes gather D[r_k, r_k+1] in parallel
__global__ __launch_bounds__(BLOCK) void k_ga_fitness(int n, int64_t pop, const double *__restrict__ D,
                                                      const int32_t *__restrict__ routes,
                                                      double *__restrict__ dist, double *__restrict__ fit, int order_mode)
{
    const int lane = threadIdx.x & 63;
    const int64_t ch = (int64_t)blockIdx.x * NWAVE + (threadIdx.x >> 6);
    if (ch >= pop) return;
    const int32_t *__restrict__ r = routes + ch * n;
    double total = 0.0;
    if (order_mode == 0) {
        // left-to-right float64 sum, exactly the reference's loop order
        for (int base = 0; base < n; base += 64) {
            const int k = base + lane;
            double d = 0.0;
            if (k < n) {
                const int a = r[k], b = r[k + 1 == n ? 0 : k + 1];
                d = D[(int64_t)a * n + b];
            }
            const int m = min(64, n - base);
            for (int l = 0; l < m; ++l) total += __shfl(d, l);
        }
    } else {
        for (int k = lane; k < n; k += 64) {
            const int a = r[k], b = r[k + 1 == n ? 0 : k + 1];
            total += D[(int64_t)a * n + b];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
    }
    if (lane == 0) {
        if (dist) dist[ch] = total;
        if (fit) fit[ch] = 1.0 / (total + 1e-6);   // GA:172
    }
}

// --------------------------------------------------------------------------------------------
// launchers (called from fcpp_api.cpp)
// --------------------------------------------------------------------------------------------
#define FCPP_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

int launch_generate(hipStream_t st, int64_t n_tiles, const DevTile *tiles, const DevField *fields,
                    const DevPrim *prims, const DevConst &cst, double *x, double *y, double *v, uint32_t *fs)
{
    if (n_tiles <= 0) return 0;
    hipLaunchKernelGGL(k_generate, dim3((unsigned)n_tiles), dim3(BLOCK), 0, st, tiles, fields, prims, cst, x, y, v, fs);
    FCPP_LAUNCH_CHECK();
    return 0;
}

int launch_curv_clamp(hipStream_t st, int64_t n_tiles, const DevTile *tiles, const DevPath *paths,
                      const DevConst &cst, int do_clamp, const double *x, const double *y, const double *v_in,
                      double *v_out, double *kappa, unsigned long long *n_adjusted)
{
    if (n_tiles <= 0) return 0;
    hipLaunchKernelGGL(k_curv_clamp, dim3((unsigned)n_tiles), dim3(BLOCK), 0, st, tiles, paths, cst, do_clamp, x, y,
                       v_in, v_out, kappa, n_adjusted);
    FCPP_LAUNCH_CHECK();
    return 0;
}

int launch_scan_tiles(hipStream_t st, int64_t n_tiles, const DevTile *tiles, const DevPath *paths, const DevConst &cst,
                      const double *x, const double *y, const double *v_in, void *agg_f, void *agg_b)
{
    if (n_tiles <= 0) return 0;
    hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)n_tiles), dim3(BLOCK), 0, st, tiles, paths, cst, x, y, v_in,
                       (Agg *)agg_f, (Agg *)agg_b);
    FCPP_LAUNCH_CHECK();
    return 0;
}

int launch_scan_spine(hipStream_t st, int64_t n_tiles, const void *agg_f, const void *agg_b, double *carry_f,
                      double *carry_b)
{
    if (n_tiles <= 0) return 0;
    hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(BLOCK), 0, st, n_tiles, (const Agg *)agg_f, (const Agg *)agg_b,
                       carry_f, carry_b);
    FCPP_LAUNCH_CHECK();
    return 0;
}

int launch_scan_apply(hipStream_t st, int64_t n_tiles, const DevTile *tiles, const DevPath *paths, const DevConst &cst,
                      int min_n, const double *x, const double *y, const double *v_in, double *v_out,
                      const double *carry_f, const double *carry_b)
{
    if (n_tiles <= 0) return 0;
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)n_tiles), dim3(BLOCK), 0, st, tiles, paths, cst, min_n, x, y, v_in,
                       v_out, carry_f, carry_b);
    FCPP_LAUNCH_CHECK();
    return 0;
}

int launch_validate(hipStream_t st, int64_t n_tiles, const DevTile *tiles, const DevPath *paths,
                    const DevField *fields, const DevConst &cst, const DevObstacles &obs, const double *x,
                    const double *y, const double *kappa, const double *v, uint32_t *fs, TilePartial *partial)
{
    if (n_tiles <= 0) return 0;
    hipLaunchKernelGGL(k_validate, dim3((unsigned)n_tiles), dim3(BLOCK), 0, st, tiles, paths, fields, cst, obs, x, y,
                       kappa, v, fs, partial);
    FCPP_LAUNCH_CHECK();
    return 0;
}

int launch_reduce_stats(hipStream_t st, int64_t n_paths, const TilePartial *partial, const int64_t *tile_first,
                        const unsigned long long *n_adjusted, fcpp_field_stats *stats)
{
    if (n_paths <= 0) return 0;
    hipLaunchKernelGGL(k_reduce_stats, dim3((unsigned)n_paths), dim3(64), 0, st, n_paths, tile_first, partial,
                       n_adjusted, stats);
    FCPP_LAUNCH_CHECK();
    return 0;
}

int launch_straight(hipStream_t st, int64_t n_seg, const double *seg, int n_pts, const int32_t *mask, double *out)
{
    const int64_t n = n_seg * n_pts;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_straight, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n_seg, seg, n_pts, mask, out);
    FCPP_LAUNCH_CHECK();
    return 0;
}

int launch_fresnel(hipStream_t st, int64_t n, const double *t, double *c, double *s)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_fresnel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, t, c, s);
    FCPP_LAUNCH_CHECK();
    return 0;
}

int launch_ga_fitness(hipStream_t st, int n, int64_t pop, const double *D, const int32_t *routes, double *dist,
                      double *fit, int order_mode)
{
    if (pop <= 0) return 0;
    hipLaunchKernelGGL(k_ga_fitness, dim3((unsigned)((pop + NWAVE - 1) / NWAVE)), dim3(BLOCK), 0, st, n, pop, D, routes,
                       dist, fit, order_mode);
    FCPP_LAUNCH_CHECK();
    return 0;
}

}  // namespace fcpp
