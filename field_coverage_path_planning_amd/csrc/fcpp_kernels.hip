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

#include <string.h>

#include "fcpp_quiet_fn.h"

namespace fcpp {

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

// The spine of a large batch in three levels (2 M tiles at cfg2 / 0.1 m would keep the single workgroup above busy for milliseconds):
// every workgroup composes the maps of SPINE_BLOCK consecutive tiles (k_spine_blocks<false>), the single-workgroup spine scans those
// block aggregates, and the blocks are scanned again with their carries (k_spine_blocks<true>).  The maps compose associatively, so
// the carries are those of the one-level scan up to the rounding of w's sums.  blockIdx.y = direction; direction 1 walks the
// tiles from the last to the first, its block aggregates are stored in reverse so that the top-level spine walks them its usual way.
static constexpr int SPINE_IPT = 8, SPINE_BLOCK = BLOCK * SPINE_IPT;

template <bool APPLY>
__global__ __launch_bounds__(BLOCK) void k_spine_blocks(int64_t n_tiles, int64_t n_blocks, const Agg *__restrict__ agg_f,
                                                        const Agg *__restrict__ agg_b, Agg *__restrict__ block_f, Agg *__restrict__ block_b,
                                                        const double *__restrict__ cin_f, const double *__restrict__ cin_b,
                                                        double *__restrict__ carry_f, double *__restrict__ carry_b)
{
    __shared__ Agg sh[NWAVE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, dir = blockIdx.y;
    const Agg *__restrict__ agg = dir ? agg_b : agg_f;
    const int64_t b = blockIdx.x, bi = dir ? (n_blocks - 1 - b) : b;
    const int64_t q0 = b * SPINE_BLOCK + (int64_t)tid * SPINE_IPT;             // first scan position of this thread
    Agg me[SPINE_IPT], ta = { FCPP_INF, 0.0 };
#pragma unroll
    for (int k = 0; k < SPINE_IPT; ++k) {
        const int64_t q = q0 + k;
        me[k] = { FCPP_INF, 0.0 };
        if (q < n_tiles) me[k] = agg[dir ? (n_tiles - 1 - q) : q];
        ta = combine_after(ta, me[k]);
    }
    Agg inc = ta;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        Agg pv = { __shfl_up(inc.c, o), __shfl_up(inc.w, o) };
        if (lane >= o) inc = combine_after(pv, inc);
    }
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    if (!APPLY) {
        if (tid == 0) {
            Agg tot = { FCPP_INF, 0.0 };
            for (int k = 0; k < NWAVE; ++k) tot = combine_after(tot, sh[k]);
            (dir ? block_b : block_f)[bi] = tot;
        }
        return;
    }
    Agg ex = { __shfl_up(inc.c, 1), __shfl_up(inc.w, 1) };
    if (lane == 0) ex = { FCPP_INF, 0.0 };
    Agg pre = { FCPP_INF, 0.0 };
    for (int k = 0; k < wave; ++k) pre = combine_after(pre, sh[k]);
    ex = combine_after(pre, ex);
    const double cin = (dir ? cin_b : cin_f)[bi];
    double *__restrict__ carry = dir ? carry_b : carry_f;
    double u = fmin(ex.c, cin + ex.w);                                          // value entering this thread's first tile
#pragma unroll
    for (int k = 0; k < SPINE_IPT; ++k) {
        const int64_t q = q0 + k;
        if (q < n_tiles) carry[dir ? (n_tiles - 1 - q) : q] = u;
        u = fmin(me[k].c, u + me[k].w);
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
struct RedShared { double d[NWAVE][9]; long long i[NWAVE][3]; };

static constexpr int VAL_LDS_VERTS = 1024;   // (staged validator: one polygon staged per workgroup)

__global__ __launch_bounds__(BLOCK) void k_validate(const DevTile *__restrict__ tiles,
                                                    const DevPath *__restrict__ paths,
                                                    const DevField *__restrict__ fields /* may be NULL */,
                                                    DevConst cst, DevObstacles obs, const double *__restrict__ x,
                                                    const double *__restrict__ y, const double *__restrict__ kappa,
                                                    const double *__restrict__ v, uint32_t *__restrict__ fsw,
                                                    TilePartial *__restrict__ partial)
{
    __shared__ RedShared R;
    __shared__ double ox[VAL_LDS_VERTS], oy[VAL_LDS_VERTS];
    const DevTile t = tiles[blockIdx.x];
    const DevPath p = paths[t.field];
    const int64_t n_main = fields ? fields[t.field].n_main : p.n;
    // stage this field's obstacle vertices in LDS (falls back to global memory if they do not fit)
    int64_t ov0 = 0, ov1 = 0; int ob0 = 0, ob1 = 0; bool obs_lds = false;
    if (fields && fields[t.field].obs_count > 0) {
        ob0 = fields[t.field].obs_first; ob1 = ob0 + fields[t.field].obs_count;
        ov0 = obs.offsets[ob0]; ov1 = obs.offsets[ob1];
        obs_lds = (ov1 - ov0) <= VAL_LDS_VERTS;
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

// G lanes per path: the lanes stride over the path's entries in a fixed assignment, then a fixed butterfly over the group.
// ids (optional): the reduction runs over partial[ids[k]], k in [tile_first[p], tile_first[p+1]) -- the fused pipeline lists only the
// tiles that can hold statistics (general tiles, wave tiles and the first tile of every quiet run; the others stay zero).
// run_count (with ids): > 0 for the first tile of a quiet run of that many points: its length / time / curvature statistics are the
// run's closed form (fcpp_quiet_fn.h), evaluated here; its partial slot only collects the flag counts k_plan_quiet adds while it
// stores the run, and is cleared again for the next step.
// path_list (optional): the paths this launch reduces.  The fused pipeline sorts its paths into three classes BY THEIR NUMBER OF
// ENTRIES -- a property of the field alone, so that a field's sums are added in the same order whatever batch it is part of --:
// up to 64 entries: 8 lanes per path (a plan at the reference's sampling has about ten); up to 8192: a wavefront per path; above
// (one path of 6e7 points has 1.2e5): a workgroup per path, k_reduce_stats_wg.
struct StatAcc {
    double a[9];
    long long b[4];
};
static_assert(sizeof(StatAcc) == 104, "red_scratch in fcpp_api.cpp is sized for 104-byte slice results");

__device__ __forceinline__ void stat_entry(StatAcc &s, int64_t t, TilePartial *__restrict__ partial, const int32_t *__restrict__ ids,
                                           const int64_t *__restrict__ run_count, const DevTile *__restrict__ tiles,
                                           const FieldStatView &fv, const DevPrim *__restrict__ prims, const DevConst &cst, int clear_counts = 0)
{
    const int64_t slot = ids ? (int64_t)ids[t] : t;
    TilePartial tp = partial[slot];
    // the fused pipeline (clear_counts): the entries of a path lie side by side in `partial`; the slot of a quiet run holds the run's
    // closed-form statistics (written once, at batch creation: k_run_consts) and collects the flag counts k_plan_quiet adds while it
    // stores the run -- cleared here for the next step
    if (clear_counts && (tp.n_outside | tp.n_in_obstacle)) { partial[slot].n_outside = 0; partial[slot].n_in_obstacle = 0; }
    const int64_t rc = run_count ? run_count[t] : 0;
    if (rc > 0) {
        const DevRun run = { (int32_t)slot, 0, rc };
        const TilePartial rp = quiet_run_partial(run, tiles[slot], fv, prims, cst);
        tp.main_len = rp.main_len; tp.main_time_pre = rp.main_time_pre; tp.main_time = rp.main_time;
        tp.head_len = rp.head_len; tp.head_time_pre = rp.head_time_pre; tp.head_time = rp.head_time;
        tp.max_kappa = rp.max_kappa; tp.max_alat = rp.max_alat; tp.max_jump = rp.max_jump;
        if (tp.n_outside | tp.n_in_obstacle) { partial[slot].n_outside = 0; partial[slot].n_in_obstacle = 0; }
    }
    s.a[0] += tp.main_len; s.a[1] += tp.main_time_pre; s.a[2] += tp.main_time;
    s.a[3] += tp.head_len; s.a[4] += tp.head_time_pre; s.a[5] += tp.head_time;
    s.a[6] = fmax(s.a[6], tp.max_kappa); s.a[7] = fmax(s.a[7], tp.max_alat); s.a[8] = fmax(s.a[8], tp.max_jump);
    s.b[0] += tp.n_viol; s.b[1] += tp.n_outside; s.b[2] += tp.n_in_obstacle; s.b[3] += tp.n_adjusted;
}

__device__ __forceinline__ void stat_store(const StatAcc &s, int64_t pth, const unsigned long long *__restrict__ n_adjusted,
                                           fcpp_field_stats *__restrict__ stats)
{
    fcpp_field_stats o;
    o.main_len_m = s.a[0]; o.main_time_pre_s = s.a[1]; o.main_time_s = s.a[2];
    o.head_len_m = s.a[3]; o.head_time_pre_s = s.a[4]; o.head_time_s = s.a[5];
    o.max_kappa = s.a[6]; o.max_alat = s.a[7]; o.max_jump = s.a[8];
    o.n_viol = s.b[0]; o.n_outside = s.b[1]; o.n_in_obstacle = s.b[2];
    o.n_adjusted = s.b[3] + (n_adjusted ? (int64_t)n_adjusted[pth] : 0);
    stats[pth] = o;
}

template <int G>
__global__ __launch_bounds__(256) void k_reduce_stats(int64_t n_list, const int32_t *__restrict__ path_list, const int64_t *__restrict__ tile_first,
                                                     TilePartial *__restrict__ partial, const unsigned long long *__restrict__ n_adjusted,
                                                     fcpp_field_stats *__restrict__ stats, const int32_t *__restrict__ ids,
                                                     const int64_t *__restrict__ run_count, const DevTile *__restrict__ tiles,
                                                     const DevField *__restrict__ fields, const DevPrim *__restrict__ prims, DevConst cst, int clear_counts)
{
    const int64_t slot = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    const int sub = threadIdx.x % G;
    StatAcc s;
#pragma unroll
    for (int k = 0; k < 9; ++k) s.a[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) s.b[k] = 0;
    const int64_t pth = slot < n_list ? (path_list ? (int64_t)path_list[slot] : slot) : -1;
    if (pth >= 0) {
        FieldStatView fv = {};
        if (fields) fv.load(fields, cst, pth);          // (the path's field: asked for beside the entry list, not through the tile records)
        for (int64_t t = tile_first[pth] + sub; t < tile_first[pth + 1]; t += G) stat_entry(s, t, partial, ids, run_count, tiles, fv, prims, cst, clear_counts);
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
#pragma unroll
        for (int k = 0; k < 6; ++k) s.a[k] += __shfl_xor(s.a[k], o);
#pragma unroll
        for (int k = 6; k < 9; ++k) s.a[k] = fmax(s.a[k], __shfl_xor(s.a[k], o));
#pragma unroll
        for (int k = 0; k < 4; ++k) s.b[k] += __shfl_xor(s.b[k], o);
    }
    if (pth >= 0 && sub == 0) stat_store(s, pth, n_adjusted, stats);
}

// ... and for paths with very many entries (one field of 10^5 chunk runs: cfg3) REDUCE_SPLIT workgroups per path, each over a contiguous
// slice of the entries, whose results a second launch adds up in slice order (k_reduce_stats_join)
static constexpr int REDUCE_SPLIT = 64;
static_assert(REDUCE_SPLIT == 64, "k_reduce_stats_join: one slice per lane");
__global__ __launch_bounds__(256) void k_reduce_stats_slice(const int32_t *__restrict__ path_list, const int64_t *__restrict__ tile_first,
                                                           TilePartial *__restrict__ partial, const int32_t *__restrict__ ids,
                                                           const int64_t *__restrict__ run_count, const DevTile *__restrict__ tiles,
                                                           const DevField *__restrict__ fields, const DevPrim *__restrict__ prims, DevConst cst,
                                                           StatAcc *__restrict__ scratch, int clear_counts)
{
    __shared__ StatAcc sh[4];
    const int64_t li = blockIdx.x / REDUCE_SPLIT;
    const int slice = blockIdx.x % REDUCE_SPLIT;
    const int64_t pth = path_list ? (int64_t)path_list[li] : li;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t t0 = tile_first[pth], ne = tile_first[pth + 1] - t0, per = (ne + REDUCE_SPLIT - 1) / REDUCE_SPLIT;
    const int64_t a = t0 + slice * per, b = min(t0 + ne, a + per);
    StatAcc s;
#pragma unroll
    for (int k = 0; k < 9; ++k) s.a[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) s.b[k] = 0;
    FieldStatView fv = {};
    if (fields) fv.load(fields, cst, pth);
    for (int64_t t = a + threadIdx.x; t < b; t += 256) stat_entry(s, t, partial, ids, run_count, tiles, fv, prims, cst, clear_counts);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int k = 0; k < 6; ++k) s.a[k] += __shfl_xor(s.a[k], o);
#pragma unroll
        for (int k = 6; k < 9; ++k) s.a[k] = fmax(s.a[k], __shfl_xor(s.a[k], o));
#pragma unroll
        for (int k = 0; k < 4; ++k) s.b[k] += __shfl_xor(s.b[k], o);
    }
    if (lane == 0) sh[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w2 = 1; w2 < 4; ++w2) {
            for (int k = 0; k < 6; ++k) s.a[k] += sh[w2].a[k];
            for (int k = 6; k < 9; ++k) s.a[k] = fmax(s.a[k], sh[w2].a[k]);
            for (int k = 0; k < 4; ++k) s.b[k] += sh[w2].b[k];
        }
        scratch[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(64) void k_reduce_stats_join(const int32_t *__restrict__ path_list, const StatAcc *__restrict__ scratch,
                                                         const unsigned long long *__restrict__ n_adjusted, fcpp_field_stats *__restrict__ stats)
{
    const int64_t li = blockIdx.x;
    const int64_t pth = path_list ? (int64_t)path_list[li] : li;
    StatAcc s = scratch[li * REDUCE_SPLIT + threadIdx.x];        // REDUCE_SPLIT == the wavefront: lane = slice, fixed butterfly
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int k = 0; k < 6; ++k) s.a[k] += __shfl_xor(s.a[k], o);
#pragma unroll
        for (int k = 6; k < 9; ++k) s.a[k] = fmax(s.a[k], __shfl_xor(s.a[k], o));
#pragma unroll
        for (int k = 0; k < 4; ++k) s.b[k] += __shfl_xor(s.b[k], o);
    }
    if (threadIdx.x == 0) stat_store(s, pth, n_adjusted, stats);
}

// a workgroup per path: every thread strides over the entries, the wavefronts' butterflies meet in LDS in wave order
__global__ __launch_bounds__(256) void k_reduce_stats_wg(int64_t n_list, const int32_t *__restrict__ path_list, const int64_t *__restrict__ tile_first,
                                                        TilePartial *__restrict__ partial, const unsigned long long *__restrict__ n_adjusted,
                                                        fcpp_field_stats *__restrict__ stats, const int32_t *__restrict__ ids,
                                                        const int64_t *__restrict__ run_count, const DevTile *__restrict__ tiles,
                                                        const DevField *__restrict__ fields, const DevPrim *__restrict__ prims, DevConst cst, int clear_counts)
{
    __shared__ StatAcc sh[4];
    const int64_t pth = path_list ? (int64_t)path_list[blockIdx.x] : (int64_t)blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    StatAcc s;
#pragma unroll
    for (int k = 0; k < 9; ++k) s.a[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) s.b[k] = 0;
    FieldStatView fv = {};
    if (fields) fv.load(fields, cst, pth);
    for (int64_t t = tile_first[pth] + threadIdx.x; t < tile_first[pth + 1]; t += 256) stat_entry(s, t, partial, ids, run_count, tiles, fv, prims, cst, clear_counts);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int k = 0; k < 6; ++k) s.a[k] += __shfl_xor(s.a[k], o);
#pragma unroll
        for (int k = 6; k < 9; ++k) s.a[k] = fmax(s.a[k], __shfl_xor(s.a[k], o));
#pragma unroll
        for (int k = 0; k < 4; ++k) s.b[k] += __shfl_xor(s.b[k], o);
    }
    if (lane == 0) sh[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w2 = 1; w2 < 4; ++w2) {
            for (int k = 0; k < 6; ++k) s.a[k] += sh[w2].a[k];
            for (int k = 6; k < 9; ++k) s.a[k] = fmax(s.a[k], sh[w2].a[k]);
            for (int k = 0; k < 4; ++k) s.b[k] += sh[w2].b[k];
        }
        stat_store(s, pth, n_adjusted, stats);
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

// corner turns as a standalone operator (MLP:1580-1608, 1024-1084, 1154-1288): one thread per (corner, sample slot)
__global__ void k_corner_turns(int64_t n, const double *__restrict__ corners, const int32_t *__restrict__ ci_arr,
                               const int32_t *__restrict__ rev_arr, double R, double L, double H, int stride,
                               double *__restrict__ out, int32_t *__restrict__ counts)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n * stride) return;
    const int64_t c = g / stride;
    const int s = (int)(g - c * stride);
    const double cx = corners[2 * c], cy = corners[2 * c + 1];
    const int ci = (ci_arr[c] >= 0 && ci_arr[c] <= 2) ? ci_arr[c] : 3;
    constexpr int NT = 15;
    const double step = kHalfPi / (double)(NT - 1);
    auto arc = [&](int k, double &x, double &y) {
        double sn, cs;
        sincos(linspace_at(0.0, kHalfPi, step, NT, k), &sn, &cs);
        corner_arc_point(ci, cx, cy, R, cs, sn, x, y);
    };
    int n_rev = 0;
    double e1x = 0, e1y = 0, dx = -1.0, dy = 0.0, len = 0;
    if (rev_arr[c]) {
        double e2x, e2y;
        arc(NT - 1, e1x, e1y); arc(NT - 2, e2x, e2y);
        const double tx = e1x - e2x, ty = e1y - e2y, nrm = sqrt(tx * tx + ty * ty);     // MLP:1188-1193
        if (nrm > 1e-6) { dx = -tx / nrm; dy = -ty / nrm; }
        double best = 0; bool have = false;                                               // MLP:1241-1281
        auto take = [&](double t) { if (t > 0 && (!have || t < best)) { best = t; have = true; } };
        if (fabs(dx) > 1e-6) { take((0 - e1x) / dx); take((L - e1x) / dx); }
        if (fabs(dy) > 1e-6) { take((0 - e1y) / dy); take((H - e1y) / dy); }
        len = have ? fmin(best, 3.0 * R) : 2.0 * R;
        n_rev = max(10, (int)(len / 0.5));                                                // MLP:1214
        if (NT + n_rev > stride) n_rev = stride - NT;
    }
    if (s == 0) { counts[2 * c] = NT; counts[2 * c + 1] = n_rev; }
    double x, y;
    if (s < NT) arc(s, x, y);
    else if (s - NT < n_rev) {
        const double t = linspace_at(0.0, len, n_rev > 1 ? len / (double)(n_rev - 1) : 0.0, n_rev, s - NT);
        x = e1x + t * dx; y = e1y + t * dy;                                               // MLP:1215-1216
    } else return;
    out[2 * g] = x; out[2 * g + 1] = y;
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
                d = ((unsigned)a < (unsigned)n && (unsigned)b < (unsigned)n) ? D[(int64_t)a * n + b] : __builtin_nan("");   // precondition, include/fcpp.h
            }
            const int m = min(64, n - base);
            const int dh = __double2hiint(d), dl = __double2loint(d);      // (left to right; a scalar lane index: v_readlane, not the LDS crossbar)
            if (m == 64) {
#pragma unroll
                for (int l = 0; l < 64; ++l) total += __hiloint2double(__builtin_amdgcn_readlane(dh, l), __builtin_amdgcn_readlane(dl, l));
            } else {
                for (int l = 0; l < m; ++l) total += __hiloint2double(__builtin_amdgcn_readlane(dh, l), __builtin_amdgcn_readlane(dl, l));
            }
        }
    } else {
        for (int k = lane; k < n; k += 64) {
            const int a = r[k], b = r[k + 1 == n ? 0 : k + 1];
            total += ((unsigned)a < (unsigned)n && (unsigned)b < (unsigned)n) ? D[(int64_t)a * n + b] : __builtin_nan("");
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
    }
    if (lane == 0) {
        if (dist) dist[ch] = total;
        if (fit) fit[ch] = 1.0 / (total + 1e-6);   // GA:172
    }
}

// centroid distance matrix (MVP:229-259, MFP:263-288): one thread per entry, rows coalesced
__global__ void k_distance_matrix(int n, const double *__restrict__ x, const double *__restrict__ y, double *__restrict__ D)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= n) return;
    const double dx = x[i] - x[j], dy = y[i] - y[j];
    D[(int64_t)i * n + j] = i == j ? 0.0 : sqrt(dx * dx + dy * dy);
}

// shortest exit -> entry connection per node pair (MFP:290-320): one wavefront per pair scans the candidate products in
// order; ties go to the first pair in (exit-major, entry-minor) order
__global__ __launch_bounds__(64) void k_best_connections(int64_t n_pairs, const int64_t *__restrict__ fo, const int64_t *__restrict__ to,
                                                         const double *__restrict__ fx, const double *__restrict__ fy,
                                                         const double *__restrict__ tx, const double *__restrict__ ty,
                                                         int32_t *__restrict__ bf, int32_t *__restrict__ bt, double *__restrict__ bd)
{
    const int64_t p = blockIdx.x;
    if (p >= n_pairs) return;
    const int lane = threadIdx.x;
    const int64_t f0 = fo[p], nf = fo[p + 1] - f0, t0 = to[p], nt = to[p + 1] - t0, total = nf * nt;
    double best = FCPP_INF;
    int64_t bk = -1;
    for (int64_t k = lane; k < total; k += 64) {
        const int64_t a = k / nt, b = k - a * nt;
        const double dx = fx[f0 + a] - tx[t0 + b], dy = fy[f0 + a] - ty[t0 + b];
        const double d = sqrt(dx * dx + dy * dy);
        if (d < best) { best = d; bk = k; }            // (k ascends per lane: the lane's first minimum)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o);
        const int64_t ok = __shfl_xor(bk, o);
        if (ok >= 0 && (bk < 0 || ob < best || (ob == best && ok < bk))) { best = ob; bk = ok; }
    }
    if (lane == 0) {
        bf[p] = bk < 0 ? -1 : (int32_t)(f0 + bk / nt);
        bt[p] = bk < 0 ? -1 : (int32_t)(t0 + bk % nt);
        bd[p] = best;
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
    FCPP_LAUNCH(k_generate, dim3((unsigned)n_tiles), dim3(BLOCK), 0, st, tiles, fields, prims, cst, x, y, v, fs);
    FCPP_LAUNCH_CHECK();
    return 0;
}

int launch_curv_clamp(hipStream_t st, int64_t n_tiles, const DevTile *tiles, const DevPath *paths,
                      const DevConst &cst, int do_clamp, const double *x, const double *y, const double *v_in,
                      double *v_out, double *kappa, unsigned long long *n_adjusted)
{
    if (n_tiles <= 0) return 0;
    FCPP_LAUNCH(k_curv_clamp, dim3((unsigned)n_tiles), dim3(BLOCK), 0, st, tiles, paths, cst, do_clamp, x, y,
                       v_in, v_out, kappa, n_adjusted);
    FCPP_LAUNCH_CHECK();
    return 0;
}

int launch_scan_tiles(hipStream_t st, int64_t n_tiles, const DevTile *tiles, const DevPath *paths, const DevConst &cst,
                      const double *x, const double *y, const double *v_in, void *agg_f, void *agg_b)
{
    if (n_tiles <= 0) return 0;
    FCPP_LAUNCH(k_scan_tiles, dim3((unsigned)n_tiles), dim3(BLOCK), 0, st, tiles, paths, cst, x, y, v_in,
                       (Agg *)agg_f, (Agg *)agg_b);
    FCPP_LAUNCH_CHECK();
    return 0;
}

int64_t spine_scratch_bytes(int64_t n_tiles)
{
    const int64_t nb = (n_tiles + SPINE_BLOCK - 1) / SPINE_BLOCK;
    return n_tiles > SPINE_BLOCK ? nb * (int64_t)(2 * sizeof(Agg) + 2 * sizeof(double)) : 0;
}

int launch_scan_spine(hipStream_t st, int64_t n_tiles, const void *agg_f, const void *agg_b, double *carry_f,
                      double *carry_b, void *scratch)
{
    if (n_tiles <= 0) return 0;
    if (n_tiles <= SPINE_BLOCK || !scratch) {       // one workgroup walks every tile aggregate
        FCPP_LAUNCH(k_scan_spine, dim3(1), dim3(BLOCK), 0, st, n_tiles, (const Agg *)agg_f, (const Agg *)agg_b, carry_f, carry_b);
        FCPP_LAUNCH_CHECK();
        return 0;
    }
    const int64_t nb = (n_tiles + SPINE_BLOCK - 1) / SPINE_BLOCK;
    Agg *block_f = (Agg *)scratch, *block_b = block_f + nb;
    double *cin_f = (double *)(block_b + nb), *cin_b = cin_f + nb;
    const dim3 grid((unsigned)nb, 2u);
    FCPP_LAUNCH(k_spine_blocks<false>, grid, dim3(BLOCK), 0, st, n_tiles, nb, (const Agg *)agg_f, (const Agg *)agg_b, block_f, block_b,
                (const double *)nullptr, (const double *)nullptr, (double *)nullptr, (double *)nullptr);
    FCPP_LAUNCH_CHECK();
    FCPP_LAUNCH(k_scan_spine, dim3(1), dim3(BLOCK), 0, st, nb, (const Agg *)block_f, (const Agg *)block_b, cin_f, cin_b);
    FCPP_LAUNCH_CHECK();
    FCPP_LAUNCH(k_spine_blocks<true>, grid, dim3(BLOCK), 0, st, n_tiles, nb, (const Agg *)agg_f, (const Agg *)agg_b, block_f, block_b,
                (const double *)cin_f, (const double *)cin_b, carry_f, carry_b);
    FCPP_LAUNCH_CHECK();
    return 0;
}

int launch_scan_apply(hipStream_t st, int64_t n_tiles, const DevTile *tiles, const DevPath *paths, const DevConst &cst,
                      int min_n, const double *x, const double *y, const double *v_in, double *v_out,
                      const double *carry_f, const double *carry_b)
{
    if (n_tiles <= 0) return 0;
    FCPP_LAUNCH(k_scan_apply, dim3((unsigned)n_tiles), dim3(BLOCK), 0, st, tiles, paths, cst, min_n, x, y, v_in,
                       v_out, carry_f, carry_b);
    FCPP_LAUNCH_CHECK();
    return 0;
}

int launch_validate(hipStream_t st, int64_t n_tiles, const DevTile *tiles, const DevPath *paths,
                    const DevField *fields, const DevConst &cst, const DevObstacles &obs, const double *x,
                    const double *y, const double *kappa, const double *v, uint32_t *fs, TilePartial *partial)
{
    if (n_tiles <= 0) return 0;
    FCPP_LAUNCH(k_validate, dim3((unsigned)n_tiles), dim3(BLOCK), 0, st, tiles, paths, fields, cst, obs, x, y,
                       kappa, v, fs, partial);
    FCPP_LAUNCH_CHECK();
    return 0;
}

// ---- fcpp_validate: geofence / obstacle flags of CALLER-SUPPLIED paths against arbitrary simple polygons -----------------------------------
// One workgroup per tile of one path.  The path's field polygon (polygon p of `field`) and its obstacle polygons are staged in LDS by
// coalesced loads (up to VAL_LDS_VERTS vertices together; beyond that they are read from global memory), every thread then tests its
// points: even-odd crossing number against the field polygon, the signed distance to its boundary against the tolerance
//     s = +dist inside, -dist outside;   FCPP_FLAG_OUTSIDE  <=>  s < -tol
// (for a convex field and tol >= 0 the half-plane rule of the planner, except in the wedges beyond its corners, where the distance to the
// corner decides), and the crossing number against every obstacle polygon (FCPP_FLAG_OBSTACLE).  Counts go to stats[p] by integer atomics.
struct DevPolySet { const int64_t *off; const double *x, *y; int64_t n; };
__device__ __forceinline__ bool pip_even_odd(double px, double py, const double *vx, const double *vy, int64_t a0, int64_t a1)
{
    bool in = false;
    for (int64_t k = a0, q = a1 - 1; k < a1; q = k++) {
        const double xi = vx[k], yi = vy[k], xj = vx[q], yj = vy[q];
        if (((yi > py) != (yj > py)) && (px < (xj - xi) * (py - yi) / (yj - yi) + xi)) in = !in;
    }
    return in;
}
// squared distance of (px, py) to the polygon's boundary (its segments, the closing one included)
__device__ __forceinline__ double poly_dist2(double px, double py, const double *vx, const double *vy, int64_t a0, int64_t a1)
{
    double best = __builtin_huge_val();
    for (int64_t k = a0, q = a1 - 1; k < a1; q = k++) {
        const double ax = vx[q], ay = vy[q], bx = vx[k] - ax, by = vy[k] - ay, wx = px - ax, wy = py - ay;
        const double len2 = bx * bx + by * by, dot = wx * bx + wy * by;
        double t = len2 > 0.0 ? dot / len2 : 0.0;
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
        const double dx = wx - t * bx, dy = wy - t * by, d2 = dx * dx + dy * dy;
        if (d2 < best) best = d2;
    }
    return best;
}
constexpr int VAL_LDS_POLYS = 128;      // obstacle polygons of a path whose bounding boxes a tile keeps in LDS (more: every polygon is tested)
constexpr int VAL_TILES_PER_WG = 16;
__global__ __launch_bounds__(BLOCK) void k_validate_polys(int64_t n_tiles, const DevTile *__restrict__ tiles, const DevPath *__restrict__ paths, DevPolySet field,
                                                          DevPolySet obst, const int64_t *__restrict__ obst_range, double tol, double a_lat,
                                                          const double *__restrict__ x, const double *__restrict__ y,
                                                          const double *__restrict__ kappa, const double *__restrict__ v,
                                                          uint32_t *__restrict__ flags, fcpp_field_stats *__restrict__ stats)
{
    __shared__ double lx[VAL_LDS_VERTS], ly[VAL_LDS_VERTS];
    __shared__ double lbb[VAL_LDS_POLYS][4];
    // a workgroup takes VAL_TILES_PER_WG consecutive tiles: a path's polygons are staged once for all of them that belong to it
    int staged_field = -1;
    for (int64_t ti = (int64_t)blockIdx.x * VAL_TILES_PER_WG; ti < n_tiles && ti < ((int64_t)blockIdx.x + 1) * VAL_TILES_PER_WG; ++ti) {
    const DevTile t = tiles[ti];
    const DevPath p = paths[t.field];
    // the path's polygons: vertices [f0, f1) of the field set, obstacle polygons [o0, o1)
    int64_t f0 = 0, f1 = 0, o0 = 0, o1 = 0, ov0 = 0, ov1 = 0;
    if (field.off && t.field < field.n) { f0 = field.off[t.field]; f1 = field.off[t.field + 1]; }
    if (obst.off && obst.n > 0) {
        o0 = obst_range ? obst_range[t.field] : 0; o1 = obst_range ? obst_range[t.field + 1] : obst.n;
        ov0 = obst.off[o0]; ov1 = obst.off[o1];
    }
    const int64_t nf = f1 - f0, no = ov1 - ov0;
    const bool staged = nf + no <= VAL_LDS_VERTS;
    const bool restage = t.field != staged_field;          // (workgroup-uniform)
    if (staged && restage) {
        __syncthreads();                                     // (the previous tile's readers are done)
        for (int k = threadIdx.x; k < (int)nf; k += BLOCK) { lx[k] = field.x[f0 + k]; ly[k] = field.y[f0 + k]; }
        for (int k = threadIdx.x; k < (int)no; k += BLOCK) { lx[nf + k] = obst.x[ov0 + k]; ly[nf + k] = obst.y[ov0 + k]; }
        __syncthreads();
    }
    const double *fx = staged ? lx : field.x + f0, *fy = staged ? ly : field.y + f0;            // field vertex k at fx[k]
    const double *ox = staged ? lx + nf : obst.x + ov0, *oy = staged ? ly + nf : obst.y + ov0;   // obstacle vertex (global index k) at ox[k - ov0]
    // One bounding box per obstacle polygon, made once per tile: a wavefront's 64 points are neighbours on the path, their own box meets
    // few of the obstacles' boxes (on a 0.05 m path usually none), and only those polygons are tested point by point -- 12.8 -> 1.x ms on
    // cfg3's 6.3e7 points against 32 obstacles (tools/bench_validate.py).
    const bool culled = staged && o1 - o0 <= VAL_LDS_POLYS;
    if (culled && restage) {
        for (int k = threadIdx.x; k < (int)(o1 - o0); k += BLOCK) {
            const int64_t a0 = obst.off[o0 + k] - ov0, a1 = obst.off[o0 + k + 1] - ov0;
            double bx0 = FCPP_INF, by0 = FCPP_INF, bx1 = -FCPP_INF, by1 = -FCPP_INF;
            for (int64_t q = a0; q < a1; ++q) { bx0 = fmin(bx0, ox[q]); bx1 = fmax(bx1, ox[q]); by0 = fmin(by0, oy[q]); by1 = fmax(by1, oy[q]); }
            lbb[k][0] = bx0; lbb[k][1] = by0; lbb[k][2] = bx1; lbb[k][3] = by1;
        }
        __syncthreads();
    }
    staged_field = t.field;
    const double tol2 = tol * tol;
    long long nout = 0, nobs = 0;
    for (int j0 = 0; j0 < t.count; j0 += BLOCK) {          // (every wavefront goes through the same rounds: the reductions below are wave-wide)
        const int j = j0 + (int)threadIdx.x;
        const bool has = j < t.count;
        const int64_t i = t.start + (has ? j : t.count - 1), g = p.off + i;
        const double px = x[g], py = y[g];
        double mnx = has ? px : FCPP_INF, mxx = has ? px : -FCPP_INF, mny = has ? py : FCPP_INF, mxy = has ? py : -FCPP_INF;
        if (culled) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                mnx = fmin(mnx, __shfl_xor(mnx, o)); mny = fmin(mny, __shfl_xor(mny, o));
                mxx = fmax(mxx, __shfl_xor(mxx, o)); mxy = fmax(mxy, __shfl_xor(mxy, o));
            }
        }
        if (!has) continue;
        uint32_t fs = 0;
        if (i > 0 && i < p.n - 1) {          // MLP:1383-1401, as k_validate counts n_viol
            const double ms = v[g] / 3.6;
            if (ms * ms * kappa[g] > a_lat) fs |= FCPP_FLAG_ALAT;
        }
        if (nf >= 3) {
            const bool in = pip_even_odd(px, py, fx, fy, 0, nf);
            bool out;
            if (in) out = tol < 0.0 && poly_dist2(px, py, fx, fy, 0, nf) < tol2;
            else out = tol < 0.0 || poly_dist2(px, py, fx, fy, 0, nf) > tol2;
            if (out) { fs |= FCPP_FLAG_OUTSIDE; ++nout; }
        }
        for (int64_t b = o0; b < o1; ++b) {
            if (culled) {          // (the same for every lane of the wavefront: its points' box against the polygon's)
                const double *bb = lbb[b - o0];
                if (bb[0] > mxx || bb[2] < mnx || bb[1] > mxy || bb[3] < mny) continue;
            }
            const int64_t a0 = obst.off[b], a1 = obst.off[b + 1];
            if (a1 - a0 >= 3 && pip_even_odd(px, py, ox, oy, a0 - ov0, a1 - ov0)) { fs |= FCPP_FLAG_OBSTACLE; ++nobs; break; }
        }
        flags[g] = fs;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { nout += __shfl_xor(nout, o); nobs += __shfl_xor(nobs, o); }
    if ((threadIdx.x & 63) == 0) {
        if (nout) atomicAdd(reinterpret_cast<unsigned long long *>(&stats[t.field].n_outside), (unsigned long long)nout);
        if (nobs) atomicAdd(reinterpret_cast<unsigned long long *>(&stats[t.field].n_in_obstacle), (unsigned long long)nobs);
    }
    }
}

int launch_validate_polys(hipStream_t st, int64_t n_tiles, const DevTile *tiles, const DevPath *paths, const int64_t *field_off, const double *field_x,
                          const double *field_y, int64_t n_field, const int64_t *obst_off, const double *obst_x, const double *obst_y, int64_t n_obst,
                          const int64_t *obst_range, double tol, double a_lat, const double *x, const double *y, const double *kappa, const double *v,
                          uint32_t *flags, fcpp_field_stats *stats)
{
    if (n_tiles <= 0) return 0;
    const DevPolySet f = { field_off, field_x, field_y, n_field }, o = { obst_off, obst_x, obst_y, n_obst };
    hipLaunchKernelGGL(k_validate_polys, dim3((unsigned)((n_tiles + VAL_TILES_PER_WG - 1) / VAL_TILES_PER_WG)), dim3(BLOCK), 0, st, n_tiles, tiles, paths, f, o, obst_range,
                       tol, a_lat, x, y, kappa, v, flags, stats);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// Batch creation (fused pipeline): the statistics slots of a batch, one per entry.  The slot of a quiet run gets the run's closed-form
// length / time / curvature statistics -- geometry and nominal speeds only, the same at every step --, every other slot zeros (wave
// tiles and general tiles overwrite theirs at every step).
__global__ void k_run_consts(int64_t n_entries, const int32_t *__restrict__ ids, const int64_t *__restrict__ run_count,
                             const DevTile *__restrict__ tiles, const DevField *__restrict__ fields, const DevPrim *__restrict__ prims, DevConst cst,
                             TilePartial *__restrict__ partial)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_entries) return;
    TilePartial tp;
    memset(&tp, 0, sizeof tp);
    const int64_t rc = run_count[e];
    if (rc > 0) {
        const DevTile tl = tiles[ids[e]];
        FieldStatView fv = {};
        fv.load(fields, cst, tl.field);
        const DevRun run = { ids[e], 0, rc };
        tp = quiet_run_partial(run, tl, fv, prims, cst);
        tp.n_viol = tp.n_outside = tp.n_in_obstacle = tp.n_adjusted = 0;
    }
    partial[e] = tp;
}

int launch_run_consts(hipStream_t st, int64_t n_entries, const int32_t *ids, const int64_t *run_count, const DevTile *tiles, const DevField *fields,
                      const DevPrim *prims, const DevConst &cst, TilePartial *partial)
{
    if (n_entries <= 0) return 0;
    hipLaunchKernelGGL(k_run_consts, dim3((unsigned)((n_entries + 255) / 256)), dim3(256), 0, st, n_entries, ids, run_count, tiles, fields, prims, cst, partial);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// The chunk lists of k_plan_quiet from the chunk groups the host listed (fcpp_tiler.cpp: derive_field, whose loop this is, with the same
// integer arithmetic).  A wavefront per SEGMENT of a group (at most CHUNK_SEGMENT chunks), a lane per chunk: chunk j of a group begins
// c_first + (j - 1) 512 points into it (the group is cut on 512-point boundaries of the batch arrays), the run it begins in is found in
// the runs' first tiles (tiles[e0 + k].start: a binary search), and its record is that run's tile with the chunk's start, count and
// statistics entry; a chunk of a layer-1 span, or one that holds the end of one run and the start of the next, goes to the span list
// with (pass, offset in the pass) of its first point.  Either list keeps chunk order: places by ballot.
__global__ __launch_bounds__(64) void k_expand_chunks(int64_t n_segments, const DevChunkGroup *__restrict__ groups, const DevTile *__restrict__ tiles,
                                                      const DevField *__restrict__ fields, DevTile *__restrict__ chunks, DevTile *__restrict__ span_chunks)
{
    if ((int64_t)blockIdx.x >= n_segments) return;
    const DevChunkGroup g = groups[blockIdx.x];
    const int lane = threadIdx.x;
    const DevField &F = fields[g.field];
    const int64_t pass = (int64_t)F.n_line + F.n_turn;
    const DevTile a = tiles[g.e0];
    const int64_t room0 = TILE_POINTS - (g.g0 % TILE_POINTS), c_first = g.total < room0 ? g.total : room0;
    int64_t nc = 0, ns = 0;
    for (int it = 0; it < g.n; it += 64) {
        const int64_t j = (int64_t)g.j0 + it + lane;
        const bool on = it + lane < g.n;
        bool span = false;
        DevTile ch = a;
        if (on) {
            const int64_t done = j == 0 ? 0 : c_first + (j - 1) * TILE_POINTS;
            const int64_t left = g.total - done;
            const int64_t c = j == 0 ? c_first : (left < TILE_POINTS ? left : TILE_POINTS);
            // the run that holds the chunk's first point: the last one that begins at or before it
            const int64_t at = a.start + done;
            int lo = 0, hi = g.n_runs - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tiles[g.e0 + mid].start <= at) lo = mid; else hi = mid - 1; }
            const DevTile tr = tiles[g.e0 + lo];
            const int64_t rc_begin = tr.start - a.start;
            const int64_t rc_end = lo + 1 < g.n_runs ? tiles[g.e0 + lo + 1].start - a.start : g.total;
            ch = tr;
            ch.start = at; ch.count = (int32_t)c; ch.stat_tile = g.e0 + lo;
            const bool one_run = done + c <= rc_end;
            if (one_run && tr.quiet != 4) ch.off0 = (int32_t)(tr.off0 + (done - rc_begin));
            else {
                ch.quiet = 4;
                ch.idx0 = (int32_t)(at / pass); ch.off0 = (int32_t)(at % pass);
            }
            span = ch.quiet == 4;
        }
        const unsigned long long ms = __ballot(on && span), mc = __ballot(on && !span), below = (1ull << lane) - 1ull;
        if (on) {
            if (span) span_chunks[g.span_base + ns + __popcll(ms & below)] = ch;
            else chunks[g.chunk_base + nc + __popcll(mc & below)] = ch;
        }
        ns += __popcll(ms); nc += __popcll(mc);
    }
}

int launch_expand_chunks(hipStream_t st, int64_t n_segments, const DevChunkGroup *groups, const DevTile *tiles, const DevField *fields,
                         DevTile *chunks, DevTile *span_chunks)
{
    if (n_segments <= 0) return 0;
    hipLaunchKernelGGL(k_expand_chunks, dim3((unsigned)n_segments), dim3(64), 0, st, n_segments, groups, tiles, fields, chunks, span_chunks);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// group = lanes per path: 8, 64, or 256 (a workgroup per path)
int launch_reduce_stats(hipStream_t st, int64_t n_list, TilePartial *partial, const int64_t *tile_first,
                        const unsigned long long *n_adjusted, fcpp_field_stats *stats, const int32_t *ids, const int64_t *run_count,
                        const DevTile *tiles, const DevField *fields, const DevPrim *prims, const DevConst *cst, const int32_t *path_list,
                        int group, void *scratch, int clear_counts)
{
    if (n_list <= 0) return 0;
    DevConst c0;
    memset(&c0, 0, sizeof c0);
    const DevConst &c = cst ? *cst : c0;
    if (group == 8)
        FCPP_LAUNCH(k_reduce_stats<8>, dim3((unsigned)((n_list + 31) / 32)), dim3(256), 0, st, n_list, path_list, tile_first, partial, n_adjusted,
                    stats, ids, run_count, tiles, fields, prims, c, clear_counts);
    else if (group == 256 && scratch) {
        FCPP_LAUNCH(k_reduce_stats_slice, dim3((unsigned)(n_list * REDUCE_SPLIT)), dim3(256), 0, st, path_list, tile_first, partial, ids, run_count,
                    tiles, fields, prims, c, (StatAcc *)scratch, clear_counts);
        hipLaunchKernelGGL(k_reduce_stats_join, dim3((unsigned)n_list), dim3(64), 0, st, path_list, (const StatAcc *)scratch, n_adjusted, stats);
    } else if (group == 256)
        FCPP_LAUNCH(k_reduce_stats_wg, dim3((unsigned)n_list), dim3(256), 0, st, n_list, path_list, tile_first, partial, n_adjusted, stats, ids,
                    run_count, tiles, fields, prims, c, clear_counts);
    else
        FCPP_LAUNCH(k_reduce_stats<64>, dim3((unsigned)((n_list + 3) / 4)), dim3(256), 0, st, n_list, path_list, tile_first, partial, n_adjusted,
                    stats, ids, run_count, tiles, fields, prims, c, clear_counts);
    FCPP_LAUNCH_CHECK();
    return 0;
}

int launch_distance_matrix(hipStream_t st, int n, const double *x, const double *y, double *D)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_distance_matrix, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, st, n, x, y, D);
    FCPP_LAUNCH_CHECK();
    return 0;
}

int launch_best_connections(hipStream_t st, int64_t n_pairs, const int64_t *fo, const int64_t *to, const double *fx, const double *fy,
                            const double *tx, const double *ty, int32_t *bf, int32_t *bt, double *bd)
{
    if (n_pairs <= 0) return 0;
    hipLaunchKernelGGL(k_best_connections, dim3((unsigned)n_pairs), dim3(64), 0, st, n_pairs, fo, to, fx, fy, tx, ty, bf, bt, bd);
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

int launch_corner_turns(hipStream_t st, int64_t n, const double *corners, const int32_t *ci, const int32_t *rev, double R, double L,
                        double H, int stride, double *out, int32_t *counts)
{
    if (n <= 0) return 0;
    const int64_t total = n * stride;
    hipLaunchKernelGGL(k_corner_turns, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, n, corners, ci, rev, R, L, H, stride, out,
                       counts);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
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
