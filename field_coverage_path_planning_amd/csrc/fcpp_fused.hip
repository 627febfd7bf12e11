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
    double ev[NWAVE], ek[NWAVE], evn[NWAVE];                   // last item of each wave: final v, kappa, nominal v
    double tr[NWAVE][TR_WORDS];
    RedSharedF R;
};

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// curvature (MLP:513-536) from the two chords and their lengths
__device__ __forceinline__ double curv_chords(double dx1, double dy1, double ds1, double dx2, double dy2, double ds2)
{
    if (ds1 < 1e-6 || ds2 < 1e-6) return 0.0;
    const double cr = dx1 * dy2 - dy1 * dx2, dt = dx1 * dx2 + dy1 * dy2;
    const double dth = atan2(cr, dt);
    return fabs(2 * dth / (ds1 + ds2));
}

__device__ __forceinline__ double clamp_speed(double v, double kappa, const DevConst &cst, int &adj)
{
    if (kappa > 1e-6) {                                           // MLP:496-504
        const double vmax_ms = sqrt(cst.a_lat / kappa) * cst.sf;
        const double vmax_kmh = vmax_ms * 3.6;
        if (v > vmax_kmh) { v = vmax_kmh; ++adj; }
    }
    return v;
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
    Agg total = { FCPP_INF, 0.0 };
    bool first = true;
    int64_t b = BACK ? edge - 64 : edge;
    for (;;) {
        const int64_t i = b + lane;
        const bool act = i >= 0 && i < n;
        GenOut g; g.x = g.y = g.v = 0; g.fs = 0;
        if (act) gen_point(f, prims, i, cst, g);
        double xm = __shfl_up(g.x, 1), ym = __shfl_up(g.y, 1), xp = __shfl_down(g.x, 1), yp = __shfl_down(g.y, 1);
        const int64_t ei = lane == 0 ? i - 1 : i + 1;
        if ((lane == 0 || lane == 63) && ei >= 0 && ei < n) {   // the chunk's two outer neighbours
            GenOut e;
            gen_point(f, prims, ei, cst, e);
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

__global__ __launch_bounds__(BLOCK) void k_plan_fused(const DevTile *__restrict__ tiles,
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

    // ---- 1. generate this thread's 8 consecutive points -------------------------------------------
    const int j0 = tid * IPT;
    double X[IPT + 2], Y[IPT + 2], vn[IPT];
    uint32_t fs[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        GenOut g; g.x = g.y = g.v = 0; g.fs = 0;
        if (j0 + k < cnt) gen_point(f, prims, s + j0 + k, cst, g);
        X[k + 1] = g.x; Y[k + 1] = g.y; vn[k] = g.v; fs[k] = g.fs;
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

    // ---- 2. segment lengths, curvature, clamp -----------------------------------------------------
    double d[IPT + 1];      // d[k] = |P(item k) - P(item k-1)|, item -1 / item IPT = the end neighbours
#pragma unroll
    for (int k = 0; k <= IPT; ++k) {
        const double dx = X[k + 1] - X[k], dy = Y[k + 1] - Y[k];
        d[k] = sqrt(dx * dx + dy * dy);
    }
    double kap[IPT], v0[IPT], c[IPT], wf[IPT], wb[IPT];
    int adj = 0;
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int64_t i = s + j0 + k;
        const bool in = j0 + k < cnt;
        double kk = 0.0;
        if (in && i > 0 && i < n - 1)
            kk = curv_chords(X[k + 1] - X[k], Y[k + 1] - Y[k], d[k], X[k + 2] - X[k + 1], Y[k + 2] - Y[k + 1], d[k + 1]);
        kap[k] = kk;
        v0[k] = in ? clamp_speed(vn[k], kk, cst, adj) : 0.0;
        const double ms = v0[k] / 3.6;
        c[k] = in ? ms * ms : FCPP_INF;
        // couplings: skipped steps (MLP:560-561, 576-577) and the path ends cut the propagation
        wf[k] = !in ? 0.0 : ((i == 0 || d[k] < 1e-6) ? FCPP_INF : two_a * d[k]);
        wb[k] = !in ? 0.0 : ((i == n - 1 || d[k + 1] < 1e-6) ? FCPP_INF : two_a * d[k + 1]);
    }

    // ---- 3. forward / backward sweeps as min-plus scans over registers ----------------------------
    Agg fa = { FCPP_INF, 0.0 }, ba = { FCPP_INF, 0.0 };
#pragma unroll
    for (int k = 0; k < IPT; ++k) { fa.c = fmin(c[k], fa.c + wf[k]); fa.w += wf[k]; }
#pragma unroll
    for (int k = IPT - 1; k >= 0; --k) { ba.c = fmin(c[k], ba.c + wb[k]); ba.w += wb[k]; }
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
    double u[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) { uf = fmin(c[k], uf + wf[k]); u[k] = uf; }
#pragma unroll
    for (int k = IPT - 1; k >= 0; --k) { ub = fmin(c[k], ub + wb[k]); u[k] = fmin(u[k], ub); }
    const double b_first = ub;   // backward value at this thread's first item
    double vf[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) vf[k] = (u[k] < c[k]) ? sqrt(u[k]) * 3.6 : v0[k];   // untouched points keep v0 exactly

    // ---- 4. previous point's final v / kappa / nominal v (for the segment metrics) ------------------
    double vprev = __shfl_up(vf[IPT - 1], 1), kprev = __shfl_up(kap[IPT - 1], 1), vnprev = __shfl_up(vn[IPT - 1], 1);
    if (lane == 63) { S.ev[wave] = vf[IPT - 1]; S.ek[wave] = kap[IPT - 1]; S.evn[wave] = vn[IPT - 1]; }
    __syncthreads();
    if (lane == 0) {
        if (wave > 0) { vprev = S.ev[wave - 1]; kprev = S.ek[wave - 1]; vnprev = S.evn[wave - 1]; }
        else if (S.back.valid) {
            // final value at s-1: forward part = carry_f, backward part = B(s) + w(s-1,s)
            const double up = fmin(carry_f, b_first + wf[0]);
            vprev = (up < S.back.u0) ? sqrt(up) * 3.6 : S.back.v0;
            kprev = S.back.kappa; vnprev = S.back.vnom;
        }
    }

    // ---- 5. validator + metrics (MLP:1290-1311, 1373-1424; geofence / obstacles) --------------------
    double s_len[2] = { 0, 0 }, s_tpre[2] = { 0, 0 }, s_t[2] = { 0, 0 }, mk = 0, ma = 0, mj = 0;
    long long nv = 0, nout = 0, nobs = 0;
    const int ob0 = f.obs_first, ob1 = f.obs_first + f.obs_count;
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int64_t i = s + j0 + k;
        if (j0 + k < cnt) {
            const double px = X[k + 1], py = Y[k + 1];
            const double vp = k == 0 ? vprev : vf[k - 1], kp = k == 0 ? kprev : kap[k - 1], vnp = k == 0 ? vnprev : vn[k - 1];
            if (i > 0 && i != f.n_main) {      // the seam main|headland belongs to neither layer
                const int layer = i > f.n_main ? 1 : 0;
                s_len[layer] += d[k];
                s_t[layer] += d[k] / fmax(((vp + vf[k]) / 2) / 3.6, 0.1);
                s_tpre[layer] += d[k] / fmax(((vnp + vn[k]) / 2) / 3.6, 0.1);
            }
            if (i > 0 && i < n - 1) {
                const double ms = vf[k] / 3.6, alat = ms * ms * kap[k];
                mk = fmax(mk, kap[k]); ma = fmax(ma, alat);
                if (alat > cst.a_lat) { ++nv; fs[k] |= FCPP_FLAG_ALAT; }
                if (i > 1) mj = fmax(mj, fabs(kap[k] - kp));
            }
            bool out = false;
#pragma unroll
            for (int e = 0; e < 4; ++e) out = out || (f.ex[e] * px + f.ey[e] * py + f.eo[e] < -cst.geofence_tol);
            if (out) { ++nout; fs[k] |= FCPP_FLAG_OUTSIDE; }
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
    long long iv[4] = { nv, nout, nobs, adj };
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int k = 0; k < 6; ++k) dv[k] += __shfl_xor(dv[k], o);
#pragma unroll
        for (int k = 6; k < 9; ++k) dv[k] = fmax(dv[k], __shfl_xor(dv[k], o));
#pragma unroll
        for (int k = 0; k < 4; ++k) iv[k] += __shfl_xor(iv[k], o);
    }
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

int launch_plan_fused(hipStream_t st, int64_t n_tiles, const DevTile *tiles, const DevField *fields,
                      const DevPrim *prims, const DevConst &cst, const DevObstacles &obs, double *x, double *y,
                      double *kappa, double *v, uint32_t *fs, TilePartial *partial)
{
    if (n_tiles <= 0) return 0;
    hipLaunchKernelGGL(k_plan_fused, dim3((unsigned)n_tiles), dim3(BLOCK), 0, st, tiles, fields, prims, cst, obs, x, y,
                       kappa, v, fs, partial);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

}  // namespace fcpp
