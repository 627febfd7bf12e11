// fcpp_sparse.hip -- pipeline B at SPARSE sampling: the general stretches of a path planned one point per lane.
//
// At the reference's own sampling (2 points per swath line, 20 per U-turn, 15 per corner arc, 20 per headland side, MLP:761-767,
// 807, 1046, 1013) consecutive samples are metres apart: a sweep constraint reaches over a handful of POINTS, not hundreds, and a
// lane that owns eight consecutive points (fcpp_fused.hip) walks across two or three primitives.  Here a wavefront owns a
// "wave tile": up to 64 consecutive points of one field,
//         [ Hb halo lanes | count output lanes | Hf halo lanes ]          Hb + count + Hf <= 64
// lane l evaluates point first + l once, neighbours come from the adjacent lanes, and everything a point needs -- chords,
// curvature, clamp, the two min-plus scans, segment metrics -- is a handful of instructions per lane with no loop over items,
// no LDS and no workgroup barrier.  The halo lanes make the tile self-contained: the HOST chooses them (Tiling::wave_tiles in
// fcpp_api.cpp, from the same primitives and turn templates) so that
//   * the sweeps cannot carry anything into the output lanes from beyond the halo: the couplings 2a|dp| summed over the halo
//     reach u_cap = (v_max/3.6)^2, or the halo ends at a skipped step (|dp| < 1e-6, MLP:560-561) or at the path's end;
//   * the point before the first output lane has its final speed and curvature (the segment metrics of MLP:1290-1311 and the
//     curvature jump of MLP:1404-1406 need them), which takes one more lane for its own stencil.
// The outermost halo lane only lends its coordinates to its neighbour's stencil; its own (possibly unclamped) speed cannot bind
// anything by the choice above.  Same arithmetic as fcpp_fused.hip (fcpp_pointfn.h) => results do not depend on which of the two
// kernels plans a stretch.  Stretches whose halos would not fit (dense sampling) stay with k_plan_fused.
#include "fcpp_pointfn.h"

namespace fcpp {

static constexpr int SP_WAVES = 4;     // wave tiles per workgroup (independent of each other: no barrier)

__global__ __launch_bounds__(64 * SP_WAVES) void k_plan_sparse(const int32_t *__restrict__ ids, int64_t n_ids,
                                                               const DevTile *__restrict__ tiles, const DevField *__restrict__ fields,
                                                               const DevPrim *__restrict__ prims, DevConst cst, DevObstacles obs,
                                                               double *__restrict__ xo, double *__restrict__ yo, double *__restrict__ ko,
                                                               double *__restrict__ vo, uint32_t *__restrict__ fso,
                                                               TilePartial *__restrict__ partial)
{
    __shared__ double obs_lds[SP_WAVES][2 * OBS_LDS_VERTS];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int64_t slot = (int64_t)blockIdx.x * SP_WAVES + wave;
    if (slot >= n_ids) return;
    const int tile_id = ids[slot];
    const DevTile tl = tiles[tile_id];
    const DevField &f = fields[tl.field];
    const int Hb = tl.stat_tile & 0xff, Hf = (tl.stat_tile >> 8) & 0xff;
    const int nl = Hb + tl.count + Hf;                       // active lanes
    const int64_t n = f.n_total, n_main = f.n_main, first = tl.start - Hb;
    const int64_t i = first + lane;
    const bool act = lane < nl && i >= 0 && i < n;
    const bool out = lane >= Hb && lane < Hb + tl.count;

    // ---- 1. the lane's point --------------------------------------------------------------------------------------------------
    double px = 0.0, py = 0.0;
    uint32_t fw = 0;
    if (act) {
        if (i < n_main) {       // layer 1: (pass, offset) from the tile's host-side decode of its first lane
            const unsigned per = (unsigned)(f.n_line + f.n_turn);
            const unsigned off = (unsigned)tl.off0 + (unsigned)lane, q = off / per;
            eval_main(f, cst, tl.idx0 + (int)q, (int)(off - q * per), px, py, fw);
        } else {                // layer 2: the tile's first primitive is known, a wave tile spans at most 8 more
            const int plast = f.prim_first + f.prim_count - 1;
            const int p0 = first >= n_main ? tl.idx0 : f.prim_first;
            int pi = p0;
#pragma unroll
            for (int k = 1; k <= 8; ++k) {
                const int pk = min(p0 + k, plast);                                  // (wave-uniform: scalar loads)
                pi += (p0 + k <= plast && i >= prims[pk].start) ? 1 : 0;
            }
            const DevPrim &p = prims[pi];
            eval_prim(p, cst, (int)(i - p.start), px, py);
            fw = p.fs;
        }
    }
    const double vn = nominal_speed(fw, cst), msn = nominal_ms(fw, cst);

    // ---- 2. chords, curvature (MLP:513-536), clamp (MLP:490-504) ----------------------------------------------------------------
    const double xm = lane_prev(px), ym = lane_prev(py), xp = lane_next(px), yp = lane_next(py);
    const bool has_prev = act && lane > 0 && i > 0;
    const double dx1 = px - xm, dy1 = py - ym;
    const double dprev = has_prev ? seg_len(dx1, dy1) : 0.0;                 // |p_i - p_(i-1)|
    const double dnext = lane_next(dprev);
    const bool interior = has_prev && lane < nl - 1 && i < n - 1;            // both neighbours are lanes of this wave
    double kappa = 0.0;
    if (interior) kappa = curv_chords(dx1, dy1, dprev, xp - px, yp - py, dnext);
    bool cl = false;
    double v0 = vn;
    if (kappa > 1e-6) v0 = clamped_speed(vn, kappa, cst, cl);
    const double ms0 = cl ? v0 / 3.6 : msn;
    const double u0 = act ? ms0 * ms0 : FCPP_INF;

    // ---- 3. sweeps (MLP:538-589) as min-plus scans over the lanes; skipped when no single step binds ----------------------------
    // w = coupling of segment (i-1, i); +inf: nothing propagates (skipped step, the wave's first lane, the path's first point)
    const double two_a = 2 * cst.a_lon;
    const double w = !act ? 0.0 : ((!has_prev || dprev < 1e-6) ? FCPP_INF : two_a * dprev);
    const double u0m = lane_prev(u0);
    const bool binds = has_prev && w < FCPP_INF && (u0m + w < u0 || u0 + w < u0m);
    double u = u0;
    if (__ballot(binds) != 0ull) {
        Agg fi = { u0, w };
        double wn = lane_next(w);                      // coupling to the next lane
        if (lane >= nl - 1) wn = act ? FCPP_INF : 0.0;
        Agg bi = { u0, wn };
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            Agg pf = { __shfl_up(fi.c, o), __shfl_up(fi.w, o) };
            Agg pb = { __shfl_down(bi.c, o), __shfl_down(bi.w, o) };
            if (lane >= o) fi = combine_after(pf, fi);
            if (lane + o < 64) bi = combine_after(pb, bi);
        }
        u = fmin(fi.c, bi.c);
    }
    // untouched points keep exactly their clamped / nominal value
    const double vfin = (u < u0) ? sqrt(u) * 3.6 : (cl ? v0 : vn);

    // ---- 4. validation flags ------------------------------------------------------------------------------------------------------
    int nout = 0, nobs = 0, nv = 0;
    if (out) {
        const double ntol = -cst.geofence_tol;
        bool o = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) o = o | (f.ex[e] * px + f.ey[e] * py + f.eo[e] < ntol);
        if (o) { fw |= FCPP_FLAG_OUTSIDE; nout = 1; }
    }
    if (f.obs_count > 0) {      // wave-uniform: bounding box of the wave's output points, then culled + LDS-staged polygon tests
        double mnx = out ? px : FCPP_INF, mxx = out ? px : -FCPP_INF, mny = out ? py : FCPP_INF, mxy = out ? py : -FCPP_INF;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mnx = fmin(mnx, __shfl_xor(mnx, o)); mny = fmin(mny, __shfl_xor(mny, o));
            mxx = fmax(mxx, __shfl_xor(mxx, o)); mxy = fmax(mxy, __shfl_xor(mxy, o));
        }
        const double ox[1] = { px }, oy[1] = { py };
        const unsigned m = obstacle_mask<1>(obs, f.obs_first, f.obs_first + f.obs_count, obs_lds[wave], mnx, mny, mxx, mxy, ox, oy, out ? 1 : 0);
        if (out && (m & 1u)) { fw |= FCPP_FLAG_OBSTACLE; nobs = 1; }
    }

    // ---- 5. metrics (MLP:1290-1311) and a_lat validation (MLP:1383-1408) on the output lanes ---------------------------------------
    const double vprev = lane_prev(vfin), kprev = lane_prev(kappa), vnprev = lane_prev(vn);
    double s_len[2] = { 0, 0 }, s_tpre[2] = { 0, 0 }, s_t[2] = { 0, 0 }, mk = 0, ma = 0, mj = 0;
    if (out && i > 0 && i != n_main) {                      // the seam main|headland belongs to neither layer
        const int layer = i > n_main ? 1 : 0;
        const double ms_pre = (vnprev == vn) ? msn : ((vnprev + vn) / 2) / 3.6;
        const double tpre = dprev / fmax(ms_pre, 0.1);
        const double t = (vprev == vnprev && vfin == vn) ? tpre : dprev / fmax(((vprev + vfin) / 2) / 3.6, 0.1);
        s_len[layer] = dprev; s_tpre[layer] = tpre; s_t[layer] = t;
    }
    if (out && i > 0 && i < n - 1) {                        // interior points of the path
        if (kappa > 0.0) {
            const double ms = vfin / 3.6, alat = ms * ms * kappa;
            mk = kappa; ma = alat;
            if (alat > cst.a_lat) { nv = 1; fw |= FCPP_FLAG_ALAT; }
        }
        if (kappa != kprev && i != 1) mj = fabs(kappa - kprev);          // |kappa_i - kappa_(i-1)| for i >= 2 (MLP:1404-1406)
    }

    // ---- 6. stores: consecutive lanes, consecutive addresses ------------------------------------------------------------------------
    if (out) {
        const int64_t g = f.pt_off + i;
        xo[g] = px; yo[g] = py; ko[g] = kappa; vo[g] = vfin; fso[g] = fw;
    }

    // ---- 7. the tile's partial statistics (fixed butterfly => run-to-run identical sums) ----------------------------------------------
    double dv[9] = { s_len[0], s_tpre[0], s_t[0], s_len[1], s_tpre[1], s_t[1], mk, ma, mj };
#pragma unroll
    for (int k = 0; k < 6; ++k) dv[k] = wave_sum_to63(dv[k]);
#pragma unroll
    for (int k = 6; k < 9; ++k) dv[k] = wave_max0_to63(dv[k]);
    // the flag counts are one bit per lane: population counts of ballots
    const long long i_v = __popcll(__ballot(nv != 0)), i_o = __popcll(__ballot(nout != 0)), i_b = __popcll(__ballot(nobs != 0)),
                    i_a = __popcll(__ballot(out && cl));
    if (lane == 63) {
        TilePartial tp;
        tp.main_len = dv[0]; tp.main_time_pre = dv[1]; tp.main_time = dv[2];
        tp.head_len = dv[3]; tp.head_time_pre = dv[4]; tp.head_time = dv[5];
        tp.max_kappa = dv[6]; tp.max_alat = dv[7]; tp.max_jump = dv[8];
        tp.n_viol = i_v; tp.n_outside = i_o; tp.n_in_obstacle = i_b; tp.n_adjusted = i_a;
        partial[tile_id] = tp;
    }
}

int launch_plan_sparse(hipStream_t st, int64_t n_ids, const int32_t *ids, const DevTile *tiles, const DevField *fields, const DevPrim *prims,
                       const DevConst &cst, const DevObstacles &obs, double *x, double *y, double *kappa, double *v, uint32_t *fs,
                       TilePartial *partial)
{
    if (n_ids <= 0) return 0;
    FCPP_LAUNCH(k_plan_sparse, dim3((unsigned)((n_ids + SP_WAVES - 1) / SP_WAVES)), dim3(64 * SP_WAVES), 0, st, ids, n_ids, tiles, fields,
                       prims, cst, obs, x, y, kappa, v, fs, partial);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

}  // namespace fcpp
