// fcpp_sparse_fn.h -- one wave tile of the sparse-sampling path as a device function (see fcpp_sparse.hip for the method), used by
// k_plan_sparse: one wave tile per wavefront, statistics per tile.
#pragma once
#include "fcpp_pointfn.h"

namespace fcpp {

// per-lane running statistics of the tiles a wavefront has planned, and its flag counts (wave-uniform)
struct SparseAcc {
    double s_len[2], s_tpre[2], s_t[2], mk, ma, mj;
    int c_viol, c_out, c_obs, c_adj;
    __device__ __forceinline__ void clear()
    {
        s_len[0] = s_len[1] = s_tpre[0] = s_tpre[1] = s_t[0] = s_t[1] = mk = ma = mj = 0.0;
        c_viol = c_out = c_obs = c_adj = 0;
    }
};

// The (at most nine) primitive records of a wave tile, copied into the wavefront's own LDS by two coalesced loads: every lane then
// reads its primitive's record from there.  (Read from global memory by every lane -- six 16-byte loads per point, 64 lanes asking
// for the same few cache lines -- the records were 190 of a wavefront's 350 accesses to the vector cache, which was the busiest unit of
// the kernel: TCP_TOTAL_CACHE_ACCESSES and TCP_PENDING_STALL_CYCLES, profiles/r03_sparse_mem_counters.txt.)
static constexpr int TILE_PRIMS_MAX = 9, PRIM_DOUBLES = (int)(sizeof(DevPrim) / sizeof(double)), TILE_STARTS_AT = TILE_PRIMS_MAX * PRIM_DOUBLES + 1,
                     TILE_PRIMS_LDS = TILE_STARTS_AT + 5;      // (behind the records: nine 32-bit sample-index bases, tile_starts)
static_assert(sizeof(DevPrim) % sizeof(double) == 0 && TILE_PRIMS_MAX * PRIM_DOUBLES <= 128, "two loads per lane stage a tile's primitives");
// The primitive of a tile's point: the number of primitives that start at or before it -- the record's thresholds as bits of a 128-bit mask
// built by the scalar unit, counted by v_bcnt below the point's own bit -- instead of eight compare / select rounds per point.
// lane l: points 2l (sa) and 2l + 1 (sb).
__device__ __forceinline__ void tile_slots2(const DevWaveTile &wt, int lane, int &sa, int &sb)
{
    unsigned long long mlo = 0, mhi = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned t = wt.thr[k];
        if (t < 64u) mlo |= 1ull << t;
        else if (t < 128u) mhi |= 1ull << (t - 64u);
    }
    const bool hi = lane >= 32;
    const unsigned long long m = hi ? mhi : mlo;
    const int p = (2 * lane) & 63;
    sa = __popcll(m & ((2ull << p) - 1ull)) + (hi ? __popcll(mlo) : 0);
    sb = sa + (int)((m >> (p + 1)) & 1ull);
}

// PACKED: `prims` is the tile's own copy of its primitives (DevFieldPack: nine slots, the tile's first primitive in slot 0), whose address
// does not depend on the tile record -- the two loads are issued before the record has arrived
template <bool PACKED = false>
__device__ __forceinline__ void stage_tile_prims(const DevWaveTile &wt, const DevPrim *__restrict__ prims, double *plds, int nl)
{
    const int lane = threadIdx.x & 63;
    if (PACKED) {
        const double *src = reinterpret_cast<const double *>(prims);
        const double v0 = src[lane], v1 = lane + 64 < TILE_PRIMS_MAX * PRIM_DOUBLES ? src[lane + 64] : 0.0;
        plds[lane] = v0;
        if (lane + 64 < TILE_PRIMS_MAX * PRIM_DOUBLES) plds[lane + 64] = v1;
    } else {
        if (wt.rel_main >= nl) return;                           // (wave-uniform) no point of layer 2 in this tile
        int np = 1;
#pragma unroll
        for (int k = 0; k < 8; ++k) np += wt.thr[k] != 255 ? 1 : 0;
        const double *src = reinterpret_cast<const double *>(prims + wt.p0);
        const int nw = np * PRIM_DOUBLES;
        if (lane < nw) plds[lane] = src[lane];
        if (lane + 64 < nw) plds[lane + 64] = src[lane + 64];
    }
    // the sample index of a point in its primitive is its index in the tile minus tile_starts[slot]: -r0 for the tile's first
    // primitive, the primitive's first point (the record's threshold) for the others
    if (lane < TILE_PRIMS_MAX) {
        const unsigned long long th = (unsigned long long)wt.thr[0] | (unsigned long long)wt.thr[1] << 8 | (unsigned long long)wt.thr[2] << 16 |
                                      (unsigned long long)wt.thr[3] << 24 | (unsigned long long)wt.thr[4] << 32 | (unsigned long long)wt.thr[5] << 40 |
                                      (unsigned long long)wt.thr[6] << 48 | (unsigned long long)wt.thr[7] << 56;
        reinterpret_cast<int *>(plds + TILE_STARTS_AT)[lane] = lane == 0 ? -wt.r0 : (int)((th >> (8 * (lane - 1))) & 255ull);
    }
    wave_sync();
}

// obs_lds: 2 * OBS_LDS_VERTS doubles of LDS owned by this wavefront (only touched when the field has obstacles); atab: the staged
// table of atan2_abs_dev (atan_tab_stage); plds: TILE_PRIMS_LDS doubles of LDS owned by this wavefront
__device__ __forceinline__ void sparse_tile(const DevWaveTile &wt, const DevField &f, const DevPrim *__restrict__ prims, const DevConst &cst,
                                            const DevObstacles &obs, double *obs_lds, const double *atab, double *plds, double *__restrict__ xo, double *__restrict__ yo,
                                            double *__restrict__ ko, double *__restrict__ vo, uint32_t *__restrict__ fso, SparseAcc &acc)
{
    const int lane = threadIdx.x & 63;
    const int Hb = wt.hb, nl = wt.hb + wt.count + wt.hf;     // active lanes
    const bool act = lane < nl;
    const bool out = lane >= Hb && lane < Hb + wt.count;
    // the lane's position on the path, relative to the tile's first lane (32-bit throughout)
    const bool in_main = lane < wt.rel_main;                 // generated from layer 1's closed form
    const bool is_first = lane == wt.rel_zero;               // the path's first point
    const bool is_last = lane == wt.rel_last;                // the path's last point
    const bool at_seam = lane == wt.rel_seam;                // first point of layer 2
    const bool is_second = lane == wt.rel_zero + 1;          // path index 1

    // ---- 1. the lane's point --------------------------------------------------------------------------------------------------
    // Everything needed to address the point's data is in the tile record: primitive records, turn template samples and the field's
    // geofence are fetched side by side, not one after the other.
    double px = 0.0, py = 0.0;
    uint32_t fw = 0;
    // layer 2: primitive and sample index from the record's lane thresholds.  The primitive record is read WHOLE (read field by field
    // inside the branches of eval_prim every primitive kind present in the wave paid a round trip of its own: phase stamps showed 36 %
    // of a wave's life there) and the corner template's sample is asked for beside it; both are requested BEFORE the layer-1 lanes of
    // a seam tile are evaluated, whose own chain of loads (field constants, turn template) then runs beside them.
    const bool in_l2 = act && !in_main;
    int r = 0;
    DevPrim p;
    double2 tc = make_double2(0.0, 0.0);
    stage_tile_prims<false>(wt, prims, plds, nl);
    if (in_l2) {
        int slot = 0;
        r = lane + wt.r0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int th = wt.thr[k];
            if (lane >= th) { ++slot; r = lane - th; }
        }
        tc = cst.tmpl_c[min(max(r, 0), cst.tmpl_nc - 1)];
        p = reinterpret_cast<const DevPrim *>(plds)[slot];
    }
    if (__ballot(act && in_main) != 0ull) {                  // (wave-uniform) layer 1: (pass, offset) from the host's decode of lane 0
        if (act && in_main) {
            const unsigned per = (unsigned)(f.n_line + f.n_turn);
            const unsigned off = (unsigned)wt.off0 + (unsigned)lane, q = off / per;
            eval_main(f, cst, wt.idx0 + (int)q, (int)(off - q * per), px, py, fw);
        }
    }
    if (in_l2) {
        int kind = p.kind;
        asm volatile("" : "+v"(tc.x), "+v"(tc.y), "+v"(kind));     // (needs both values: neither load may be sunk behind the other's wait)
        eval_prim_lanes(p, cst, r, px, py, tc);
        fw = p.fs;
    }
    // nominal speed: layer 1 by kind (swath / U-turn), layer 2 from the primitive record (the host sets it by the same table as
    // nominal_speed()); in m/s by the exact reciprocal division, i.e. the tabulated v / 3.6 bit for bit
    const double vn = in_l2 ? p.v_nom : (((fw & FCPP_KIND_MASK) == FCPP_KIND_SWATH) ? cst.v_work : cst.v_turn);
    const double msn = div36(vn);

    // ---- 2. chords, curvature (MLP:513-536), clamp (MLP:490-504) ----------------------------------------------------------------
    const double xm = lane_prev(px), ym = lane_prev(py), xp = lane_next(px), yp = lane_next(py);
    const bool has_prev = act && lane > 0;                                    // (lane > 0 => not the path's first point)
    const double dx1 = px - xm, dy1 = py - ym;
    const double dprev = has_prev ? seg_len_fast(dx1, dy1) : 0.0;                 // |p_i - p_(i-1)|
    const double dnext = lane_next(dprev);
    const bool interior = has_prev && lane < nl - 1 && !is_last;             // both neighbours are lanes of this wave
    double kappa = 0.0;
    if (interior) kappa = curv_chords_atan(dx1, dy1, dprev, xp - px, yp - py, dnext, atab);
    bool cl = false;
    double v0 = vn;
    if (kappa > 1e-6) v0 = clamped_speed_fast(vn, kappa, cst, cl);
    const double ms0 = cl ? div36(v0) : msn;
    const double u0 = act ? ms0 * ms0 : FCPP_INF;

    // ---- 3. sweeps (MLP:538-589); skipped when no single step binds ----------------------------------------------------------------
    // w = coupling of segment (i-1, i); +inf: nothing propagates (skipped step, the wave's first lane, the path's first point).
    // At this sampling a constraint reaches one to three points, so the sweeps run as a relaxation u_i = min(u_i, u_(i-1) + w_i,
    // u_(i+1) + w_(i+1)) until a ballot reports no change: every round moves all constraints one point on, in both directions,
    // ~10 instructions a round, and the sums accumulate point by point as in the reference's loops.  A tile that has not settled after
    // SWEEP_ROUNDS rounds (dense stretches) finishes with the two min-plus scans, started from where the relaxation got to (same
    // fixed point).
    constexpr int SWEEP_ROUNDS = 5;
    const double two_a = 2 * cst.a_lon;
    const double w = (!has_prev || dprev < 1e-6) ? FCPP_INF : two_a * dprev;      // (has_prev is false on the lanes beyond the tile)
    const double u0m = lane_prev(u0);
    const bool binds = has_prev && w < FCPP_INF && (u0m + w < u0 || u0 + w < u0m);
    double u = u0;
    if (__ballot(binds) != 0ull) {
        double wn = lane_next(w);                      // coupling to the next lane
        if (lane >= nl - 1) wn = FCPP_INF;
        bool settled = false;
#pragma unroll 1
        for (int round = 0; round < SWEEP_ROUNDS; ++round) {
            const double m = min_raw(u, min_raw(lane_prev(u) + w, lane_next(u) + wn));    // (lane 0: 0 + inf; the last lane: 0 + inf)
            const bool moved = m < u;
            u = m;
            if (__ballot(moved) == 0ull) { settled = true; break; }
        }
        if (!settled) {
            Agg fi = { u, w }, bi = { u, wn };
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                Agg pf = { __shfl_up(fi.c, o), __shfl_up(fi.w, o) };
                Agg pb = { __shfl_down(bi.c, o), __shfl_down(bi.w, o) };
                if (lane >= o) fi = combine_after(pf, fi);
                if (lane + o < 64) bi = combine_after(pb, bi);
            }
            u = fmin(fi.c, bi.c);
        }
    }
    // untouched points keep exactly their clamped / nominal value; the square root only in waves where a sweep lowered somebody
    double vfin = cl ? v0 : vn;
    const bool lowered = u < u0;
    if (__ballot(lowered) != 0ull) vfin = lowered ? fsqrt_pos(u) * 3.6 : vfin;

    // ---- 4. validation flags ------------------------------------------------------------------------------------------------------
    bool o_out = false, o_obs = false, o_viol = false;
    // (wave-uniform: tiles whose output points the tiler found safely inside the polygon -- tiler_inside, fcpp_tilefn.h -- skip the test: most headland tiles)
    if (!wt.inside && out) {
        const double ntol = -cst.geofence_tol;
        bool o = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) o = o | (f.ex[e] * px + f.ey[e] * py + f.eo[e] < ntol);
        if (o) { fw |= FCPP_FLAG_OUTSIDE; o_out = true; }
    }
    if (f.obs_count > 0) {      // wave-uniform: bounding box of the wave's output points, then culled + LDS-staged polygon tests
        double mnx = out ? px : FCPP_INF, mxx = out ? px : -FCPP_INF, mny = out ? py : FCPP_INF, mxy = out ? py : -FCPP_INF;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mnx = fmin(mnx, __shfl_xor(mnx, o)); mny = fmin(mny, __shfl_xor(mny, o));
            mxx = fmax(mxx, __shfl_xor(mxx, o)); mxy = fmax(mxy, __shfl_xor(mxy, o));
        }
        const double ox[1] = { px }, oy[1] = { py };
        const unsigned m = obstacle_mask<1>(obs, f.obs_first, f.obs_first + f.obs_count, obs_lds, mnx, mny, mxx, mxy, ox, oy, out ? 1 : 0);
        if (out && (m & 1u)) { fw |= FCPP_FLAG_OBSTACLE; o_obs = true; }
    }

    // ---- 5. metrics (MLP:1290-1311) and a_lat validation (MLP:1383-1408) on the output lanes ---------------------------------------
    const double vprev = lane_prev(vfin), kprev = lane_prev(kappa), vnprev = lane_prev(vn);
    {
        const bool seg = out && !is_first && !at_seam;        // the seam main|headland belongs to neither layer
        const bool l0 = lane < wt.rel_seam;                   // layer 1 (statistics index 0)
        double tpre = 0.0, t = 0.0;
        if (seg) {
            const double ms_pre = (vnprev == vn) ? msn : div36((vnprev + vn) / 2);
            tpre = fdiv(dprev, fmax(ms_pre, 0.1));
        }
        // the time at the planned speeds differs from the one at nominal speeds only where a speed was changed: the second division
        // only in waves that hold such a segment
        const bool changed = seg && !(vprev == vnprev && vfin == vn);
        t = tpre;
        if (__ballot(changed) != 0ull) t = changed ? fdiv(dprev, fmax(div36((vprev + vfin) / 2), 0.1)) : tpre;
        const double len = seg ? dprev : 0.0;
        // (static indices: a per-lane index into the accumulator arrays would put them in scratch memory)
        acc.s_len[0] += l0 ? len : 0.0; acc.s_tpre[0] += l0 ? tpre : 0.0; acc.s_t[0] += l0 ? t : 0.0;
        acc.s_len[1] += l0 ? 0.0 : len; acc.s_tpre[1] += l0 ? 0.0 : tpre; acc.s_t[1] += l0 ? 0.0 : t;
    }
    if (out && !is_first && !is_last) {                     // interior points of the path
        if (kappa > 0.0) {
            // (v / 3.6)^2 kappa with the final speed: an untouched point's v / 3.6 is ms0 (the clamped value / 3.6, or the tabulated nominal one)
            const double ms = lowered ? div36(vfin) : ms0, alat = ms * ms * kappa;
            acc.mk = max_raw(acc.mk, kappa); acc.ma = max_raw(acc.ma, alat);
            if (alat > cst.a_lat) { o_viol = true; fw |= FCPP_FLAG_ALAT; }
        }
        if (kappa != kprev && !is_second) acc.mj = max_raw(acc.mj, fabs(kappa - kprev));          // |kappa_i - kappa_(i-1)| for i >= 2 (MLP:1404-1406)
    }

    // ---- 6. stores: consecutive lanes, consecutive addresses ------------------------------------------------------------------------
    if (out) {
        const int64_t g = wt.out_base + lane;
        xo[g] = px; yo[g] = py; ko[g] = kappa; vo[g] = vfin; fso[g] = fw;
    }

    // the flag counts are one bit per lane: population counts of ballots
    acc.c_viol += __popcll(__ballot(o_viol)); acc.c_out += __popcll(__ballot(o_out)); acc.c_obs += __popcll(__ballot(o_obs));
    acc.c_adj += __popcll(__ballot(out && cl));
}

}  // namespace fcpp
