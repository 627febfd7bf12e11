// fcpp_sparse2_fn.h -- the wave tile of fcpp_sparse.hip with TWO consecutive points per lane: a wavefront owns up to 128 consecutive
// points of one field, lane l the points 2l ("a") and 2l + 1 ("b") of the tile.  Same arithmetic per point as sparse_tile
// (fcpp_sparse_fn.h), so a point's results do not depend on which of the two plans it; what changes is the cost per point: the halo
// points (about ten per tile, whatever its size), the lane moves (a's successor and b's predecessor are the lane's own registers) and
// the wave-wide reductions of the statistics are paid once per 128 points instead of once per 64.
#pragma once
#include "fcpp_sparse_fn.h"

namespace fcpp {

// Diagnostic build only (-DFCPP_DIAG_SPARSE, never shipped; tools/diag_sparse.py): shader-clock cycles of every section of sparse_tile2,
// summed over the wave tiles of all launches since the last read (fcpp_diag_sparse, fcpp_sparse.hip).
#ifdef FCPP_DIAG_SPARSE
static constexpr int SP_DIAG_SLOTS = 1 << 16;
__device__ unsigned g_sparse_diag[SP_DIAG_SLOTS * 16];     // one row per wave tile (claimed through g_sparse_diag_next): cycles per section
__device__ unsigned g_sparse_diag_next;
__device__ int g_sparse_stop = 99;                         // leave sparse_tile2 after this section (instruction counts per section: rocprofv3 --pmc)
template <class T> __device__ __forceinline__ void sp_pin(T &v) { asm volatile("" : "+v"(v)); }
#define SP_STAMP_S(k, sval) do { int s_ = __builtin_amdgcn_readfirstlane((int)(sval)); asm volatile("" : "+s"(s_)); const unsigned long long t_now_ = __builtin_readcyclecounter(); \
                                t_sec_[k] = (unsigned)(t_now_ - t_prev_) + (unsigned)(s_ & 0); t_prev_ = t_now_; } while (0)
template <class T, class... R> __device__ __forceinline__ void sp_pin(T &v, R &...r) { sp_pin(v); sp_pin(r...); }
// (the values a section produces go through an empty volatile asm, so the compiler can move their computation neither behind the
// stamp nor, for what depends on them, before it)
#define SP_STAMP(k, ...) do { sp_pin(__VA_ARGS__); const unsigned long long t_now_ = __builtin_readcyclecounter(); \
                              t_sec_[k] = (unsigned)(t_now_ - t_prev_); t_prev_ = t_now_; \
                              if (stop_ == k) return; } while (0)
#define SP_STAMP_BEGIN() unsigned t_sec_[15] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }; int stop_ = __builtin_amdgcn_readfirstlane(g_sparse_stop); asm volatile("" : "+s"(stop_)); \
                         unsigned long long t_prev_ = __builtin_readcyclecounter(); t_sec_[13] = (unsigned)(t_prev_ >> 4)
#define SP_STAMP_END() do { t_sec_[14] = (unsigned)(t_prev_ >> 4); unsigned row_ = 0; if ((threadIdx.x & 63) == 0) row_ = atomicAdd(&g_sparse_diag_next, 1u); \
                            row_ = __builtin_amdgcn_readfirstlane(row_); \
                            if ((threadIdx.x & 63) < 15 && row_ < SP_DIAG_SLOTS) { unsigned v_ = 0; \
                                for (int k_ = 0; k_ < 15; ++k_) if ((int)(threadIdx.x & 63) == k_) v_ = t_sec_[k_]; \
                                g_sparse_diag[row_ * 16 + (threadIdx.x & 63)] = v_; } } while (0)
#else
#define SP_STAMP(k, ...) do { } while (0)
#define SP_STAMP_BEGIN() do { } while (0)
#define SP_STAMP_S(k, sval) do { } while (0)
#define SP_STAMP_END() do { } while (0)
#endif

// one of the lane's two points
struct SparsePt {
    double px, py, vn, d, kappa, v0, ms0, u0, u, w, vfin;
    uint32_t fw;
    bool act, out, cl, lowered, has_prev, interior;    // (where the point lies on the path -- first, second, last, at the seam -- is compared where it is asked: a predicate kept from the start to the metrics holds a scalar register pair all that way, and the kernel spills those)
};

// the point `rel` (index in the tile) of a lane: coordinates, flag word, nominal speed
__device__ __forceinline__ void sparse2_point(const DevWaveTile &wt, const DevField &f, const double *plds, const DevConst &cst, int rel,
                                              int slot, int nl, SparsePt &q)
{
    q.act = rel < nl;
    q.out = rel >= wt.hb && rel < wt.hb + wt.count;
    const bool in_main = rel < wt.rel_main;
    q.px = q.py = 0.0;
    q.fw = 0;
    const bool in_l2 = q.act && !in_main;
    int r = 0;
    DevPrim p;
    double2 tc = make_double2(0.0, 0.0);
    if (in_l2) {
        r = rel - reinterpret_cast<const int *>(plds + TILE_STARTS_AT)[slot];
        tc = cst.tmpl_c[min(max(r, 0), cst.tmpl_nc - 1)];
        p = reinterpret_cast<const DevPrim *>(plds)[slot];
    }
    if (__ballot(q.act && in_main) != 0ull) {
        if (q.act && in_main) {
            const unsigned per = (unsigned)(f.n_line + f.n_turn);
            const unsigned off = (unsigned)wt.off0 + (unsigned)rel, qq = off / per;
            eval_main(f, cst, wt.idx0 + (int)qq, (int)(off - qq * per), q.px, q.py, q.fw);
        }
    }
    if (in_l2) {
        int kind = p.kind;
        asm volatile("" : "+v"(tc.x), "+v"(tc.y), "+v"(kind));
        eval_prim_lanes(p, cst, r, q.px, q.py, tc);
        q.fw = p.fs;
    }
    q.vn = in_l2 ? p.v_nom : (((q.fw & FCPP_KIND_MASK) == FCPP_KIND_SWATH) ? cst.v_work : cst.v_turn);
}

// obs_lds: 2 * OBS_LDS_VERTS doubles of LDS owned by this wavefront (only touched when the field has obstacles); atab: the staged
// table of atan2_abs_dev (atan_tab_stage); plds: TILE_PRIMS_LDS doubles of LDS owned by this wavefront (stage_tile_prims)
template <bool PACKED = false>
__device__ __forceinline__ void sparse_tile2(const DevWaveTile &wt, const DevField &f, const DevPrim *__restrict__ prims, const DevConst &cst,
                                             const DevObstacles &obs, double *obs_lds, const double *atab, double *plds, double *__restrict__ xo, double *__restrict__ yo,
                                             double *__restrict__ ko, double *__restrict__ vo, uint32_t *__restrict__ fso, SparseAcc &acc)
{
    const int lane = threadIdx.x & 63;
    const int nl = wt.hb + wt.count + wt.hf;                 // active points
    const int ra = 2 * lane, rb = 2 * lane + 1;
    SparsePt A, B;
    SP_STAMP_BEGIN();
#ifdef FCPP_DIAG_SPARSE
    if (stop_ == -1 || stop_ == -5) return;
#endif
    SP_STAMP_S(9, wt.hb);
    SP_STAMP_S(10, f.n_line);
    stage_tile_prims<PACKED>(wt, prims, plds, nl);
    int slot_a, slot_b;
    tile_slots2(wt, lane, slot_a, slot_b);
    sparse2_point(wt, f, plds, cst, ra, slot_a, nl, A);
#ifdef FCPP_DIAG_SPARSE
    sp_pin(A.px, A.py, A.vn, A.fw);
    if (stop_ == -3) return;
#endif
    sparse2_point(wt, f, plds, cst, rb, slot_b, nl, B);
    SP_STAMP(0, A.px, A.py, B.px, B.py, A.vn, B.vn, A.fw, B.fw);

    // ---- chords, curvature (MLP:513-536), clamp (MLP:490-504) ----------------------------------------------------------------------
    // a's predecessor is the previous lane's b, its successor the lane's own b; b's predecessor is a, its successor the next lane's a
    const double xm = lane_prev(B.px), ym = lane_prev(B.py), xn = lane_next(A.px), yn = lane_next(A.py);
    A.has_prev = A.act && lane > 0;
    B.has_prev = B.act;
    const double dxa = A.px - xm, dya = A.py - ym, dxb = B.px - A.px, dyb = B.py - A.py;
    A.d = A.has_prev ? seg_len_fast(dxa, dya) : 0.0;
    B.d = B.has_prev ? seg_len_fast(dxb, dyb) : 0.0;
    const double dn_b = lane_next(A.d);                      // |next lane's a - b|
    A.interior = A.has_prev && ra < nl - 1 && ra != wt.rel_last;
    B.interior = B.has_prev && rb < nl - 1 && rb != wt.rel_last;
    A.kappa = B.kappa = 0.0;
    SP_STAMP(1, A.d, B.d);
    if (A.interior) A.kappa = curv_chords_atan(dxa, dya, A.d, dxb, dyb, B.d, atab);
    if (B.interior) B.kappa = curv_chords_atan(dxb, dyb, B.d, xn - B.px, yn - B.py, dn_b, atab);
    SP_STAMP(2, A.kappa, B.kappa);
    A.cl = B.cl = false;
    A.v0 = A.vn; B.v0 = B.vn;
    if (A.kappa > 1e-6) A.v0 = clamped_speed_fast(A.vn, A.kappa, cst, A.cl);
    if (B.kappa > 1e-6) B.v0 = clamped_speed_fast(B.vn, B.kappa, cst, B.cl);
    A.ms0 = div36(A.cl ? A.v0 : A.vn);
    B.ms0 = div36(B.cl ? B.v0 : B.vn);
    A.u0 = A.act ? A.ms0 * A.ms0 : FCPP_INF;
    B.u0 = B.act ? B.ms0 * B.ms0 : FCPP_INF;

    // ---- sweeps (MLP:538-589) as a relaxation over the tile's 128 points, two per lane ----------------------------------------------
    SP_STAMP(3, A.u0, B.u0, A.v0, B.v0);
    constexpr int SWEEP_ROUNDS = 5;
    const double two_a = 2 * cst.a_lon;
    A.w = (!A.has_prev || A.d < 1e-6) ? FCPP_INF : two_a * A.d;        // coupling (previous lane's b, a)
    B.w = (!B.has_prev || B.d < 1e-6) ? FCPP_INF : two_a * B.d;        // coupling (a, b)
    const double ubm0 = lane_prev(B.u0);
    const bool binds = (A.has_prev && A.w < FCPP_INF && (ubm0 + A.w < A.u0 || A.u0 + A.w < ubm0)) ||
                       (B.has_prev && B.w < FCPP_INF && (A.u0 + B.w < B.u0 || B.u0 + B.w < A.u0));
    A.u = A.u0; B.u = B.u0;
    if (__ballot(binds) != 0ull) {
        double wn = lane_next(A.w);                  // coupling (b, next lane's a)
        if (rb >= nl - 1) wn = FCPP_INF;
        bool settled = false;
#pragma unroll 1
        for (int round = 0; round < SWEEP_ROUNDS; ++round) {
            const double ubm = lane_prev(B.u), uan = lane_next(A.u);
            const double ma = min_raw(A.u, min_raw(ubm + A.w, B.u + B.w));
            const double mb = min_raw(B.u, min_raw(A.u + B.w, uan + wn));
            const bool moved = ma < A.u || mb < B.u;
            A.u = ma; B.u = mb;
            if (__ballot(moved) == 0ull) { settled = true; break; }
        }
        if (!settled) {
            // the two min-plus scans over the lanes' composite maps (a then b forwards, b then a backwards), from where the relaxation got to
            const Agg fa = { A.u, A.w }, fb = { B.u, B.w }, ba = { A.u, B.w }, bb = { B.u, wn };
            Agg fi = combine_after(fa, fb), bi = combine_after(bb, ba);
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                Agg pf = { __shfl_up(fi.c, o), __shfl_up(fi.w, o) };
                Agg pb = { __shfl_down(bi.c, o), __shfl_down(bi.w, o) };
                if (lane >= o) fi = combine_after(pf, fi);
                if (lane + o < 64) bi = combine_after(pb, bi);
            }
            double fprev = __shfl_up(fi.c, 1), bnext = __shfl_down(bi.c, 1);       // the results at the neighbours' facing points
            if (lane == 0) fprev = FCPP_INF;
            if (lane == 63) bnext = FCPP_INF;
            const double fwd_a = fmin(A.u, fprev + A.w), fwd_b = fi.c;
            const double bwd_b = fmin(B.u, bnext + wn), bwd_a = bi.c;
            A.u = fmin(fwd_a, bwd_a); B.u = fmin(fwd_b, bwd_b);
        }
    }
    SP_STAMP(4, A.u, B.u);
    A.vfin = A.cl ? A.v0 : A.vn;
    B.vfin = B.cl ? B.v0 : B.vn;
    A.lowered = A.u < A.u0; B.lowered = B.u < B.u0;
    if (__ballot(A.lowered || B.lowered) != 0ull) {
        A.vfin = A.lowered ? fsqrt_pos(A.u) * 3.6 : A.vfin;
        B.vfin = B.lowered ? fsqrt_pos(B.u) * 3.6 : B.vfin;
    }

    // ---- validation flags --------------------------------------------------------------------------------------------------------------
    SP_STAMP(5, A.vfin, B.vfin);
    bool a_out = false, b_out = false, a_obs = false, b_obs = false, a_viol = false, b_viol = false;
    if (!wt.inside) {
        const double ntol = -cst.geofence_tol;
        if (A.out) {
            bool o = false;
#pragma unroll
            for (int e = 0; e < 4; ++e) o = o | (f.ex[e] * A.px + f.ey[e] * A.py + f.eo[e] < ntol);
            if (o) { A.fw |= FCPP_FLAG_OUTSIDE; a_out = true; }
        }
        if (B.out) {
            bool o = false;
#pragma unroll
            for (int e = 0; e < 4; ++e) o = o | (f.ex[e] * B.px + f.ey[e] * B.py + f.eo[e] < ntol);
            if (o) { B.fw |= FCPP_FLAG_OUTSIDE; b_out = true; }
        }
    }
    if (f.obs_count > 0) {
        double mnx = FCPP_INF, mxx = -FCPP_INF, mny = FCPP_INF, mxy = -FCPP_INF;
        if (A.out) { mnx = A.px; mxx = A.px; mny = A.py; mxy = A.py; }
        if (B.out) { mnx = fmin(mnx, B.px); mxx = fmax(mxx, B.px); mny = fmin(mny, B.py); mxy = fmax(mxy, B.py); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mnx = fmin(mnx, __shfl_xor(mnx, o)); mny = fmin(mny, __shfl_xor(mny, o));
            mxx = fmax(mxx, __shfl_xor(mxx, o)); mxy = fmax(mxy, __shfl_xor(mxy, o));
        }
        // (obstacle_mask tests the points [0, nvalid) of a lane: a lane whose a is a halo point and whose b is an output tests both)
        const double ox[2] = { A.px, B.px }, oy[2] = { A.py, B.py };
        const unsigned m = obstacle_mask<2>(obs, f.obs_first, f.obs_first + f.obs_count, obs_lds, mnx, mny, mxx, mxy, ox, oy, B.out ? 2 : (A.out ? 1 : 0));
        if (A.out && (m & 1u)) { A.fw |= FCPP_FLAG_OBSTACLE; a_obs = true; }
        if (B.out && (m & 2u)) { B.fw |= FCPP_FLAG_OBSTACLE; b_obs = true; }
    }

    // ---- metrics (MLP:1290-1311) and a_lat validation (MLP:1383-1408) on the output points ------------------------------------------------
    SP_STAMP(6, A.fw, B.fw);
    const double vprev_a = lane_prev(B.vfin), kprev_a = lane_prev(B.kappa), vnprev_a = lane_prev(B.vn);
    // (all but the one tile of a field that holds the seam between the layers have their output points in ONE layer: three sums per
    // lane -- slot 0, handed to the tile's layer below -- instead of six with two selects each)
    const bool mixed = wt.rel_seam > wt.hb && wt.rel_seam < wt.hb + wt.count;
    auto metrics = [&](const SparsePt &q, int rel, double vprev, double vnprev, double kprev, bool &viol, uint32_t &fw) {
        const bool is_first = rel == wt.rel_zero;
        const bool seg = q.out && !is_first && rel != wt.rel_seam;
        const bool l0 = rel < wt.rel_seam;
        double tpre = 0.0, t = 0.0;
        if (seg) {
            const double ms_pre = div36((vnprev + q.vn) / 2);       // ((a + a) / 2 == a exactly: no case for equal speeds)
            tpre = fdiv(q.d, fmax(ms_pre, 0.1));
        }
        const bool changed = seg && !(vprev == vnprev && q.vfin == q.vn);
        t = tpre;
        if (__ballot(changed) != 0ull) t = changed ? fdiv(q.d, fmax(div36((vprev + q.vfin) / 2), 0.1)) : tpre;
        const double len = seg ? q.d : 0.0;
        if (mixed) {
            acc.s_len[0] += l0 ? len : 0.0; acc.s_tpre[0] += l0 ? tpre : 0.0; acc.s_t[0] += l0 ? t : 0.0;
            acc.s_len[1] += l0 ? 0.0 : len; acc.s_tpre[1] += l0 ? 0.0 : tpre; acc.s_t[1] += l0 ? 0.0 : t;
        } else { acc.s_len[0] += len; acc.s_tpre[0] += tpre; acc.s_t[0] += t; }
        if (q.out && !is_first && rel != wt.rel_last) {
            if (q.kappa > 0.0) {
                const double ms = div36(q.vfin), alat = ms * ms * q.kappa;       // (not lowered: vfin is the value ms0 was taken from)
                acc.mk = max_raw(acc.mk, q.kappa); acc.ma = max_raw(acc.ma, alat);
                if (alat > cst.a_lat) { viol = true; fw |= FCPP_FLAG_ALAT; }
            }
            if (q.kappa != kprev && rel != wt.rel_zero + 1) acc.mj = max_raw(acc.mj, fabs(q.kappa - kprev));
        }
    };
    // (a first, then b: the per-lane accumulators add a's terms before b's; the order of the additions is fixed by the tile alone)
    metrics(A, ra, vprev_a, vnprev_a, kprev_a, a_viol, A.fw);
    metrics(B, rb, A.vfin, A.vn, A.kappa, b_viol, B.fw);
    if (!mixed && wt.rel_seam <= wt.hb) {        // (the tile lies in layer 2: the sums collected in slot 0 are its)
        acc.s_len[1] = acc.s_len[0]; acc.s_tpre[1] = acc.s_tpre[0]; acc.s_t[1] = acc.s_t[0];
        acc.s_len[0] = 0.0; acc.s_tpre[0] = 0.0; acc.s_t[0] = 0.0;
    }

    // ---- stores: the lane's two points are neighbours in memory ----------------------------------------------------------------------------
    SP_STAMP(7, acc.s_len[0], acc.s_len[1], acc.s_t[0], acc.s_t[1], acc.s_tpre[0], acc.s_tpre[1], acc.mk, acc.ma, acc.mj, A.fw, B.fw);
    const int64_t g = wt.out_base + ra;
#ifdef FCPP_DIAG_SPARSE
    { unsigned long long px_ = (unsigned long long)xo, pf_ = (unsigned long long)fso; asm volatile("" : "+s"(px_), "+s"(pf_)); SP_STAMP_S(11, (int)(px_ ^ pf_)); }
#endif
    // (a lane whose two points are both outputs -- all but the one or two at the ends of the output range -- writes them as one 16-byte
    // element per array: a store instruction then covers 1 KiB of consecutive bytes, every 64-byte line of it whole, instead of every
    // other 8 bytes of it; the element is 8-byte aligned, which global memory instructions accept)
    typedef double pair_f64 __attribute__((ext_vector_type(2), aligned(8)));
    typedef uint32_t pair_u32 __attribute__((ext_vector_type(2), aligned(4)));
    if (A.out && B.out) {
        pair_f64 vx = { A.px, B.px }, vy = { A.py, B.py }, vk = { A.kappa, B.kappa }, vv = { A.vfin, B.vfin };
        pair_u32 vf = { A.fw, B.fw };
#ifndef FCPP_SPARSE_NT
#define FCPP_SPARSE_NT 1     // (0: a build for the A/B)
#endif
        if (FCPP_SPARSE_NT) {
            __builtin_nontemporal_store(vx, reinterpret_cast<pair_f64 *>(xo + g)); __builtin_nontemporal_store(vy, reinterpret_cast<pair_f64 *>(yo + g));
            __builtin_nontemporal_store(vk, reinterpret_cast<pair_f64 *>(ko + g)); __builtin_nontemporal_store(vv, reinterpret_cast<pair_f64 *>(vo + g));
            __builtin_nontemporal_store(vf, reinterpret_cast<pair_u32 *>(fso + g));
        } else {
            *reinterpret_cast<pair_f64 *>(xo + g) = vx; *reinterpret_cast<pair_f64 *>(yo + g) = vy;
            *reinterpret_cast<pair_f64 *>(ko + g) = vk; *reinterpret_cast<pair_f64 *>(vo + g) = vv;
            *reinterpret_cast<pair_u32 *>(fso + g) = vf;
        }
    } else {
        if (A.out) { xo[g] = A.px; yo[g] = A.py; ko[g] = A.kappa; vo[g] = A.vfin; fso[g] = A.fw; }
        if (B.out) { xo[g + 1] = B.px; yo[g + 1] = B.py; ko[g + 1] = B.kappa; vo[g + 1] = B.vfin; fso[g + 1] = B.fw; }
    }

    SP_STAMP_S(12, 0);
    acc.c_viol += __popcll(__ballot(a_viol)) + __popcll(__ballot(b_viol));
    acc.c_out += __popcll(__ballot(a_out)) + __popcll(__ballot(b_out));
    acc.c_obs += __popcll(__ballot(a_obs)) + __popcll(__ballot(b_obs));
    acc.c_adj += __popcll(__ballot(A.out && A.cl)) + __popcll(__ballot(B.out && B.cl));
    SP_STAMP(8, acc.c_viol, acc.c_out, acc.c_obs, acc.c_adj);
    SP_STAMP_END();
}

}  // namespace fcpp
