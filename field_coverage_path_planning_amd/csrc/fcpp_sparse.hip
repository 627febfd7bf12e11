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
#include "fcpp_sparse2_fn.h"
#include "fcpp_quiet_fn.h"

namespace fcpp {

// SP_WAVES wave tiles per workgroup (independent of each other: no barrier).  Four for launches of a few rounds of workgroups (the
// headline's 32 768 tiles: 41 us, 43 with two, 45-48 with one), two for long launches (cfg5's 622 016 tiles: 661-682 us, 674-688 with four,
// 737-752 with eight); measured on the same box, tools/ab_knob.py per build.
// PTS: points per lane -- 1: wave tiles of up to 64 points (sparse_tile), 2: of up to 128 (sparse_tile2, fcpp_sparse2_fn.h)
template <int SP_WAVES, int PTS>
__global__ __launch_bounds__(64 * SP_WAVES) void k_plan_sparse(const DevWaveTile *__restrict__ wtiles, const int32_t *__restrict__ ids, int64_t n_wtiles,
                                                               const DevField *__restrict__ fields, const DevPrim *__restrict__ prims,
                                                               DevConst cst, DevObstacles obs, double *__restrict__ xo, double *__restrict__ yo,
                                                               double *__restrict__ ko, double *__restrict__ vo, uint32_t *__restrict__ fso,
                                                               TilePartial *__restrict__ partial, int xcd_map)
{
    __shared__ double obs_lds[SP_WAVES][2 * OBS_LDS_VERTS];
    __shared__ double atab[ATAN_TAB_DOUBLES];
    __shared__ double plds[SP_WAVES][TILE_PRIMS_LDS];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    // XCD-aware order (round 5): workgroups go round robin to the chip's eight XCDs, each with an L2 of its own, and the tiles of ONE field
    // -- five on cfg5, side by side in the list -- read the same field record and primitives: workgroup b takes the slots of position
    // (b mod 8) x (grid / 8) + b / 8, so that every XCD walks one contiguous eighth of the list and a field's records are fetched into one
    // L2 instead of five (the grid is a multiple of eight: launch_plan_sparse)
    const unsigned bid = xcd_map ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int64_t slot = (int64_t)bid * SP_WAVES + wave;
    if (slot >= n_wtiles) return;
    atan_tab_stage(atab);
    const DevWaveTile wt = wtiles[ids ? (int64_t)ids[slot] : slot];      // (ids: the wave tiles of the fields k_plan_sparse_fields does not take)
    SparseAcc acc;
    acc.clear();
    if (PTS == 2) sparse_tile2<false>(wt, fields[wt.field], prims, cst, obs, obs_lds[wave], atab, plds[wave], xo, yo, ko, vo, fso, acc);
    else sparse_tile(wt, fields[wt.field], prims, cst, obs, obs_lds[wave], atab, plds[wave], xo, yo, ko, vo, fso, acc);

    // the tile's partial statistics: the three sums of a layer (and the three maxima) go through the wave together (wave4_to_hi),
    // groups in which every lane holds zero (the other layer, tiles without curvature) are skipped by a ballot
    double g[3] = { 0.0, 0.0, 0.0 };
    // (which layers the output points lie in is in the tile record: scalar compares)
    const int out0 = wt.hb, out1 = wt.hb + wt.count;
    if (out0 < wt.rel_seam) g[0] = wave4_to_hi<0>(acc.s_len[0], acc.s_tpre[0], acc.s_t[0], 0.0);
    if (out1 > wt.rel_seam + 1) g[1] = wave4_to_hi<0>(acc.s_len[1], acc.s_tpre[1], acc.s_t[1], 0.0);     // (the seam point itself belongs to neither)
    if (__ballot(acc.mk != 0.0 || acc.mj != 0.0) != 0ull) g[2] = wave4_to_hi<1>(acc.mk, acc.ma, acc.mj, 0.0);  // (a_lat > 0 needs kappa > 0)
    const int vs = WAVE4_SLOT(lane);
    if (lane >= 60 && vs < 3) {          // TilePartial: 9 doubles (len, time_pre, time per layer; max kappa, a_lat, jump), 4 counters
        double *tp = reinterpret_cast<double *>(&partial[wt.tile]);
        tp[vs] = g[0]; tp[3 + vs] = g[1]; tp[6 + vs] = g[2];
    }
    if (lane == 63) {
        TilePartial &tp = partial[wt.tile];
        tp.n_viol = acc.c_viol; tp.n_outside = acc.c_out; tp.n_in_obstacle = acc.c_obs; tp.n_adjusted = acc.c_adj;
    }
}

// One workgroup per FIELD (DevFieldWork) of W wavefronts, W >= the number of wave tiles of any field of the launch: wavefront w
// plans the field's w-th wave tile (two points per lane; a wavefront without a tile leaves at once), then wavefront 0 reduces the
// field's statistics -- the tiles' partial results through LDS, the closed-form statistics of the field's quiet runs from one record
// summed at batch creation (k_work_totals), the flag counts k_plan_quiet has added to the runs' slots in this step (this launch comes
// after the streaming kernels) from memory.  No k_reduce_stats launch for such fields, no global partial slots for their tiles, no
// cross-workgroup synchronisation: everything the reduction needs is the workgroup's own or final.
// The span phase of k_plan_sparse_fields: wavefront v writes chunk v of the field's span -- at most W chunks (the tiler fuses only such
// spans: FUSED_SPAN_CHUNKS), one per wavefront and no loop: as a loop over a wavefront's chunks the chunk writer's forty pinned field
// constants stayed live around it, 95 vector registers instead of 73.
template <int W, bool OBS>
__device__ __forceinline__ void field_span_chunk(const DevFieldPack &pk, int wave, const DevConst &cst, const DevObstacles &obs, double *obs_lds,
                                                 double *tlds, double *__restrict__ xo, double *__restrict__ yo, double *__restrict__ ko,
                                                 double *__restrict__ vo, uint32_t *__restrict__ fso, unsigned long long *cnt)
{
    static_assert(W == FUSED_SPAN_CHUNKS, "one chunk per wavefront");
    const int64_t S = pk.span_points;
    const DevField &f = pk.field;
    const unsigned per = (unsigned)(f.n_line + f.n_turn);
    const int r0 = (int)(f.pt_off & (TILE_POINTS - 1));
    const int c_first = (int)(S < TILE_POINTS - r0 ? S : TILE_POINTS - r0);
    const int n_chunks = (int)((r0 + S + TILE_POINTS - 1) / TILE_POINTS);
    if (wave >= n_chunks) return;
    wave_sync();
    stage_turn_template(cst, tlds);
    DevTile tl;
    const unsigned start = wave == 0 ? 0u : (unsigned)(c_first + (wave - 1) * TILE_POINTS);
    tl.field = pk.work.field; tl.start = start; tl.quiet = 4; tl.stat_tile = 0;
    tl.count = wave == 0 ? c_first : (int)((S - start < TILE_POINTS) ? S - start : TILE_POINTS);
    // (wave-uniform values the compiler computes on the vector unit -- there is no scalar division -- go back to scalar registers)
    tl.idx0 = __builtin_amdgcn_readfirstlane((int)(start / per)); tl.off0 = __builtin_amdgcn_readfirstlane((int)(start - (unsigned)tl.idx0 * per));
    tl.count = __builtin_amdgcn_readfirstlane(tl.count);
    quiet_tile<16, true, OBS>(tl, &f, nullptr, cst, obs, obs_lds, tlds, xo, yo, ko, vo, fso, cnt, cnt + 1);
}

template <int W, bool OBS, bool SPANS>
__device__ __forceinline__ void plan_sparse_fields_body(const DevFieldPack *__restrict__ packs,
                                                              DevConst cst, double *__restrict__ xo, double *__restrict__ yo,
                                                              double *__restrict__ ko, double *__restrict__ vo, uint32_t *__restrict__ fso,
                                                              DevObstacles obs, TilePartial *__restrict__ partial,
                                                              const TilePartial *__restrict__ totals, fcpp_field_stats *__restrict__ stats)
{
    __shared__ double obs_lds[W][OBS ? 2 * OBS_LDS_VERTS : 1];
    __shared__ TilePartial red[FIELD_WORK_TILES];
    __shared__ double atab[ATAN_TAB_DOUBLES];
    // a wavefront's own LDS: its tile's primitives (stage_tile_prims) while it plans the tile, the turn template (stage_turn_template) while
    // it writes chunks of the field's span
    static_assert(3 * TMPL_LDS >= TILE_PRIMS_LDS, "one area serves both phases");
    __shared__ double wlds[W][3 * TMPL_LDS];
    __shared__ unsigned long long span_cnt[W][2];        // flag counts (outside the geofence, inside an obstacle) of the span chunks a wavefront wrote
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    atan_tab_stage(atab);
    // everything of the field at addresses that follow from the workgroup's index (DevFieldPack)
    const DevFieldPack &pk = packs[blockIdx.x];
    const DevFieldWork w = pk.work;
#ifdef FCPP_DIAG_SPARSE
    if (g_sparse_stop == -2) return;
#endif
    auto plan_tile = [&](const int t) {
        const DevWaveTile wt = pk.tile[t];
        SparseAcc acc;
        acc.clear();
        sparse_tile2<true>(wt, pk.field, pk.prims[t], cst, obs, obs_lds[wave], atab, wlds[wave], xo, yo, ko, vo, fso, acc);
        double g[3] = { 0.0, 0.0, 0.0 };
        const int out0 = wt.hb, out1 = wt.hb + wt.count;
        if (out0 < wt.rel_seam) g[0] = wave4_to_hi<0>(acc.s_len[0], acc.s_tpre[0], acc.s_t[0], 0.0);
        if (out1 > wt.rel_seam + 1) g[1] = wave4_to_hi<0>(acc.s_len[1], acc.s_tpre[1], acc.s_t[1], 0.0);
        if (__ballot(acc.mk != 0.0 || acc.mj != 0.0) != 0ull) g[2] = wave4_to_hi<1>(acc.mk, acc.ma, acc.mj, 0.0);
        const int vs = WAVE4_SLOT(lane);
        if (lane >= 60 && vs < 3) {
            double *tp = reinterpret_cast<double *>(&red[t]);
            tp[vs] = g[0]; tp[3 + vs] = g[1]; tp[6 + vs] = g[2];
        }
        if (lane == 63) { red[t].n_viol = acc.c_viol; red[t].n_outside = acc.c_out; red[t].n_in_obstacle = acc.c_obs; red[t].n_adjusted = acc.c_adj; }
    };
    // (one tile per wavefront, no loop: as a loop over the field's tiles -- fewer wavefronts per field -- the compiler allots this kernel
    // 93-104 vector registers instead of 71, four or five resident wavefronts per SIMD instead of seven: measured 40-44 us instead of
    // 36.5 on the headline)
    // ---- the field's layer-1 span (all complete passes, closed form: fcpp_quiet_fn.h), cut into chunks on 512-point boundaries of the
    // batch arrays as k_plan_quiet's chunks are, wavefront v the chunk v (the tiler fuses spans of at most W chunks): the vector work of
    // the wave tiles and the stores of the span share the compute units instead of one launch after the other, and the field's statistics
    // need nothing from another launch.  (Workgroups taking the two phases in alternating order -- so that the resident ones are
    // spread over both -- measured no better: 61.9 vs 58.8 us.)  SPANS = false: the instance for batches without fused spans (65 instead of
    // 73 vector registers: seven resident wavefronts per SIMD instead of six).
    if (lane < 2) span_cnt[wave][lane] = 0;
    auto span_phase = [&]() {
        // (the pack's address goes through an opaque register here: the span phase's loads of the field's constants -- some forty values
        // it pins in registers -- cannot be hoisted above the tile phase, where they would sit in registers the wave tile needs: 133 vector
        // registers for the kernel instead of the larger of the two phases' 65 and 73)
        const DevFieldPack *pk2 = &pk;
        asm volatile("" : "+s"(pk2) : : "memory");
        if (pk2->span_points > 0) field_span_chunk<W, OBS>(*pk2, wave, cst, obs, obs_lds[wave], wlds[wave], xo, yo, ko, vo, fso, &span_cnt[wave][0]);
    };
    if (wave < w.n_tiles) plan_tile(wave);
    if (SPANS) span_phase();
    // What the reduction needs from memory is asked for BEFORE the barrier and arrives while the other wavefronts finish their work:
    // lane v < 13 the v-th of the thirteen 8-byte components of the field's run totals (k_work_totals: the closed-form statistics of
    // its quiet runs, the same at every step), lane e < n_entries the two flag counts k_plan_quiet has added to entry e's slot in this
    // step (runs other than the span, whose chunks are k_plan_quiet's: this launch comes after the streaming kernels; the slot of an
    // entry that is a wave tile or the fused span is never written: zeros).
    static_assert(sizeof(TilePartial) == 13 * 8 && sizeof(fcpp_field_stats) == 13 * 8, "statistics records are thirteen 8-byte components");
    unsigned long long run_v = 0, c_out = 0, c_obs = 0;
    if (wave == 0) {
        if (lane < 13) run_v = reinterpret_cast<const unsigned long long *>(&totals[blockIdx.x])[lane];
        if (lane < w.n_entries) { const TilePartial &slot = partial[w.e_first + lane]; c_out = (unsigned long long)slot.n_outside; c_obs = (unsigned long long)slot.n_in_obstacle; }
    }
    // (the barrier orders the tiles' results and the span counts in LDS only: no wait for the loads above or for anybody's stores)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (wave != 0) return;
    if ((c_out | c_obs) != 0ull) { partial[w.e_first + lane].n_outside = 0; partial[w.e_first + lane].n_in_obstacle = 0; }   // collected anew in the next step
    if (lane < W) { c_out += span_cnt[lane][0]; c_obs += span_cnt[lane][1]; }
    if (__ballot((c_out | c_obs) != 0ull) != 0ull) {             // (rare) points of the field's runs were flagged in this step
#pragma unroll
        for (int o = FIELD_WORK_ENTRIES / 2; o > 0; o >>= 1) { c_out += __shfl_xor(c_out, o); c_obs += __shfl_xor(c_obs, o); }
    }
    if (lane >= 13) return;
    // component v = the runs' total, then the tiles' results in tile order: sums (0-5), maxima (6-8), counts (9-12)
    const unsigned long long *rv = reinterpret_cast<const unsigned long long *>(red) + lane;
    unsigned long long bits;
    if (lane < 9) {
        double a = __longlong_as_double((long long)run_v);
        for (int t = 0; t < w.n_tiles; ++t) {
            const double x = __longlong_as_double((long long)rv[13 * t]);
            a = lane < 6 ? a + x : max_raw(a, x);
        }
        bits = (unsigned long long)__double_as_longlong(a);
    } else {
        bits = run_v;
        for (int t = 0; t < w.n_tiles; ++t) bits += rv[13 * t];
        if (lane == 10) bits += c_out;
        if (lane == 11) bits += c_obs;
    }
    reinterpret_cast<unsigned long long *>(&stats[w.field])[lane] = bits;
}

// OBS: the batch has obstacle polygons (without them the polygon tests of both phases are compiled out)
template <int W, bool OBS, bool SPANS>
__global__ __launch_bounds__(64 * W) void k_plan_sparse_fields(const DevFieldPack *__restrict__ packs, DevConst cst, double *__restrict__ xo, double *__restrict__ yo,
                                                              double *__restrict__ ko, double *__restrict__ vo, uint32_t *__restrict__ fso,
                                                              DevObstacles obs, TilePartial *__restrict__ partial,
                                                              const TilePartial *__restrict__ totals, fcpp_field_stats *__restrict__ stats)
{
    plan_sparse_fields_body<W, OBS, SPANS>(packs, cst, xo, yo, ko, vo, fso, obs, partial, totals, stats);
}
// the same held to eight resident wavefronts per SIMD (the excess registers spilled): A/B under FCPP_TUNE=1 FCPP_FW_OCC8=1
template <int W, bool OBS, bool SPANS>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_plan_sparse_fields_occ8(const DevFieldPack *__restrict__ packs, DevConst cst, double *__restrict__ xo, double *__restrict__ yo,
                                                              double *__restrict__ ko, double *__restrict__ vo, uint32_t *__restrict__ fso,
                                                              DevObstacles obs, TilePartial *__restrict__ partial,
                                                              const TilePartial *__restrict__ totals, fcpp_field_stats *__restrict__ stats)
{
    plan_sparse_fields_body<W, OBS, SPANS>(packs, cst, xo, yo, ko, vo, fso, obs, partial, totals, stats);
}

// Batch creation: per field of field_work, the closed-form statistics of its quiet runs (their slots, k_run_consts) summed in the order of
// the field's entries -- the same at every step, so k_plan_sparse_fields adds its tiles' results to ONE record instead of walking the slots
__global__ void k_work_totals(int64_t n_work, const DevFieldWork *__restrict__ work, const int64_t *__restrict__ stat_run,
                              const TilePartial *__restrict__ partial, TilePartial *__restrict__ totals)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_work) return;
    const DevFieldWork w = work[i];
    TilePartial t;
    memset(&t, 0, sizeof t);
    for (int e = 0; e < w.n_entries; ++e) {
        if (stat_run[w.e_first + e] == 0) continue;          // a wave tile
        const TilePartial p = partial[w.e_first + e];
        t.main_len += p.main_len; t.main_time_pre += p.main_time_pre; t.main_time += p.main_time;
        t.head_len += p.head_len; t.head_time_pre += p.head_time_pre; t.head_time += p.head_time;
        t.max_kappa = fmax(t.max_kappa, p.max_kappa); t.max_alat = fmax(t.max_alat, p.max_alat); t.max_jump = fmax(t.max_jump, p.max_jump);
        t.n_viol += p.n_viol; t.n_adjusted += p.n_adjusted;
    }
    totals[i] = t;
}

int launch_work_totals(hipStream_t st, int64_t n_work, const DevFieldWork *work, const int64_t *stat_run, const TilePartial *partial, TilePartial *totals)
{
    if (n_work <= 0) return 0;
    hipLaunchKernelGGL(k_work_totals, dim3((unsigned)((n_work + 255) / 256)), dim3(256), 0, st, n_work, work, stat_run, partial, totals);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int launch_plan_sparse_fields(hipStream_t st, int64_t n_work, const DevFieldPack *packs, const DevConst &cst, const DevObstacles &obs, double *x,
                              double *y, double *kappa, double *v, uint32_t *fs, TilePartial *partial, int waves, const TilePartial *totals,
                              fcpp_field_stats *stats, bool spans)
{
    if (n_work <= 0) return 0;
#define FCPP_FW(K, W, OB, SP) FCPP_LAUNCH((K<W, OB, SP>), dim3((unsigned)n_work), dim3(64 * W), 0, st, packs, cst, x, y, kappa, v, fs, obs, partial, totals, stats)
    if (waves != 4) return (int)hipErrorInvalidValue;       // (instances for 5, 6, 8 wavefronts: measured slower than the open list, fcpp_internal.h)
    const bool has_obs = obs.offsets != nullptr;
    if (tune_int("FCPP_FW_OCC8", 0) && !has_obs && !spans) FCPP_FW(k_plan_sparse_fields_occ8, 4, false, false);
    else if (has_obs && spans) FCPP_FW(k_plan_sparse_fields, 4, true, true);
    else if (has_obs) FCPP_FW(k_plan_sparse_fields, 4, true, false);
    else if (spans) FCPP_FW(k_plan_sparse_fields, 4, false, true);
    else FCPP_FW(k_plan_sparse_fields, 4, false, false);
#undef FCPP_FW
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int launch_plan_sparse(hipStream_t st, int64_t n_wtiles, const DevWaveTile *wtiles, const DevField *fields, const DevPrim *prims,
                       const DevConst &cst, const DevObstacles &obs, double *x, double *y, double *kappa, double *v, uint32_t *fs,
                       TilePartial *partial, int points_per_lane, const int32_t *ids)
{
    if (n_wtiles <= 0) return 0;
    const bool two = points_per_lane == 2;
    const int wpb_k = tune_int("FCPP_SPARSE_WPB", 0);         // (FCPP_TUNE=1 only: 2 or 4 wave tiles per workgroup, tools/ab_knob.py)
    static const int xcd_map = getenv("FCPP_XCD_MAP") ? atoi(getenv("FCPP_XCD_MAP")) : 1;       // (0: workgroup b takes slot b -- the A/B)
    auto grid8 = [&](int64_t per) { const int64_t g = (n_wtiles + per - 1) / per; return dim3((unsigned)(xcd_map ? (g + 7) / 8 * 8 : g)); };
    if (wpb_k == 2 || (wpb_k != 4 && n_wtiles >= (two ? 65536 : 131072))) {
        if (two) FCPP_LAUNCH((k_plan_sparse<2, 2>), grid8(2), dim3(128), 0, st, wtiles, ids, n_wtiles, fields, prims, cst, obs, x, y, kappa, v, fs, partial, xcd_map);
        else FCPP_LAUNCH((k_plan_sparse<2, 1>), grid8(2), dim3(128), 0, st, wtiles, ids, n_wtiles, fields, prims, cst, obs, x, y, kappa, v, fs, partial, xcd_map);
    } else {
        if (two) FCPP_LAUNCH((k_plan_sparse<4, 2>), grid8(4), dim3(256), 0, st, wtiles, ids, n_wtiles, fields, prims, cst, obs, x, y, kappa, v, fs, partial, xcd_map);
        else FCPP_LAUNCH((k_plan_sparse<4, 1>), grid8(4), dim3(256), 0, st, wtiles, ids, n_wtiles, fields, prims, cst, obs, x, y, kappa, v, fs, partial, xcd_map);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

}  // namespace fcpp

#ifdef FCPP_DIAG_SPARSE
// diagnostic build only: rows[0 .. n) x 16 = cycles per section of sparse_tile2 of the wave tiles planned since the last call (at most
// `cap` rows are copied); returns the number of rows claimed, clears the counter
extern "C" long long fcpp_diag_sparse(unsigned *rows, long long cap)
{
    unsigned n = 0, zero = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(fcpp::g_sparse_diag_next), sizeof n) != hipSuccess) return -1;
    const long long m = n < (unsigned)fcpp::SP_DIAG_SLOTS ? n : fcpp::SP_DIAG_SLOTS;
    const long long c = m < cap ? m : cap;
    if (c > 0 && hipMemcpyFromSymbol(rows, HIP_SYMBOL(fcpp::g_sparse_diag), (size_t)c * 16 * sizeof(unsigned)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(fcpp::g_sparse_diag_next), &zero, sizeof zero) != hipSuccess) return -1;
    return n;
}
extern "C" int fcpp_diag_sparse_stop(int section)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(fcpp::g_sparse_stop), &section, sizeof section) == hipSuccess ? 0 : -1;
}
#endif
