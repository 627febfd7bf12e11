// fcpp_device.h -- interface between the C-ABI glue (fcpp_api.cpp) and the kernels (fcpp_kernels.hip)
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <stdint.h>

#include "fcpp_geom.h"
#include "fcpp_internal.h"

namespace fcpp {

struct DevPath { int64_t off, n; };   // one path = points [off, off + n) of the SoA arrays

struct DevObstacles {                 // batch obstacle polygons, CSR, device pointers
    const int64_t *offsets;
    const double *x, *y;
    const double *bbox;               // 4 doubles per polygon: min_x, min_y, max_x, max_y (fused kernels: tile-level culling)
};

// by-value kernel argument: scalars + pointer to the two clothoid-arc-clothoid unit shapes
// (shapes[0]: 180 degrees, shapes[1]: 90 degrees; device memory, NULL for the standalone operators)
struct DevConst {
    double a_lat, a_lon, sf, geofence_tol;
    double v_work, v_turn, v_head;
    double u_cap;              // (max nominal speed / 3.6)^2: no sweep constraint can bind above this
    double ms_work, ms_turn, ms_head, ms_rev;   // the four nominal speeds in m/s (v / 3.6, IEEE division on the host)
    double inv_sf36;           // 1 / (safety_factor * 3.6), for the clamp pre-test
    const CacShape *shapes;
    const double2 *tmpl_u, *tmpl_c;   // turn templates of the batch (fused kernel), see TurnTemplates
    int tmpl_n, tmpl_nc;              // samples of the U-turn template / of the corner template
    // quiet U-turns (k_plan_quiet kind 3): per-batch constants of the turn shape; index 0 / 1 = passes ascending / descending in y
    double turn_kappa_last[2];        // curvature at the turn's last sample (its stencil spans the jump to the next swath line)
    double turn_len, turn_time;       // sum of the turn's segment lengths, and that over the nominal turn speed
    double turn_max_kappa[2], turn_max_jump[2];
    double turn_jump[2];              // length of the jump from the turn's last sample to the next line's first point (the closed-form cut: fcpp_cutfn.h)
    const double2 *field_junc;        // per field: (curvature of the first point of a line that follows a U-turn, length of the jump from the turn's end)
    const double2 *tmpl_u_dk;         // per U-turn sample k: (|t_k - t_(k-1)|, curvature at t_k), the shape's own segment lengths / curvatures
};

// Tuning knobs (extra LDS per workgroup = fewer resident waves, wave tiles per workgroup, ...) for tools/ab_knob.py, which flips them
// between runs of one process on identical memory.  They are LIVE only in a process started with FCPP_TUNE=1 (checked once, when the
// library is first used); otherwise every knob is its default and no launch path reads the environment.  Callers clamp the values.
inline bool tune_enabled()
{
    static const bool on = [] { const char *v = getenv("FCPP_TUNE"); return v && v[0] == '1'; }();
    return on;
}
inline int tune_int(const char *name, int dflt)
{
    if (!tune_enabled()) return dflt;
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

// Per-kernel device timing (fcpp_batch_set_profiling): when the caller has armed g_launch_prof, the next launcher call dispatches
// its kernel with hipExtLaunchKernelGGL(start, stop): the two events take the dispatch's own begin / end time stamps -- no marker
// packets between the kernels, so the timed step runs as it does unprofiled -- and the launcher disarms it.
struct LaunchProf { hipEvent_t start = nullptr, stop = nullptr; };
extern thread_local LaunchProf g_launch_prof;

// every launcher returns 0 or a hipError_t value
int launch_generate(hipStream_t st, int64_t n_tiles, const DevTile *tiles, const DevField *fields,
                    const DevPrim *prims, const DevConst &cst, double *x, double *y, double *v, uint32_t *fs);
int launch_curv_clamp(hipStream_t st, int64_t n_tiles, const DevTile *tiles, const DevPath *paths,
                      const DevConst &cst, int do_clamp, const double *x, const double *y, const double *v_in,
                      double *v_out, double *kappa, unsigned long long *n_adjusted);
int launch_scan_tiles(hipStream_t st, int64_t n_tiles, const DevTile *tiles, const DevPath *paths, const DevConst &cst,
                      const double *x, const double *y, const double *v_in, void *agg_f, void *agg_b);
// scratch: spine_scratch_bytes(n_tiles) bytes of device memory (0 / NULL: small batches, one workgroup walks all tiles)
int64_t spine_scratch_bytes(int64_t n_tiles);
int launch_scan_spine(hipStream_t st, int64_t n_tiles, const void *agg_f, const void *agg_b, double *carry_f,
                      double *carry_b, void *scratch = nullptr);
int launch_scan_apply(hipStream_t st, int64_t n_tiles, const DevTile *tiles, const DevPath *paths, const DevConst &cst,
                      int min_n, const double *x, const double *y, const double *v_in, double *v_out,
                      const double *carry_f, const double *carry_b);
int launch_validate(hipStream_t st, int64_t n_tiles, const DevTile *tiles, const DevPath *paths,
                    const DevField *fields, const DevConst &cst, const DevObstacles &obs, const double *x,
                    const double *y, const double *kappa, const double *v, uint32_t *fs, TilePartial *partial);
// fcpp_validate: geofence / obstacle flags of caller-supplied paths against arbitrary simple polygons (device CSR tables; see k_validate_polys)
int launch_validate_polys(hipStream_t st, int64_t n_tiles, const DevTile *tiles, const DevPath *paths, const int64_t *field_off, const double *field_x,
                          const double *field_y, int64_t n_field, const int64_t *obst_off, const double *obst_x, const double *obst_y, int64_t n_obst,
                          const int64_t *obst_range, double tol, double a_lat, const double *x, const double *y, const double *kappa, const double *v,
                          uint32_t *flags, fcpp_field_stats *stats);
// ids / run_count / path_list / group (lanes per path: 8, 64 or 256): see k_reduce_stats; tiles, fields, prims, cst only with run_count
int launch_reduce_stats(hipStream_t st, int64_t n_list, TilePartial *partial, const int64_t *tile_first,
                        const unsigned long long *n_adjusted, fcpp_field_stats *stats, const int32_t *ids = nullptr,
                        const int64_t *run_count = nullptr, const DevTile *tiles = nullptr, const DevField *fields = nullptr,
                        const DevPrim *prims = nullptr, const DevConst *cst = nullptr, const int32_t *path_list = nullptr, int group = 64,
                        void *scratch = nullptr,       // scratch: 64 x 104 bytes per path of the list (group 256 only: sliced reduction)
                        int clear_counts = 0);         // fused pipeline: entries lie side by side (no ids), flag counts of run slots are cleared
// batch creation: closed-form statistics of the quiet runs into their slots, zeros into the others (see k_run_consts)
// the chunk lists of k_plan_quiet from the host's chunk groups (host-built images; see k_expand_chunks)
int launch_expand_chunks(hipStream_t st, int64_t n_segments, const DevChunkGroup *groups, const DevTile *tiles, const DevField *fields,
                         DevTile *chunks, DevTile *span_chunks);
int launch_run_consts(hipStream_t st, int64_t n_entries, const int32_t *ids, const int64_t *run_count, const DevTile *tiles, const DevField *fields,
                      const DevPrim *prims, const DevConst &cst, TilePartial *partial);
int launch_build_templates(hipStream_t st, const TurnTemplates &tt, const CacShape *shapes, void *tu, void *tc);
// ids: tile indices the launch covers (NULL = all tiles in order)
int launch_plan_fused(hipStream_t st, int variant, int64_t n_tiles, const int32_t *ids, const DevTile *tiles,
                      const DevField *fields, const DevPrim *prims, const DevConst &cst, const DevObstacles &obs, double *x,
                      double *y, double *kappa, double *v, uint32_t *fs, TilePartial *partial);
// one wavefront per wave tile; statistics go to partial[wtiles[k].tile]
int launch_plan_sparse(hipStream_t st, int64_t n_wtiles, const DevWaveTile *wtiles, const DevField *fields, const DevPrim *prims,
                       const DevConst &cst, const DevObstacles &obs, double *x, double *y, double *kappa, double *v, uint32_t *fs,
                       TilePartial *partial, int points_per_lane = 1,        // 1: tiles of <= 64 points, 2: of <= 128 (fcpp_sparse2_fn.h)
                       const int32_t *ids = nullptr);                       // ids: the launch covers wtiles[ids[k]] (NULL: all in order)
// one workgroup per field of `work`: its wave tiles planned (two points per lane) and its statistics reduced, written to stats[field]
int launch_plan_sparse_fields(hipStream_t st, int64_t n_work, const DevFieldPack *packs, const DevConst &cst, const DevObstacles &obs, double *x,
                              double *y, double *kappa, double *v, uint32_t *fs, TilePartial *partial, int waves, const TilePartial *totals,
                              fcpp_field_stats *stats, bool spans);   // waves: of FIELD_WORK_WAVES; spans: some packs carry a fused span (DevFieldPack.span_points)
int launch_work_totals(hipStream_t st, int64_t n_work, const DevFieldWork *work, const int64_t *stat_run, const TilePartial *partial, TilePartial *totals);
int launch_distance_matrix(hipStream_t st, int n, const double *x, const double *y, double *D);
int launch_best_connections(hipStream_t st, int64_t n_pairs, const int64_t *fo, const int64_t *to, const double *fx, const double *fy,
                            const double *tx, const double *ty, int32_t *bf, int32_t *bt, double *bd);
int launch_build_template_metrics(hipStream_t st, int n, const void *tmpl, void *dk);
int launch_field_junctions(hipStream_t st, int64_t n_fields, const DevField *fields, const DevConst &cst, void *junc);
int launch_plan_quiet(hipStream_t st, int64_t n_chunks, const DevTile *chunks, int kinds, const DevField *fields, const DevPrim *prims,
                      const DevConst &cst, const DevObstacles &obs, double *x, double *y, double *kappa, double *v, uint32_t *fs,
                      TilePartial *partial);
int launch_straight(hipStream_t st, int64_t n_seg, const double *seg, int n_pts, const int32_t *mask, double *out);
int launch_corner_turns(hipStream_t st, int64_t n, const double *corners, const int32_t *ci, const int32_t *rev, double R, double L,
                        double H, int stride, double *out, int32_t *counts);
int launch_fresnel(hipStream_t st, int64_t n, const double *t, double *c, double *s);
int launch_ga_fitness(hipStream_t st, int n, int64_t pop, const double *D, const int32_t *routes, double *dist,
                      double *fit, int order_mode);

}  // namespace fcpp
