// fcpp_devplan.h -- the setup of a batch ON THE DEVICE (fcpp_devplan.hip): what fcpp_host.cpp (one field's plan) and fcpp_tiler.cpp (its
// path cut into kernel work) do on the host's cores, done by the GPU -- the plan call of the reference, plan_complete_coverage
// (MLP:387-465), plans a NEW field every time, so the setup belongs on the clock and off the host.  Batches with obstacle_mode = FLAG: at the
// reference's own sampling (rounds 4-5) and at dense sampling when no field has obstacles (round 5).
//
//   k_plan_fields16    sixteen lanes per field, four fields per wavefront: fcpp_planfn.h's plan function restated lane-parallel, the same
//                      float64 operations in the same order -> fcpp_field_info, DevField, its primitives (k_plan_fields: the one-thread
//                      original, FCPP_PLAN_SERIAL=1 and fcpp_plan_points)
//   k_tile_fields<0>   one wavefront per field, the tiler's cut, counting its records.  Sparse sampling: span of whole passes + the general
//                      stretch cut in closed form (fcpp_cutfn.h) or by the window cut; dense sampling (the DENSE instance): span + a lane per
//                      quiet zone (last line, headland straights) + a lane per stretch between them (general tiles, or wave tiles by the
//                      window cut, a stretch after the other)
//   k_scan_*           exclusive scans over the fields of the per-field counts; the totals go to the host's pinned memory (polled)
//   k_tile_fields<1>   the records written at their scanned positions: the tables of ImageLayout, equal byte for byte to what BatchTiler::fill
//                      writes on the host (tests/test_gpu_devplan.py)
// Only the 128-byte fcpp_field records go to the device and the totals (one small copy) and fcpp_field_info come back.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "fcpp_device.h"
#include "fcpp_internal.h"
#include "fcpp_planfn.h"
#include "fcpp_tilefn.h"
#include "fcpp_cutfn.h"

namespace fcpp {

// columns of the per-field count table (counts[col][field]) and of its exclusive scan (bases[col][field]); totals[col] = sum
enum PlanCol : int {
    PC_POINTS = 0, PC_PRIMS, PC_TILES, PC_WAVE, PC_GENERAL, PC_STAT, PC_SPAN, PC_WORK, PC_OPEN, PC_CLS0, PC_CLS1, PC_CLS2, PC_CLS3,
    PC_RUNS, PC_SPAN_PTS, PC_WAVE_PTS, PC_WORK_WAVE_PTS, PC_WAVE_INSIDE, PC_WORK_SPAN_PTS, PC_SPAN_F, PC_UNFUSABLE,
    PC_CHUNKS, PC_CHUNK_PTS,      // dense sampling: chunk records / points of the quiet runs of straights (the last swath line, headland straights)
    PC_COLS
};
// (PC_SPAN: span chunks when no span is fused; PC_SPAN_F: when every fusable span is; PC_UNFUSABLE: fields of field work whose span has too
// many chunks to be fused -- the host fuses all or none, k_tile_fields<true> is told which: DevTileConsts.fuse_spans)
// totals[PC_COLS + k]: flags the kernels raise -- a flag holds the generation number (DevTileConsts.gen, counted up by the context) of the
// last counting phase that raised it, so nothing has to be cleared between batches: raised in this phase <=> flag == gen
enum PlanFlag : int { PF_FALLBACK = 0, PF_BAD_OBSTACLES = 1, PF_OVER_CAPACITY = 2, PF_COUNT = 3 };
// behind the flags: [PX_ARRIVED] the columns of the counting phase's last scan that have published their totals (device only), [PX_DONE] the
// generation number of the phase whose totals are complete -- written to the host's copy by the last column to arrive: the host polls it
// (a word in its own pinned memory) instead of waiting for an event
constexpr int PX_ARRIVED = PC_COLS + PF_COUNT, PX_DONE = PC_COLS + PF_COUNT + 1, PLAN_TOTALS = PC_COLS + PF_COUNT + 2;
// (PF_OVER_CAPACITY: a speculative setup -- tables laid out by per-field capacities, the fill pass enqueued before the host has the totals
// -- met a field beyond them: SPEC_* below; the host then lays the tables out from the totals and fills them again)
constexpr int SPEC_SPAN_CHUNKS = 16;         // span chunks per field a speculative layout has room for (tiles: 1 + DEVPLAN_KEEP_TILES, wave / general tiles: DEVPLAN_KEEP_TILES)

// what the device tiler needs to know about the batch (TileConsts of fcpp_tiler.h with the templates on the device)
struct DevTileConsts {
    const Pt2 *tu, *tc;           // the batch's U-turn / corner templates (device)
    int32_t nu, nc;
    int32_t turn_quiet, wave_factor, field_work_tiles, max_prims, fuse_spans;
    int32_t no_bases;             // counting pass of a small batch: the fields' point offsets are not known yet (ONE scan, after the pass)
    int32_t speculative;          // the tables are laid out by capacities: the counting pass and the scan watch them, the fill pass decides
                                  // the fusing of spans itself and does nothing when a flag of this generation is up
    int32_t closed_cut;           // the general stretches of fields with a closed-form span are cut in closed form (fcpp_cutfn.h); 0: by the window cut of round 4
    int32_t dense;                // sample_spacing > 0: span of all complete passes + quiet runs of the straights + general tiles between them (k_tile_fields, "dense")
    int64_t span_line_max;        // TileConsts.span_line_max
    double two_a, u_cap, c_line, fence_margin;
    int64_t reduce_wg_max;
    int64_t gen;                  // this counting phase's generation number (> 0)
    int64_t f0, f1;               // counting pass: the fields [f0, f1) (launch_devplan_count sets them: a large batch is counted in chunks)
    CutConsts cut;                // the closed-form cut (fcpp_cutfn.h), templates and chord tables on the device
};

// the device planner's scratch: one allocation the context keeps (grow-only); all pointers device
struct DevPlanScratch {
    fcpp_field *fields_in;        // n (copied from the host, unless the device reads the caller's pinned records where they lie)
    fcpp_field_info *info;        // n (copied back)
    DevField *fields_tmp;         // n: pt_off / prim_first still relative to the field
    DevPrim *prims_tmp;           // n x max_prims
    int64_t *counts, *bases;      // PC_COLS x n
    int64_t *blk_sums;            // PC_COLS x blocks of 1024 fields
    int64_t *totals;              // PLAN_TOTALS
    // the counting pass keeps the first DEVPLAN_KEEP_TILES wave tiles of every field (records with field-relative indices): the fill pass
    // copies and rebases them instead of cutting the field again (fields with more are cut again)
    DevTile *keep_tiles;          // n x DEVPLAN_KEEP_ROWS
    DevWaveTile *keep_wtiles;     // n x DEVPLAN_KEEP_WROWS
};
int64_t devplan_small_blocks();
size_t devplan_scratch_layout(int64_t n, int max_prims, DevPlanScratch *offsets_as_pointers /* offsets from 0, cast to pointers */);

// the tables the fill pass writes (pointers into the batch's slab, laid out by the host from the totals)
struct DevPlanTables {
    DevField *fields; DevPrim *prims; DevTile *tiles; DevWaveTile *wtiles; int32_t *general_ids; DevTile *span_chunks; DevTile *chunks;
    int32_t *stat_ids; int64_t *stat_first, *stat_run; int32_t *red_paths; DevFieldWork *field_work; DevFieldPack *field_packs; int32_t *open_wave_ids;
    double *seg; int32_t *seg_mask;
    // what batch creation computes once from the tables (k_field_junctions, k_run_consts, k_work_totals on the host path), done by the
    // field's own wavefront here; and the field's fcpp_field_info, kept with the batch for fcpp_batch_info
    TilePartial *partial; double2 *field_junc; TilePartial *work_totals; fcpp_field_info *info;
};

// the device tiler's LDS window over a field's general stretch (it slides), and the most primitives a field may have (8-bit indices in
// that window); a batch whose vehicle needs more (31+ headland loops) is set up on the host
constexpr int DEVPLAN_WINDOW = 576;
constexpr int DEVPLAN_PRIMS_CAP = 255;
constexpr int DEVPLAN_KEEP_TILES = 8;
constexpr int DEVPLAN_KEEP_ROWS = CUT_TILES_MAX;      // rows per field of the kept-tile arrays: every tile of a closed-form cut is kept (the fill pass never cuts such a field again)
static_assert(DEVPLAN_KEEP_ROWS >= DEVPLAN_KEEP_TILES, "the window cut keeps its first DEVPLAN_KEEP_TILES tiles in the same rows");
// rows per field of the kept WAVE-tile array: one more, whose 64 bytes hold a dense field's wave tiles per stretch (byte j: stretch j; 255: the
// stretch is the general kernel's) -- the counting pass's verdicts, read by the fill pass
constexpr int DEVPLAN_KEEP_WROWS = DEVPLAN_KEEP_ROWS + 1;
static_assert(sizeof(DevWaveTile) == 64, "a row of the kept wave tiles holds the 64 stretches' bytes");

// Small batches (at most 8192 fields) are counted WITHOUT the fields' point offsets: what depends on a span's alignment in the batch arrays
// -- its chunk count, hence whether its field's workgroup can write it (PC_SPAN, PC_SPAN_F, PC_WORK_SPAN_PTS, PC_UNFUSABLE) -- is derived by the
// one scan that follows the pass, from the offsets it has just computed (span_counts): one launch and one scan fewer in front of the pass.
// phase 1: plan + count.  Enqueues k_plan_fields, the scans and the counting pass; afterwards totals[] holds the sums and the flags, and so
// does totals_host (pinned host memory the device can write, or null) once the stream has got there: the scans write it themselves.
// fields: the records as the device reaches them (s.fields_in after a copy, or the caller's pinned memory).
// side, ev[n_ev]: a second stream and events for the experiment FCPP_COUNT_CHUNKS (a large batch's counting pass in chunks beside its planner's:
// measured slower, see launch_devplan_count); null / 0: one stream.
int launch_devplan_count(hipStream_t st, int64_t n, const PlanConsts &pc, const DevTileConsts &tc, const DevPlanScratch &s, const fcpp_field *fields,
                         int64_t n_polys, int check_obstacles, int64_t *totals_host, hipStream_t side = nullptr, hipEvent_t *ev = nullptr, int n_ev = 0);
// (tc.dense: the counting pass needs the fields' point offsets -- the chunks of every quiet run lie on 512-point boundaries of the batch arrays --:
// planner, scan of the points, pass, scan of the other columns)
// sizing only (fcpp_plan_points): k_plan_fields without primitives; counts[PC_POINTS][field] = points of the field
int launch_devplan_points(hipStream_t st, int64_t n, const PlanConsts &pc, const DevPlanScratch &s, const fcpp_field *fields);
// phase 2: the tables.  `bases` / `totals` as phase 1 left them.
int launch_devplan_fill(hipStream_t st, int64_t n, const DevTileConsts &tc, const DevConst &cst, const DevPlanScratch &s, const DevPlanTables &t);
// fcpp_math.h on the device (tests): fn 0 sincos, 1 atan2(a, b), 2 acos(a), 3 hypot(a, b)
int launch_debug_math(hipStream_t st, int fn, int64_t n, const double *a, const double *b, double *out0, double *out1);

}  // namespace fcpp
