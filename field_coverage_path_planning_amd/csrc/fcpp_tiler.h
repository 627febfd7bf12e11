// fcpp_tiler.h -- the host-side tiler of the fused pipeline: cuts every field's path into work for the four single-pass kernels
// (spans and closed-form runs for k_plan_quiet, wave tiles for k_plan_sparse, general tiles for k_plan_fused) plus the lists
// k_reduce_stats walks, and lays all of it out as ONE image that goes to the device in one copy.
//
// Pure C++ (no HIP): fields are tiled block by block on the host's cores (fcpp_parallel.h), the blocks' records are merged in block
// order, so the image does not depend on the thread count.  A field's cut depends on the field alone, never on its position in the
// batch; only indices (field, tile slot, primitive, output offset) differ between two equal fields.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "fcpp_internal.h"
#include "fcpp_tilefn.h"
#include "fcpp_cutfn.h"

namespace fcpp {

// what the tiler needs to know about the batch besides the host plan
struct TileConsts {
    const Pt2 *tu = nullptr, *tc = nullptr;   // host copies of the U-turn / corner templates (nu / nc samples)
    int nu = 0, nc = 0;
    bool templates_ok = false;                // the copies are there (wave tiles need them to size their halos)
    bool turn_quiet = false;                  // U-turns of this batch are closed form (closed_form_turns, fcpp_api.cpp)
    double two_a = 0.0, u_cap = 0.0, c_line = 0.0;     // 2 a_lon, (v_max / 3.6)^2, (v_work / 3.6)^2
    double fence_margin = 1e-7;               // a point whose edge functions are all at least this cannot be flagged by the device's geofence test (1e-7 - geofence_tol)
    int wave_factor = 24;                     // wave tiles where wave_factor * 2a * line step >= u_cap
    int wave_points = 128;                    // points per wave tile: 64 (one per lane) or 128 (two per lane, fcpp_sparse2_fn.h)
    bool field_work = true;                   // fields with few wave tiles and nothing else general: planned and reduced by one workgroup (DevFieldWork)
    int field_work_tiles = FIELD_WORK_TILES;  // ... at most this many (<= FIELD_WORK_TILES)
    bool fuse_spans = true;                   // ... and that workgroup also writes the field's layer-1 span (its chunks are then not in k_plan_quiet's list)
    int64_t reduce_wg_max = 1024;             // statistic entries one workgroup reduces; beyond: 64 workgroups + join
    bool closed_cut = false;                  // reference sampling: the general stretch of a field with a closed-form span is cut by fcpp_cutfn.h (as the device planner cuts it)
    CutConsts cut = {};                       // ... with these constants (host copies of the templates and their chord tables)
    // a field's complete passes (line + closed-form U-turn) form ONE span -- one run, one statistics entry, chunks decoded by (pass, offset) -- when
    // its lines' quiet zones are shorter than span_line_max; fields with obstacles keep 64 (their lines and turns stay runs of their own: a line's
    // chunks test the obstacles once per line, a span's chunks pair by pair -- cfg3: 0.36 vs 0.43 ms a step).  Round 5: unlimited for fields without
    // obstacles, whatever the sampling -- 2P - 1 runs per field become one (cfg2 at 0.5 m: a 20 MB image -> 6 MB, the plan call 3.1 -> 2.4 ms,
    // the step itself 1.23 -> 1.10 ms); FCPP_DENSE_SPAN=0 keeps round 4's runs for the A/B.
    int64_t span_line_max = INT64_MAX;
    bool device_chunks = false;               // the chunk lists of k_plan_quiet are expanded on the device from chunk groups (fcpp_batch_create; false: written by the host, the checker)
};

// the tables of the fused pipeline inside one allocation; all offsets in bytes from the image's start, 256-byte aligned
struct ImageLayout {
    size_t fields = 0, prims = 0, tiles = 0, wtiles = 0, general_ids = 0, chunks = 0, span_chunks = 0, stat_ids = 0, stat_first = 0,
           stat_run = 0, red_paths = 0, field_work = 0, field_packs = 0, open_wave_ids = 0, chunk_groups = 0, obs_off = 0, obs_x = 0, obs_y = 0, obs_bbox = 0, seg = 0, seg_mask = 0;
    size_t upload_bytes = 0;                  // [0, upload_bytes) is built on the host and copied
    size_t partial = 0, red_scratch = 0, field_junc = 0, work_totals = 0, info = 0, own_stats = 0;      // device-only scratch behind it (own_stats: fcpp_batch_plan's statistics records when the caller brings none)
    bool info_on_device = false;              // the batch was set up on the device: its fcpp_field_info records live in the slab (info)
    size_t total_bytes = 0;
    int64_t n_fields = 0, n_prims = 0, n_tiles = 0, n_wave = 0, n_general = 0, n_chunks = 0, n_span_chunks = 0, n_runs = 0, n_stat = 0;
    int64_t n_chunk_groups = 0;               // host-built images: the chunk lists are expanded on the device from this many groups (0: the lists are in the image)
    int64_t n_red[4] = { 0, 0, 0, 0 };       // fields reduced by k_reduce_stats, by class (fields of field_work are in none)
    int64_t n_field_work = 0, n_open_wave = 0;  // fields planned AND reduced by one workgroup each / wave tiles of the other fields
    int64_t n_polys = 0, n_poly_verts = 0;
    int64_t quiet_points = 0, span_points = 0, chunk_points = 0, wave_points = 0;
    int64_t work_wave_points = 0;             // the part of wave_points in fields of field_work
    int64_t unfusable_work = 0;               // fields of field work whose span has more than FUSED_SPAN_CHUNKS chunks
    int64_t work_span_points = 0;             // points of layer-1 spans written by k_plan_sparse_fields (not part of span_points: those are k_plan_quiet's)
    int64_t n_work[4] = { 0, 0, 0, 0 };       // fields of field_work by class (field_work_class: wavefronts of the workgroup); n_field_work = their sum
    int64_t wave_fail[5] = { 0, 0, 0, 0, 0 }; // diagnostics: stretches refused for wave tiles, by reason
    int64_t wave_inside = 0;                  // wave tiles whose outputs the host found inside the geofence
    int wave_tile_points = 64;                // points per wave tile (TileConsts.wave_points)
};

// offsets of the tables from their counts; the obstacle part of the image (both also used by the device-side setup, fcpp_api.cpp)
void layout_image(ImageLayout &lay);
// (rebase: dst holds the image from byte `rebase` on -- the obstacle region alone when rebase = lay.obs_off)
void fill_obstacles(const fcpp_polys *polys, const ImageLayout &lay, unsigned char *dst, size_t rebase = 0);

struct BlockTiles;      // a block's records before the merge (fcpp_tiler.cpp)

class BatchTiler {
public:
    BatchTiler();
    ~BatchTiler();
    // phase 1: tile every block of `hp` (side by side), size the image.  `polys` (may be NULL): the batch's obstacle table, copied into
    // the image with one bounding box per polygon.
    int plan(const HostPlan &hp, const TileConsts &tc, const fcpp_polys *polys, ImageLayout &lay, std::string &err);
    // phase 2: write the image into `dst` (lay.upload_bytes bytes; pinned host memory in fcpp_batch_create), side by side
    void fill(const HostPlan &hp, const fcpp_polys *polys, const ImageLayout &lay, unsigned char *dst) const;
private:
    int plan_impl(const HostPlan &hp, const TileConsts &tc, const fcpp_polys *polys, ImageLayout &lay, std::string &err);
    std::vector<BlockTiles> *blocks_;
};

}  // namespace fcpp
