// fcpp_internal.h -- structures shared by the host-side setup (fcpp_host.cpp) and the HIP kernels.
//
// Data layout in HBM:
//   * outputs: SoA float64 x[], y[], kappa[], v[] and uint32 flagseg[], one element per path point,
//     fields laid out back to back (field f owns [pt_off, pt_off + n_total)).
//   * per-field descriptor DevField (O(1) per field: the boustrophedon layer is decoded in closed
//     form from the point index) and a short list of DevPrim primitives for the headland layer.
//   * tile table: one entry per workgroup = (field, first point, count); tiles never straddle fields.
#pragma once
#include <stdint.h>

#include "../../include/fcpp.h"

namespace fcpp {

constexpr int TILE_POINTS = 512;   // points per tile: one wavefront x 8 points in the fused kernel, 256 threads x 2 in the staged ones

enum PrimKind : int32_t { PRIM_POINT = 0, PRIM_LINSPACE = 1, PRIM_ARC = 2, PRIM_RAY = 3, PRIM_CAC = 4, PRIM_UTURN = 5 };

// one primitive: a run of `n` points starting at path index `start` -- the headland layer (MLP:943-1084) and, with obstacle-aware
// swaths (fcpp_options.obstacle_mode), layer 1 as well (sub-swaths, detour legs, U-turns)
struct DevPrim {
    int64_t start;   // index in the field's concatenated path (>= gen_main)
    int32_t n;
    int32_t kind;    // PrimKind
    int32_t form;    // PRIM_ARC: corner index 0..3 selecting the quadrant formula (MLP:1049-1060)
    uint32_t fs;     // flag/segment base word
    double v_nom;    // nominal speed of the run [km/h]
    // PRIM_POINT    a0,a1 = x,y
    // PRIM_LINSPACE a0..a3 = x0,y0,x1,y1 ; a4,a5 = step_x, step_y (numpy.linspace step)
    // PRIM_ARC      a0,a1 = corner x,y ; a2 = R ; a3 = theta_end ; a4 = theta step
    // PRIM_RAY      a0,a1 = origin ; a2,a3 = unit direction ; a4 = length ; a5 = t step
    // PRIM_CAC      a0,a1 = start ; a2 = heading ; a3 = signed heading change ; a4 = Re ; a5 = s step ; a6 = total length
    // PRIM_UTURN    a0 = x anchor, a1 = y of the pass, in the frame of layer 1 ; a2,a3 = rot cos, sin ; a4,a5 = rot centre ;
    //               form: bit 0 turn right, bit 1 rotate back, bit 2 clothoid model (samples = the batch's U-turn template, MLP:791-830)
    double a[7];
};

// closed-form description of one field's plan
struct DevField {
    int64_t pt_off;      // first point in the batch arrays
    int64_t n_main;      // points of layer 1
    int64_t gen_main;    // points of layer 1 that are generated in closed form from the point index (n_main; 0 when layer 1 is a list
                         // of primitives: obstacle-aware swaths)
    int64_t n_total;     // n_main + n_head
    // layer 1 (MLP:720-789) in the rotated frame
    double lsx, lex;     // line_start_x, line_end_x (MLP:736-737)
    double line_step;    // numpy.linspace step lsx -> lex over n_line points
    double min_x, max_x, min_y;
    double W, R;
    double turn_step;    // arcs: pi/(n_turn-1) ; CAC: T/(n_turn-1)
    double turn_end;     // arcs: pi ; CAC: total length T
    double turn_Re;      // CAC effective radius
    double rot_cos, rot_sin, rot_cx, rot_cy;  // rotate-back (MLP:709-714)
    double v_work, v_turn;
    int32_t P;           // num_passes
    int32_t n_line, n_turn;
    int32_t reverse_order, start_from_right, rotated;
    int32_t turn_model;
    int32_t prim_first, prim_count;   // headland primitives
    // validator
    int32_t obs_first, obs_count;     // obstacle polygons (batch polygon table)
    int32_t span_inside;              // 1: the bounding box of layer 1's lines and U-turns lies inside the geofence with the tiler's margin --
                                      // no point of its closed-form span can be flagged, the span kernels skip the four edge tests per point
    double ex[4], ey[4], eo[4];       // field edges as inward unit normals: inside <=> ex*px + ey*py + eo >= -tol
};

struct DevTile {
    int32_t field;
    int32_t count;       // <= TILE_POINTS
    int64_t start;       // first point of the tile inside the field's path
    int32_t idx0, off0;  // layer 1 only: pass position idx and offset inside the pass of `start` (start = idx0*per + off0)
    int32_t quiet;       // 1: the tile and its sweep neighbourhood lie on ONE swath line (closed-form results, see fcpp_fused.hip)
    int32_t stat_tile;   // general tiles and quiet chunks: the statistics entry (slot of `partial`) that collects this tile's / chunk's results
};

// A chunk group = consecutive quiet runs of one field whose points are cut TOGETHER on 512-point boundaries of the batch arrays (one run;
// or the swath lines and U-turns of layer 1 that follow each other).  The host lists the groups -- in SEGMENTS of at most
// CHUNK_SEGMENT chunks, each with the places its records go to -- and the device expands them into the chunk records of k_plan_quiet
// (k_expand_chunks, a wavefront per segment): at dense sampling the records are most of the batch's image, the segments a hundredth.
constexpr int CHUNK_SEGMENT = 256;
struct DevChunkGroup {
    int32_t field;
    int32_t e0;          // statistics entry of the group's first run (its tile = tiles[e0], tiles[e0 + k].start = where run k begins)
    int32_t n_runs;
    int32_t j0, n;       // this segment: chunks [j0, j0 + n) of the group
    int32_t _pad;
    int64_t g0;          // the group's first point in the batch arrays
    int64_t total;       // points of the group
    int64_t chunk_base;  // where the segment's records go: those that lie in one run of a straight / U-turn primitive (list `chunks`) ...
    int64_t span_base;   // ... and those of layer-1 spans or across runs (list `span_chunks`), each list in chunk order
};
static_assert(sizeof(DevChunkGroup) == 56, "layout of the image");

// A quiet run = one quiet zone of a straight primitive (consecutive quiet tiles).  Its points are STORED by chunks cut on
// 512-point boundaries of the batch arrays (aligned 1 KiB stores), its length / time statistics are one closed form for the
// whole run, credited to the run's first tile -- so the statistics do not depend on where the field sits in the batch.
struct DevRun {
    int32_t tile;        // first tile of the run (geometry: field, idx0, off0, kind)
    int32_t _pad;
    int64_t count;       // points in the run
};

// A wave tile of the sparse kernel (fcpp_sparse.hip) as ONE self-contained 64-byte record: everything the wavefront needs to place
// its 64 lanes on the path, so that after this single fetch the field's geofence, the primitives and the turn template samples are
// all loaded side by side (the kernel is latency-bound on its dependent loads otherwise).  Lane l holds path point first + l.
struct DevWaveTile {
    int64_t out_base;        // index of lane 0's point in the batch arrays (pt_off + first)
    int32_t field;
    int32_t tile;            // the tile's statistics entry = its slot in the partial statistics (the entries of a field lie side by side)
    uint8_t count;           // output lanes ...
    uint8_t hb, hf;          // ... after hb halo lanes and before hf halo lanes (hb + count + hf <= 64)
    uint8_t inside;          // 1: the tiler found every output point inside every edge of the field polygon by more than the device's test can fire at (fcpp_tilefn.h:
                             // tiler_inside: 1e-7 m - geofence_tol + 256 ulps of the coordinates): no geofence test needed
    int32_t rel_main;        // gen_main - first: lanes below it are generated from layer 1's closed form (clamped to [-2, 1 << 30])
    int32_t rel_seam;        // n_main - first: the lane of the first point of layer 2 (clamped likewise)
    int32_t rel_last;        // (n_total - 1) - first: the lane of the path's last point (clamped likewise); first == 0 <=> rel_zero == 0
    int32_t rel_zero;        // -first clamped: the lane of the path's first point (0) or negative
    int32_t idx0, off0;      // layer 1: (pass position, offset in the pass) of lane 0
    int32_t p0;              // layer 2: primitive (batch-wide index) of the first layer-2 lane ...
    int32_t r0;              // ... and that lane's sample index in it is  lane + r0
    uint8_t thr[8];          // lane at which primitive p0 + 1 + k starts (255 = not in this tile)
};
static_assert(sizeof(DevWaveTile) == 64, "DevWaveTile is fetched as one 64-byte record");

// A field whose general points are all in wave tiles, at most FIELD_WORK_TILES of them: ONE workgroup of k_plan_sparse_fields plans
// its tiles, a wavefront each, and then reduces the field's statistics itself: its tiles' partial results through LDS, its quiet runs'
// from one record summed at batch creation, so such fields need no k_reduce_stats launch.  (Workgroups of four wavefronts; a wavefront without a tile leaves at once.  Measured with eight-wave
// workgroups for four-tile fields whose idle wavefronts waited at the barrier: headline 60 instead of 35 us.)
constexpr int FIELD_WORK_TILES = 4, FIELD_WORK_ENTRIES = 16;
// The launches of k_plan_sparse_fields, one per class of fields by their number of wave tiles, workgroups of as many wavefronts.  Only
// the first class is in use: with fields of five to eight tiles in classes of their own (workgroups of 5, 6, 8 wavefronts) cfg5 -- nine
// fields in ten have five tiles -- took 2.20 instead of 1.97 ms: such fields stay with k_plan_sparse and k_reduce_stats.
constexpr int FIELD_WORK_WAVES[4] = { 4, 5, 6, 8 };
inline int field_work_class(int n_tiles) { return n_tiles <= 4 ? 0 : (n_tiles == 5 ? 1 : (n_tiles == 6 ? 2 : 3)); }
struct DevFieldWork {
    int32_t field;
    int32_t n_tiles;         // wave tiles wtiles[w_first .. w_first + n_tiles)
    int32_t w_first;
    int32_t e_first;         // statistics entries [e_first, e_first + n_entries): the field's runs and tiles in path order
    int32_t n_entries;
    int32_t fused_span;      // points of the field's layer-1 span when the field's workgroup writes it too (at most FUSED_SPAN_CHUNKS chunks), else 0
    int32_t _pad[2];
};
static_assert(sizeof(DevFieldWork) == 32, "DevFieldWork is fetched as one 32-byte record");

// Everything ONE workgroup of k_plan_sparse_fields needs of its field, side by side: the work record, the field's (at most four) wave tiles,
// its descriptor, and for every tile a copy of the (at most nine) primitives its points lie in.  Every address follows from the
// workgroup's index alone, so a wavefront asks for its tile, the field's constants and its primitives AT ONCE -- through the separate
// tables the chain was work -> wave tile -> field / primitives, each hop a trip to memory that the wavefront's 800 vector instructions
// waited behind (profiles/r03_sparse_sections.txt: four fifths of a wave tile's cycles passed before its points were there).
constexpr int PACK_TILE_PRIMS = 9;
constexpr int FUSED_SPAN_CHUNKS = 4;       // a span of at most this many 512-point chunks is written by its field's own workgroup (k_plan_sparse_fields)
constexpr int TMPL_LDS_SAMPLES = 64;      // turn-template samples a wavefront stages in LDS (fcpp_pointfn.h: TMPL_LDS)
struct DevFieldPack {
    DevFieldWork work;                                  //    0
    int64_t span_points;                                //   32: points of the field's layer-1 span that this workgroup writes too (0: none, or
                                                        //       the span's chunks are k_plan_quiet's: TileConsts.fuse_spans off)
    int32_t _pad0[6];                                   //   40
    DevWaveTile tile[FIELD_WORK_TILES];                 //   64: unused tiles zero
    DevField field;                                     //  320
    double _pad1;                                       //  632
    DevPrim prims[FIELD_WORK_TILES][PACK_TILE_PRIMS];   //  640: tile t's primitives from wtiles[t].p0 on (slots beyond its last primitive zero)
    double _pad2[4];                                    // 3808
};
static_assert(sizeof(DevField) == 312 && sizeof(DevPrim) == 88 && sizeof(DevFieldPack) == 3840 && sizeof(DevFieldPack) % 128 == 0,
              "a pack is thirty 128-byte lines");

// batch-wide turn templates: every field of a batch shares the vehicle and the sampling options, hence the number of
// samples and the shape of its U-turns (nu) and corner turns (nc)
struct TurnTemplates {
    int32_t turn_model, nu, nc, _pad;
    double R;
    double u_end, u_step, u_Re;   // U-turn: arcs end = pi (angle), clothoid end = total length
    double c_end, c_step, c_Re;   // corner turn: arcs end = pi/2, clothoid end = total length
};

// per-tile partial statistics (reduced per field in a fixed order => run-to-run identical sums)
struct TilePartial {
    double main_len, main_time_pre, main_time, head_len, head_time_pre, head_time;
    double max_kappa, max_alat, max_jump;
    int64_t n_viol, n_outside, n_in_obstacle, n_adjusted;
};

}  // namespace fcpp

// ---- host-side plan of a batch (fcpp_host.cpp) ----------------------------------------------
#ifdef __cplusplus
#include <string>
#include <vector>
namespace fcpp {
// The setup of a batch is cut into blocks of PLAN_BLOCK_FIELDS consecutive fields, planned (and tiled, fcpp_tiler.h) side by side on the
// host's cores; a block owns the primitives of its fields.  The cut is fixed, so what is built does not depend on the thread count.
constexpr int64_t PLAN_BLOCK_FIELDS = 64;
struct PlanBlock {
    int64_t f0 = 0, f1 = 0;          // fields [f0, f1)
    std::vector<DevPrim> prims;      // the block's primitives: fields with identical constructor arguments share one list
    int64_t prim_base = 0;           // index of prims[0] in the batch-wide primitive table
    int64_t point_base = 0;          // first point of field f0 in the batch arrays
    int64_t points = 0;
};
struct HostPlan {
    std::vector<fcpp_field_info> info;
    std::vector<DevField> fields;    // pt_off and prim_first are batch-wide
    std::vector<PlanBlock> blocks;   // (empty without want_device)
    std::vector<int32_t> same_as;    // per field: an earlier field of the same block with equal constructor arguments (its plan was copied,
                                     // its primitives are shared), or -1; with want_device only
    TurnTemplates tt;
    int64_t total_points = 0, total_prims = 0;
    const DevPrim *prims_of(int64_t field) const      // the primitives of a field: [0, fields[field].prim_count)
    {
        const PlanBlock &b = blocks[(size_t)(field / PLAN_BLOCK_FIELDS)];
        return b.prims.data() + (fields[(size_t)field].prim_first - b.prim_base);
    }
};
// Builds info (+ device descriptors when want_device) for n fields; returns FCPP_OK or FCPP_E*.  Threaded over blocks of fields.
int build_host_plan(const fcpp_vehicle &veh, const fcpp_options &opt, int64_t n, const fcpp_field *fields, const fcpp_polys *polys,
                    bool want_device, HostPlan &out, std::string &err);
// the batch-wide turn templates' description (sample counts, shape parameters) for a vehicle and options; validates both
int plan_templates(const fcpp_vehicle &veh, const fcpp_options &opt, TurnTemplates &tt, std::string &err);
// the polygon table of a batch: counts, offsets (start at 0, non-decreasing) and coordinate pointers; FCPP_OK or FCPP_ESIZE / FCPP_EINVAL
int validate_polys(const fcpp_polys *polys, std::string &err);
// host+device Fresnel / CAC helpers live in fcpp_geom.h
}  // namespace fcpp
#endif
