/*
 * fcpp.h -- C ABI of libfcpp.so, the MI355X (gfx950) coverage-path geometry engine.
 *
 * The reference (qwagrox/field-coverage-path-planning @ 2025-10-24) is pure Python and has no
 * FFI: its boundary is the class surface of multi_layer_planner_v3.py ("MLP") and
 * genetic_algorithm_solver.py ("GA").  Each entry point below names the reference
 * method(s) it replaces; the Python mirror of that surface
 * (field_coverage_path_planning_amd/multi_layer_planner_v3.py) binds them through ctypes, and
 * INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only; every function returns FCPP_OK (0) or a negative FCPP_E* code and
 *     records a message retrievable with fcpp_last_error() (thread-local).
 *   - pointers named *_dev are DEVICE pointers (hipMalloc / torch CUDA tensors); all others
 *     are host pointers.  Path arrays are SoA float64: x[], y[], kappa[], v[] (km/h) plus
 *     one uint32 flag/segment word per point.
 *   - work is enqueued on the context's HIP stream (fcpp_ctx_set_stream); calls that return
 *     host data synchronise that stream themselves, the others are asynchronous.
 *   - a context is not thread-safe; distinct contexts are independent.
 *   - there is NO CPU fallback: without a usable HIP device every compute entry fails with
 *     FCPP_EHIP.
 */
#ifndef FCPP_H
#define FCPP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FCPP_ABI_VERSION 5

enum {
    FCPP_OK = 0,
    FCPP_EINVAL = -1,       /* the reference's ValueError: no field given / headland wider than field (MLP:135,597-598) */
    FCPP_EHEADLAND = -2,    /* a headland loop's inset polygon is empty (MLP:967-969 followed by the vstack at :939) */
    FCPP_EUNSUPPORTED = -3, /* not a convex quadrilateral, or a decision only GEOS could take: the corner-gap test `gap.area > 0.1` (MLP:1070) where
                               0.1 m^2 lies between the areas for the exact and for GEOS' polygonal buffer -- a band of ~0.04 m of working width */
    FCPP_EHIP = -4,         /* HIP runtime failure / no device */
    FCPP_ENOMEM = -5,
    FCPP_ESIZE = -6         /* negative / inconsistent sizes */
};

/* ---- VehicleParams (MLP:29-39), same field order and defaults ------------------------ */
typedef struct fcpp_vehicle {
    double working_width;          /* 3.2  */
    double min_turn_radius;        /* 8.0  */
    double max_work_speed_kmh;     /* 9.0  */
    double max_headland_speed_kmh; /* 15.0 */
    double headland_turn_speed_kmh;/* 4.0  */
    double max_lateral_accel;      /* 2.0  */
    double max_longitudinal_accel; /* 1.5  */
    double safety_factor;          /* 0.85 */
} fcpp_vehicle;

/* ---- sampling / validation options (build-defined; all-zero + geofence_tol = reference) -- */
enum { FCPP_TURN_ARC = 0, FCPP_TURN_CLOTHOID = 1 };
typedef struct fcpp_options {
    int32_t turn_model;     /* FCPP_TURN_ARC: circular arcs as MLP:807-825,1046-1062; _CLOTHOID: line->clothoid->arc->clothoid->line */
    int32_t clothoid_fit;   /* 0: kappa_max = 1/R ; 1: scale so the turn ends where the reference arc ends */
    double sample_spacing;  /* 0: the reference's fixed counts (2/20/15/20, reverse 0.5 m); >0: uniform arc-length spacing [m] */
    double clothoid_frac;   /* share of a turn's heading change spent in its two clothoids, in [0,1] */
    double geofence_tol;    /* a point further than this outside the field polygon is flagged [m] */
    int32_t obstacle_mode;  /* FCPP_OBSTACLES_FLAG: obstacles only set the validity flag of the points inside them (the reference: its
                               swath generator ignores the differenced work area, MLP:731-732); FCPP_OBSTACLES_AVOID: the swaths
                               of layer 1 are clipped against the obstacles and re-routed around them (below) */
    int32_t ring_order;     /* the order in which Shapely's `buffer(-offset).exterior.coords[:-1]` lists the four inset corners of a headland
                               loop (MLP:964-972) -- a GEOS fact the reference neither documents nor tests, and its corner formulas index
                               the list (MLP:1049-1060): FCPP_RING_AS_VERTICES = in the order of the field vertices (the documented
                               intent 0=LL, 1=LR, 2=UR, 3=UL of MLP:957; reproduces every number the reference publishes);
                               FCPP_RING_REVERSED = the other way round from the same first vertex (0, 3, 2, 1: LL, UL, UR, LR -- what a
                               clockwise shell starting at the first vertex lists).  INTEGRATION.md shows how a user with Shapely
                               checks which one their GEOS produces.  Host-side permutation only. */
} fcpp_options;
enum { FCPP_RING_AS_VERTICES = 0, FCPP_RING_REVERSED = 1 };

/* Obstacle-aware swaths (what README_en.md:156-178 promises and MLP:601-609 prepares: obstacles expanded by
 * working_width / 2 and taken out of the work area).  Build-defined -- the reference has no code for it.  In the frame of layer 1
 * (MLP:686-687) every obstacle is represented by the bounding box of its vertices grown by W/2 on every side; grown boxes that overlap
 * or touch are merged into their common bounding box until no two do, so the boxes are disjoint.  A swath line whose y lies strictly
 * inside a box is CLIPPED there and led around the obstacle (segment kind FCPP_KIND_DETOUR, nominal speed headland_turn_speed_kmh; legs
 * sampled like the reverse fills at the reference's sampling: 0.5 m, at least 2 points per leg):
 *   - an obstacle whose box was not merged: along its W/2-grown POLYGON (round 4) -- the convex hull of its vertices, every edge moved
 *     W/2 outwards, neighbours joined at their mitre point (a corner sharper than 60 degrees: a square cap W/2 beyond it), clipped to the
 *     grown box; it contains every point within W/2 of the hull.
 *     The swath is worked up to where its line meets the polygon, the way around is the shorter of the polygon's upper and lower chain
 *     between the two meeting points that stays inside the y-range of the main work area (never longer than the box's three legs), and a
 *     line that passes clear of the polygon is not interrupted;
 *   - merged boxes: three straight legs along the box -- up or down its near side, along its top or bottom (the closer one unless it lies
 *     outside the y-range of the main work area), back along its far side.
 * The legs stay inside their own box and the boxes are disjoint: no leg enters another obstacle.  A box (or polygon) with room on neither
 * side is refused.  Sub-swaths and legs are numpy.linspace runs between their end points in field coordinates.
 * End zones (round 4): the turn after a pass starts where its line ends and occupies a zone beyond that end -- the reference's half
 * circle about (max_x, y): 2 R along the line and R above it; the clothoid turn: the extents of its shape.  A box that meets that zone,
 * or either of the two lines within it, moves the turn inwards until the zone is free (again if the moved zone meets another box): both
 * passes end / start there, the U-turn is the same shape translated, and the strip beyond stays unworked.  Only the free ends -- the
 * start of the first pass, the end of the last -- are not moved: a box there, or one that leaves no side to pass, refuses the field
 * with FCPP_EUNSUPPORTED (its status; the other fields of the batch are planned).
 * Headland (round 4): a headland straight that crosses a grown box is cut at the box and led around it along the box's boundary -- from
 * where it enters to where it leaves, the shorter way whose box corners stay at least W/2 inside the field -- as FCPP_KIND_DETOUR legs
 * carrying the loop's FCPP_FLAG_HEADLAND; the pieces of the straight keep its sample density (length / 19 per step at the reference's
 * sampling).  A box over an end of a straight or within 2 R of a loop corner (where the corner turns are), across a reverse fill, or
 * with no way around inside the field refuses the field. */
enum { FCPP_OBSTACLES_FLAG = 0, FCPP_OBSTACLES_AVOID = 1 };

/* ---- one field = one planner instance (ctor arguments, MLP:63-72) ---------------------- */
typedef struct fcpp_field {
    double vx[4], vy[4];        /* field_vertices; for field_length/field_width: (0,0),(L,0),(L,H),(0,H) (MLP:127-132) */
    int32_t from_vertices;      /* 1 = field_vertices=..., 0 = field_length/field_width */
    int32_t has_start, has_end; /* start_point / end_point given */
    double start_x, start_y, end_x, end_y;
    int32_t n_obstacles;        /* obstacles=[...] : polygons obstacle_first .. +n_obstacles of the batch polygon table */
    int32_t _pad;
    int64_t obstacle_first;
} fcpp_field;

/* obstacle polygons of a whole batch, CSR layout, host pointers */
typedef struct fcpp_polys {
    int64_t n_polys;
    const int64_t *offsets;     /* n_polys + 1 */
    const double *x, *y;        /* offsets[n_polys] vertices */
} fcpp_polys;

/* ---- what the host-side setup decides per field (integers: bit-exact parity) ------------ */
typedef struct fcpp_field_info {
    int64_t point_offset;       /* first point of this field in the batch arrays */
    int64_t n_main, n_head;     /* len(main_work['path']), len(headland['path']) */
    int32_t n_swaths;           /* num_passes, MLP:739 */
    int32_t n_loops;            /* ceil(R / W), MLP:916 */
    int32_t start_corner;       /* MLP:360-385 */
    int32_t reverse_order, start_from_right; /* MLP:631-668 */
    int32_t rotated;            /* |rotation_angle| > 0.01, MLP:686 */
    int32_t start_kept, end_kept; /* after _validate_point, MLP:322-343 */
    int32_t shape;              /* 0 rectangle, 1 parallelogram, 2 other (MLP:137-163) */
    int32_t n_reverse[4];       /* reverse-fill points appended at corner c of the outer loop */
    int32_t status;             /* FCPP_OK or the error this field raises (its n_main = n_head = 0) */
    double corner_angles[4];    /* degrees, MLP:165-192 */
    double field_length, field_width, headland_width, rotation_angle;
    double approach_from[2], approach_to[2];    /* MLP:437-441 (valid if start_kept) */
    double departure_from[2], departure_to[2];  /* MLP:443-447 (valid if end_kept) */
} fcpp_field_info;

/* ---- per-field results reduced on the device ---------------------------------------- */
typedef struct fcpp_field_stats {
    double main_len_m, main_time_pre_s, main_time_s;   /* MLP:616-628 and :423-426 */
    double head_len_m, head_time_pre_s, head_time_s;   /* MLP:882-895 and :428-431 */
    double max_kappa, max_alat, max_jump;              /* verify_curvature_constraints over main||headland, MLP:1396-1408 */
    int64_t n_viol;          /* a_lat > max_lateral_accel, MLP:1401 */
    int64_t n_outside;       /* geofence: points outside the field polygon */
    int64_t n_in_obstacle;   /* points inside an obstacle polygon */
    int64_t n_adjusted;      /* points slowed by the curvature clamp, MLP:502-504 */
} fcpp_field_stats;

/* ---- flag / segment word -------------------------------------------------------------- */
enum {
    FCPP_KIND_SWATH = 0, FCPP_KIND_UTURN = 1, FCPP_KIND_HEAD_START = 2, FCPP_KIND_HEAD_STRAIGHT = 3,
    FCPP_KIND_CORNER = 4, FCPP_KIND_REVERSE = 5, FCPP_KIND_DETOUR = 6
};
#define FCPP_KIND_MASK 7u
#define FCPP_FLAG_HEADLAND 8u     /* layer 2 */
#define FCPP_FLAG_ALAT 16u        /* lateral acceleration above max_lateral_accel at this point */
#define FCPP_FLAG_OUTSIDE 32u     /* outside the field polygon (geofence) */
#define FCPP_FLAG_OBSTACLE 64u    /* inside an obstacle polygon */
#define FCPP_INDEX_SHIFT 8        /* bits 8..31: swath index i (layer 1) or loop*8 + corner/side (layer 2) */

typedef struct fcpp_ctx fcpp_ctx;
typedef struct fcpp_batch fcpp_batch;

/* ---- library / context ---------------------------------------------------------------- */
const char *fcpp_last_error(void);
int fcpp_abi_version(void);
void fcpp_vehicle_default(fcpp_vehicle *v);   /* VehicleParams() defaults, MLP:31-38 */
void fcpp_options_default(fcpp_options *o);   /* reference behaviour, geofence_tol = 1e-6 */
int fcpp_ctx_create(int device_id, fcpp_ctx **ctx);
int fcpp_ctx_destroy(fcpp_ctx *ctx);
int fcpp_ctx_set_stream(fcpp_ctx *ctx, void *hip_stream); /* a hipStream_t; NULL = HIP's default stream.  A new context starts on a private non-blocking stream */
int fcpp_ctx_synchronize(fcpp_ctx *ctx);
/* Where the setup of a batch runs (fcpp_batch_create: every field's __init__ and O(1) decisions, MLP:63-107, 591-668, 898-1084, and the
 * cut of its path into kernel work).  FCPP_SETUP_AUTO: on the DEVICE for batches of 16 fields and more with obstacle_mode = FLAG -- at the
 * reference's own sampling (sample_spacing = 0) and, since round 5, at any uniform sampling when no field of the batch has obstacles;
 * only the fcpp_field records go up, fcpp_field_info comes back -- and on the host's cores otherwise (obstacle-aware swaths, dense
 * sampling of fields with obstacles, fields beyond the device planner's limits, a handful of fields).  _HOST: always on the host (the checker of the device
 * path: both build the same tables, byte for byte).  _DEVICE: batches the device planner does not take fail with FCPP_EUNSUPPORTED.
 * The environment variable FCPP_SETUP=host|device sets the initial mode of new contexts. */
enum { FCPP_SETUP_AUTO = 0, FCPP_SETUP_HOST = 1, FCPP_SETUP_DEVICE = 2 };
int fcpp_ctx_set_setup(fcpp_ctx *ctx, int mode);
/* device memory for hosts without their own allocator (torch users pass tensor pointers instead) */
int fcpp_malloc(fcpp_ctx *ctx, int64_t bytes, void **dev_ptr);
int fcpp_free(fcpp_ctx *ctx, void *dev_ptr);
/* Output arrays for fcpp_batch_run, and the placement rule measured on MI355X: the hot kernels write x, y, kappa, v
 * and flagseg side by side, and five write streams within a few GiB of each other in device memory reach 4.6 TB/s where the same streams
 * >= 12-24 GiB apart reach 6.3-6.6 TB/s (small batches -- a few hundred MB of output -- live in the caches and do not care).
 *
 * fcpp_ctx_reserve_outputs gives the context an ARENA for that: ONE device allocation of 4 x pitch + lane bytes, made once (an allocation
 * of this size takes the driver seconds: it belongs to context creation, not to a plan call), five lanes `pitch_bytes` apart (0:
 * FCPP_OUTPUT_PITCH) of `lane_bytes` each (0: one pitch).  fcpp_outputs_alloc(pitch_bytes = 0) then places array k in lane k, first fit
 * among the live allocations -- any number of live batches share the arena, each with its arrays a pitch apart.  Nothing is reserved
 * unless the caller asks: without an arena (or for arrays larger than a lane) pitch_bytes = 0 gives one allocation with the arrays back to
 * back, pitch_bytes > 0 one allocation of 4 x pitch + array bytes of the caller's own.  Free all five with fcpp_outputs_free(ctx, x).
 * fcpp_ctx_reserve_outputs again re-sizes the arena (no live allocations allowed); fcpp_ctx_destroy releases it. */
#define FCPP_OUTPUT_PITCH ((int64_t)24 << 30)
int fcpp_ctx_reserve_outputs(fcpp_ctx *ctx, int64_t lane_bytes, int64_t pitch_bytes);
int fcpp_ctx_outputs_info(const fcpp_ctx *ctx, int64_t *lane_bytes, int64_t *pitch_bytes, int64_t *live_bytes);   /* 0, 0, 0 without an arena */
int fcpp_outputs_alloc(fcpp_ctx *ctx, int64_t n_points, int64_t pitch_bytes, double **x_dev, double **y_dev, double **kappa_dev,
                       double **v_dev, uint32_t **flagseg_dev);
int fcpp_outputs_free(fcpp_ctx *ctx, double *x_dev);
int fcpp_memcpy_h2d(fcpp_ctx *ctx, void *dst_dev, const void *src, int64_t bytes);
int fcpp_memcpy_d2h(fcpp_ctx *ctx, void *dst, const void *src_dev, int64_t bytes);

/* ---- planner: TwoLayerPathPlannerV37.__init__ + plan_complete_coverage (MLP:63-107, 387-465) ---- */
/* Host-only sizing and decisions for n_fields planners (no GPU needed): __init__ (MLP:63-107),
 * _select_best_start_corner (:360-385), _determine_optimal_pass_order (:631-668), num_passes (:739),
 * num_loops (:916), reverse-fill lengths (:1154-1288).  Fields that raise get info[i].status < 0. */
int fcpp_plan_count(const fcpp_vehicle *veh, const fcpp_options *opt, int64_t n_fields,
                    const fcpp_field *fields, const fcpp_polys *obstacles /* may be NULL unless obstacle_mode = AVOID */,
                    fcpp_field_info *info_out);
/* Sizing alone, for a job sharded over several GPUs (contiguous blocks of fields cut on the point counts): points_out[i] =
 * n_main + n_head of field i, 0 for a field that raises.  Runs on the device where the device-side setup takes the batch (fcpp_ctx_set_setup;
 * only the field records go up, 8 bytes per field come back), else on the host's cores like fcpp_plan_count.  Synchronises. */
int fcpp_plan_points(fcpp_ctx *ctx, const fcpp_vehicle *veh, const fcpp_options *opt, int64_t n_fields, const fcpp_field *fields,
                     const fcpp_polys *obstacles, int64_t *points_out);
/* Same setup, plus the per-field descriptors and the kernels' work lists on the device.  `fields` is read only during the call; it may be
 *   - DEVICE memory (round 5; engine.FieldTable.to_device(), a caller whose field table lives on the GPU): the device-side setup reads the
 *     records where they lie, nothing crosses PCIe in front of its first kernel (the headline's plan call 0.145 -> 0.138 ms); the writes that
 *     made the records must be ordered before the context's stream.  Records on another device are copied over.  The library's host paths
 *     (fewer than 16 fields, AVOID mode, FCPP_SETUP_HOST, fcpp_plan_count) copy them back first;
 *   - PINNED host memory (hipHostMalloc / hipHostRegister, a torch tensor with pin_memory): read by the device-side setup across PCIe where
 *     they lie -- no copy goes before its first kernel (fcpp_plan_points likewise);
 *   - pageable host memory: copied to the device first. */
int fcpp_batch_create(fcpp_ctx *ctx, const fcpp_vehicle *veh, const fcpp_options *opt, int64_t n_fields,
                      const fcpp_field *fields, const fcpp_polys *obstacles, fcpp_batch **batch);
int fcpp_batch_info(const fcpp_batch *batch, fcpp_field_info *info_out /* n_fields, may be NULL */,
                    int64_t *total_points);
/* Where the time of fcpp_batch_create went (wall-clock milliseconds on the host): the plan call of the reference,
 * plan_complete_coverage (MLP:387-465), times its field setup together with the generation -- fcpp_batch_create + one fcpp_batch_run is
 * that call for a whole batch, and bench.py reports it as such (end to end) beside the time of the step alone. */
typedef struct fcpp_setup_times {
    double host_plan_ms;    /* __init__ + the O(1) decisions of every field (MLP:63-107, 591-668, 898-1084), blocks of fields on the host's cores */
    double templates_ms;    /* turn templates sampled on the device and copied back (close to 0 when the context already had them) */
    double tiler_ms;        /* the paths cut into kernel work (quiet runs, spans, wave tiles with their halos, general tiles) */
    double image_ms;        /* device allocation + the tables written into the context's pinned staging memory */
    double h2d_ms;          /* the one host-to-device copy, the per-field junction kernel, and the stream drained */
    double total_ms;        /* the whole call */
    int64_t image_bytes;    /* bytes copied to the device */
    int32_t threads;        /* host threads that took part (FCPP_THREADS; default: the machine's, at most 16) */
    int32_t device_setup;   /* 1: the setup ran on the device (fcpp_ctx_set_setup): host_plan_ms = field records up + plan + counting pass + totals
                               back, image_ms = layout + allocation + obstacle table, tiler_ms = tables written + per-batch constants +
                               fcpp_field_info back, h2d_ms = 0, image_bytes = bytes copied to the device */
} fcpp_setup_times;
int fcpp_batch_setup_times(const fcpp_batch *batch, fcpp_setup_times *out);
/* The hot path: sample every path point (MLP:720-830, 898-1084, 1154-1218, 1580-1608), curvature
 * (MLP:513-536), curvature clamp (MLP:467-511), forward/backward sweeps (MLP:538-589), validator
 * (MLP:1373-1424 + geofence / obstacle flags) and metrics (MLP:1290-1311).  Outputs are device
 * arrays of total_points elements; stats_dev has n_fields entries.
 * mode 0: staged pipeline (one kernel per operator, 7 launches); mode 1: fused single-pass kernels
 * (each point is written once, nothing is read back: closed-form runs and spans, wave tiles at sparse sampling,
 * general tiles) -- same results. */
int fcpp_batch_run(fcpp_batch *batch, double *x_dev, double *y_dev, double *kappa_dev, double *v_dev,
                   uint32_t *flagseg_dev, fcpp_field_stats *stats_dev, int mode);
/* The reference's plan call for a whole batch in ONE entry: plan_complete_coverage (MLP:387-465) sets a NEW field up and generates its path in
 * one call, and so does this -- fcpp_batch_create, output arrays for the batch's points (fcpp_outputs_alloc(pitch_bytes = 0): from the
 * context's arena when it has one, else an allocation of their own; released with fcpp_outputs_free(ctx, *x_dev)) and one fcpp_batch_run
 * (mode 1), with nothing of the caller's between the three: the step is enqueued the moment the setup's totals have sized the arrays.
 * stats_dev: n_fields records of the caller's, or NULL: the batch's own (fcpp_batch_own_stats; they live as long as the batch).
 * Asynchronous like fcpp_batch_run: the arrays are complete when the context's stream is.
 * The batch stays valid for further fcpp_batch_run calls on the same arrays (or others) until fcpp_batch_destroy.  On an error nothing is
 * left allocated.  bench.py's headline step is this call + the stream drained. */
int fcpp_batch_plan(fcpp_ctx *ctx, const fcpp_vehicle *veh, const fcpp_options *opt, int64_t n_fields, const fcpp_field *fields,
                    const fcpp_polys *obstacles, fcpp_field_stats *stats_dev, fcpp_batch **batch, double **x_dev, double **y_dev,
                    double **kappa_dev, double **v_dev, uint32_t **flagseg_dev, int64_t *total_points);
/* the statistics records a batch keeps for callers that bring none (n_fields records inside the batch's own device allocation) */
int fcpp_batch_own_stats(const fcpp_batch *batch, fcpp_field_stats **stats_dev);
/* _generate_approach_path / _generate_departure_path (MLP:1313-1355): 50 points each, AoS (x,y) per
 * field at [field*100 .. +100); rows of fields without a kept start/end point are left untouched. */
int fcpp_batch_connectors(fcpp_batch *batch, double *approach_xy_dev, double *departure_xy_dev);
int fcpp_batch_destroy(fcpp_batch *batch);
/* Per-kernel device timing with HIP events (bench.py's roofline leg).  enable = k > 0: every k-th fcpp_batch_run from now on
 * dispatches each of its kernels with a start and a stop event of its own (hipExtLaunchKernel: the dispatch's time stamps, no
 * marker packets between the kernels; up to 256 runs are kept); such a run costs ~15 us more than a plain one, hence the stride.
 * fcpp_batch_stage_times synchronises, returns the summed milliseconds per stage over the recorded runs, their number, and
 * clears the record. */
int fcpp_batch_set_profiling(fcpp_batch *batch, int enable);
int fcpp_batch_stage_times(fcpp_batch *batch, int max_stages, double *ms_sum_out, int *n_stages_out, int *n_runs_out);
const char *fcpp_batch_stage_name(int mode, int stage);
/* points one launch of stage `stage` of pipeline `mode` processes (the stages of fcpp_batch_stage_name; mode 1: the closed-form spans,
 * the closed-form runs, the wave tiles of the sparse kernel, the general tiles, and all points for the reduction) */
int fcpp_batch_stage_points(const fcpp_batch *batch, int mode, int stage, int64_t *points);
/* how the fused pipeline (mode 1) splits the batch: points in closed-form runs and spans (k_plan_quiet) / all other points */
int fcpp_batch_point_split(const fcpp_batch *batch, int64_t *quiet_points, int64_t *general_points);
/* how the per-field statistics of the fused pipeline are reduced: paths per class of k_reduce_stats, by statistic entries of a path
 * (<= 64: 8 lanes, <= 256: a wavefront, <= 1024: a workgroup, more: 64 workgroups + join); classes_out[4] */
int fcpp_batch_reduce_classes(const fcpp_batch *batch, int64_t *classes_out);

/* ---- standalone operators on caller-supplied paths (CSR offsets, n_paths+1, device) -------
 * The offsets size the launches, so the host needs them: offsets_host (n_paths + 1 values, same content as offsets_dev) spares
 * the call a device-to-host copy and a stream synchronisation; NULL = the library reads offsets_dev back itself.  The tile table
 * built from the offsets is kept in the context and reused while consecutive calls bring the same offsets (compared by
 * content when offsets_host is given). */
/* _calculate_curvature for every interior point (MLP:513-536); end points get 0 */
int fcpp_curvature(fcpp_ctx *ctx, int64_t n_paths, const int64_t *offsets_dev, int64_t total_points,
                   const double *x_dev, const double *y_dev, double *kappa_dev, const int64_t *offsets_host);
/* _apply_curvature_based_speed_limit incl. _smooth_speed_profile (MLP:467-589); paths with fewer
 * than 3 points are returned unchanged (MLP:480-481).  clamp=0 runs _smooth_speed_profile only.
 * v_out_dev may alias v_in_dev; kappa_dev and n_adjusted_dev (int64[n_paths]) may be NULL. */
int fcpp_speed_plan(fcpp_ctx *ctx, const fcpp_vehicle *veh, int clamp, int64_t n_paths,
                    const int64_t *offsets_dev, int64_t total_points, const double *x_dev,
                    const double *y_dev, const double *v_in_dev, double *v_out_dev, double *kappa_dev,
                    int64_t *n_adjusted_dev, const int64_t *offsets_host);
/* verify_curvature_constraints (MLP:1373-1424) + _calculate_path_length/_calculate_work_time
 * (MLP:1290-1311) per path; stats_dev[n_paths] uses the main_* members for the whole path. */
int fcpp_verify(fcpp_ctx *ctx, const fcpp_vehicle *veh, int64_t n_paths, const int64_t *offsets_dev,
                int64_t total_points, const double *x_dev, const double *y_dev, const double *v_dev,
                fcpp_field_stats *stats_dev, const int64_t *offsets_host);
/* The validator on CALLER-SUPPLIED paths (README_en.md:183 "Electronic Fence Boundary Checking" -- the reference has the
 * bounds test of its start / end points only, MLP:322-343): per point the lateral-acceleration flag of verify_curvature_constraints
 * (MLP:1383-1401), the geofence flag against the path's field polygon and the obstacle flag against its obstacle polygons; per path the
 * statistics of fcpp_verify plus n_outside / n_in_obstacle.  Polygons are arbitrary simple polygons (host CSR tables, copied by the call):
 *   field_polys (NULL: no geofence): polygon p is the field of path p (n_polys == n_paths; a polygon of < 3 vertices: no geofence for that path).
 *       A point is FCPP_FLAG_OUTSIDE iff its signed distance to the polygon's boundary (+ inside, - outside; even-odd rule) is below
 *       -opt->geofence_tol -- for a convex field and a tolerance >= 0 the planner's own rule except beyond the corners, where the distance
 *       to the corner decides instead of the distances to the two edge lines.
 *   obstacles (NULL: none) with obstacle_offsets (n_paths + 1 values: path p is tested against the polygons [obstacle_offsets[p],
 *       obstacle_offsets[p + 1]); NULL: every path against all of them).  A point inside one (even-odd) is FCPP_FLAG_OBSTACLE.
 * flags_dev: total_points words, overwritten (FCPP_FLAG_ALAT | _OUTSIDE | _OBSTACLE).  Only veh's limits and opt->geofence_tol are read.
 * offsets_host as for the other standalone operators.  Synchronises. */
int fcpp_validate(fcpp_ctx *ctx, const fcpp_vehicle *veh, const fcpp_options *opt, int64_t n_paths, const int64_t *offsets_dev,
                  int64_t total_points, const double *x_dev, const double *y_dev, const double *v_dev, const fcpp_polys *field_polys,
                  const fcpp_polys *obstacles, const int64_t *obstacle_offsets, uint32_t *flags_dev, fcpp_field_stats *stats_dev,
                  const int64_t *offsets_host);
/* numpy.linspace straight segments (MLP:1013-1022, 1313-1355): seg_dev = n_seg x (x0,y0,x1,y1),
 * out_xy_dev = n_seg x n_points x 2 */
int fcpp_straight_segments(fcpp_ctx *ctx, int64_t n_seg, const double *seg_dev, int32_t n_points,
                           double *out_xy_dev);
/* _generate_corner_turn_arc (MLP:1580-1608) and _generate_corner_turn_with_reverse (MLP:1024-1084 with
 * _generate_optimal_reverse_path, MLP:1154-1218, and _calculate_distance_to_boundary, MLP:1220-1288) for n corners at once:
 * corner k = corners_dev[2k .. 2k+1], quadrant formula corner_index_dev[k] (0 LL, 1 LR, 2 UR, else UL; MLP:1049-1060).
 * The 15-point quarter arc of radius min_turn_radius goes to out_xy_dev[k * stride * 2 ..]; where with_reverse_dev[k] != 0
 * the reverse fill follows it: backwards along the arc's end tangent to the nearest side of the box [0, field_length] x
 * [0, field_width], at most 3R (2R when no side lies ahead), max(10, int(len / 0.5)) points.  counts_dev[2k], [2k+1] = points of
 * the arc and of the reverse fill.  stride (points per corner in out_xy_dev) must be >= 15 + max(10, int(3R / 0.5)).
 * The Shapely-dependent decision `gap.area > 0.1` (MLP:1070) stays with the caller (with_reverse_dev). */
int fcpp_corner_turns(fcpp_ctx *ctx, const fcpp_vehicle *veh, int64_t n, const double *corners_dev,
                      const int32_t *corner_index_dev, const int32_t *with_reverse_dev, double field_length,
                      double field_width, int32_t stride, double *out_xy_dev, int32_t *counts_dev);
/* Fresnel integrals C(t), S(t) = int_0^t cos|sin(pi u^2/2) du (README_en.md:111-120 promises the
 * clothoid; the reference has no code for it) */
int fcpp_fresnel(fcpp_ctx *ctx, int64_t n, const double *t_dev, double *c_dev, double *s_dev);

/* ---- GeneticAlgorithmSolver._calculate_distance / _calculate_fitness (GA:168-181) ---------- */
/* routes_dev: pop x n_nodes int32 permutations; D_dev: n_nodes x n_nodes float64 row-major.
 * order_mode 0 = left-to-right summation (bit-exact with the reference), 1 = tree reduction.
 * Precondition: every gene lies in [0, n_nodes); a chromosome that violates it gets distance = fitness = NaN (no out-of-range
 * read happens). */
int fcpp_ga_fitness(fcpp_ctx *ctx, int32_t n_nodes, int64_t pop, const double *D_dev,
                    const int32_t *routes_dev, double *dist_dev, double *fit_dev, int order_mode);

/* ---- GeneticAlgorithmSolver.solve: the evolution loop on the device (GA:64-115, 183-268) --------------
 * selection (GA:183-196), OX crossover (GA:198-242), swap mutation (GA:244-252), elitism (GA:254-268), fitness
 * (GA:168-181), best tracking and the convergence test (GA:90-113) for up to max_generations, whole generations on the
 * GPU.  The reference draws from the unseeded stdlib `random`; here every decision comes from the counter-based
 * generator Philox4x32-10 with key = seed and counter = (generation, pair, stream, block):
 *   stream 1 + s, word j : tournament candidate j of slot s = word % population_size, duplicates are redrawn
 *   stream 3             : crossover iff unit(w0, w1) < crossover_rate; cut points i = w2 % n, j = w3 % (n-1), j += (j >= i)
 *   stream 4 + c         : mutation of child c iff unit(w0, w1) < mutation_rate; positions as for the cut points
 *   unit(a, b) = ((a >> 5) * 2^26 + (b >> 6)) / 2^53
 * so a run is reproducible and identical to oracle/fcpp_oracle.c: orc_ga_evolve.  Ties in the elitism order go to the
 * larger index.  routes_dev must hold permutations of 0 .. n_nodes-1 (checked on the device before the first generation:
 * FCPP_EINVAL otherwise).  population_size must be even (the reference grows an odd population by one per generation, GA:203),
 * elite_size < population_size, tournament_size <= min(64, population_size), 2 <= n_nodes <= 2048. */
typedef struct fcpp_ga_config {   /* GAConfig, GA:20-29, + seed */
    int32_t population_size, max_generations;
    double crossover_rate, mutation_rate;
    int32_t elite_size, tournament_size, convergence_threshold, _pad;
    uint64_t seed;
} fcpp_ga_config;
typedef struct fcpp_ga_result {   /* the `stats` of GA:122-127 */
    int32_t generations;          /* generation + 1 */
    int32_t convergence_gen;      /* generation - generations_without_improvement */
    double best_distance, best_fitness;
} fcpp_ga_result;
/* routes_dev: pop x n int32, in = the initial population, out = the final one; best_route_dev: n int32 (as found, not yet
 * rotated to start at node 0, GA:118-120); hist_dev: NULL or 2 * max_generations doubles = best_fitness_history then
 * avg_fitness_history (GA:106-107; entries [0, generations) of each half are written).  Synchronises the stream. */
int fcpp_ga_evolve(fcpp_ctx *ctx, int32_t n_nodes, const fcpp_ga_config *cfg, const double *D_dev, int32_t *routes_dev,
                   int32_t *best_route_dev, double *hist_dev, fcpp_ga_result *result);

/* ---- scheduler inputs -----------------------------------------------------------
 * MultiVehiclePlanner._build_distance_matrix (MVP:229-259) / MultiFieldPlanner._calculate_distance_matrix (MFP:263-288):
 * D[i][j] = sqrt((x_i - x_j)^2 + (y_i - y_j)^2), 0 on the diagonal; node 0 is the depot, the others field centroids.
 * D_dev: n x n float64 row-major (the layout fcpp_ga_fitness / fcpp_ga_evolve take). */
int fcpp_distance_matrix(fcpp_ctx *ctx, int32_t n, const double *x_dev, const double *y_dev, double *D_dev);
/* MultiFieldPlanner._find_best_connection (MFP:290-320) for a batch of consecutive node pairs: pair p connects one of the exit
 * candidates [from_off[p], from_off[p+1]) of its first node with one of the entry candidates [to_off[p], to_off[p+1]) of its
 * second node; the shortest pair wins, the FIRST one in (exit-major, entry-minor) order among equals (the reference's
 * `distance < best_distance`).  Outputs per pair: the winning candidate indices (into fx/fy and tx/ty) and the distance
 * (-1, -1, +inf when a candidate list is empty).  All arrays on the device. */
int fcpp_best_connections(fcpp_ctx *ctx, int64_t n_pairs, const int64_t *from_off_dev, const int64_t *to_off_dev,
                          const double *fx_dev, const double *fy_dev, const double *tx_dev, const double *ty_dev,
                          int32_t *best_from_dev, int32_t *best_to_dev, double *best_dist_dev);

/* ---- coverage rasterisation -------------------------------------------------
 * Replaces the Shapely calls of verify_corner_coverage_grid_based (MLP:1426-1509: `LineString(path).buffer(W/2)
 * .contains(Point)` per 0.1 m grid cell of a 2R x 2R corner square, first for the turn, then for the reverse fill on
 * the cells still open) and of _calculate_coverage_rate (MLP:1357-1371: area(path.buffer(W/2) & area) / area(area)),
 * by one sampled operator: a job is a regular grid of sample points, a polyline A and an optional polyline B (stored
 * right after A in px/py).  A sample is covered by a polyline iff its distance to one of the polyline's segments
 * (consecutive point pairs, jumps included -- as in LineString(path)) is < radius (strict = 1, Shapely's `contains`)
 * or <= radius (strict = 0).  The distance test is division-free and the same in the oracle and on the GPU: with
 * a -> b the segment, p the sample, dot = (p-a).(b-a), len2 = |b-a|^2:
 *     dot <= 0     : |p-a|^2              < radius^2
 *     dot >= len2  : |p-b|^2              < radius^2
 *     otherwise    : ((b-a) x (p-a))^2    < radius^2 * len2
 * B is only evaluated on samples A left open (MLP:1489-1497).  counts (3 per job): samples in the region, of those
 * covered by A, of those covered by A or B.  grid (optional): one byte per sample, row-major [j][i] like the
 * reference's grid[j, i] (MLP:1483), bit 0 = A, bit 1 = B (and not A); samples outside the region stay 0. */
typedef struct fcpp_cover_job {
    double ox, oy;        /* sample (i, j) lies at (ox + (i + shift) * res, oy + (j + shift) * res) */
    double res, shift;    /* MLP:1449, 1477-1478: res = 0.1, shift = 0 (cell corners); 0.5 = cell centres for area estimates */
    double radius;        /* W / 2 (MLP:1471, 1363) */
    int32_t nx, ny;       /* samples per row, rows */
    int32_t n_a, n_b;     /* points of polyline A and of polyline B (0 = none) */
    int64_t pts_first;    /* index of A's first point in px / py */
    int64_t grid_first;   /* offset of this job's nx * ny bytes in `grid`, or -1: counts only */
    int32_t strict;       /* 1: distance < radius, 0: distance <= radius */
    int32_t region;       /* 0: every sample counts; 1: samples inside `outer` and not inside `inner` */
    double outer[12];     /* 4 half-planes (a, b, c): inside <=> a*x + b*y + c >= 0 for all four */
    double inner[12];
} fcpp_cover_job;

/* jobs: host array; px, py: device (n_pts points, e.g. the x / y arrays of fcpp_batch_run); grid_dev may be NULL when no job
 * asks for it; counts_dev: 3 * n_jobs int64 (zeroed by the call).  Runs on the context's stream and synchronises it. */
int fcpp_cover_grid(fcpp_ctx *ctx, int64_t n_jobs, const fcpp_cover_job *jobs, int64_t n_pts, const double *px_dev,
                    const double *py_dev, uint8_t *grid_dev, int64_t *counts_dev);

/* ---- the final gather of a job sharded over the GPUs of a node ----------------------------------
 * Fields are independent: every rank plans a contiguous block of them (cut on fcpp_plan_points) with its own context and batch, and the
 * only exchange is this gather of the blocks' results on one rank.  For each of n_arrays arrays (elem_bytes[a] bytes per element: 8 for
 * x / y / kappa / v, 4 for flagseg, sizeof(fcpp_field_stats) for the statistics with counts in fields) rank r contributes counts_per_rank[r]
 * elements from send_dev[a]; the root receives them in rank order into recv_dev[a] (sum of the counts elements; NULL on the other ranks).
 * One ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on the context's stream over the caller's communicator (`nccl_comm`: an
 * ncclComm_t of RCCL) -- point to point, every peer over its own xGMI link to the root, no ring and no staging copy; the root's own block
 * is a device-to-device copy (flags bit 0: it, too, goes through the communicator -- a one-GPU test of the RCCL path).  Asynchronous: the
 * arrays are complete when the stream is.  RCCL is looked up in the process at the first call, not linked: FCPP_EUNSUPPORTED without
 * it.  (The Python mirror gathers through torch.distributed instead -- sharding.py -- whose process group owns the communicator.) */
int fcpp_gather(fcpp_ctx *ctx, void *nccl_comm, int rank, int world, int root, int n_arrays, const void *const *send_dev, const int32_t *elem_bytes,
                const int64_t *counts_per_rank, void *const *recv_dev, int flags);

/* ---- diagnostics (tests/) -----------------------------------------------------------------------
 * The setup's transcendentals (csrc/fcpp_math.h: plain IEEE operations so that host and device agree bit for bit) evaluated on the host /
 * on the device: fn 0 = sin and cos of a -> out0, out1; 1 = atan2(a, b); 2 = acos(a); 3 = hypot(a, b) -> out0.  The _dev variant takes
 * device pointers and synchronises. */
int fcpp_debug_math(int fn, int64_t n, const double *a, const double *b, double *out0, double *out1);
int fcpp_debug_math_dev(fcpp_ctx *ctx, int fn, int64_t n, const double *a_dev, const double *b_dev, double *out0_dev, double *out1_dev);
/* One of a batch's device tables copied to the host (dst = NULL: only its size in *bytes_out): 0 field descriptors, 1 primitives, 2 tiles,
 * 3 wave tiles, 4 general tile ids, 5 chunks, 6 span chunks, 7 statistics entry -> tile, 8 first entry per field, 9 run length per entry,
 * 10 reduction lists, 11 field work, 12 open wave tile ids, 13 connector segments, 14 connector masks, 15 statistics slots (after batch
 * creation: the closed-form statistics of the quiet runs), 16 junction constants, 17 run totals per field of field work, 18-21 obstacle
 * offsets / x / y / bounding boxes, 22 the packs of k_plan_sparse_fields.  tests/test_gpu_devplan.py compares the tables of a batch set up on the device with those of the
 * same batch set up on the host. */
int fcpp_batch_debug_table(const fcpp_batch *batch, int table, void *dst, int64_t cap_bytes, int64_t *bytes_out);

#ifdef __cplusplus
}
#endif
#endif /* FCPP_H */
