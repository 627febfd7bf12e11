// fcpp_api.cpp -- the C ABI declared in include/fcpp.h: argument checking, device buffers, launches.
// No CPU compute path exists here: every operator ends in a HIP kernel launch or fails with FCPP_EHIP.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "fcpp_cover.h"
#include "fcpp_ga.h"
#include "fcpp_device.h"
#include "fcpp_devplan.h"
#include "fcpp_internal.h"
#include "fcpp_parallel.h"
#include "fcpp_tiler.h"

using namespace fcpp;

namespace fcpp { thread_local LaunchProf g_launch_prof; }

namespace {
thread_local std::string g_err;

int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return fail(FCPP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));                   \
    } while (0)
#define LAUNCHCHK(expr)                                                                                  \
    do {                                                                                                 \
        int e_ = (expr);                                                                                 \
        if (e_ != 0)                                                                                     \
            return fail(FCPP_EHIP, std::string(#expr) + ": " + hipGetErrorString((hipError_t)e_));       \
    } while (0)

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    hipError_t alloc(size_t count)
    {
        release();
        n = count;
        if (count == 0) return hipSuccess;
        return hipMalloc((void **)&p, count * sizeof(T));
    }
    hipError_t upload(const std::vector<T> &h, hipStream_t st)
    {
        hipError_t e = alloc(h.size());
        if (e != hipSuccess || h.empty()) return e;
        return hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st);
    }
};

DevConst make_const(const fcpp_vehicle &veh, const fcpp_options &opt)
{
    DevConst c;
    c.a_lat = veh.max_lateral_accel; c.a_lon = veh.max_longitudinal_accel; c.sf = veh.safety_factor;
    c.geofence_tol = opt.geofence_tol;
    c.v_work = veh.max_work_speed_kmh; c.v_turn = veh.headland_turn_speed_kmh; c.v_head = veh.max_headland_speed_kmh;
    const double vm = std::max(std::max(veh.max_work_speed_kmh, veh.max_headland_speed_kmh),
                               std::max(veh.headland_turn_speed_kmh, 2.5)) / 3.6;
    c.u_cap = vm * vm;
    c.inv_sf36 = 1.0 / (c.sf * 3.6);
    c.ms_work = c.v_work / 3.6; c.ms_turn = c.v_turn / 3.6; c.ms_head = c.v_head / 3.6; c.ms_rev = 2.5 / 3.6;
    c.shapes = nullptr; c.tmpl_u = nullptr; c.tmpl_c = nullptr; c.tmpl_u_dk = nullptr; c.field_junc = nullptr;
    c.tmpl_n = 0; c.tmpl_nc = 1;
    c.turn_kappa_last[0] = c.turn_kappa_last[1] = c.turn_len = c.turn_time = 0.0;
    c.turn_max_kappa[0] = c.turn_max_kappa[1] = c.turn_max_jump[0] = c.turn_max_jump[1] = 0.0;
    c.turn_jump[0] = c.turn_jump[1] = 0.0;
    return c;
}

// plain tile table of a set of paths (the staged pipeline and the standalone operators): tiles never straddle paths, hold at most
// TILE_POINTS points, a path is cut into near-equal tiles.  (The fused pipeline's tiler lives in fcpp_tiler.cpp.)
struct Tiling {
    std::vector<DevPath> paths;
    std::vector<DevTile> tiles;
    std::vector<int64_t> tile_first;
    void build(int64_t n_paths, const int64_t *offsets)
    {
        paths.resize((size_t)n_paths);
        tile_first.assign((size_t)n_paths + 1, 0);
        tiles.clear();
        for (int64_t p = 0; p < n_paths; ++p) {
            const int64_t n = offsets[p + 1] - offsets[p];
            paths[(size_t)p] = { offsets[p], n };
            tile_first[(size_t)p] = (int64_t)tiles.size();
            if (n <= 0) continue;
            const int64_t k = (n + TILE_POINTS - 1) / TILE_POINTS, base = n / k, rem = n % k;
            int64_t a = 0;
            for (int64_t i = 0; i < k; ++i) {
                const int64_t c = base + (i < rem ? 1 : 0);
                DevTile t;
                t.field = (int32_t)p; t.start = a; t.count = (int32_t)c; t.quiet = 0; t.stat_tile = 0; t.idx0 = 0; t.off0 = 0;
                tiles.push_back(t);
                a += c;
            }
        }
        tile_first[(size_t)n_paths] = (int64_t)tiles.size();
    }
};

struct DevTiling {
    DevBuf<DevPath> paths;
    DevBuf<DevTile> tiles;
    DevBuf<int64_t> tile_first;
    DevBuf<char> agg_f, agg_b;   // Agg = 2 doubles
    DevBuf<double> carry_f, carry_b;
    DevBuf<char> spine;          // scratch of the three-level spine (large batches)
    DevBuf<TilePartial> partial;
    DevBuf<unsigned long long> n_adj;
    int64_t n_tiles = 0, n_paths = 0;
    hipError_t upload(const Tiling &t, hipStream_t st)
    {
        n_tiles = (int64_t)t.tiles.size(); n_paths = (int64_t)t.paths.size();
        hipError_t e;
        if ((e = paths.upload(t.paths, st)) != hipSuccess) return e;
        if ((e = tiles.upload(t.tiles, st)) != hipSuccess) return e;
        if ((e = tile_first.upload(t.tile_first, st)) != hipSuccess) return e;
        if ((e = agg_f.alloc((size_t)n_tiles * 16)) != hipSuccess) return e;
        if ((e = agg_b.alloc((size_t)n_tiles * 16)) != hipSuccess) return e;
        if ((e = carry_f.alloc((size_t)n_tiles)) != hipSuccess) return e;
        if ((e = carry_b.alloc((size_t)n_tiles)) != hipSuccess) return e;
        if ((e = spine.alloc((size_t)spine_scratch_bytes(n_tiles))) != hipSuccess) return e;
        if ((e = partial.alloc((size_t)n_tiles)) != hipSuccess) return e;
        if ((e = n_adj.alloc((size_t)n_paths)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;   // the staging vectors die here
        return hipSuccess;
    }
};

// The turn templates of a batch (every field shares the vehicle and the sampling options, hence the shape and the sample count of
// its U-turns and corner turns): sampled once on the device (k_build_templates), copied back for the tiler's halo sizing and the
// closed-form test.  A context keeps the last set: a caller that creates batch after batch with one vehicle (the planner mirror does,
// one per plan_complete_coverage) pays for the two launches and the copy once.
struct TemplateSet {
    TurnTemplates tt;
    double clothoid_frac = 0.0;
    DevBuf<CacShape> shapes;                       // [0] 180-degree, [1] 90-degree clothoid-arc-clothoid unit shapes
    DevBuf<double2> tmpl_u, tmpl_c, tmpl_u_dk;     // sampled turn templates (fcpp_fused.hip), (segment length, curvature) per U-turn sample
    DevBuf<double2> tmpl_c_dk;                     // the same per corner-turn sample (the closed-form cut's chord table, fcpp_cutfn.h)
    std::vector<double2> h_tu, h_tc, h_dk, h_dkc;  // host copies
    double tc_lo[2] = { 0.0, 0.0 }, tc_hi[2] = { 0.0, 0.0 };      // the corner template's box (from h_tc; template_box())
    bool box_done = false;
    void template_box()                            // (needs the host copies: after the stream that fetched them has been drained)
    {
        if (box_done) return;
        for (int d = 0; d < 2; ++d) { tc_lo[d] = 0.0; tc_hi[d] = 0.0; }
        for (size_t k = 0; k < h_tc.size(); ++k) {
            tc_lo[0] = std::min(tc_lo[0], h_tc[k].x); tc_hi[0] = std::max(tc_hi[0], h_tc[k].x);
            tc_lo[1] = std::min(tc_lo[1], h_tc[k].y); tc_hi[1] = std::max(tc_hi[1], h_tc[k].y);
        }
        box_done = true;
    }
    bool same(const TurnTemplates &o, double frac) const { return memcmp(&tt, &o, sizeof tt) == 0 && clothoid_frac == frac; }
};
}  // namespace

namespace { struct PathTiling; }

struct fcpp_ctx {
    int device = 0;
    PathTiling *paths_cache = nullptr;     // tile table of the standalone operators' last path set (make_tiling)
    hipStream_t own = nullptr, stream = nullptr;
    // side stream of the fused pipeline: the ALU-bound kernels (wave tiles, general tiles) run beside the HBM-bound streaming
    // kernels of the same step; ev_fork / ev_join order the two streams inside a step
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // batch setup (fcpp_batch_create): the image of a batch's tables is built in pinned host memory that the context keeps (grow-only,
    // up to kStageMax; larger images go through a pageable buffer), and the last destroyed batch's device allocation is kept for the
    // next one (up to kSpareMax): a caller that plans batch after batch allocates nothing after the first
    void *stage = nullptr; size_t stage_cap = 0;
    // the stream of the last asynchronous copy out of `stage`: whoever writes that memory next drains it first (a caller that plans batch
    // after batch has drained it long before: a query; no event -- a record between two kernels holds the second back by 5 us)
    hipStream_t stage_stream = nullptr; bool stage_busy = false;
    hipError_t stage_wait()
    {
        if (!stage_busy) return hipSuccess;
        stage_busy = false;
        hipError_t e = hipStreamSynchronize(stage_stream);
        if (e != hipSuccess) { (void)hipGetLastError(); e = hipDeviceSynchronize(); }      // (that stream is gone)
        return e;
    }
    void *spare = nullptr; size_t spare_cap = 0;
    std::shared_ptr<TemplateSet> templates;         // the last batch's turn templates
    fcpp_setup_times last_setup = {};
    // device-side setup (fcpp_devplan.h): FCPP_SETUP_AUTO / _HOST / _DEVICE; its scratch (grow-only) and a small pinned block for the
    // totals that come back in the middle of it
    int setup_mode = FCPP_SETUP_AUTO;
    // (up to FOUR scratch allocations, one per stream that sets batches up: a caller that plans batch k + 1 on a second stream while batch k's step
    // still runs on the first -- the sustained rate of bench.py -- must not wait for that step because its fill pass shared the scratch)
    struct PlanSlot { void *p = nullptr; size_t cap = 0; hipStream_t stream = nullptr; bool pending = false; uint64_t tick = 0; };
    static constexpr int kPlanSlots = 4;
    PlanSlot plan_slots[kPlanSlots];
    uint64_t plan_tick = 0;
    int plan_cur = 0;                               // the slot of the setup in progress / of the last one
    void *verify_scratch = nullptr;                 // sliced reduction of the standalone operators' long paths (reduce_paths)
    size_t verify_scratch_cap = 0;
    int64_t *plan_totals_host = nullptr;            // pinned, PC_COLS + PF_COUNT values: the scans of the counting phase write them here
    unsigned long long *ga_mirror = nullptr;        // pinned, one word: (converged << 32) | generations of the running fcpp_ga_evolve (GaState::mirror)
    int64_t plan_gen = 0;                           // generation number of the last counting phase (PlanFlag, fcpp_devplan.h)
    // the stream the last device-side setup was enqueued on (its fill pass may still read the scratch): a setup on ANOTHER stream records
    // ev_plan there and waits for it -- lazily, when that other stream shows up: an event recorded between two kernels of the plan call
    // costs 5 us of device time between them (round 5: the three records of a plan call were 16 of its 158 us)
    hipEvent_t ev_plan = nullptr;
    hipEvent_t ev_chunk[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };      // a large batch's counting pass in chunks beside its planner (launch_devplan_count)
    // the output arena (fcpp_ctx_reserve_outputs): ONE allocation of 4 x pitch + lane bytes; array k of every batch's outputs lies in lane k
    // (lanes `pitch` apart), placed first-fit among the live allocations of the lane -- all five arrays of an allocation at the same offset
    void *arena = nullptr; size_t arena_pitch = 0, arena_lane = 0;
    struct ArenaBlock { size_t off, len; hipStream_t last = nullptr; bool used = false; };      // last: the stream of the last fcpp_batch_run that wrote the block
    std::vector<ArenaBlock> arena_live;             // sorted by off
    std::vector<hipEvent_t> ev_pool;                // setup events of destroyed batches (fcpp_batch::ev_setup), reused: no event is created per plan call
};

// device pointers of a batch's tables: all inside ONE allocation laid out by the tiler (fcpp_tiler.h: ImageLayout)
struct FusedTables {
    DevField *fields = nullptr; DevPrim *prims = nullptr; DevTile *tiles = nullptr; DevWaveTile *wave_tiles = nullptr;
    int32_t *general_ids = nullptr; DevTile *chunks = nullptr, *span_chunks = nullptr;
    int32_t *stat_ids = nullptr; int64_t *stat_first = nullptr, *stat_run = nullptr; int32_t *red_paths = nullptr;
    DevFieldWork *field_work = nullptr; DevFieldPack *field_packs = nullptr; int32_t *open_wave_ids = nullptr;
    int64_t *obs_off = nullptr; double *obs_x = nullptr, *obs_y = nullptr, *obs_bbox = nullptr;
    double *seg = nullptr; int32_t *seg_mask = nullptr;
    TilePartial *partial = nullptr; char *red_scratch = nullptr; double2 *field_junc = nullptr;
    TilePartial *work_totals = nullptr;
};

struct fcpp_batch {
    fcpp_ctx *ctx = nullptr;
    fcpp_vehicle veh;
    fcpp_options opt;
    int64_t n_fields = 0;
    HostPlan hp;               // info + field descriptors (the blocks' primitive lists are dropped once the image is built)
    DevConst cst;
    void *slab = nullptr; size_t slab_bytes = 0;
    ImageLayout lay;           // counts and offsets of the tables in the slab
    FusedTables t;             // fused pipeline (mode 1): quiet runs, wave tiles, general tiles, reduction lists
    DevTiling til0;            // staged pipeline (mode 0): plain near-equal tiles, built on its first run
    bool til0_built = false;
    std::shared_ptr<TemplateSet> templates;
    // optional per-stage HIP-event timing (fcpp_batch_set_profiling)
    int profiling = 0;           // 0: off; k > 0: every k-th run carries the per-kernel events
    int64_t run_counter = 0;
    std::vector<hipEvent_t> events;   // kProfRuns x kStages x (start, stop)
    std::vector<unsigned char> ev_set; // kProfRuns x kStages: the stage launched a kernel in that run
    int prof_runs = 0;
    int last_mode = 0;
    bool two_streams = true;     // ALU-bound kernels of a step on the context's side stream (FCPP_ONE_STREAM=1 in the environment: off)
    int two_stream_max = 512;    // ... when there are at most this many general tiles (FCPP_TWO_STREAM_MAX, read at batch creation)
    int sparse_beside_max = 0;   // ... or at most this many wave tiles (FCPP_SPARSE_BESIDE_MAX; 0 = never: measured, see fcpp_batch_run)
    fcpp_setup_times setup = {};
    const fcpp_field_info *info_dev = nullptr;     // device-side setup: the records in the slab; hp.info is filled from them on demand
    // every stream this batch's kernels were enqueued on (callers re-bind the context's stream between calls: engine.py binds torch's
    // current stream): fcpp_batch_destroy drains them all before the tables go back to the context as the next batch's allocation
    std::vector<hipStream_t> used_streams;
    void note_stream(hipStream_t s) { if (std::find(used_streams.begin(), used_streams.end(), s) == used_streams.end()) used_streams.push_back(s); }
    // the device-side setup returns with its last kernels still in the stream it was enqueued on: whoever reads the tables on ANOTHER stream
    // (a caller that re-binds the context's stream between create and run) waits for this event first
    // (the event is recorded on the setup's stream when such a stream first shows up -- everything enqueued there so far, the setup included,
    // lies before it -- not by the setup itself: see fcpp_ctx::ev_plan)
    hipStream_t setup_stream = nullptr;
    bool setup_pending = false;
    hipEvent_t ev_setup = nullptr;
    std::vector<hipStream_t> setup_seen;           // streams already ordered behind the setup
    hipError_t wait_setup(hipStream_t s)
    {
        if (!setup_pending || s == setup_stream || std::find(setup_seen.begin(), setup_seen.end(), s) != setup_seen.end()) return hipSuccess;
        if (!ev_setup) {
            if (!ctx->ev_pool.empty()) { ev_setup = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); }
            else { const hipError_t e = hipEventCreateWithFlags(&ev_setup, hipEventDisableTiming); if (e != hipSuccess) return e; }
        }
        hipError_t e = hipEventRecord(ev_setup, setup_stream);
        if (e != hipSuccess) {          // (the setup's stream is gone: whatever ran there has been drained or is drained now)
            (void)hipGetLastError();
            e = hipDeviceSynchronize();
            if (e == hipSuccess) setup_pending = false;
            return e;
        }
        e = hipStreamWaitEvent(s, ev_setup, 0);
        if (e == hipSuccess) setup_seen.push_back(s);
        return e;
    }
    ~fcpp_batch() { for (hipEvent_t e : events) (void)hipEventDestroy(e); if (ev_setup) (void)hipEventDestroy(ev_setup); }
};

namespace {
constexpr int kStages = 7;
constexpr int kProfRuns = 256;
const char *const kStageNames[2][kStages] = {
    { "k_generate", "k_curv_clamp", "k_scan_tiles", "k_scan_spine", "k_scan_apply", "k_validate", "k_reduce_stats" },
    { "k_plan_quiet_spans", "k_plan_quiet", "k_plan_sparse", "k_plan_fused", "k_reduce_stats", "k_plan_sparse_fields", "" } };
const int kStageCount[2] = { 7, 6 };
}

// Are the U-turns of this batch closed form?  A turn is a translate / mirror of the template t[0..nu); its neighbours are the
// swath line it leaves (whose last point is the turn's first: a skipped step, nothing propagates across it) and the next
// line, reached by a jump J from the turn's last sample.  If every sample keeps the nominal turn speed under the curvature
// clamp, the next line's first point keeps the work speed, and 2a|J| is too long for the sweeps to bind across the jump, then
// turn points are template + translation, curvature and segment lengths are the shape's own (dk), speeds are nominal.
// Fills the per-batch constants the quiet kernel and the run statistics need.  Index 0 / 1: passes ascending / descending in y.
static bool closed_form_turns(const fcpp_vehicle &veh, const TurnTemplates &tt, const std::vector<double2> &t,
                              const std::vector<double2> &dk, DevConst &c)
{
    const int nu = tt.nu;
    const double W = veh.working_width, R = veh.min_turn_radius;
    // the turn's first sample must be the line's last point: template offset 0 (clothoid: x = (max_x - R) + t.x) or R (arcs: x = max_x - t.x)
    if (nu < 3 || fabs(t[0].x - (tt.turn_model == FCPP_TURN_ARC ? R : 0.0)) > 1e-9 || fabs(t[0].y) > 1e-9) return false;
    const double q_t = c.v_turn * c.inv_sf36, q_w = c.v_work * c.inv_sf36, lim = c.a_lat * 0.999;
    double len = 0.0, maxk = 0.0, maxj = 0.0;
    for (int k = 1; k < nu; ++k) len += dk[(size_t)k].x;
    for (int k = 1; k + 1 < nu; ++k) {
        maxk = std::max(maxk, dk[(size_t)k].y);
        maxj = std::max(maxj, fabs(dk[(size_t)k].y - dk[(size_t)k - 1].y));
    }
    if (maxk * q_t * q_t >= lim) return false;
    // the turn's last chord in the frame of a right turn (world x grows to the right)
    const bool arc = tt.turn_model == FCPP_TURN_ARC;
    const double sx = arc ? -1.0 : 1.0;                    // arcs: px = max_x - t.x; clothoid: px = (max_x - R) + t.x
    const double c1x = sx * (t[(size_t)nu - 1].x - t[(size_t)nu - 2].x), c1y = t[(size_t)nu - 1].y - t[(size_t)nu - 2].y;
    const double d1 = sqrt(c1x * c1x + c1y * c1y);
    const double u_t = c.ms_turn * c.ms_turn, u_w = c.ms_work * c.ms_work;
    for (int v = 0; v < 2; ++v) {
        // jump to the first point of the next line: x back to max_x - R, y to the next pass (+W, or -W in top-down order)
        const double jx = arc ? (-R + t[(size_t)nu - 1].x) : -t[(size_t)nu - 1].x;
        const double jy = (v == 0 ? W : -W) - t[(size_t)nu - 1].y;
        const double dj = sqrt(jx * jx + jy * jy);
        if (!(dj >= 1e-3) || d1 < 1e-6) return false;
        if (u_t + 2 * c.a_lon * dj < u_w * (1.0 + 1e-9)) return false;          // the sweeps must not bind across the jump
        const double k_last = fabs(2 * atan2_fd(c1x * jy - c1y * jx, c1x * jx + c1y * jy) / (d1 + dj));
        // first point of the next line: chords J and the line heading (-x after a right turn); its curvature is largest for step -> 0
        const double k_first_max = fabs(2 * atan2_fd(jx * 0.0 - jy * -1.0, jx * -1.0 + jy * 0.0) / dj);
        if (k_last * q_t * q_t >= lim || k_first_max * q_w * q_w >= lim) return false;
        c.turn_kappa_last[v] = k_last;
        c.turn_jump[v] = dj;
        c.turn_max_kappa[v] = std::max(maxk, k_last);
        c.turn_max_jump[v] = std::max(std::max(maxj, fabs(dk[(size_t)nu - 2].y - (nu >= 3 ? dk[(size_t)nu - 3].y : 0.0))), fabs(k_last - dk[(size_t)nu - 2].y));
    }
    c.turn_len = len;
    c.turn_time = len / std::max(c.ms_turn, 0.1);
    return true;
}

static void free_paths_cache(fcpp_ctx *c);

// the constants of the closed-form cut (fcpp_cutfn.h) for a batch: the templates and their chord tables as the HOST copies or as the device
// arrays (the same values), the rest from the batch's constants (closed_form_turns has run)
static CutConsts make_cut_consts(TemplateSet &ts, bool device, bool turn_quiet, int wave_factor, double two_a, double u_cap, double c_line,
                                 double fence_margin, const DevConst &cst)
{
    CutConsts cc;
    memset(&cc, 0, sizeof cc);
    ts.template_box();
    if (device) {
        cc.tu = reinterpret_cast<const Pt2 *>(ts.tmpl_u.p); cc.tc = reinterpret_cast<const Pt2 *>(ts.tmpl_c.p);
        cc.dk_u = reinterpret_cast<const Pt2 *>(ts.tmpl_u_dk.p); cc.dk_c = reinterpret_cast<const Pt2 *>(ts.tmpl_c_dk.p);
    } else {
        cc.tu = reinterpret_cast<const Pt2 *>(ts.h_tu.data()); cc.tc = reinterpret_cast<const Pt2 *>(ts.h_tc.data());
        cc.dk_u = reinterpret_cast<const Pt2 *>(ts.h_dk.data()); cc.dk_c = reinterpret_cast<const Pt2 *>(ts.h_dkc.data());
    }
    cc.nu = ts.tt.nu; cc.nc = ts.tt.nc; cc.turn_quiet = turn_quiet ? 1 : 0; cc.wave_factor = wave_factor;
    cc.two_a = two_a; cc.u_cap = u_cap; cc.c_line = c_line; cc.fence_margin = fence_margin;
    cc.jump[0] = cst.turn_jump[0]; cc.jump[1] = cst.turn_jump[1];
    for (int d = 0; d < 2; ++d) { cc.tc_lo[d] = ts.tc_lo[d]; cc.tc_hi[d] = ts.tc_hi[d]; }
    // the shortest chord of either template (host copies of the device's chord tables: the same values whichever side cuts)
    cc.u_step_min = cc.c_step_min = HUGE_VAL;
    for (size_t k = 1; k < ts.h_dk.size(); ++k) cc.u_step_min = std::min(cc.u_step_min, ts.h_dk[k].x);
    for (size_t k = 1; k < ts.h_dkc.size(); ++k) cc.c_step_min = std::min(cc.c_step_min, ts.h_dkc[k].x);
    if (ts.h_dk.size() < 2) cc.u_step_min = 0.0;
    if (ts.h_dkc.size() < 2) cc.c_step_min = 0.0;
    return cc;
}

extern "C" {

const char *fcpp_last_error(void) { return g_err.c_str(); }
int fcpp_abi_version(void) { return FCPP_ABI_VERSION; }

void fcpp_vehicle_default(fcpp_vehicle *v)
{
    v->working_width = 3.2; v->min_turn_radius = 8.0; v->max_work_speed_kmh = 9.0;
    v->max_headland_speed_kmh = 15.0; v->headland_turn_speed_kmh = 4.0; v->max_lateral_accel = 2.0;
    v->max_longitudinal_accel = 1.5; v->safety_factor = 0.85;
}

void fcpp_options_default(fcpp_options *o)
{
    o->turn_model = FCPP_TURN_ARC; o->clothoid_fit = 1; o->sample_spacing = 0.0; o->clothoid_frac = 0.5;
    o->geofence_tol = 1e-6; o->obstacle_mode = FCPP_OBSTACLES_FLAG; o->ring_order = FCPP_RING_AS_VERTICES;
}

int fcpp_ctx_create(int device_id, fcpp_ctx **out)
{
    if (!out) return fail(FCPP_EINVAL, "ctx out pointer is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(FCPP_EHIP, "no HIP device available: libfcpp has no CPU fallback");
    if (device_id < 0 || device_id >= n) return fail(FCPP_EINVAL, "device_id out of range");
    HIPCHK(hipSetDevice(device_id));
    fcpp_ctx *c = new (std::nothrow) fcpp_ctx();
    if (!c) return fail(FCPP_ENOMEM, "out of host memory");
    c->device = device_id;
    e = hipStreamCreateWithFlags(&c->own, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    if (e != hipSuccess) {
        if (c->own) (void)hipStreamDestroy(c->own);
        if (c->side) (void)hipStreamDestroy(c->side);
        if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
        delete c;
        return fail(FCPP_EHIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    c->stream = c->own;
    if (const char *e = getenv("FCPP_SETUP")) c->setup_mode = !strcmp(e, "host") ? FCPP_SETUP_HOST : (!strcmp(e, "device") ? FCPP_SETUP_DEVICE : FCPP_SETUP_AUTO);
    *out = c;
    return FCPP_OK;
}

int fcpp_ctx_set_setup(fcpp_ctx *c, int mode)
{
    if (!c || mode < FCPP_SETUP_AUTO || mode > FCPP_SETUP_DEVICE) return fail(FCPP_EINVAL, "bad arguments");
    c->setup_mode = mode;
    return FCPP_OK;
}

int fcpp_ctx_destroy(fcpp_ctx *c)
{
    if (!c) return FCPP_OK;
    (void)hipSetDevice(c->device);
    if (c->own) { (void)hipStreamSynchronize(c->own); (void)hipStreamDestroy(c->own); }
    if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    free_paths_cache(c);
    (void)c->stage_wait();
    if (c->stage) (void)hipHostFree(c->stage);
    if (c->spare) (void)hipFree(c->spare);
    for (auto &sl : c->plan_slots) if (sl.p) { (void)hipDeviceSynchronize(); (void)hipFree(sl.p); sl.p = nullptr; }
    if (c->arena) { (void)hipDeviceSynchronize(); (void)hipFree(c->arena); }
    if (c->ev_plan) (void)hipEventDestroy(c->ev_plan);
    for (hipEvent_t e : c->ev_chunk) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->plan_totals_host) (void)hipHostFree(c->plan_totals_host);
    if (c->ga_mirror) (void)hipHostFree(c->ga_mirror);
    if (c->verify_scratch) (void)hipFree(c->verify_scratch);
    c->templates.reset();
    delete c;
    return FCPP_OK;
}

int fcpp_ctx_set_stream(fcpp_ctx *c, void *s)
{
    if (!c) return fail(FCPP_EINVAL, "ctx is NULL");
    c->stream = (hipStream_t)s;   // NULL = HIP's default stream
    return FCPP_OK;
}

int fcpp_ctx_synchronize(fcpp_ctx *c)
{
    if (!c) return fail(FCPP_EINVAL, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return FCPP_OK;
}

int fcpp_malloc(fcpp_ctx *c, int64_t bytes, void **p)
{
    if (!c || !p || bytes < 0) return fail(FCPP_EINVAL, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    *p = nullptr;
    if (bytes == 0) return FCPP_OK;
    HIPCHK(hipMalloc(p, (size_t)bytes));
    return FCPP_OK;
}

int fcpp_free(fcpp_ctx *c, void *p)
{
    if (!c) return fail(FCPP_EINVAL, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    if (p) HIPCHK(hipFree(p));
    return FCPP_OK;
}

// ---- output arrays with the placement rule of DESIGN.md section 2 (HISTORY.md "Where the five arrays lie").  Measured on MI355X: the streaming kernels
// write x, y, kappa, v and flagseg side by side, and five write streams that lie within a few GiB of each other in (physical) device
// memory run at 4.6 TB/s, the same streams >= 12-24 GiB apart at 6.3-6.6 TB/s (tools/placement_pitch.py: the class follows the pitch
// and nothing else).  The context's ARENA is the remedy that several live batches can share: one allocation made once
// (fcpp_ctx_reserve_outputs: five lanes `pitch` apart), array k of every fcpp_outputs_alloc in lane k.
int fcpp_ctx_reserve_outputs(fcpp_ctx *c, int64_t lane_bytes, int64_t pitch_bytes)
{
    if (!c || lane_bytes < 0 || pitch_bytes < 0) return fail(FCPP_EINVAL, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    if (c->arena) {
        if (!c->arena_live.empty()) return fail(FCPP_EINVAL, "the output arena has live allocations");
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipFree(c->arena));
        c->arena = nullptr; c->arena_pitch = c->arena_lane = 0;
    }
    size_t pitch = pitch_bytes > 0 ? (size_t)pitch_bytes : (size_t)FCPP_OUTPUT_PITCH;
    size_t lane = lane_bytes > 0 ? (size_t)lane_bytes : pitch;
    pitch = (pitch + 4095) / 4096 * 4096; lane = (lane + 4095) / 4096 * 4096;
    if (lane > pitch) pitch = lane;                      // lanes must not overlap
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    const size_t reserve = (size_t)8 << 30;
    if (free_b < 4 * pitch + lane + reserve) return fail(FCPP_ENOMEM, "not enough free device memory for the output arena (4 x pitch + lane + 8 GiB)");
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, 4 * pitch + lane);
    if (e != hipSuccess) return fail(FCPP_ENOMEM, std::string("output arena: ") + hipGetErrorString(e));
    c->arena = p; c->arena_pitch = pitch; c->arena_lane = lane;
    c->arena_live.clear();
    return FCPP_OK;
}

int fcpp_ctx_outputs_info(const fcpp_ctx *c, int64_t *lane_bytes, int64_t *pitch_bytes, int64_t *live_bytes)
{
    if (!c) return fail(FCPP_EINVAL, "ctx is NULL");
    if (lane_bytes) *lane_bytes = (int64_t)c->arena_lane;
    if (pitch_bytes) *pitch_bytes = (int64_t)c->arena_pitch;
    if (live_bytes) { size_t s = 0; for (const auto &blk : c->arena_live) s += blk.len; *live_bytes = (int64_t)s; }
    return FCPP_OK;
}

int fcpp_outputs_alloc(fcpp_ctx *c, int64_t n_points, int64_t pitch_bytes, double **x, double **y, double **kappa, double **v, uint32_t **flagseg)
{
    if (!c || n_points < 0 || !x || !y || !kappa || !v || !flagseg) return fail(FCPP_EINVAL, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    const size_t S = (((size_t)n_points * 8 + 4095) / 4096) * 4096;
    char *b = nullptr;
    size_t P = 0;
    if (pitch_bytes == 0 && c->arena && S <= c->arena_lane) {
        // first fit in the lanes (the same offset in all five): the lowest gap between live allocations that holds the array
        const size_t need = std::max<size_t>(S, 4096);
        size_t off = 0;
        size_t at = 0;
        for (; at < c->arena_live.size(); ++at) {
            if (c->arena_live[at].off - off >= need) break;
            off = c->arena_live[at].off + c->arena_live[at].len;
        }
        if (off + need <= c->arena_lane) {
            c->arena_live.insert(c->arena_live.begin() + (long)at, fcpp_ctx::ArenaBlock{ off, need, nullptr, false });
            b = static_cast<char *>(c->arena) + off;
            P = c->arena_pitch;
        }
    }
    if (!b) {
        // without an arena (or beyond it): ONE allocation of its own, the arrays pitch_bytes apart -- back to back when no pitch is asked
        // for (the slow placement class for arrays of GBs, but nothing is taken from the device that the arrays do not need)
        P = pitch_bytes > 0 ? (size_t)pitch_bytes : S;
        if (P < S) P = S;
        P = (P + 4095) / 4096 * 4096;
        void *slab = nullptr;
        hipError_t e = hipMalloc(&slab, std::max<size_t>(4 * P + S, 4096));
        if (e != hipSuccess) return fail(FCPP_ENOMEM, std::string("output arrays: ") + hipGetErrorString(e));
        b = static_cast<char *>(slab);
    }
    *x = reinterpret_cast<double *>(b); *y = reinterpret_cast<double *>(b + P); *kappa = reinterpret_cast<double *>(b + 2 * P);
    *v = reinterpret_cast<double *>(b + 3 * P); *flagseg = reinterpret_cast<uint32_t *>(b + 4 * P);
    return FCPP_OK;
}

int fcpp_outputs_free(fcpp_ctx *c, double *x)
{
    if (!c) return fail(FCPP_EINVAL, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    if (!x) return FCPP_OK;
    char *p = reinterpret_cast<char *>(x);
    if (c->arena && p >= static_cast<char *>(c->arena) && p < static_cast<char *>(c->arena) + c->arena_lane) {
        const size_t off = (size_t)(p - static_cast<char *>(c->arena));
        for (size_t k = 0; k < c->arena_live.size(); ++k)
            if (c->arena_live[k].off == off) {
                // (steps that write the block may still be queued on the stream they ran on, and first fit may hand the same offset to a
                // batch on another stream at once: the block goes back only when that stream has finished with it -- a query when it is idle)
                const fcpp_ctx::ArenaBlock blk = c->arena_live[k];
                if (blk.used && hipStreamQuery(blk.last) != hipSuccess) { (void)hipGetLastError(); HIPCHK(hipStreamSynchronize(blk.last)); }
                c->arena_live.erase(c->arena_live.begin() + (long)k);
                return FCPP_OK;
            }
        return fail(FCPP_EINVAL, "not an allocation of the output arena");
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipFree(x));
    return FCPP_OK;
}

int fcpp_memcpy_h2d(fcpp_ctx *c, void *dst, const void *src, int64_t bytes)
{
    if (!c || bytes < 0) return fail(FCPP_EINVAL, "bad arguments");
    if (bytes == 0) return FCPP_OK;
    HIPCHK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return FCPP_OK;
}

int fcpp_memcpy_d2h(fcpp_ctx *c, void *dst, const void *src, int64_t bytes)
{
    if (!c || bytes < 0) return fail(FCPP_EINVAL, "bad arguments");
    if (bytes == 0) return FCPP_OK;
    HIPCHK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return FCPP_OK;
}

namespace {
// Field records that live in DEVICE memory (round 5: a table kept on the GPU -- engine.FieldTable.to_device(), a caller whose fields are
// made there -- is read by the device-side setup where it lies, like pinned records, without crossing PCIe): the HOST paths (host plan,
// fcpp_plan_count, a batch the device planner hands back) read a copy.  -> `host` = fields, or copy.data()
int host_fields(const fcpp_field *fields, int64_t n_fields, std::vector<fcpp_field> &copy, const fcpp_field *&host)
{
    host = fields;
    if (n_fields <= 0 || !fields) return FCPP_OK;
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof at);
    if (hipPointerGetAttributes(&at, fields) != hipSuccess) { (void)hipGetLastError(); return FCPP_OK; }      // (pageable memory, or no GPU at all)
    if (at.type != hipMemoryTypeDevice) return FCPP_OK;
    try { copy.resize((size_t)n_fields); } catch (const std::bad_alloc &) { return fail(FCPP_ENOMEM, "out of host memory"); }
    HIPCHK(hipMemcpy(copy.data(), fields, (size_t)n_fields * sizeof(fcpp_field), hipMemcpyDeviceToHost));
    host = copy.data();
    return FCPP_OK;
}
}  // namespace

int fcpp_plan_count(const fcpp_vehicle *veh, const fcpp_options *opt, int64_t n_fields, const fcpp_field *fields,
                    const fcpp_polys *obstacles, fcpp_field_info *info_out)
{
    if (!veh || !opt || n_fields < 0 || (n_fields > 0 && (!fields || !info_out)))
        return fail(FCPP_EINVAL, "bad arguments");
    HostPlan hp;
    std::string err;
    std::vector<fcpp_field> fcopy;
    int rc = host_fields(fields, n_fields, fcopy, fields);
    if (rc != FCPP_OK) return rc;
    rc = build_host_plan(*veh, *opt, n_fields, fields, obstacles, false, hp, err);
    if (rc != FCPP_OK) return fail(rc, err);
    if (n_fields) memcpy(info_out, hp.info.data(), (size_t)n_fields * sizeof(fcpp_field_info));
    return FCPP_OK;
}

namespace {
int plan_scratch(fcpp_ctx *c, int64_t n_fields, int max_prims, hipStream_t st, DevPlanScratch &s, std::string &err);
int device_fields(const fcpp_field *fields, int64_t n_fields, hipStream_t st, const DevPlanScratch &s, const fcpp_field *&dev, std::string &err);
}

// Sizing for a sharded job: points per field.  On the device for the batches the device planner takes (k_plan_fields without primitives,
// the counts copied back), else on the host's cores.
int fcpp_plan_points(fcpp_ctx *c, const fcpp_vehicle *veh, const fcpp_options *opt, int64_t n_fields, const fcpp_field *fields,
                     const fcpp_polys *obstacles, int64_t *points_out)
{
    if (!c || !veh || !opt || n_fields < 0 || (n_fields > 0 && (!fields || !points_out))) return fail(FCPP_EINVAL, "bad arguments");
    if (n_fields == 0) return FCPP_OK;
    std::string err;
    PlanConsts pc;
    TurnTemplates tt;
    int rc = plan_prepare(*veh, *opt, pc, tt, err);
    if (rc != FCPP_OK) return fail(rc, err);
    const bool on_device = c->setup_mode != FCPP_SETUP_HOST && opt->sample_spacing == 0.0 && opt->obstacle_mode == FCPP_OBSTACLES_FLAG &&
                           pc.max_prims <= DEVPLAN_PRIMS_CAP && !tune_enabled();
    if (!on_device) {
        HostPlan hp;
        std::vector<fcpp_field> fcopy;
        if ((rc = host_fields(fields, n_fields, fcopy, fields)) != FCPP_OK) return rc;
        rc = build_host_plan(*veh, *opt, n_fields, fields, obstacles, false, hp, err);
        if (rc != FCPP_OK) return fail(rc, err);
        for (int64_t i = 0; i < n_fields; ++i) points_out[i] = hp.info[(size_t)i].n_main + hp.info[(size_t)i].n_head;
        return FCPP_OK;
    }
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    DevPlanScratch s;
    if ((rc = plan_scratch(c, n_fields, pc.max_prims, st, s, err)) != FCPP_OK) return fail(rc, err);
    const fcpp_field *dev_fields = nullptr;
    if ((rc = device_fields(fields, n_fields, st, s, dev_fields, err)) != FCPP_OK) return fail(rc, err);
    {   // (whatever was launched may still be reading the caller's pinned records: drained before an error goes back)
        const int lrc = launch_devplan_points(st, n_fields, pc, s, dev_fields);
        if (lrc) { (void)hipStreamSynchronize(st); return fail(FCPP_EHIP, std::string("launch_devplan_points: ") + hipGetErrorString((hipError_t)lrc)); }
    }
    HIPCHK(hipMemcpyAsync(points_out, s.counts + (int64_t)PC_POINTS * n_fields, (size_t)n_fields * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return FCPP_OK;
}

// the context's turn templates for (tt, clothoid_frac): the kept set, or a new one sampled on the device (asynchronous: the caller
// synchronises the stream before it reads the host copies; `fresh` tells it to)
static int get_templates(fcpp_ctx *c, const TurnTemplates &tt, const fcpp_options &opt, hipStream_t st, std::shared_ptr<TemplateSet> &out, bool &fresh)
{
    fresh = false;
    if (c->templates && c->templates->same(tt, opt.clothoid_frac)) { out = c->templates; return FCPP_OK; }
    std::shared_ptr<TemplateSet> ts(new (std::nothrow) TemplateSet());
    if (!ts) return fail(FCPP_ENOMEM, "out of host memory");
    ts->tt = tt; ts->clothoid_frac = opt.clothoid_frac;
    const int nu = tt.nu, nc = tt.nc;
    std::vector<CacShape> shp = { make_cac_shape(kPi, opt.clothoid_frac), make_cac_shape(kHalfPi, opt.clothoid_frac) };
    HIPCHK(ts->shapes.upload(shp, st));
    HIPCHK(hipStreamSynchronize(st));             // (shp dies with this scope)
    HIPCHK(ts->tmpl_u.alloc((size_t)nu));
    HIPCHK(ts->tmpl_c.alloc((size_t)nc));
    HIPCHK(ts->tmpl_u_dk.alloc((size_t)nu));
    HIPCHK(ts->tmpl_c_dk.alloc((size_t)nc));
    LAUNCHCHK(launch_build_templates(st, tt, ts->shapes.p, ts->tmpl_u.p, ts->tmpl_c.p));
    LAUNCHCHK(launch_build_template_metrics(st, nu, ts->tmpl_u.p, ts->tmpl_u_dk.p));
    LAUNCHCHK(launch_build_template_metrics(st, nc, ts->tmpl_c.p, ts->tmpl_c_dk.p));
    ts->h_tu.resize((size_t)nu); ts->h_tc.resize((size_t)nc); ts->h_dk.resize((size_t)nu); ts->h_dkc.resize((size_t)nc);
    if (nu > 0) {
        HIPCHK(hipMemcpyAsync(ts->h_tu.data(), ts->tmpl_u.p, (size_t)nu * sizeof(double2), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(ts->h_dk.data(), ts->tmpl_u_dk.p, (size_t)nu * sizeof(double2), hipMemcpyDeviceToHost, st));
    }
    if (nc > 0) {
        HIPCHK(hipMemcpyAsync(ts->h_tc.data(), ts->tmpl_c.p, (size_t)nc * sizeof(double2), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(ts->h_dkc.data(), ts->tmpl_c_dk.p, (size_t)nc * sizeof(double2), hipMemcpyDeviceToHost, st));
    }
    out = ts;
    fresh = true;
    return FCPP_OK;
}

namespace {
constexpr size_t kStageMax = (size_t)2 << 30;      // pinned setup memory a context keeps at most
constexpr size_t kSpareMax = (size_t)1 << 30;      // device allocation of a destroyed batch kept for the next one at most
double ms_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
}

namespace {
thread_local double g_trace_totals_ms = 0.0;      // (FCPP_TRACE_PLAN: when the last device setup saw its totals, ms since try_device_setup began)
constexpr int kNotOnDevice = 1;      // try_device_setup: this batch is the host's (reason in err)

// pointers of the fused pipeline's tables inside the slab
void bind_tables(fcpp_batch *b)
{
    unsigned char *d = static_cast<unsigned char *>(b->slab);
    const ImageLayout &lay = b->lay;
    FusedTables &t = b->t;
    t.fields = reinterpret_cast<DevField *>(d + lay.fields); t.prims = reinterpret_cast<DevPrim *>(d + lay.prims);
    t.tiles = reinterpret_cast<DevTile *>(d + lay.tiles); t.wave_tiles = reinterpret_cast<DevWaveTile *>(d + lay.wtiles);
    t.general_ids = reinterpret_cast<int32_t *>(d + lay.general_ids);
    t.chunks = reinterpret_cast<DevTile *>(d + lay.chunks); t.span_chunks = reinterpret_cast<DevTile *>(d + lay.span_chunks);
    t.stat_ids = reinterpret_cast<int32_t *>(d + lay.stat_ids); t.stat_first = reinterpret_cast<int64_t *>(d + lay.stat_first);
    t.stat_run = reinterpret_cast<int64_t *>(d + lay.stat_run); t.red_paths = reinterpret_cast<int32_t *>(d + lay.red_paths);
    t.field_work = reinterpret_cast<DevFieldWork *>(d + lay.field_work); t.field_packs = reinterpret_cast<DevFieldPack *>(d + lay.field_packs);
    t.open_wave_ids = reinterpret_cast<int32_t *>(d + lay.open_wave_ids);
    if (lay.n_polys > 0) {
        t.obs_off = reinterpret_cast<int64_t *>(d + lay.obs_off); t.obs_x = reinterpret_cast<double *>(d + lay.obs_x);
        t.obs_y = reinterpret_cast<double *>(d + lay.obs_y); t.obs_bbox = reinterpret_cast<double *>(d + lay.obs_bbox);
    }
    t.seg = reinterpret_cast<double *>(d + lay.seg); t.seg_mask = reinterpret_cast<int32_t *>(d + lay.seg_mask);
    t.partial = reinterpret_cast<TilePartial *>(d + lay.partial); t.red_scratch = d ? reinterpret_cast<char *>(d + lay.red_scratch) : nullptr;
    t.field_junc = reinterpret_cast<double2 *>(d + lay.field_junc);
    t.work_totals = reinterpret_cast<TilePartial *>(d + lay.work_totals);
}

// the batch's device allocation: the context's spare if it is large enough
int take_slab(fcpp_ctx *c, fcpp_batch *b, std::string &err)
{
    const ImageLayout &lay = b->lay;
    if (c->spare && c->spare_cap >= lay.total_bytes) { b->slab = c->spare; b->slab_bytes = c->spare_cap; c->spare = nullptr; c->spare_cap = 0; return FCPP_OK; }
    hipError_t e = hipMalloc(&b->slab, std::max<size_t>(lay.total_bytes, 256));
    if (e != hipSuccess) { b->slab = nullptr; err = std::string("batch tables: ") + hipGetErrorString(e); return FCPP_ENOMEM; }
    b->slab_bytes = std::max<size_t>(lay.total_bytes, 256);
    return FCPP_OK;
}

#define DEVCHK(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return FCPP_EHIP; } \
    } while (0)

// the context's planner scratch for n fields (grow-only), ordered behind the last fill pass that read it
int plan_scratch(fcpp_ctx *c, int64_t n_fields, int max_prims, hipStream_t st, DevPlanScratch &s, std::string &err)
{
    // the slot: this stream's, else an unused one, else the one used longest ago -- ordered behind whatever another stream left in it
    int k = -1;
    for (int i = 0; i < fcpp_ctx::kPlanSlots; ++i) if (c->plan_slots[i].p && c->plan_slots[i].stream == st) k = i;
    if (k < 0) for (int i = 0; i < fcpp_ctx::kPlanSlots; ++i) if (!c->plan_slots[i].p) { k = i; break; }
    if (k < 0) { k = 0; for (int i = 1; i < fcpp_ctx::kPlanSlots; ++i) if (c->plan_slots[i].tick < c->plan_slots[k].tick) k = i; }
    fcpp_ctx::PlanSlot &sl = c->plan_slots[k];
    if (sl.pending && sl.stream != st) {       // (that batch's fill pass, on another stream, may still read the scratch)
        if (!c->ev_plan) DEVCHK(hipEventCreateWithFlags(&c->ev_plan, hipEventDisableTiming));
        if (hipEventRecord(c->ev_plan, sl.stream) == hipSuccess) DEVCHK(hipStreamWaitEvent(st, c->ev_plan, 0));
        else { (void)hipGetLastError(); DEVCHK(hipDeviceSynchronize()); }        // (that stream is gone)
        sl.pending = false;
    }
    sl.stream = st; sl.tick = ++c->plan_tick;
    c->plan_cur = k;
    DevPlanScratch off;
    const size_t need = devplan_scratch_layout(n_fields, max_prims, &off);
    if (sl.cap < need) {
        if (sl.p) { DEVCHK(hipDeviceSynchronize()); (void)hipFree(sl.p); sl.p = nullptr; sl.cap = 0; }
        const size_t want = need + need / 4;
        if (hipMalloc(&sl.p, want) != hipSuccess) { (void)hipGetLastError(); sl.p = nullptr; err = "out of device memory for the planner's scratch"; return FCPP_ENOMEM; }
        sl.cap = want;
        // (the flags live at the start of the allocation and are never cleared again: they hold generation numbers)
        DEVCHK(hipMemsetAsync(sl.p, 0, PLAN_TOTALS * sizeof(int64_t), st));
    }
    if (!c->plan_totals_host)
        { DEVCHK(hipHostMalloc((void **)&c->plan_totals_host, PLAN_TOTALS * sizeof(int64_t), hipHostMallocMapped | hipHostMallocCoherent)); memset(c->plan_totals_host, 0, PLAN_TOTALS * sizeof(int64_t)); }
    unsigned char *sb = static_cast<unsigned char *>(sl.p);
    s.fields_in = reinterpret_cast<fcpp_field *>(sb + (size_t)off.fields_in); s.info = reinterpret_cast<fcpp_field_info *>(sb + (size_t)off.info);
    s.fields_tmp = reinterpret_cast<DevField *>(sb + (size_t)off.fields_tmp); s.prims_tmp = reinterpret_cast<DevPrim *>(sb + (size_t)off.prims_tmp);
    s.counts = reinterpret_cast<int64_t *>(sb + (size_t)off.counts); s.bases = reinterpret_cast<int64_t *>(sb + (size_t)off.bases);
    s.blk_sums = reinterpret_cast<int64_t *>(sb + (size_t)off.blk_sums); s.totals = reinterpret_cast<int64_t *>(sb + (size_t)off.totals);
    s.keep_tiles = reinterpret_cast<DevTile *>(sb + (size_t)off.keep_tiles); s.keep_wtiles = reinterpret_cast<DevWaveTile *>(sb + (size_t)off.keep_wtiles);
    return FCPP_OK;
}

// The field records as the device reaches them.  Records in pinned host memory (hipHostMalloc / hipHostRegister: a torch tensor with
// pin_memory, engine.FieldTable) are read by k_plan_fields where they lie -- the kernel's first load is the transfer, 128 bytes per
// thread, and no copy command goes before it; anything else is copied to the scratch first.  The callers drain the stream before they return, so
// the caller's memory is not touched afterwards either way.
int device_fields(const fcpp_field *fields, int64_t n_fields, hipStream_t st, const DevPlanScratch &s, const fcpp_field *&dev, std::string &err)
{
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof at);
    void *dp = nullptr;
    if (hipPointerGetAttributes(&at, fields) == hipSuccess && at.type == hipMemoryTypeHost &&
        hipHostGetDevicePointer(&dp, const_cast<fcpp_field *>(fields), 0) == hipSuccess && dp) {
        dev = static_cast<const fcpp_field *>(dp);
        return FCPP_OK;
    }
    // (records in DEVICE memory: read where they lie when they are this context's device's, copied from the peer otherwise)
    int cur = -1;
    if (at.type == hipMemoryTypeDevice && hipGetDevice(&cur) == hipSuccess && at.device == cur) { dev = fields; return FCPP_OK; }
    (void)hipGetLastError();          // (pageable memory: "invalid value" on some runtimes, hipMemoryTypeUnregistered on others)
    DEVCHK(hipMemcpyAsync(s.fields_in, fields, (size_t)n_fields * sizeof(fcpp_field), at.type == hipMemoryTypeDevice ? hipMemcpyDefault : hipMemcpyHostToDevice, st));
    dev = s.fields_in;
    return FCPP_OK;
}

// FCPP_OK: the batch is set up (tables on the device, info on the host); kNotOnDevice: not this path's batch; else the error.
// fresh_templates: the batch's template set is still on its way back from the device (cleared once this path has waited for it)
int try_device_setup(fcpp_ctx *c, fcpp_batch *b, int64_t n_fields, const fcpp_field *fields, const fcpp_polys *obstacles, bool &fresh_templates,
                     std::string &err)
{
    const fcpp_options &opt = b->opt;
    fcpp_setup_times &tm = b->setup;
    if (c->setup_mode == FCPP_SETUP_HOST) { err = "host setup requested"; return kNotOnDevice; }
    if (n_fields <= 0) { err = "empty batch"; return kNotOnDevice; }
    // (a handful of fields: the device's chain of five dependent launches costs ~110 us whatever the batch, the host's plan + tiler + one copy
    // ~65 us + 3 us per field -- a single 500 x 200 m planner's plan call 0.49 -> 0.44 ms)
    {
        const char *e = getenv("FCPP_SMALL_BATCH");          // (fields below which `auto` takes the host: 0 sends every batch it can take to the device -- tools/fuzz_parity.py)
        const int64_t small = e ? atoll(e) : 16;
        if (c->setup_mode == FCPP_SETUP_AUTO && n_fields < small) { err = "a handful of fields: set up by the host"; return kNotOnDevice; }
    }
    const bool dense = opt.sample_spacing != 0.0;          // (round 5: span + quiet runs of the straights + general tiles, k_tile_fields' dense block)
    if (dense && getenv("FCPP_DENSE_DEVICE") && atoi(getenv("FCPP_DENSE_DEVICE")) == 0) { err = "FCPP_DENSE_DEVICE=0: dense sampling set up by the host"; return kNotOnDevice; }
    if (opt.obstacle_mode != FCPP_OBSTACLES_FLAG) { err = "obstacle-aware swaths are planned on the host"; return kNotOnDevice; }
    if (tune_enabled()) { err = "FCPP_TUNE: the tuning knobs are the host tiler's"; return kNotOnDevice; }
    PlanConsts pc;
    TurnTemplates tt;
    int rc = plan_prepare(b->veh, opt, pc, tt, err);
    if (rc != FCPP_OK) return rc;
    if ((rc = validate_polys(obstacles, err)) != FCPP_OK) return rc;
    if (pc.max_prims > DEVPLAN_PRIMS_CAP) { err = "too many headland loops for the device planner"; return kNotOnDevice; }
    hipStream_t st = c->stream;
    auto t0 = std::chrono::steady_clock::now();
    const auto t_call = t0;

    // templates on the host (closed-form test); a fresh set is on its way back
    if (fresh_templates) { DEVCHK(hipStreamSynchronize(st)); fresh_templates = false; }
    c->templates = b->templates;
    const TemplateSet &ts = *b->templates;
    b->cst = make_const(b->veh, opt);
    b->cst.shapes = ts.shapes.p;
    b->cst.tmpl_u = ts.tmpl_u.p; b->cst.tmpl_c = ts.tmpl_c.p; b->cst.tmpl_u_dk = ts.tmpl_u_dk.p;
    b->cst.tmpl_n = (int)ts.tt.nu; b->cst.tmpl_nc = std::max(1, (int)ts.tt.nc);
    const bool turn_quiet = ts.tt.nu >= 3 && closed_form_turns(b->veh, ts.tt, ts.h_tu, ts.h_dk, b->cst);
    tm.templates_ms += ms_since(t0);

    // scratch + the field records
    t0 = std::chrono::steady_clock::now();
    DevPlanScratch s;
    if ((rc = plan_scratch(c, n_fields, pc.max_prims, st, s, err)) != FCPP_OK) return rc;
    const fcpp_field *dev_fields = nullptr;
    if ((rc = device_fields(fields, n_fields, st, s, dev_fields, err)) != FCPP_OK) return rc;

    DevTileConsts tc;
    tc.tu = reinterpret_cast<const Pt2 *>(ts.tmpl_u.p); tc.tc = reinterpret_cast<const Pt2 *>(ts.tmpl_c.p);
    tc.nu = ts.tt.nu; tc.nc = ts.tt.nc;
    tc.turn_quiet = turn_quiet; tc.wave_factor = 24; tc.field_work_tiles = FIELD_WORK_TILES; tc.max_prims = pc.max_prims;
    tc.fuse_spans = ts.tt.nu <= TMPL_LDS_SAMPLES; tc.no_bases = 0;
    tc.two_a = 2 * b->cst.a_lon; tc.u_cap = b->cst.u_cap; tc.c_line = b->cst.ms_work * b->cst.ms_work;
    tc.fence_margin = 1e-7 - opt.geofence_tol;
    tc.reduce_wg_max = 1024;
    tc.cut = make_cut_consts(*b->templates, true, turn_quiet, tc.wave_factor, tc.two_a, tc.u_cap, tc.c_line, tc.fence_margin, b->cst);
    const int64_t n_polys = obstacles ? obstacles->n_polys : 0;
    tc.gen = ++c->plan_gen;
    int64_t *tot = c->plan_totals_host;
    // ---- the tables' layout.  SPECULATIVE for small batches (at most 8192 fields: the counting phase's one-scan form): laid out by per-field
    // CAPACITIES before anything has run, so that the fill pass can be enqueued right behind the last scan -- the host then waits for the
    // totals (they size the output arrays and the steps' launches) while the fill pass already runs, instead of the device waiting for the
    // host's round trip in between.  A field beyond the capacities (PF_OVER_CAPACITY: more than eight wave tiles, a span of more than
    // sixteen chunks: big fields, many headland loops) makes the fill pass a no-op; the tables are then laid out from the totals and filled
    // again, as for large batches.  Within a table the records lie packed either way: only where each table begins differs.
    ImageLayout &lay = b->lay;
    auto common_layout = [&](ImageLayout &l) {
        l = ImageLayout();
        l.n_fields = n_fields; l.wave_tile_points = 128; l.n_chunks = 0;
        l.n_polys = n_polys; l.n_poly_verts = n_polys > 0 ? obstacles->offsets[n_polys] : 0;
        l.info_on_device = true;
    };
    auto upload_obstacles = [&]() -> int {
        if (lay.n_polys <= 0) return FCPP_OK;
        // (the obstacle region of the image -- it alone, offsets rebased -- through the context's pinned staging memory, pageable when that
        // cannot be had; the copy out of the staging memory is asynchronous: the next writer of that memory drains this stream first)
        const size_t o0 = lay.obs_off, o1 = lay.seg, nb = o1 - o0;
        std::vector<unsigned char> tmp;
        unsigned char *img = nullptr;
        if (nb <= kStageMax) {
            DEVCHK(c->stage_wait());
            if (c->stage_cap < nb) {
                if (c->stage) { (void)hipHostFree(c->stage); c->stage = nullptr; c->stage_cap = 0; }
                if (hipHostMalloc(&c->stage, nb + nb / 4, hipHostMallocDefault) == hipSuccess) c->stage_cap = nb + nb / 4;
                else { c->stage = nullptr; (void)hipGetLastError(); }
            }
            if (c->stage_cap >= nb) img = static_cast<unsigned char *>(c->stage);
        }
        if (!img) { try { tmp.resize(nb); } catch (const std::bad_alloc &) { err = "out of host memory"; return FCPP_ENOMEM; } img = tmp.data(); }
        fill_obstacles(obstacles, lay, img, o0);
        DEVCHK(hipMemcpyAsync(static_cast<unsigned char *>(b->slab) + o0, img, nb, hipMemcpyHostToDevice, st));
        if (!tmp.empty()) DEVCHK(hipStreamSynchronize(st));
        else {
            c->stage_stream = st; c->stage_busy = true;
        }
        return FCPP_OK;
    };
    DevPlanTables T;
    auto launch_fill = [&]() -> int {
        bind_tables(b);
        T.fields = b->t.fields; T.prims = b->t.prims; T.tiles = b->t.tiles; T.wtiles = b->t.wave_tiles; T.general_ids = b->t.general_ids;
        T.span_chunks = b->t.span_chunks; T.chunks = b->t.chunks; T.stat_ids = b->t.stat_ids; T.stat_first = b->t.stat_first; T.stat_run = b->t.stat_run;
        T.red_paths = b->t.red_paths; T.field_work = b->t.field_work; T.field_packs = b->t.field_packs; T.open_wave_ids = b->t.open_wave_ids; T.seg = b->t.seg; T.seg_mask = b->t.seg_mask;
        T.partial = b->t.partial; T.field_junc = b->t.field_junc; T.work_totals = b->t.work_totals;
        T.info = reinterpret_cast<fcpp_field_info *>(static_cast<unsigned char *>(b->slab) + lay.info);
        b->cst.field_junc = b->t.field_junc;
        // (the fill pass also computes what the host path launches k_field_junctions, k_run_consts and k_work_totals for, field by field)
        const int frc = launch_devplan_fill(st, n_fields, tc, b->cst, s, T);
        if (frc) { err = std::string("launch_devplan_fill: ") + hipGetErrorString((hipError_t)frc); return FCPP_EHIP; }
        return FCPP_OK;
    };
    const bool exact_only = getenv("FCPP_SETUP_EXACT") != nullptr;              // (the checker of the speculative layout: tests/test_gpu_devplan.py)
    bool spec = (n_fields + 1023) / 1024 <= devplan_small_blocks() && !exact_only && !dense;
    if (spec) {
        common_layout(lay);
        const int64_t K = DEVPLAN_KEEP_TILES;
        lay.n_prims = n_fields * pc.max_prims;
        lay.n_tiles = lay.n_stat = n_fields * (1 + K); lay.n_wave = lay.n_general = lay.n_open_wave = n_fields * K;
        lay.n_span_chunks = n_fields * SPEC_SPAN_CHUNKS; lay.n_runs = n_fields;
        lay.n_red[0] = n_fields; lay.n_work[0] = lay.n_field_work = n_fields;
        layout_image(lay);
        if (lay.total_bytes > ((size_t)1 << 30)) spec = false;
    }
    tc.speculative = spec ? 1 : 0;
    tc.closed_cut = (getenv("FCPP_WINDOW_CUT") || dense) ? 0 : 1;          // (FCPP_WINDOW_CUT=1: round 4's cut on both sides -- the A/B of the two cuts)
    tc.dense = dense ? 1 : 0;
    tc.span_line_max = (getenv("FCPP_DENSE_SPAN") && atoll(getenv("FCPP_DENSE_SPAN")) <= 0) ? 64 : INT64_MAX;
    if (spec) {
        if ((rc = take_slab(c, b, err)) != FCPP_OK) return rc;
        bind_tables(b);
        if ((rc = upload_obstacles()) != FCPP_OK) return rc;
    }
    if (!spec && c->side && !c->ev_chunk[0])
        for (hipEvent_t &e : c->ev_chunk) DEVCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    tc.f0 = 0; tc.f1 = n_fields;
    int lrc = launch_devplan_count(st, n_fields, pc, tc, s, dev_fields, n_polys, obstacles != nullptr, tot, spec ? nullptr : c->side, c->ev_chunk, 5);
    if (lrc) { (void)hipStreamSynchronize(st); err = std::string("launch_devplan_count: ") + hipGetErrorString((hipError_t)lrc); return FCPP_EHIP; }   // (drained: the caller's pinned records may still be read)
    if (spec && (rc = launch_fill()) != FCPP_OK) { (void)hipStreamSynchronize(st); return rc; }
    // (the last scan has written the totals and the flags to `tot`, then the phase's generation number to tot[PX_DONE]: polled -- a word
    // of the host's own pinned memory, there a microsecond after the kernel wrote it -- with the drained stream as the fallback; a speculative
    // fill pass runs on.  No event behind the scan: a record between two kernels holds the second one back by 5 us.  Batches laid out from
    // their totals poll as well: nothing else is in the stream, and the poll sees the word ~10 us before a drained stream reports)
    {
        volatile int64_t *done = tot + PX_DONE;
        const auto t_poll = std::chrono::steady_clock::now();
        bool seen = false;
        for (int spin = 0; !seen; ++spin) {
            seen = *done == tc.gen;
            if (!seen && (spin & 1023) == 1023 && ms_since(t_poll) > 2.0) break;
        }
        if (!seen) DEVCHK(hipStreamSynchronize(st));
        std::atomic_thread_fence(std::memory_order_acquire);
        g_trace_totals_ms = ms_since(t_call);
    }
    tm.host_plan_ms = ms_since(t0);          // (the plan and the counting pass, on the device)
    if (tot[PC_COLS + PF_BAD_OBSTACLES] == tc.gen) { err = "field obstacle range outside the polygon table"; return FCPP_ESIZE; }
    if (tot[PC_COLS + PF_FALLBACK] == tc.gen) { err = "a field beyond the device planner's limits (general stretch, primitives)"; return kNotOnDevice; }
    if (tot[PC_POINTS] > kCountCap) { err = "batch too large"; return FCPP_ESIZE; }
    if (tot[PC_PRIMS] > ((int64_t)1 << 26)) { err = "too many path primitives in one batch: split the batch"; return FCPP_ESIZE; }
    if (tot[PC_TILES] > INT32_MAX) { err = "too many tiles in one batch: split the batch"; return FCPP_ESIZE; }

    // the spans of fields of field work are written by the fields' own workgroups (k_plan_sparse_fields) when ALL of them are short enough for
    // that -- the span launch of such a batch disappears -- and by k_plan_quiet otherwise (a mix loses: measured on cfg2 at the reference's
    // sampling, a third of its spans fusable, 0.0435 instead of 0.0392 ms)
    t0 = std::chrono::steady_clock::now();
    const bool fuse = tc.fuse_spans && tot[PC_UNFUSABLE] == 0 && tot[PC_WORK_SPAN_PTS] > 0;
    auto counts_from_totals = [&](ImageLayout &l) {
        l.n_prims = tot[PC_PRIMS];
        l.n_tiles = tot[PC_TILES]; l.n_wave = tot[PC_WAVE]; l.n_general = tot[PC_GENERAL]; l.n_stat = tot[PC_STAT];
        l.n_span_chunks = fuse ? tot[PC_SPAN_F] : tot[PC_SPAN]; l.n_runs = tot[PC_RUNS];
        for (int k = 0; k < 4; ++k) l.n_red[k] = tot[PC_CLS0 + k];
        l.n_work[0] = tot[PC_WORK]; l.n_field_work = tot[PC_WORK]; l.n_open_wave = tot[PC_OPEN];
        l.quiet_points = tot[PC_SPAN_PTS] + tot[PC_CHUNK_PTS]; l.span_points = tot[PC_SPAN_PTS] - (fuse ? tot[PC_WORK_SPAN_PTS] : 0); l.work_span_points = fuse ? tot[PC_WORK_SPAN_PTS] : 0;
        l.n_chunks = tot[PC_CHUNKS]; l.chunk_points = tot[PC_CHUNK_PTS]; l.wave_points = tot[PC_WAVE_PTS];
        l.work_wave_points = tot[PC_WORK_WAVE_PTS]; l.wave_inside = tot[PC_WAVE_INSIDE];
    };
    if (spec && tot[PC_COLS + PF_OVER_CAPACITY] != tc.gen) {
        counts_from_totals(lay);             // (the tables begin where the capacities put them; what they hold is what the totals say)
        tm.image_ms = ms_since(t0);
    } else {
        // the image's layout from the totals, the allocation, the obstacle table, the fill pass
        ImageLayout ex;
        common_layout(ex);
        counts_from_totals(ex);
        layout_image(ex);
        if (b->slab && b->slab_bytes < ex.total_bytes) {       // (a speculative slab that is too small: its fill pass was a no-op, but it is in the stream)
            DEVCHK(hipStreamSynchronize(st));
            (void)hipFree(b->slab); b->slab = nullptr; b->slab_bytes = 0;
        }
        lay = ex;
        if (!b->slab && (rc = take_slab(c, b, err)) != FCPP_OK) return rc;
        bind_tables(b);
        if ((rc = upload_obstacles()) != FCPP_OK) return rc;
        tm.image_ms = ms_since(t0);
        t0 = std::chrono::steady_clock::now();
        tc.speculative = 0;
        tc.fuse_spans = fuse;
        if ((rc = launch_fill()) != FCPP_OK) return rc;
    }
    tm.image_bytes = (int64_t)((size_t)n_fields * sizeof(fcpp_field) + (lay.n_polys > 0 ? lay.seg - lay.obs_off : 0));
    c->plan_slots[c->plan_cur].pending = true;
    b->setup_stream = st; b->setup_pending = true;
    // fcpp_field_info stays on the device until somebody asks (fcpp_batch_info); the stream is NOT drained: a step enqueued next runs
    // right behind the setup
    b->info_dev = T.info;
    b->hp.info.clear();
    b->hp.tt = tt;
    b->hp.total_points = tot[PC_POINTS]; b->hp.total_prims = tot[PC_PRIMS];
    b->hp.fields.clear(); b->hp.blocks.clear(); b->hp.same_as.clear();
    tm.tiler_ms = ms_since(t0);              // (the fill pass, the setup kernels and the copy back of the field records)
    tm.h2d_ms = 0.0;
    tm.device_setup = 1;
    return FCPP_OK;
}
#undef DEVCHK
}  // namespace

int fcpp_batch_create(fcpp_ctx *c, const fcpp_vehicle *veh, const fcpp_options *opt, int64_t n_fields,
                      const fcpp_field *fields, const fcpp_polys *obstacles, fcpp_batch **out)
{
    if (!c || !veh || !opt || !out || n_fields < 0 || (n_fields > 0 && !fields))
        return fail(FCPP_EINVAL, "bad arguments");
    if (n_fields > INT32_MAX) return fail(FCPP_ESIZE, "too many fields");
    *out = nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    HIPCHK(hipSetDevice(c->device));
    std::unique_ptr<fcpp_batch> b(new (std::nothrow) fcpp_batch());
    if (!b) return fail(FCPP_ENOMEM, "out of host memory");
    b->ctx = c; b->veh = *veh; b->opt = *opt; b->n_fields = n_fields;
    b->two_streams = !(tune_enabled() && getenv("FCPP_ONE_STREAM") != nullptr);
    b->two_stream_max = std::max(0, std::min(tune_int("FCPP_TWO_STREAM_MAX", 512), 1 << 20));
    b->sparse_beside_max = std::max(0, std::min(tune_int("FCPP_SPARSE_BESIDE_MAX", 0), 1 << 30));
    fcpp_setup_times &tm = b->setup;
    tm.threads = WorkerPool::width();
    std::string err;
    hipStream_t st = c->stream;
    b->note_stream(st);

    // ---- 1. the turn templates go first: their two launches and the copy back run on the device while the host plans the fields
    auto t0 = std::chrono::steady_clock::now();
    bool fresh = false;
    {
        TurnTemplates tt;
        int rc = plan_templates(*veh, *opt, tt, err);
        if (rc != FCPP_OK) return fail(rc, err);
        rc = get_templates(c, tt, *opt, st, b->templates, fresh);
        if (rc != FCPP_OK) return rc;
    }
    tm.templates_ms = ms_since(t0);

    // ---- 2a. the setup on the DEVICE (fcpp_devplan.h) for batches at the reference's own sampling: the field records go up, the plan
    // function the host would run (fcpp_planfn.h) runs one thread per field, the tiler's cut one wavefront per field, the totals come
    // back once (they size the tables and the output arrays), the tables are written in place.  The host plans nothing per field.
    {
        int rc = try_device_setup(c, b.get(), n_fields, fields, obstacles, fresh, err);
        if (rc == FCPP_OK) {
            tm.total_ms = ms_since(t_begin);
            c->last_setup = tm;
            *out = b.release();
            return FCPP_OK;
        }
        if (b->slab) { (void)hipStreamSynchronize(c->stream); (void)hipFree(b->slab); b->slab = nullptr; }
        if (rc != kNotOnDevice) return fail(rc, err);
        if (c->setup_mode == FCPP_SETUP_DEVICE) return fail(FCPP_EUNSUPPORTED, "FCPP_SETUP_DEVICE: " + err);
        err.clear();
    }

    // ---- 2. host plan: __init__ + the O(1) decisions of every field, blocks of fields side by side (fcpp_host.cpp)
    t0 = std::chrono::steady_clock::now();
    std::vector<fcpp_field> fcopy;             // (records in device memory: the host plans a copy)
    int rc = host_fields(fields, n_fields, fcopy, fields);
    if (rc != FCPP_OK) { if (fresh) (void)hipStreamSynchronize(st); return rc; }
    rc = build_host_plan(*veh, *opt, n_fields, fields, obstacles, true, b->hp, err);
    if (rc != FCPP_OK) { if (fresh) (void)hipStreamSynchronize(st); return fail(rc, err); }
    tm.host_plan_ms = ms_since(t0);

    // ---- 3. templates on the host; are the U-turns of this batch closed form?
    t0 = std::chrono::steady_clock::now();
    if (fresh) HIPCHK(hipStreamSynchronize(st));
    c->templates = b->templates;
    const TemplateSet &ts = *b->templates;
    b->cst = make_const(*veh, *opt);
    b->cst.shapes = ts.shapes.p;
    b->cst.tmpl_u = ts.tmpl_u.p; b->cst.tmpl_c = ts.tmpl_c.p; b->cst.tmpl_u_dk = ts.tmpl_u_dk.p;
    b->cst.tmpl_n = (int)ts.tt.nu; b->cst.tmpl_nc = std::max(1, (int)ts.tt.nc);
    const bool turn_quiet = ts.tt.nu >= 3 && closed_form_turns(*veh, ts.tt, ts.h_tu, ts.h_dk, b->cst);
    tm.templates_ms += ms_since(t0);

    // ---- 4. tiler: every field's path cut into work for the four kernels, blocks side by side (fcpp_tiler.cpp)
    t0 = std::chrono::steady_clock::now();
    TileConsts tc;
    tc.tu = reinterpret_cast<const Pt2 *>(ts.h_tu.data()); tc.tc = reinterpret_cast<const Pt2 *>(ts.h_tc.data());
    tc.nu = ts.tt.nu; tc.nc = ts.tt.nc; tc.templates_ok = true;
    tc.turn_quiet = turn_quiet;
    tc.two_a = 2 * b->cst.a_lon; tc.u_cap = b->cst.u_cap; tc.c_line = b->cst.ms_work * b->cst.ms_work;
    // (the device flags a point whose edge function is below -geofence_tol; host and device evaluate a point by the same formulas and
    // differ by roundings of ~1e-12 m: 1e-7 m of slack is five orders of magnitude of safety -- and lets the points that lie ON the boundary,
    // the ends of the reverse fills, three per field of the metric's size, pass with the default tolerance of 1e-6 m: with the millimetre of margin
    // of rounds 2-3a those three points sent half of the headline's wave tiles through the geofence test)
    tc.fence_margin = 1e-7 - opt->geofence_tol;
    // (tuning knobs, read once per batch and clamped; both change which kernel plans a stretch or which reduction class a path falls
    // into, i.e. the order of its sums at the last bit: diagnostic builds only)
    tc.wave_factor = std::max(0, std::min(tune_int("FCPP_WAVE_FACTOR", 24), 64));
    tc.reduce_wg_max = std::max(256, std::min(tune_int("FCPP_REDUCE_WG_MAX", 1024), 1 << 20));
    tc.wave_points = tune_int("FCPP_WAVE_POINTS", 128) == 64 ? 64 : 128;
    tc.field_work = tune_int("FCPP_FIELD_WORK", 1) != 0;
    tc.field_work_tiles = std::max(1, std::min(tune_int("FCPP_FIELD_WORK_TILES", FIELD_WORK_TILES), FIELD_WORK_TILES));
    tc.fuse_spans = tc.field_work && ts.tt.nu <= TMPL_LDS_SAMPLES && tune_int("FCPP_FUSE_SPANS", 1) != 0;
    // the chunk lists are expanded on the device from the host's chunk groups; FCPP_HOST_CHUNKS=1 (the checker, tests/test_gpu_devplan.py) and
    // the FCPP_CHUNK_SPREAD diagnostic keep the host's own lists
    tc.device_chunks = !(getenv("FCPP_HOST_CHUNKS") && atoi(getenv("FCPP_HOST_CHUNKS")) != 0) && !getenv("FCPP_CHUNK_SPREAD");
    // the reference's sampling: the general stretch of every field with a closed-form span is cut in closed form (fcpp_cutfn.h), as the device
    // planner cuts it -- host-built and device-built tables stay equal byte for byte
    tc.closed_cut = opt->sample_spacing == 0.0 && opt->obstacle_mode == FCPP_OBSTACLES_FLAG && tc.wave_points == CUT_WAVE_LANES && !getenv("FCPP_WINDOW_CUT");
    tc.cut = make_cut_consts(*b->templates, false, turn_quiet, tc.wave_factor, tc.two_a, tc.u_cap, tc.c_line, tc.fence_margin, b->cst);
    if (getenv("FCPP_DENSE_SPAN") && atoll(getenv("FCPP_DENSE_SPAN")) <= 0) tc.span_line_max = 64;          // (round 4's runs: the A/B)
    BatchTiler tiler;
    ImageLayout &lay = b->lay;
    rc = tiler.plan(b->hp, tc, obstacles, lay, err);
    if (rc != FCPP_OK) return fail(rc, err);
    if (tc.fuse_spans && lay.unfusable_work > 0) {       // all spans of fields of field work are fused, or none (as on the device path)
        tc.fuse_spans = false;
        rc = tiler.plan(b->hp, tc, obstacles, lay, err);
        if (rc != FCPP_OK) return fail(rc, err);
    }
    tm.tiler_ms = ms_since(t0);
    if (getenv("FCPP_DEBUG_TILING"))
        fprintf(stderr, "[fcpp] tiling: %lld tiles; wave-tile stretches refused: back halo %lld, forward halo %lld, too few outputs %lld, "
                "primitive span %lld; %lld wave tiles, %lld of them inside the geofence by the host's test\n", (long long)lay.n_tiles,
                (long long)lay.wave_fail[0], (long long)lay.wave_fail[1], (long long)lay.wave_fail[2], (long long)lay.wave_fail[3],
                (long long)lay.n_wave, (long long)lay.wave_inside);

    // ---- 5. the image: one device allocation (the context's spare if it is large enough), the tables written into pinned memory
    t0 = std::chrono::steady_clock::now();
    if ((rc = take_slab(c, b.get(), err)) != FCPP_OK) return fail(rc, err);
    struct SlabGuard {       // (an error below must not leak the allocation)
        fcpp_batch *b; bool armed = true;
        ~SlabGuard() { if (armed && b->slab) { (void)hipFree(b->slab); b->slab = nullptr; } }
    } guard{ b.get() };
    unsigned char *img = nullptr;
    std::vector<unsigned char> pageable;
    if (lay.upload_bytes <= kStageMax) {
        HIPCHK(c->stage_wait());
        if (c->stage_cap < lay.upload_bytes) {
            if (c->stage) { (void)hipHostFree(c->stage); c->stage = nullptr; c->stage_cap = 0; }
            const size_t want = std::min(kStageMax, std::max<size_t>(lay.upload_bytes + lay.upload_bytes / 4, (size_t)1 << 20));
            if (hipHostMalloc(&c->stage, want, hipHostMallocDefault) == hipSuccess) c->stage_cap = want;
            else { c->stage = nullptr; (void)hipGetLastError(); }
        }
        if (c->stage_cap >= lay.upload_bytes) img = static_cast<unsigned char *>(c->stage);
    }
    if (!img) {
        try { pageable.resize(lay.upload_bytes); } catch (const std::bad_alloc &) { return fail(FCPP_ENOMEM, "out of host memory"); }
        img = pageable.data();
    }
    tiler.fill(b->hp, obstacles, lay, img);
    bind_tables(b.get());
    // (the primitives live in the image now; the host keeps the per-field records for fcpp_batch_info and the staged pipeline's tiling)
    for (PlanBlock &blk : b->hp.blocks) std::vector<DevPrim>().swap(blk.prims);
    std::vector<int32_t>().swap(b->hp.same_as);
    tm.image_ms = ms_since(t0);
    tm.image_bytes = (int64_t)lay.upload_bytes;

    // ---- 6. one copy, the per-field junction constants, and the stream drained (the staging memory is the context's)
    t0 = std::chrono::steady_clock::now();
    if (lay.upload_bytes > 0) HIPCHK(hipMemcpyAsync(b->slab, img, lay.upload_bytes, hipMemcpyHostToDevice, st));
    if (n_fields > 0) {
        b->cst.field_junc = b->t.field_junc;
        LAUNCHCHK(launch_field_junctions(st, n_fields, b->t.fields, b->cst, b->t.field_junc));
        LAUNCHCHK(launch_expand_chunks(st, lay.n_chunk_groups, reinterpret_cast<const DevChunkGroup *>(static_cast<unsigned char *>(b->slab) + lay.chunk_groups),
                                       b->t.tiles, b->t.fields, b->t.chunks, b->t.span_chunks));
        // the statistics slots: closed-form statistics of the quiet runs (the same at every step), zeros elsewhere
        LAUNCHCHK(launch_run_consts(st, lay.n_stat, b->t.stat_ids, b->t.stat_run, b->t.tiles, b->t.fields, b->t.prims, b->cst, b->t.partial));
        LAUNCHCHK(launch_work_totals(st, lay.n_field_work, b->t.field_work, b->t.stat_run, b->t.partial, b->t.work_totals));
    }
    // The stream is NOT drained when the image went through the context's pinned staging memory (round 5; as the device-side setup leaves it):
    // a step enqueued next runs right behind the copy and the setup kernels, a consumer on another stream is ordered behind them (wait_setup),
    // the staging memory's next writer drains this stream first (stage_wait).  A pageable image is freed on return: drained.
    if (!pageable.empty() || lay.upload_bytes == 0) HIPCHK(hipStreamSynchronize(st));
    else { c->stage_stream = st; c->stage_busy = true; b->setup_stream = st; b->setup_pending = true; }
    tm.h2d_ms = ms_since(t0);
    tm.total_ms = ms_since(t_begin);
    c->last_setup = tm;
    guard.armed = false;
    *out = b.release();
    return FCPP_OK;
}

// fcpp_field_info of a batch set up on the device: copied back when first needed
static int ensure_info(fcpp_batch *b)
{
    if (!b->info_dev || !b->hp.info.empty() || b->n_fields == 0) return FCPP_OK;
    HIPCHK(hipSetDevice(b->ctx->device));
    try { b->hp.info.resize((size_t)b->n_fields); } catch (const std::bad_alloc &) { return fail(FCPP_ENOMEM, "out of host memory"); }
    for (hipStream_t s : b->used_streams) HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipMemcpy(b->hp.info.data(), b->info_dev, (size_t)b->n_fields * sizeof(fcpp_field_info), hipMemcpyDeviceToHost));
    return FCPP_OK;
}

int fcpp_batch_setup_times(const fcpp_batch *b, fcpp_setup_times *out)
{
    if (!b || !out) return fail(FCPP_EINVAL, "bad arguments");
    *out = b->setup;
    return FCPP_OK;
}

int fcpp_batch_info(const fcpp_batch *b, fcpp_field_info *info_out, int64_t *total_points)
{
    if (!b) return fail(FCPP_EINVAL, "batch is NULL");
    if (info_out && b->n_fields) { const int rc = ensure_info(const_cast<fcpp_batch *>(b)); if (rc) return rc; }
    if (info_out && b->n_fields) memcpy(info_out, b->hp.info.data(), (size_t)b->n_fields * sizeof(fcpp_field_info));
    if (total_points) *total_points = b->hp.total_points;
    return FCPP_OK;
}

int fcpp_batch_run(fcpp_batch *b, double *x, double *y, double *kappa, double *v, uint32_t *fs,
                   fcpp_field_stats *stats, int mode)
{
    if (!b) return fail(FCPP_EINVAL, "batch is NULL");
    // mode 1 (default): quiet tiles (k_plan_quiet, streaming) and general tiles (k_plan_fused) as two launches.
    // Tuning only: modes 12/13/14 = mode 1 with k_plan_fused compiled for a minimum of 2/3/4 waves per SIMD (default 3).
    // (Measured and dropped: both tile kinds in one grid, and the two kernels on two streams -- the HBM-bound and the
    // ALU-bound kernel do not overlap usefully, the sum of the two launches is the faster schedule.)
    int variant = 3;
    if (mode >= 12 && mode <= 14) { variant = mode - 10; mode = 1; }
    if (mode != 0 && mode != 1) return fail(FCPP_EINVAL, "unknown pipeline mode");
    if (mode != b->last_mode) { b->prof_runs = 0; b->last_mode = mode; }
    if (b->n_fields == 0) return FCPP_OK;
    if (b->hp.total_points > 0 && (!x || !y || !kappa || !v || !fs)) return fail(FCPP_EINVAL, "output pointer is NULL");
    if (!stats) return fail(FCPP_EINVAL, "stats pointer is NULL");
    HIPCHK(hipSetDevice(b->ctx->device));
    hipStream_t st = b->ctx->stream;
    b->note_stream(st);
    HIPCHK(b->wait_setup(st));
    if (fcpp_ctx *c = b->ctx; c->arena && x) {      // output arrays from the context's arena: their block remembers the stream that writes it (fcpp_outputs_free)
        const char *p = reinterpret_cast<const char *>(x), *a0 = static_cast<const char *>(c->arena);
        if (p >= a0 && p < a0 + c->arena_lane) {
            const size_t off = (size_t)(p - a0);
            for (auto &blk : c->arena_live) if (off >= blk.off && off < blk.off + blk.len) { blk.last = st; blk.used = true; break; }
        }
    }
    if (mode == 0 && !b->til0_built) {      // the staged pipeline's own tiling: plain tiles of at most TILE_POINTS points
        { const int rc = ensure_info(b); if (rc) return rc; }
        Tiling t0;
        std::vector<int64_t> offs((size_t)b->n_fields + 1, 0);
        for (int64_t i = 0; i < b->n_fields; ++i) offs[(size_t)i + 1] = offs[(size_t)i] + b->hp.info[(size_t)i].n_main + b->hp.info[(size_t)i].n_head;
        t0.build(b->n_fields, offs.data());
        HIPCHK(b->til0.upload(t0, st));
        b->til0_built = true;
    }
    DevObstacles obs = { b->t.obs_off, b->t.obs_x, b->t.obs_y, b->t.obs_bbox };
    hipEvent_t *ev = nullptr;
    unsigned char *evs = nullptr;
    if (b->profiling > 0 && b->prof_runs < kProfRuns && (b->run_counter++ % b->profiling) == 0) {
        ev = &b->events[(size_t)b->prof_runs * kStages * 2];
        evs = &b->ev_set[(size_t)b->prof_runs * kStages];
    }
    // a stage = one kernel; profiled, its dispatch carries a start and a stop event (fcpp_device.h: LaunchProf)
#define STAGE(k, call)                                                                  \
    do {                                                                                \
        if (ev) { g_launch_prof.start = ev[2 * (k)]; g_launch_prof.stop = ev[2 * (k) + 1]; } \
        const int e_ = (call);                                                          \
        if (ev) { evs[k] = g_launch_prof.start == nullptr; g_launch_prof = LaunchProf(); }   \
        if (e_ != 0) return fail(FCPP_EHIP, std::string(#call) + ": " + hipGetErrorString((hipError_t)e_)); \
    } while (0)
    if (mode == 1) {
        const FusedTables &t = b->t;
        const ImageLayout &lay = b->lay;
        // Two streams inside the step where the general tiles are FEW and long-lived (dense sampling of a single large field: a dozen
        // tiles that walk long halos, cfg3: 5873 points in 0.37 ms): beside the HBM-bound streaming kernel they cost nothing (cfg3
        // 0.87 -> 0.51 ms).  Not when the general kernel can fill the chip itself -- side by side it takes compute units from the
        // streaming kernel (cfg2 at 0.1 m, 5950 general tiles: 6.5 vs 5.6 ms) -- and not at sparse sampling: k_plan_sparse and the
        // span kernel get in each other's way (cfg5 2.41 vs 2.29 ms, cfg1 x 4096 0.114 vs 0.098 ms; measured again with the final kernels: 2.46 vs 2.40
        // and 0.091 vs 0.085 ms, k_plan_sparse 1.07 instead of 0.69 ms; also with the span kernel held to four or five
        // waves per SIMD).
        hipStream_t sd = st;
        const bool two = b->two_streams && lay.span_points + lay.chunk_points > 0 &&
                         ((lay.n_general > 0 && lay.n_general <= b->two_stream_max && lay.n_wave == 0) ||
                          (b->sparse_beside_max > 0 && lay.n_wave > 0 && lay.n_wave <= b->sparse_beside_max));
        if (two) {
            sd = b->ctx->side;
            HIPCHK(hipEventRecord(b->ctx->ev_fork, st));
            HIPCHK(hipStreamWaitEvent(sd, b->ctx->ev_fork, 0));
        }
        // the wave tiles of fields that k_plan_sparse_fields does not take (all of them when there are none of those)
        const bool fw = lay.n_field_work > 0;
        if (!fw) STAGE(2, launch_plan_sparse(sd, lay.n_wave, t.wave_tiles, t.fields, t.prims, b->cst, obs, x, y, kappa, v, fs, t.partial, lay.wave_tile_points / 64));
        else STAGE(2, launch_plan_sparse(sd, lay.n_open_wave, t.wave_tiles, t.fields, t.prims, b->cst, obs, x, y, kappa, v, fs, t.partial, lay.wave_tile_points / 64,
                                         t.open_wave_ids));
        STAGE(3, launch_plan_fused(sd, variant, lay.n_general, t.general_ids, t.tiles, t.fields, t.prims, b->cst, obs, x,
                                   y, kappa, v, fs, t.partial));
        if (two) HIPCHK(hipEventRecord(b->ctx->ev_join, sd));
        STAGE(0, launch_plan_quiet(st, lay.n_span_chunks, t.span_chunks, 16, t.fields, t.prims, b->cst, obs, x, y, kappa, v, fs, t.partial));
        // (straights and U-turns in ONE launch: measured 4 % faster on identical memory than an instance each, tools/ab_quiet.py)
        STAGE(1, launch_plan_quiet(st, lay.n_chunks, t.chunks, 14, t.fields, t.prims, b->cst, obs, x, y, kappa, v, fs, t.partial));
        // fields planned and reduced by one workgroup each: after the streaming kernels, whose flag counts their reduction reads
        if (fw) {       // (one launch per class of fields; normally one class holds them all: the stage's events time the first launch)
            int64_t off = 0;
            bool first = true;
            for (int c = 0; c < 4; ++c) {
                if (lay.n_work[c] == 0) continue;
                if (first) STAGE(5, launch_plan_sparse_fields(st, lay.n_work[c], t.field_packs + off, b->cst, obs, x, y, kappa, v, fs, t.partial,
                                                              FIELD_WORK_WAVES[c], t.work_totals + off, stats, lay.work_span_points > 0));
                else LAUNCHCHK(launch_plan_sparse_fields(st, lay.n_work[c], t.field_packs + off, b->cst, obs, x, y, kappa, v, fs, t.partial,
                                                         FIELD_WORK_WAVES[c], t.work_totals + off, stats, lay.work_span_points > 0));
                first = false;
                off += lay.n_work[c];
            }
        }
        if (two) HIPCHK(hipStreamWaitEvent(st, b->ctx->ev_join, 0));
        // (four classes of paths by their number of entries; normally one of them holds every path of a batch: the stage's events
        // time the first launch)
        {
            const int groups[4] = { 8, 64, 256, 256 };     // (16 or 32 lanes for the first class: 49 -> 53 / 79 us on cfg5)
            const int32_t *pl = t.red_paths;
            bool first = true;
            for (int c = 0; c < 4; ++c) {
                if (lay.n_red[c] == 0) continue;
                // (a class that holds every path lists them in order: no list, one dependent load less in a latency-bound kernel)
                const int32_t *list = lay.n_red[c] == lay.n_fields ? nullptr : pl;
                if (first)
                    STAGE(4, launch_reduce_stats(st, lay.n_red[c], t.partial, t.stat_first, nullptr, stats, nullptr, nullptr, nullptr,
                                                 nullptr, nullptr, &b->cst, list, groups[c], c == 3 ? t.red_scratch : nullptr, 1));
                else
                    LAUNCHCHK(launch_reduce_stats(st, lay.n_red[c], t.partial, t.stat_first, nullptr, stats, nullptr, nullptr, nullptr,
                                                  nullptr, nullptr, &b->cst, list, groups[c], c == 3 ? t.red_scratch : nullptr, 1));
                first = false;
                pl += lay.n_red[c];
            }
        }
        if (ev) ++b->prof_runs;
        return FCPP_OK;
    }
    DevTiling &t = b->til0;
    HIPCHK(hipMemsetAsync(t.n_adj.p, 0, (size_t)t.n_paths * sizeof(unsigned long long), st));
    STAGE(0, launch_generate(st, t.n_tiles, t.tiles.p, b->t.fields, b->t.prims, b->cst, x, y, v, fs));
    STAGE(1, launch_curv_clamp(st, t.n_tiles, t.tiles.p, t.paths.p, b->cst, 1, x, y, v, v, kappa, t.n_adj.p));
    STAGE(2, launch_scan_tiles(st, t.n_tiles, t.tiles.p, t.paths.p, b->cst, x, y, v, t.agg_f.p, t.agg_b.p));
    STAGE(3, launch_scan_spine(st, t.n_tiles, t.agg_f.p, t.agg_b.p, t.carry_f.p, t.carry_b.p, t.spine.p));
    STAGE(4, launch_scan_apply(st, t.n_tiles, t.tiles.p, t.paths.p, b->cst, 3, x, y, v, v, t.carry_f.p, t.carry_b.p));
    STAGE(5, launch_validate(st, t.n_tiles, t.tiles.p, t.paths.p, b->t.fields, b->cst, obs, x, y, kappa, v, fs, t.partial.p));
    STAGE(6, launch_reduce_stats(st, t.n_paths, t.partial.p, t.tile_first.p, t.n_adj.p, stats));
#undef STAGE
    if (ev) ++b->prof_runs;
    return FCPP_OK;
}

int fcpp_batch_plan(fcpp_ctx *c, const fcpp_vehicle *veh, const fcpp_options *opt, int64_t n_fields, const fcpp_field *fields,
                    const fcpp_polys *obstacles, fcpp_field_stats *stats, fcpp_batch **batch, double **x, double **y, double **kappa, double **v,
                    uint32_t **fs, int64_t *total_points)
{
    if (!batch || !x || !y || !kappa || !v || !fs) return fail(FCPP_EINVAL, "bad arguments");
    *batch = nullptr; *x = *y = *kappa = *v = nullptr; *fs = nullptr;
    fcpp_batch *b = nullptr;
    static const bool trace = getenv("FCPP_TRACE_PLAN") != nullptr;      // (diagnostic: the host's side of a plan call, microseconds since its start, to stderr)
    const auto t_begin = std::chrono::steady_clock::now();
    int rc = fcpp_batch_create(c, veh, opt, n_fields, fields, obstacles, &b);
    if (rc != FCPP_OK) return rc;
    const double t_create = trace ? ms_since(t_begin) : 0.0;
    const int64_t total = b->hp.total_points;
    rc = fcpp_outputs_alloc(c, total, 0, x, y, kappa, v, fs);
    const double t_alloc = trace ? ms_since(t_begin) : 0.0;
    if (rc == FCPP_OK) {
        if (!stats && b->slab) stats = reinterpret_cast<fcpp_field_stats *>(static_cast<unsigned char *>(b->slab) + b->lay.own_stats);
        rc = fcpp_batch_run(b, *x, *y, *kappa, *v, *fs, stats, 1);
        if (trace) fprintf(stderr, "[fcpp] plan call: entered at %.1f us (CLOCK_MONOTONIC mod 1 s), totals seen %.1f us, create returns %.1f, arrays %.1f, step enqueued %.1f\n",
                           (double)(std::chrono::duration_cast<std::chrono::nanoseconds>(t_begin.time_since_epoch()).count() % 1000000000ll) / 1e3,
                           g_trace_totals_ms * 1e3, t_create * 1e3, t_alloc * 1e3, ms_since(t_begin) * 1e3);
        if (rc != FCPP_OK) { const std::string keep = g_err; (void)hipStreamSynchronize(c->stream); (void)fcpp_outputs_free(c, *x); g_err = keep; }
    }
    if (rc != FCPP_OK) {
        const std::string keep = g_err;
        (void)fcpp_batch_destroy(b);
        *x = *y = *kappa = *v = nullptr; *fs = nullptr;
        g_err = keep;
        return rc;
    }
    *batch = b;
    if (total_points) *total_points = total;
    return FCPP_OK;
}

int fcpp_batch_own_stats(const fcpp_batch *b, fcpp_field_stats **stats)
{
    if (!b || !stats) return fail(FCPP_EINVAL, "bad arguments");
    *stats = b->slab ? reinterpret_cast<fcpp_field_stats *>(static_cast<unsigned char *>(b->slab) + b->lay.own_stats) : nullptr;
    return FCPP_OK;
}

int fcpp_batch_set_profiling(fcpp_batch *b, int enable)
{
    if (!b) return fail(FCPP_EINVAL, "batch is NULL");
    HIPCHK(hipSetDevice(b->ctx->device));
    if (enable && b->events.empty()) {
        b->events.resize((size_t)kProfRuns * kStages * 2);
        b->ev_set.assign((size_t)kProfRuns * kStages, 0);
        for (hipEvent_t &e : b->events) HIPCHK(hipEventCreate(&e));
    }
    b->profiling = enable > 0 ? enable : 0;
    b->prof_runs = 0; b->run_counter = 0;
    return FCPP_OK;
}

int fcpp_batch_stage_times(fcpp_batch *b, int max_stages, double *ms_sum, int *n_stages, int *n_runs)
{
    if (!b || !ms_sum || max_stages < kStages) return fail(FCPP_EINVAL, "bad arguments");
    HIPCHK(hipSetDevice(b->ctx->device));
    HIPCHK(hipStreamSynchronize(b->ctx->stream));
    const int ns = kStageCount[b->last_mode];
    for (int k = 0; k < kStages; ++k) ms_sum[k] = 0.0;
    for (int r = 0; r < b->prof_runs; ++r) {
        hipEvent_t *ev = &b->events[(size_t)r * kStages * 2];
        for (int k = 0; k < ns; ++k) {
            if (!b->ev_set[(size_t)r * kStages + k]) continue;      // the stage had nothing to launch
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1]));
            ms_sum[k] += ms;
        }
    }
    if (n_stages) *n_stages = ns;
    if (n_runs) *n_runs = b->prof_runs;
    b->prof_runs = 0;
    return FCPP_OK;
}

int fcpp_batch_stage_points(const fcpp_batch *b, int mode, int stage, int64_t *points)
{
    if (!b || !points || mode < 0 || mode > 1 || stage < 0 || stage >= kStageCount[mode]) return fail(FCPP_EINVAL, "bad arguments");
    const ImageLayout &t = b->lay;
    const int64_t all = b->hp.total_points;
    if (mode == 0) { *points = all; return FCPP_OK; }         // every staged kernel sees every point
    const int64_t per_stage[6] = { t.span_points, t.chunk_points, t.wave_points - t.work_wave_points, all - t.quiet_points - t.wave_points, all,
                                   t.work_wave_points + t.work_span_points };
    *points = per_stage[stage];
    return FCPP_OK;
}

int fcpp_batch_point_split(const fcpp_batch *b, int64_t *quiet_points, int64_t *general_points)
{
    if (!b) return fail(FCPP_EINVAL, "batch is NULL");
    const int64_t q = b->lay.quiet_points;
    if (quiet_points) *quiet_points = q;
    if (general_points) *general_points = b->hp.total_points - q;
    return FCPP_OK;
}

int fcpp_batch_reduce_classes(const fcpp_batch *b, int64_t *classes_out)
{
    if (!b || !classes_out) return fail(FCPP_EINVAL, "bad arguments");
    for (int c = 0; c < 4; ++c) classes_out[c] = b->lay.n_red[c];
    return FCPP_OK;
}

const char *fcpp_batch_stage_name(int mode, int stage)
{
    return (mode >= 0 && mode < 2 && stage >= 0 && stage < kStages) ? kStageNames[mode][stage] : "";
}

int fcpp_batch_connectors(fcpp_batch *b, double *approach_xy, double *departure_xy)
{
    if (!b) return fail(FCPP_EINVAL, "batch is NULL");
    if (b->n_fields == 0) return FCPP_OK;
    HIPCHK(hipSetDevice(b->ctx->device));
    hipStream_t st = b->ctx->stream;
    b->note_stream(st);
    HIPCHK(b->wait_setup(st));
    if (approach_xy) LAUNCHCHK(launch_straight(st, b->n_fields, b->t.seg, 50, b->t.seg_mask, approach_xy));
    if (departure_xy)
        LAUNCHCHK(launch_straight(st, b->n_fields, b->t.seg + 4 * b->n_fields, 50, b->t.seg_mask + b->n_fields, departure_xy));
    return FCPP_OK;
}

int fcpp_batch_destroy(fcpp_batch *b)
{
    if (!b) return FCPP_OK;
    fcpp_ctx *c = b->ctx;
    (void)hipSetDevice(c->device);
    // (the tables may become the next batch's: nothing that reads them may still be running -- on whichever streams this batch ran)
    // (every stream the batch was created, stepped or read on is noted in used_streams; the context's CURRENT stream is not drained for its own
    // sake: it may be another batch's -- a caller with two plan calls in flight on two streams would wait for the other call's step here)
    for (hipStream_t s : b->used_streams) (void)hipStreamSynchronize(s);
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (b->slab) {      // the larger of this allocation and the context's spare stays for the next batch
        if (b->slab_bytes <= kSpareMax && b->slab_bytes > c->spare_cap) {
            if (c->spare) (void)hipFree(c->spare);
            c->spare = b->slab; c->spare_cap = b->slab_bytes;
        } else (void)hipFree(b->slab);
        b->slab = nullptr;
    }
    if (b->ev_setup && c->ev_pool.size() < 64) { c->ev_pool.push_back(b->ev_setup); b->ev_setup = nullptr; }
    delete b;
    return FCPP_OK;
}

// ---- standalone operators -------------------------------------------------------------------
namespace {
// The tile table of a path set, kept in the context between calls of the standalone operators: a caller that plans and verifies
// the same paths (the planner mirror does: speed plan, verify, verify again) pays for the host-side tiling and its upload once.
struct PathTiling {
    std::vector<int64_t> offs;
    DevTiling dt;
};

// offsets on the host: the caller's copy, or read back from the device (one copy + synchronisation); the tile table is rebuilt
// only when they differ from the cached set's
int make_tiling(fcpp_ctx *c, int64_t n_paths, const int64_t *offsets_dev, const int64_t *offsets_host, int64_t total, DevTiling **out)
{
    if (n_paths < 0 || total < 0 || n_paths > INT32_MAX) return fail(FCPP_ESIZE, "bad sizes");
    std::vector<int64_t> offs((size_t)n_paths + 1, 0);
    if (offsets_host) memcpy(offs.data(), offsets_host, offs.size() * sizeof(int64_t));
    else {
        HIPCHK(hipMemcpyAsync(offs.data(), offsets_dev, offs.size() * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    if (offs[0] != 0 || offs.back() != total) return fail(FCPP_ESIZE, "offsets do not span [0, total_points]");
    for (int64_t p = 0; p < n_paths; ++p)
        if (offs[(size_t)p + 1] < offs[(size_t)p]) return fail(FCPP_ESIZE, "offsets must be non-decreasing");
    if (!c->paths_cache || c->paths_cache->offs != offs) {
        PathTiling *pt = new (std::nothrow) PathTiling();
        if (!pt) return fail(FCPP_ENOMEM, "out of host memory");
        Tiling til;
        til.build(n_paths, offs.data());
        hipError_t e = pt->dt.upload(til, c->stream);
        if (e != hipSuccess) { delete pt; return fail(FCPP_EHIP, std::string("tile table upload: ") + hipGetErrorString(e)); }
        pt->offs.swap(offs);
        // (work of earlier calls on the old table has completed: every standalone operator synchronises before it returns)
        delete c->paths_cache;
        c->paths_cache = pt;
    }
    *out = &c->paths_cache->dt;
    return FCPP_OK;
}

}  // namespace
static void free_paths_cache(fcpp_ctx *c) { delete c->paths_cache; c->paths_cache = nullptr; }
namespace {

DevConst const_from_vehicle(const fcpp_vehicle &veh)
{
    fcpp_options o;
    fcpp_options_default(&o);
    return make_const(veh, o);
}
}  // namespace

int fcpp_curvature(fcpp_ctx *c, int64_t n_paths, const int64_t *offsets, int64_t total, const double *x,
                   const double *y, double *kappa, const int64_t *offsets_host)
{
    if (!c || (!offsets && !offsets_host) || (total > 0 && (!x || !y || !kappa))) return fail(FCPP_EINVAL, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    DevTiling *dtp = nullptr;
    int rc = make_tiling(c, n_paths, offsets, offsets_host, total, &dtp);
    if (rc) return rc;
    DevTiling &dt = *dtp;
    fcpp_vehicle veh;
    fcpp_vehicle_default(&veh);
    DevConst cst = const_from_vehicle(veh);
    DevBuf<double> vtmp;
    HIPCHK(vtmp.alloc((size_t)total));
    HIPCHK(hipMemsetAsync(vtmp.p, 0, (size_t)total * sizeof(double), c->stream));
    LAUNCHCHK(launch_curv_clamp(c->stream, dt.n_tiles, dt.tiles.p, dt.paths.p, cst, 0, x, y, vtmp.p, vtmp.p, kappa, nullptr));
    HIPCHK(hipStreamSynchronize(c->stream));
    return FCPP_OK;
}

int fcpp_speed_plan(fcpp_ctx *c, const fcpp_vehicle *veh, int clamp, int64_t n_paths, const int64_t *offsets,
                    int64_t total, const double *x, const double *y, const double *v_in, double *v_out, double *kappa,
                    int64_t *n_adjusted, const int64_t *offsets_host)
{
    if (!c || !veh || (!offsets && !offsets_host) || (total > 0 && (!x || !y || !v_in || !v_out))) return fail(FCPP_EINVAL, "bad arguments");
    if (!(veh->max_longitudinal_accel > 0) || !(veh->max_lateral_accel > 0)) return fail(FCPP_EINVAL, "accelerations must be positive");
    HIPCHK(hipSetDevice(c->device));
    DevTiling *dtp = nullptr;
    int rc = make_tiling(c, n_paths, offsets, offsets_host, total, &dtp);
    if (rc) return rc;
    DevTiling &dt = *dtp;
    DevConst cst = const_from_vehicle(*veh);
    hipStream_t st = c->stream;
    if (n_paths) HIPCHK(hipMemsetAsync(dt.n_adj.p, 0, (size_t)n_paths * sizeof(unsigned long long), st));
    LAUNCHCHK(launch_curv_clamp(st, dt.n_tiles, dt.tiles.p, dt.paths.p, cst, clamp ? 1 : 0, x, y, v_in, v_out, kappa, dt.n_adj.p));
    LAUNCHCHK(launch_scan_tiles(st, dt.n_tiles, dt.tiles.p, dt.paths.p, cst, x, y, v_out, dt.agg_f.p, dt.agg_b.p));
    LAUNCHCHK(launch_scan_spine(st, dt.n_tiles, dt.agg_f.p, dt.agg_b.p, dt.carry_f.p, dt.carry_b.p, dt.spine.p));
    LAUNCHCHK(launch_scan_apply(st, dt.n_tiles, dt.tiles.p, dt.paths.p, cst, clamp ? 3 : 2, x, y, v_out, v_out,
                                dt.carry_f.p, dt.carry_b.p));
    if (n_adjusted && n_paths)
        HIPCHK(hipMemcpyAsync(n_adjusted, dt.n_adj.p, (size_t)n_paths * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    return FCPP_OK;
}

// The standalone operators' statistics: a path per 64 lanes -- or, for a few LONG paths (one path of 6e7 points is 123 000 tiles: 1.2 ms
// through one wavefront), every path sliced over 64 workgroups and joined (the fused pipeline's class-3 reduction: 10 us).
static int reduce_paths(fcpp_ctx *c, hipStream_t st, DevTiling &dt, fcpp_field_stats *stats)
{
    if (dt.n_paths > 0 && dt.n_paths <= 64 && dt.n_tiles / dt.n_paths > 2048) {
        const size_t need = (size_t)dt.n_paths * 64 * 104;
        if (c->verify_scratch_cap < need) {
            if (c->verify_scratch) { HIPCHK(hipStreamSynchronize(st)); (void)hipFree(c->verify_scratch); c->verify_scratch = nullptr; c->verify_scratch_cap = 0; }
            HIPCHK(hipMalloc(&c->verify_scratch, need));
            c->verify_scratch_cap = need;
        }
        LAUNCHCHK(launch_reduce_stats(st, dt.n_paths, dt.partial.p, dt.tile_first.p, nullptr, stats, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 256,
                                      c->verify_scratch, 0));
        return FCPP_OK;
    }
    LAUNCHCHK(launch_reduce_stats(st, dt.n_paths, dt.partial.p, dt.tile_first.p, nullptr, stats));
    return FCPP_OK;
}

int fcpp_verify(fcpp_ctx *c, const fcpp_vehicle *veh, int64_t n_paths, const int64_t *offsets, int64_t total,
                const double *x, const double *y, const double *v, fcpp_field_stats *stats, const int64_t *offsets_host)
{
    if (!c || !veh || (!offsets && !offsets_host) || !stats || (total > 0 && (!x || !y || !v))) return fail(FCPP_EINVAL, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    DevTiling *dtp = nullptr;
    int rc = make_tiling(c, n_paths, offsets, offsets_host, total, &dtp);
    if (rc) return rc;
    DevTiling &dt = *dtp;
    DevConst cst = const_from_vehicle(*veh);
    hipStream_t st = c->stream;
    DevBuf<double> kap, vtmp;
    HIPCHK(kap.alloc((size_t)total));
    HIPCHK(vtmp.alloc((size_t)total));
    LAUNCHCHK(launch_curv_clamp(st, dt.n_tiles, dt.tiles.p, dt.paths.p, cst, 0, x, y, v, vtmp.p, kap.p, nullptr));
    DevObstacles obs = { nullptr, nullptr, nullptr, nullptr };
    LAUNCHCHK(launch_validate(st, dt.n_tiles, dt.tiles.p, dt.paths.p, nullptr, cst, obs, x, y, kap.p, v, nullptr, dt.partial.p));
    { const int rrc = reduce_paths(c, st, dt, stats); if (rrc) return rrc; }
    HIPCHK(hipStreamSynchronize(st));
    return FCPP_OK;
}

int fcpp_validate(fcpp_ctx *c, const fcpp_vehicle *veh, const fcpp_options *opt, int64_t n_paths, const int64_t *offsets, int64_t total,
                  const double *x, const double *y, const double *v, const fcpp_polys *field_polys, const fcpp_polys *obstacles,
                  const int64_t *obstacle_offsets, uint32_t *flags, fcpp_field_stats *stats, const int64_t *offsets_host)
{
    if (!c || !veh || !opt || (!offsets && !offsets_host) || !stats || (total > 0 && (!x || !y || !v || !flags))) return fail(FCPP_EINVAL, "bad arguments");
    if (!isfinite(opt->geofence_tol)) return fail(FCPP_EINVAL, "geofence_tol must be finite");
    std::string err;
    int rc = validate_polys(field_polys, err);
    if (rc == FCPP_OK) rc = validate_polys(obstacles, err);
    if (rc != FCPP_OK) return fail(rc, err);
    if (field_polys && field_polys->n_polys != n_paths) return fail(FCPP_ESIZE, "field_polys must hold one polygon per path");
    const int64_t n_obst = obstacles ? obstacles->n_polys : 0;
    if (obstacle_offsets) {
        if (obstacle_offsets[0] < 0 || obstacle_offsets[n_paths] > n_obst) return fail(FCPP_ESIZE, "obstacle_offsets outside the obstacle table");
        for (int64_t p = 0; p < n_paths; ++p)
            if (obstacle_offsets[p + 1] < obstacle_offsets[p]) return fail(FCPP_ESIZE, "obstacle_offsets must be non-decreasing");
    }
    HIPCHK(hipSetDevice(c->device));
    DevTiling *dtp = nullptr;
    rc = make_tiling(c, n_paths, offsets, offsets_host, total, &dtp);
    if (rc) return rc;
    DevTiling &dt = *dtp;
    DevConst cst = const_from_vehicle(*veh);
    hipStream_t st = c->stream;
    // curvature, a_lat flags and the metrics of fcpp_verify; then the polygon tests
    DevBuf<double> kap, vtmp;
    HIPCHK(kap.alloc((size_t)total));
    HIPCHK(vtmp.alloc((size_t)total));
    LAUNCHCHK(launch_curv_clamp(st, dt.n_tiles, dt.tiles.p, dt.paths.p, cst, 0, x, y, v, vtmp.p, kap.p, nullptr));
    DevObstacles none = { nullptr, nullptr, nullptr, nullptr };
    LAUNCHCHK(launch_validate(st, dt.n_tiles, dt.tiles.p, dt.paths.p, nullptr, cst, none, x, y, kap.p, v, nullptr, dt.partial.p));
    { const int rrc = reduce_paths(c, st, dt, stats); if (rrc) return rrc; }
    // the polygon tables: one upload (field vertices, obstacle vertices, their offsets, the per-path obstacle ranges)
    const int64_t nfv = field_polys && n_paths > 0 ? field_polys->offsets[n_paths] : 0, nov = n_obst > 0 ? obstacles->offsets[n_obst] : 0;
    std::vector<double> hv;
    std::vector<int64_t> hi;
    try {
        hv.reserve((size_t)(2 * (nfv + nov)));
        if (nfv) { hv.insert(hv.end(), field_polys->x, field_polys->x + nfv); hv.insert(hv.end(), field_polys->y, field_polys->y + nfv); }
        if (nov) { hv.insert(hv.end(), obstacles->x, obstacles->x + nov); hv.insert(hv.end(), obstacles->y, obstacles->y + nov); }
        if (field_polys) hi.insert(hi.end(), field_polys->offsets, field_polys->offsets + n_paths + 1);
        if (n_obst) hi.insert(hi.end(), obstacles->offsets, obstacles->offsets + n_obst + 1);
        if (obstacle_offsets && n_obst) hi.insert(hi.end(), obstacle_offsets, obstacle_offsets + n_paths + 1);
    } catch (const std::bad_alloc &) { return fail(FCPP_ENOMEM, "out of host memory"); }
    DevBuf<double> dv;
    DevBuf<int64_t> di;
    HIPCHK(dv.upload(hv, st));
    HIPCHK(di.upload(hi, st));
    const double *fx = dv.p, *fy = dv.p ? dv.p + nfv : nullptr, *ox = dv.p ? dv.p + 2 * nfv : nullptr, *oy = dv.p ? dv.p + 2 * nfv + nov : nullptr;
    const int64_t *foff = field_polys ? di.p : nullptr;
    const int64_t *ooff = n_obst ? di.p + (field_polys ? n_paths + 1 : 0) : nullptr;
    const int64_t *orng = (obstacle_offsets && n_obst) ? ooff + n_obst + 1 : nullptr;
    LAUNCHCHK(launch_validate_polys(st, dt.n_tiles, dt.tiles.p, dt.paths.p, foff, fx, fy, field_polys ? n_paths : 0, ooff, ox, oy, n_obst, orng,
                                    opt->geofence_tol, cst.a_lat, x, y, kap.p, v, flags, stats));
    HIPCHK(hipStreamSynchronize(st));      // (the staging vectors die here)
    return FCPP_OK;
}

int fcpp_straight_segments(fcpp_ctx *c, int64_t n_seg, const double *seg, int32_t n_points, double *out)
{
    if (!c || n_seg < 0 || n_points < 1 || (n_seg > 0 && (!seg || !out))) return fail(FCPP_EINVAL, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    LAUNCHCHK(launch_straight(c->stream, n_seg, seg, n_points, nullptr, out));
    return FCPP_OK;
}

int fcpp_corner_turns(fcpp_ctx *c, const fcpp_vehicle *veh, int64_t n, const double *corners, const int32_t *ci, const int32_t *rev,
                      double L, double H, int32_t stride, double *out, int32_t *counts)
{
    if (!c || !veh || n < 0 || (n > 0 && (!corners || !ci || !rev || !out || !counts))) return fail(FCPP_EINVAL, "bad arguments");
    const double R = veh->min_turn_radius;
    if (!(R > 0)) return fail(FCPP_EINVAL, "min_turn_radius must be positive");
    const int64_t need = 15 + std::max<int64_t>(10, (int64_t)(3.0 * R / 0.5));
    if (stride < need) return fail(FCPP_ESIZE, "stride too small for 15 + max(10, int(3R / 0.5)) points");
    HIPCHK(hipSetDevice(c->device));
    LAUNCHCHK(launch_corner_turns(c->stream, n, corners, ci, rev, R, L, H, stride, out, counts));
    return FCPP_OK;
}

int fcpp_fresnel(fcpp_ctx *c, int64_t n, const double *t, double *cc, double *ss)
{
    if (!c || n < 0 || (n > 0 && (!t || !cc || !ss))) return fail(FCPP_EINVAL, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    LAUNCHCHK(launch_fresnel(c->stream, n, t, cc, ss));
    return FCPP_OK;
}

int fcpp_distance_matrix(fcpp_ctx *c, int32_t n, const double *x, const double *y, double *D)
{
    if (!c || n < 0 || n > 65535 || (n > 0 && (!x || !y || !D))) return fail(FCPP_EINVAL, "bad arguments (0 <= n <= 65535)");
    HIPCHK(hipSetDevice(c->device));
    LAUNCHCHK(launch_distance_matrix(c->stream, n, x, y, D));
    return FCPP_OK;
}

int fcpp_best_connections(fcpp_ctx *c, int64_t n_pairs, const int64_t *fo, const int64_t *to, const double *fx, const double *fy,
                          const double *tx, const double *ty, int32_t *bf, int32_t *bt, double *bd)
{
    if (!c || n_pairs < 0 || n_pairs > 0x7fffffffLL || (n_pairs > 0 && (!fo || !to || !bf || !bt || !bd)))
        return fail(FCPP_EINVAL, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    LAUNCHCHK(launch_best_connections(c->stream, n_pairs, fo, to, fx, fy, tx, ty, bf, bt, bd));
    return FCPP_OK;
}

int fcpp_ga_evolve(fcpp_ctx *c, int32_t n, const fcpp_ga_config *cfg, const double *D, int32_t *routes, int32_t *best_route,
                   double *hist, fcpp_ga_result *result)
{
    if (!c || !cfg || !D || !routes || !best_route || !result) return fail(FCPP_EINVAL, "bad arguments");
    const int pop = cfg->population_size;
    if (n < 2 || n > GA_MAX_NODES) return fail(FCPP_EUNSUPPORTED, "n_nodes must be in [2, 2048]");
    if (pop < 2 || (pop & 1) || cfg->elite_size < 0 || cfg->elite_size >= pop || cfg->tournament_size < 1 ||
        cfg->tournament_size > 64 || cfg->tournament_size > pop || cfg->max_generations < 0)
        return fail(FCPP_EINVAL, "population_size must be even and > elite_size; 1 <= tournament_size <= min(64, population_size)");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    DevBuf<int32_t> scratch;
    DevBuf<double> fd;             // fitness / distance of both buffers
    DevBuf<GaState> state;
    HIPCHK(scratch.alloc((size_t)pop * n));
    HIPCHK(fd.alloc((size_t)pop * 4));
    HIPCHK(state.alloc(1));
    HIPCHK(hipMemsetAsync(state.p, 0, sizeof(GaState), st));
    // the run's progress in a word of pinned memory the bookkeeping role writes (GaState::mirror): followed without draining the stream
    if (!c->ga_mirror && hipHostMalloc((void **)&c->ga_mirror, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { c->ga_mirror = nullptr; (void)hipGetLastError(); }
    volatile unsigned long long *mir = c->ga_mirror;
    void *mirror_dp = nullptr;                      // (the word as the device addresses it; lives until the copy below has read it)
    if (mir) {
        *mir = 0ull;
        if (hipHostGetDevicePointer(&mirror_dp, c->ga_mirror, 0) == hipSuccess && mirror_dp) {
            HIPCHK(hipMemcpyAsync(reinterpret_cast<unsigned char *>(state.p) + offsetof(GaState, mirror), &mirror_dp, sizeof mirror_dp, hipMemcpyHostToDevice, st));
            HIPCHK(hipStreamSynchronize(st));       // (once per run, in front of its first launch)
        } else { mir = nullptr; (void)hipGetLastError(); }
    }
    {   // the initial population must consist of permutations: the kernels index D and their LDS marks by gene
        DevBuf<int32_t> bad;
        int32_t hb = 0;
        HIPCHK(bad.alloc(1));
        HIPCHK(hipMemsetAsync(bad.p, 0, sizeof(int32_t), st));
        LAUNCHCHK(launch_ga_check_perm(st, n, pop, routes, bad.p));
        HIPCHK(hipMemcpyAsync(&hb, bad.p, sizeof hb, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (hb) return fail(FCPP_EINVAL, "routes must hold permutations of 0 .. n_nodes-1");
    }
    int32_t *buf[2] = { routes, scratch.p };
    double *fit[2] = { fd.p, fd.p + 2 * (size_t)pop }, *dist[2] = { fd.p + pop, fd.p + 3 * (size_t)pop };
    LAUNCHCHK(launch_ga_fitness(st, n, pop, D, routes, dist[0], fit[0], 0));                       // GA:64
    GaState h = {};
    // the pipelined convergence check (round 5b): has the bookkeeping of generation g - 63 been done (mirror word: generations >= g - 63)?
    // -> 1: the run has converged, 0: go on, -1: no mirror word / not seen within 2 s (the caller falls back to a copy back + drained stream)
    auto follow = [&](int g) -> int {
        if (!mir) return -1;
        const auto t_wait = std::chrono::steady_clock::now();
        for (int spin = 0;; ++spin) {
            const unsigned long long w = *mir;
            if ((w >> 32) != 0) return 1;
            if ((long long)(w & 0xffffffffull) >= (long long)g - 63) return 0;
            if ((spin & 4095) == 4095 && ms_since(t_wait) > 2000.0) return -1;
        }
    };
    if (ga_generation_fits(n, pop)) {
        // One launch per generation (k_ga_generation): the population in buffer a -> its statistics and elites (the bookkeeping of
        // generation g - 1) and its children (generation g) at once; the launch after the last generation only evaluates the final
        // population.  cfg4: 77 -> 55 us per generation (45 us with the DPP arg-max of the elite selection).
        for (int g = 0; g <= cfg->max_generations; ++g) {
            const int a = g & 1, b = a ^ 1;
            LAUNCHCHK(launch_ga_generation(st, n, pop, D, buf[a], fit[a], dist[a], buf[b], fit[b], dist[b], *cfg, g, state.p, best_route, hist));
            if ((g & 31) == 31) {                      // the kernel is a no-op once converged; stop launching it
                // Round 5b: the check is PIPELINED.  The bookkeeping role of launch L leaves `generations` = L in the mirror word, so the
                // host waits -- on that word, not on the stream -- only until launch g - 63 is through: two blocks of 32 launches are in
                // flight at most, the device never runs dry at a check (it used to: a copy back and a drained stream every 32 generations,
                // ~20 us of idle device each = 0.5 us of a generation's 10), and a converged run is noticed within 64 launches (no-ops by then).
                const int f = follow(g);
                if (f == 1) break;
                if (f == 0) continue;
                HIPCHK(hipMemcpyAsync(&h, state.p, sizeof h, hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                if (h.converged) break;
            }
        }
    } else {
    LAUNCHCHK(launch_ga_stats_elite(st, n, pop, buf[0], fit[0], dist[0], buf[1], fit[1], dist[1], *cfg, -1, state.p, best_route, hist));
    // Larger tours: two launches per generation on two streams.  Population g + 1 (buffer b) = the children k_ga_pairs(g) makes of population g (buffer a) + the elites of population
    // g, which k_ga_stats_elite(g - 1) copied into b's last rows while it evaluated population g.  So k_ga_stats_elite(g) -- statistics
    // of population g + 1, its elites into a's last rows -- and k_ga_pairs(g + 1) -- children of population g + 1 into a's other rows --
    // need the same two predecessors (k_ga_pairs(g), k_ga_stats_elite(g - 1)) and write disjoint rows: they run side by side, the pairs on the context's stream, the
    // single-workgroup bookkeeping (the longer of the two) on its side stream, a generation costs the longer kernel instead of the sum
    // (measured on cfg4, which normally takes the one-launch path: 77 -> 72 us, the cross-stream waits cost ~13 us a generation).  The pairs of the generation that follows convergence may still run (they read the flag at their start):
    // they write the buffer that is NOT the final population.
    struct Events {
        hipEvent_t p[2] = { nullptr, nullptr }, s[2] = { nullptr, nullptr };
        ~Events() { for (int k = 0; k < 2; ++k) { if (p[k]) (void)hipEventDestroy(p[k]); if (s[k]) (void)hipEventDestroy(s[k]); } }
    } ev;
    for (int k = 0; k < 2; ++k) {
        HIPCHK(hipEventCreateWithFlags(&ev.p[k], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ev.s[k], hipEventDisableTiming));
    }
    hipStream_t sd = c->side;
    HIPCHK(hipEventRecord(ev.s[1], st));              // k_ga_stats_elite(-1) ran on the main stream
    HIPCHK(hipStreamWaitEvent(sd, ev.s[1], 0));
    HIPCHK(hipEventRecord(ev.s[0], st));              // (so that every event a stream waits on has been recorded)
    for (int g = 0; g < cfg->max_generations; ++g) {
        const int a = g & 1, b = a ^ 1;               // generation g: buffer a -> buffer b
        HIPCHK(hipStreamWaitEvent(st, ev.s[g & 1], 0));                // the elites of population g are in a: k_ga_stats_elite(g - 2)
        LAUNCHCHK(launch_ga_pairs(st, n, pop, D, buf[a], fit[a], buf[b], fit[b], dist[b], *cfg, g, state.p));
        HIPCHK(hipEventRecord(ev.p[g & 1], st));
        HIPCHK(hipStreamWaitEvent(sd, ev.p[g & 1], 0));                 // the children of population g are in b
        LAUNCHCHK(launch_ga_stats_elite(sd, n, pop, buf[b], fit[b], dist[b], buf[a], fit[a], dist[a], *cfg, g, state.p, best_route, hist));
        HIPCHK(hipEventRecord(ev.s[g & 1], sd));
        if ((g & 31) == 31) {                          // the kernels are no-ops once converged; stop launching them (pipelined check, as above)
            const int f = follow(g);
            if (f == 1) break;
            if (f == 0) continue;
            HIPCHK(hipMemcpyAsync(&h, state.p, sizeof h, hipMemcpyDeviceToHost, sd));
            HIPCHK(hipStreamSynchronize(sd));
            if (h.converged) break;
        }
    }
    HIPCHK(hipStreamSynchronize(sd));
    }
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpyAsync(&h, state.p, sizeof h, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (h.generations & 1)                             // the last completed generation wrote the scratch buffer
        HIPCHK(hipMemcpyAsync(routes, scratch.p, (size_t)pop * n * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    result->generations = h.generations;                               // GA:122-127 (generation + 1)
    result->convergence_gen = h.generations - 1 - h.gwi;
    result->best_distance = h.best_dist;
    result->best_fitness = h.best_fit;
    return FCPP_OK;
}

int fcpp_cover_grid(fcpp_ctx *c, int64_t n_jobs, const fcpp_cover_job *jobs, int64_t n_pts, const double *px, const double *py,
                    uint8_t *grid, int64_t *counts)
{
    if (!c || n_jobs < 0 || (n_jobs > 0 && (!jobs || !counts))) return fail(FCPP_EINVAL, "bad arguments");
    if (n_jobs == 0) return FCPP_OK;
    std::vector<DevCoverJob> dj((size_t)n_jobs);
    int64_t tiles = 0;
    for (int64_t k = 0; k < n_jobs; ++k) {
        const fcpp_cover_job &j = jobs[k];
        if (j.nx < 0 || j.ny < 0 || j.n_a < 0 || j.n_b < 0 || !(j.res > 0) || !(j.radius >= 0) || j.pts_first < 0 ||
            j.pts_first + j.n_a + j.n_b > n_pts || ((j.n_a > 0 || j.n_b > 0) && (!px || !py)) || (j.grid_first >= 0 && !grid))
            return fail(FCPP_EINVAL, "bad coverage job");
        DevCoverJob &d = dj[(size_t)k];
        d.ox = j.ox; d.oy = j.oy; d.res = j.res; d.shift = j.shift; d.radius = j.radius;
        d.nx = j.nx; d.ny = j.ny; d.n_a = j.n_a; d.n_b = j.n_b; d.pts_first = j.pts_first; d.grid_first = j.grid_first;
        d.strict = j.strict; d.region = j.region;
        memcpy(d.outer, j.outer, sizeof d.outer); memcpy(d.inner, j.inner, sizeof d.inner);
        d.tiles_x = (j.nx + 63) / 64; d.tiles_y = (j.ny + 63) / 64;
        d.tile_first = tiles;
        tiles += (int64_t)d.tiles_x * d.tiles_y;
    }
    if (tiles > 0x7fffffffLL) return fail(FCPP_ESIZE, "coverage grids too large for one call");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    DevBuf<DevCoverJob> dev;
    HIPCHK(dev.upload(dj, st));
    HIPCHK(hipMemsetAsync(counts, 0, (size_t)n_jobs * 3 * sizeof(int64_t), st));
    LAUNCHCHK(launch_cover(st, n_jobs, tiles, dev.p, px, py, grid, reinterpret_cast<unsigned long long *>(counts)));
    HIPCHK(hipStreamSynchronize(st));      // the job table dies here
    return FCPP_OK;
}

int fcpp_ga_fitness(fcpp_ctx *c, int32_t n_nodes, int64_t pop, const double *D, const int32_t *routes, double *dist,
                    double *fit, int order_mode)
{
    if (!c || n_nodes < 1 || pop < 0 || (pop > 0 && (!D || !routes)) || (order_mode != 0 && order_mode != 1))
        return fail(FCPP_EINVAL, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    LAUNCHCHK(launch_ga_fitness(c->stream, n_nodes, pop, D, routes, dist, fit, order_mode));
    return FCPP_OK;
}

// ---- the final gather of a sharded job over RCCL -------------------------------------------------------------------------
// RCCL is not linked: its four entry points are looked up in the process (a caller that holds an ncclComm_t has the library loaded;
// PyTorch's ROCm wheels bundle their own copy), then by name.
extern "C++" {
namespace {
struct Rccl {
    int (*group_start)() = nullptr;
    int (*group_end)() = nullptr;
    int (*send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    bool ok() const { return group_start && group_end && send && recv; }
};
const Rccl &rccl()
{
    static const Rccl r = [] {
        Rccl q;
        auto look = [&](void *h) {
            q.group_start = reinterpret_cast<int (*)()>(dlsym(h, "ncclGroupStart"));
            q.group_end = reinterpret_cast<int (*)()>(dlsym(h, "ncclGroupEnd"));
            q.send = reinterpret_cast<int (*)(const void *, size_t, int, int, void *, hipStream_t)>(dlsym(h, "ncclSend"));
            q.recv = reinterpret_cast<int (*)(void *, size_t, int, int, void *, hipStream_t)>(dlsym(h, "ncclRecv"));
        };
        look(RTLD_DEFAULT);
        for (const char *name : { "librccl.so.1", "librccl.so" }) {
            if (q.ok()) break;
            if (void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) look(h);
        }
        return q;
    }();
    return r;
}
}  // namespace
}  // extern "C++"

int fcpp_gather(fcpp_ctx *c, void *nccl_comm, int rank, int world, int root, int n_arrays, const void *const *send_dev, const int32_t *elem_bytes,
                const int64_t *counts_per_rank, void *const *recv_dev, int flags)
{
    if (!c || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || n_arrays < 0 || !counts_per_rank ||
        (n_arrays > 0 && (!send_dev || !elem_bytes)) || (rank == root && n_arrays > 0 && !recv_dev))
        return fail(FCPP_EINVAL, "bad arguments");
    for (int r = 0; r < world; ++r) if (counts_per_rank[r] < 0) return fail(FCPP_ESIZE, "negative count");
    for (int a = 0; a < n_arrays; ++a) if (elem_bytes[a] <= 0) return fail(FCPP_ESIZE, "element size must be positive");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const bool self_through_comm = (flags & 1) != 0;
    const bool need_comm = world > 1 || (self_through_comm && counts_per_rank[rank] > 0);
    if (need_comm && (!nccl_comm || !rccl().ok())) return fail(FCPP_EUNSUPPORTED, nccl_comm ? "RCCL (ncclSend / ncclRecv) is not available in this process" : "nccl_comm is NULL");
    std::vector<int64_t> first((size_t)world + 1, 0);
    for (int r = 0; r < world; ++r) first[(size_t)r + 1] = first[(size_t)r] + counts_per_rank[r];
    // one group: the root posts a receive per peer and array straight into the slice of its full array, every peer one send per array --
    // point to point, each peer over its own xGMI link to the root (no ring, no staging, no concatenation)
    bool grouped = false;
#define NCCLCHK(expr) do { const int e_ = (expr); if (e_ != 0) { if (grouped) (void)rccl().group_end(); return fail(FCPP_EHIP, std::string(#expr) + ": RCCL error " + std::to_string(e_)); } } while (0)
    // (the root's own block: a plain copy, enqueued BEFORE the group opens -- an error return inside an open group would leave the communicator in group mode)
    if (rank == root && !self_through_comm && counts_per_rank[rank] > 0)
        for (int a = 0; a < n_arrays; ++a) {
            const size_t eb = (size_t)elem_bytes[a];
            char *dst = static_cast<char *>(recv_dev[a]) + (size_t)first[(size_t)rank] * eb;
            if (dst != send_dev[a]) HIPCHK(hipMemcpyAsync(dst, send_dev[a], (size_t)counts_per_rank[rank] * eb, hipMemcpyDeviceToDevice, st));
        }
    if (need_comm) { NCCLCHK(rccl().group_start()); grouped = true; }
    for (int a = 0; a < n_arrays; ++a) {
        const size_t eb = (size_t)elem_bytes[a];
        if (rank == root) {
            char *full = static_cast<char *>(recv_dev[a]);
            for (int r = 0; r < world; ++r) {
                const size_t bytes = (size_t)counts_per_rank[r] * eb;
                if (bytes == 0) continue;
                char *dst = full + (size_t)first[(size_t)r] * eb;
                if (r == rank && !self_through_comm) continue;
                else NCCLCHK(rccl().recv(dst, bytes, /* ncclInt8 */ 0, r, nccl_comm, st));
            }
            if (self_through_comm && counts_per_rank[rank] > 0)
                NCCLCHK(rccl().send(send_dev[a], (size_t)counts_per_rank[rank] * eb, 0, rank, nccl_comm, st));
        } else if (counts_per_rank[rank] > 0) {
            NCCLCHK(rccl().send(send_dev[a], (size_t)counts_per_rank[rank] * eb, 0, root, nccl_comm, st));
        }
    }
    if (grouped) { grouped = false; NCCLCHK(rccl().group_end()); }
#undef NCCLCHK
    return FCPP_OK;
}

// ---- diagnostics (tests) -------------------------------------------------------------------------------------------------------------
int fcpp_debug_math(int fn, int64_t n, const double *a, const double *b, double *out0, double *out1)
{
    if (fn < 0 || fn > 5 || n < 0 || (n > 0 && (!a || !out0)) || ((fn == 1 || fn == 3 || fn == 5) && n > 0 && !b) || ((fn == 0 || fn == 4) && n > 0 && !out1))
        return fail(FCPP_EINVAL, "bad arguments");
    for (int64_t i = 0; i < n; ++i) {
        if (fn == 0) fc_sincos(a[i], out0[i], out1[i]);
        else if (fn == 4) fc_sincos_cr(a[i], out0[i], out1[i]);
        else if (fn == 5) out0[i] = fc_atan2_cr(a[i], b[i]);
        else if (fn == 1) out0[i] = atan2_fd(a[i], b[i]);
        else if (fn == 2) out0[i] = fc_acos(a[i]);
        else out0[i] = fc_hypot(a[i], b[i]);
    }
    return FCPP_OK;
}

int fcpp_debug_math_dev(fcpp_ctx *c, int fn, int64_t n, const double *a, const double *b, double *out0, double *out1)
{
    if (!c || fn < 0 || fn > 5 || n < 0) return fail(FCPP_EINVAL, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    LAUNCHCHK(launch_debug_math(c->stream, fn, n, a, b, out0, out1));
    HIPCHK(hipStreamSynchronize(c->stream));
    return FCPP_OK;
}

int fcpp_batch_debug_table(const fcpp_batch *b, int table, void *dst, int64_t cap, int64_t *bytes_out)
{
    if (!b || !bytes_out) return fail(FCPP_EINVAL, "bad arguments");
    const ImageLayout &l = b->lay;
    const size_t n = (size_t)l.n_fields;
    const size_t off[] = { l.fields, l.prims, l.tiles, l.wtiles, l.general_ids, l.chunks, l.span_chunks, l.stat_ids, l.stat_first, l.stat_run, l.red_paths,
                           l.field_work, l.open_wave_ids, l.seg, l.seg_mask, l.partial, l.field_junc, l.work_totals, l.obs_off, l.obs_x, l.obs_y, l.obs_bbox, l.field_packs };
    const size_t len[] = { n * sizeof(DevField), (size_t)l.n_prims * sizeof(DevPrim), (size_t)l.n_tiles * sizeof(DevTile), (size_t)l.n_wave * sizeof(DevWaveTile),
                           (size_t)l.n_general * 4, (size_t)l.n_chunks * sizeof(DevTile), (size_t)l.n_span_chunks * sizeof(DevTile), (size_t)l.n_stat * 4,
                           (n + 1) * 8, (size_t)l.n_stat * 8, n * 4, (size_t)l.n_field_work * sizeof(DevFieldWork), (size_t)l.n_open_wave * 4, n * 64, n * 8,
                           (size_t)l.n_stat * sizeof(TilePartial), n * 16, (size_t)l.n_field_work * sizeof(TilePartial),
                           l.n_polys > 0 ? (size_t)(l.n_polys + 1) * 8 : 0, (size_t)l.n_poly_verts * 8, (size_t)l.n_poly_verts * 8, (size_t)l.n_polys * 32,
                           (size_t)l.n_field_work * sizeof(DevFieldPack) };
    constexpr int kTables = (int)(sizeof(off) / sizeof(off[0]));
    if (table < 0 || table >= kTables) return fail(FCPP_EINVAL, "no such table");
    *bytes_out = (int64_t)len[table];
    if (!dst) return FCPP_OK;
    if (cap < (int64_t)len[table]) return fail(FCPP_ESIZE, "buffer too small");
    if (len[table] == 0 || n == 0) return FCPP_OK;
    HIPCHK(hipSetDevice(b->ctx->device));
    for (hipStream_t s : b->used_streams) HIPCHK(hipStreamSynchronize(s));          // (the setup's stream is one of them)
    HIPCHK(hipStreamSynchronize(b->ctx->stream));
    HIPCHK(hipMemcpy(dst, static_cast<const unsigned char *>(b->slab) + off[table], len[table], hipMemcpyDeviceToHost));
    return FCPP_OK;
}

}  // extern "C"
