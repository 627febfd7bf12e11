// fcpp_api.cpp -- the C ABI declared in include/fcpp.h: argument checking, device buffers, launches.
// No CPU compute path exists here: every operator ends in a HIP kernel launch or fails with FCPP_EHIP.
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "fcpp_cover.h"
#include "fcpp_ga.h"
#include "fcpp_device.h"
#include "fcpp_internal.h"

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
    return c;
}

// tile table of a set of paths
struct Tiling {
    std::vector<DevPath> paths;
    std::vector<DevTile> tiles;
    std::vector<int64_t> tile_first;
    std::vector<int64_t> pass_len;      // per path: points per pass of layer 1 (swath line + U-turn), 0 without structure
    // structure of a field's path, for the fused pipeline's tiling
    struct QuietInfo {
        int64_t n_line, n_turn, P, n_main, gen_main;
        double line_step_len, c_line;          // |numpy step| of the swath lines, their nominal u = (v/3.6)^2
        const DevPrim *prims; int prim_count, prim_index0;  // layer 2: the field's primitives, index of the first one in the batch
        double two_a;
        bool enable;
        bool turn_quiet;       // U-turns are closed form and swath lines are isolated by them (see fcpp_batch_create)
        // geometry for the wave tiles of the sparse kernel (fcpp_sparse.hip): the field, host copies of the batch's turn templates
        const DevField *df = nullptr;
        const double2 *tu = nullptr, *tc = nullptr;
        double u_cap = 0.0;
        double fence_margin = 1e-3;   // a point this far inside every edge of the field polygon cannot be flagged by the device's test
        bool wave_ok = false;  // sampling sparse enough for halos of a few lanes
    };

    // Host evaluation of path point i of a field (same formulas as eval_main / eval_prim in fcpp_pointfn.h, from the host copies of
    // the templates).  Only distances between consecutive points are taken from it, to size the halos of the wave tiles.
    static void host_point(const QuietInfo &q, int64_t i, double &px, double &py)
    {
        const DevField &f = *q.df;
        if (i < f.gen_main) {
            const int64_t per = (int64_t)f.n_line + f.n_turn;
            const int64_t idx = i / per, off = i - idx * per;
            const int64_t pi = f.reverse_order ? (f.P - 1 - idx) : idx;
            const double y = f.min_y + (double)pi * f.W;
            const bool go_left = f.start_from_right ? ((idx & 1) == 0) : ((idx & 1) == 1);
            if (off < f.n_line) {
                px = go_left ? linspace_at(f.lex, f.lsx, -f.line_step, f.n_line, off) : linspace_at(f.lsx, f.lex, f.line_step, f.n_line, off);
                py = y;
            } else {
                const double2 t = q.tu[off - f.n_line];
                const bool turn_right = !go_left;
                if (f.turn_model == FCPP_TURN_ARC) px = turn_right ? (f.max_x - t.x) : (f.min_x + t.x);
                else px = turn_right ? ((f.max_x - f.R) + t.x) : ((f.min_x + f.R) - t.x);
                py = y + t.y;
            }
            if (f.rotated) {
                const double tx = px - f.rot_cx, ty = py - f.rot_cy;
                px = (tx * f.rot_cos - ty * f.rot_sin) + f.rot_cx;
                py = (tx * f.rot_sin + ty * f.rot_cos) + f.rot_cy;
            }
            return;
        }
        const DevPrim &p = q.prims[prim_of(q, i)];
        const int64_t r = i - p.start;
        if (p.kind == PRIM_LINSPACE) { px = linspace_at(p.a[0], p.a[2], p.a[4], p.n, r); py = linspace_at(p.a[1], p.a[3], p.a[5], p.n, r); }
        else if (p.kind == PRIM_POINT) { px = p.a[0]; py = p.a[1]; }
        else if (p.kind == PRIM_RAY) { const double t = linspace_at(0.0, p.a[4], p.a[5], p.n, r); px = p.a[0] + t * p.a[2]; py = p.a[1] + t * p.a[3]; }
        else if (p.kind == PRIM_UTURN) {
            const double2 t = q.tu[r];
            const bool turn_right = p.form & 1;
            if (!(p.form & 4)) px = turn_right ? (p.a[0] - t.x) : (p.a[0] + t.x);
            else px = turn_right ? (p.a[0] + t.x) : (p.a[0] - t.x);
            py = p.a[1] + t.y;
            if (p.form & 2) {
                const double tx = px - p.a[4], ty = py - p.a[5];
                px = (tx * p.a[2] - ty * p.a[3]) + p.a[4];
                py = (tx * p.a[3] + ty * p.a[2]) + p.a[5];
            }
        }
        else {
            const double2 t = q.tc[r];
            const int ci = p.kind == PRIM_ARC ? p.form : ((p.form + 3) & 3);
            if (ci == 0)      { px = p.a[0] + t.x; py = p.a[1] + t.y; }
            else if (ci == 1) { px = p.a[0] - t.y; py = p.a[1] + t.x; }
            else if (ci == 2) { px = p.a[0] - t.x; py = p.a[1] - t.y; }
            else              { px = p.a[0] + t.y; py = p.a[1] - t.x; }
        }
    }
    static int prim_of(const QuietInfo &q, int64_t i)      // index (within the field) of the headland primitive that holds point i
    {
        int a = 0, b = q.prim_count - 1;
        while (a < b) { const int m = (a + b + 1) >> 1; if (q.prims[m].start <= i) a = m; else b = m - 1; }
        return a;
    }

    // Wave tiles of the sparse kernel for the general stretch [a, b) of path p: [ Hb halo | count outputs | Hf halo ] <= WAVE_LANES lanes.
    // The halos are sized from the path's own step lengths (see fcpp_sparse.hip): backwards from the point before the first output
    // (whose final speed the segment metrics need) until the couplings 2a|dp| add up to u_cap, a skipped step or the path's start,
    // plus one lane for the stencil of the outermost point; forwards likewise from the last output.  Every tile takes as many
    // outputs as fit.  false = some tile would hold fewer than 8 outputs (dense sampling): the stretch stays with k_plan_fused.
    static constexpr int WAVE_HALO_MAX = 40;
    static constexpr int WAVE_LANES = 64;                // fcpp_sparse.hip: one wavefront per wave tile
    mutable int64_t wave_fail[5] = { 0, 0, 0, 0, 0 };   // diagnostics (FCPP_DEBUG_TILING): stretches that did not fit, by reason
    mutable std::vector<double> wave_d;                  // scratch: step lengths of the stretch and its surroundings
    mutable std::vector<double> wave_x, wave_y;          // scratch: the points themselves (index i - (lo - 1))
    mutable std::vector<DevWaveTile> wtiles;             // the wave tiles' records, in the order of their DevTile entries
    bool wave_tiles(int64_t p, const QuietInfo &q, int64_t a, int64_t b, std::vector<DevTile> &out) const
    {
        const DevField &f = *q.df;
        const int64_t n = f.n_total, per = (int64_t)f.n_line + f.n_turn;
        const double cap = q.u_cap * (1.0 + 1e-9) + 1e-12;
        // d[i - lo] = |p_i - p_(i-1)| for the stretch and WAVE_HALO_MAX + 2 points either side
        const int64_t lo = std::max<int64_t>(a - WAVE_HALO_MAX - 2, 1), hi = std::min<int64_t>(b + WAVE_HALO_MAX + 2, n);   // i in [lo, hi)
        if (hi - lo > (int64_t)1 << 22) { ++wave_fail[4]; return false; }
        std::vector<double> &d = wave_d;
        d.resize((size_t)std::max<int64_t>(hi - lo, 0));
        std::vector<double> &hx = wave_x, &hy = wave_y;
        hx.resize(d.size() + 1); hy.resize(d.size() + 1);
        {
            double x0, y0, x1, y1;
            if (hi > lo) { host_point(q, lo - 1, x0, y0); hx[0] = x0; hy[0] = y0; }
            for (int64_t i = lo; i < hi; ++i) {
                host_point(q, i, x1, y1);
                const double dx = x1 - x0, dy = y1 - y0;
                d[(size_t)(i - lo)] = sqrt(dx * dx + dy * dy);
                hx[(size_t)(i - lo + 1)] = x1; hy[(size_t)(i - lo + 1)] = y1;
                x0 = x1; y0 = y1;
            }
        }
        // every output point of [s, s + c) well inside the field polygon?  (host and device evaluate a point with the same formulas;
        // their roundings differ by ~1e-12 m, the margin is a millimetre: the device's test of such a point cannot fire)
        auto all_inside = [&](int64_t s, int64_t c) -> bool {
            for (int64_t i = s; i < s + c; ++i) {
                if (i < lo - 1 || i >= hi) return false;
                const double px = hx[(size_t)(i - lo + 1)], py = hy[(size_t)(i - lo + 1)];
                for (int e = 0; e < 4; ++e)
                    if (!(f.ex[e] * px + f.ey[e] * py + f.eo[e] >= q.fence_margin)) return false;
            }
            return true;
        };
        auto dist = [&](int64_t i) { return d[(size_t)(i - lo)]; };     // lo <= i < hi by the halo bound below
        // (a step within 0.1 % of the 1e-6 threshold counts neither as skipped nor as a coupling)
        auto back_halo = [&](int64_t s) -> int {
            if (s == 0) return 0;
            const int64_t j = s - 1;
            int64_t m = j;
            double acc = 0.0;
            for (;;) {
                if (m == 0) return (int)(j + 1);
                const double dm = dist(m);
                if (dm < 0.999e-6) return (int)(j - (m - 1) + 1);
                if (dm > 1.001e-6) acc += q.two_a * dm;
                --m;
                if (acc >= cap) return (int)(j - m + 1);
                if (j - m + 1 > WAVE_HALO_MAX) return -1;
            }
        };
        auto fwd_halo = [&](int64_t e) -> int {
            if (e == n - 1) return 0;
            int64_t m = e;
            double acc = 0.0;
            for (;;) {
                const double dm = dist(m + 1);
                if (dm < 0.999e-6) return (int)(m + 1 - e);
                if (dm > 1.001e-6) acc += q.two_a * dm;
                ++m;
                if (m == n - 1 || acc >= cap) return (int)(m - e);
                if (m - e > WAVE_HALO_MAX) return -1;
            }
        };
        const size_t mark = out.size(), mark_w = wtiles.size();
        for (int64_t s = a; s < b;) {
            const int Hb = back_halo(s);
            if (Hb < 0) { ++wave_fail[0]; out.resize(mark); wtiles.resize(mark_w); return false; }
            // the largest count whose forward halo still fits
            int64_t c = std::min<int64_t>(b - s, WAVE_LANES - Hb);
            int Hf = -1;
            for (; c >= 1; --c) {
                Hf = fwd_halo(s + c - 1);
                if (Hf >= 0 && Hb + c + Hf <= WAVE_LANES) break;
            }
            if (c < std::min<int64_t>(8, b - s)) { ++wave_fail[Hf < 0 ? 1 : 2]; out.resize(mark); wtiles.resize(mark_w); return false; }
            const int64_t first = s - Hb, last = s + c - 1 + Hf;
            DevTile t;
            t.field = (int32_t)p; t.count = (int32_t)c; t.start = s; t.quiet = 5; t.stat_tile = Hb | (Hf << 16);
            if (first < f.gen_main) { t.idx0 = (int32_t)(first / per); t.off0 = (int32_t)(first % per); }
            else { t.idx0 = q.prim_index0 + prim_of(q, first); t.off0 = 0; }
            // the self-contained record: layer-1 decode of lane 0, and where the (at most 8) further primitives start among the lanes
            DevWaveTile wt;
            memset(&wt, 0, sizeof wt);
            auto clampi = [](int64_t v) { return (int32_t)std::max<int64_t>(-2, std::min<int64_t>(v, (int64_t)1 << 30)); };
            wt.out_base = f.pt_off + first; wt.field = (int32_t)p; wt.tile = (int32_t)out.size();
            wt.count = (uint8_t)c; wt.hb = (uint8_t)Hb; wt.hf = (uint8_t)Hf; wt.inside = all_inside(s, c) ? 1 : 0;
            wt.rel_main = clampi(f.gen_main - first); wt.rel_seam = clampi(f.n_main - first); wt.rel_last = clampi(n - 1 - first);
            wt.rel_zero = clampi(-first);
            wt.idx0 = t.idx0; wt.off0 = t.off0;
            for (int k = 0; k < 8; ++k) wt.thr[k] = 255;
            if (last >= f.gen_main) {
                const int64_t fl2 = std::max<int64_t>(first, f.gen_main);      // first primitive-generated point of the tile
                const int pa = prim_of(q, fl2), pb = prim_of(q, last);
                if (pb - pa > 8) { ++wave_fail[3]; out.resize(mark); wtiles.resize(mark_w); return false; }
                wt.p0 = q.prim_index0 + pa;
                wt.r0 = (int32_t)(first - q.prims[pa].start);
                for (int k = pa + 1; k <= pb; ++k) wt.thr[k - pa - 1] = (uint8_t)(q.prims[k].start - first);
            }
            wtiles.push_back(wt);
            out.push_back(t);
            s += c;
        }
        return true;
    }

    // Tiles never straddle paths and hold at most TILE_POINTS points.  Without structure information (standalone operators)
    // a path is cut into near-equal tiles.  With it (planner batches) every straight primitive -- swath lines of layer 1,
    // headland straights of layer 2 -- is cut as
    //     [ need | quiet zone ............................. | need ]
    // quiet zone = samples whose sweep neighbourhood stays on the straight: `need` samples span u_nominal / (2a) metres, the
    // farthest a slower point can lower speeds that are nominal for this straight.  Quiet zones become "quiet" tiles
    // (closed-form kernel); everything else -- turns, corner arcs, reverse fills and the margins around them -- becomes
    // general tiles.  The cut depends only on the field itself, never on its position in the batch.
    void build(int64_t n_paths, const int64_t *offsets, const QuietInfo *quiet = nullptr)
    {
        paths.resize((size_t)n_paths);
        tile_first.assign((size_t)n_paths + 1, 0);
        pass_len.assign((size_t)n_paths, 0);
        tiles.clear();
        wtiles.clear();
        for (int64_t p = 0; p < n_paths; ++p) {
            const int64_t n = offsets[p + 1] - offsets[p];
            paths[(size_t)p] = { offsets[p], n };
            tile_first[(size_t)p] = (int64_t)tiles.size();
            const QuietInfo *q = (quiet && quiet[p].enable) ? &quiet[p] : nullptr;
            const int64_t per = q ? q->n_line + q->n_turn : 0;
            pass_len[(size_t)p] = per;
            auto emit = [&](int64_t s, int64_t cnt, int kind, int64_t i0, int64_t o0) {
                DevTile t;
                t.field = (int32_t)p; t.start = s; t.count = (int32_t)cnt; t.quiet = kind; t.stat_tile = 0;
                t.idx0 = (int32_t)i0; t.off0 = (int32_t)o0;
                tiles.push_back(t);
            };
            auto emit_general = [&](int64_t a, int64_t b) {
                const int64_t len = b - a;
                if (len <= 0) return;
                if (q && q->wave_ok && wave_tiles(p, *q, a, b, tiles)) return;
                const int64_t k = (len + TILE_POINTS - 1) / TILE_POINTS, base = len / k, rem = len % k;
                for (int64_t i = 0; i < k; ++i) {
                    const int64_t c = base + (i < rem ? 1 : 0);
                    const bool in1 = per > 0 && a < q->gen_main;      // layer-1 decode of the tile start for the general kernel
                    emit(a, c, 0, in1 ? a / per : 0, in1 ? a % per : 0);
                    a += c;
                }
            };
            // near-equal quiet tiles of at most TILE_POINTS - 2 points (the kernel stores aligned PAIRS; a tile that starts on an
            // odd global index needs one pair more than half its points)
            auto emit_quiet = [&](int64_t zs, int64_t Z, int kind, int64_t i0, int64_t o0) {
                const int64_t cap = TILE_POINTS - 2, k = (Z + cap - 1) / cap, base = Z / k, rem = Z % k;
                for (int64_t i = 0; i < k; ++i) { const int64_t c = base + (i < rem ? 1 : 0); emit(zs, c, kind, i0, o0); zs += c; o0 += c; }
            };
            auto need_for = [&](double c_nom, double step_len) -> int64_t {
                if (!(step_len >= 1e-6)) return -1;
                return (int64_t)(c_nom / (q->two_a * step_len)) + 3;
            };
            int64_t pos = 0;
            if (q) {
                const int64_t need1 = need_for(q->c_line, q->line_step_len);
                if (need1 >= 0 && per > 0 && q->gen_main > 0) {
                    // With closed-form U-turns nothing propagates into a swath line from the turns around it (a turn starts on
                    // the line's last point: a skipped step; the jump back from the turn's end is too long to bind): all
                    // complete passes (line + turn) form ONE quiet span, whatever the sampling.  The last line ends at the
                    // seam to the headland layer: it is cut like any other straight, without a margin at its start.
                    // Dense sampling keeps lines and turns as runs of their own (cheaper per point: no pass decode); the span is for
                    // short lines -- the reference's own sampling has 2 points per line and 20 per turn.
                    int64_t first_idx = 0;
                    const bool span = q->turn_quiet && q->P >= 2 && q->n_line - need1 < 64 && (q->P - 1) * per < (int64_t)0x7fffffff;
                    if (span) {
                        const int64_t S = (q->P - 1) * per, cap = TILE_POINTS - 2, k = (S + cap - 1) / cap, base = S / k, rem = S % k;
                        int64_t a = 0;
                        for (int64_t i = 0; i < k; ++i) { const int64_t c = base + (i < rem ? 1 : 0); emit(a, c, 4, a / per, a % per); a += c; }
                        pos = S; first_idx = q->P - 1;
                    }
                    for (int64_t idx = first_idx; idx < q->P; ++idx) {
                        // closed-form turns, dense sampling: the whole line is a quiet run, and so is the turn after it
                        const bool full = q->turn_quiet && !span;
                        const int64_t need_s = (full || (span && idx > 0)) ? 0 : need1, need_e = (full && idx < q->P - 1) ? 0 : need1;
                        const int64_t L0 = idx * per, zs = L0 + need_s, Z = q->n_line - need_s - need_e;
                        if (Z < 64) break;
                        emit_general(pos, zs);
                        emit_quiet(zs, Z, 1, idx, need_s);
                        pos = zs + Z;
                        if (full && idx < q->P - 1) { emit_quiet(L0 + q->n_line, q->n_turn, 3, idx, 0); pos = L0 + per; }
                    }
                }
                for (int k = 0; k < q->prim_count; ++k) {
                    const DevPrim &pr = q->prims[k];
                    if (pr.kind != PRIM_LINSPACE) continue;
                    const double ms = pr.v_nom / 3.6;
                    const int64_t need2 = need_for(ms * ms, sqrt(pr.a[4] * pr.a[4] + pr.a[5] * pr.a[5]));
                    if (need2 < 0) continue;
                    const int64_t zs = pr.start + need2, Z = (int64_t)pr.n - 2 * need2;
                    if (Z < 64 || zs < pos) continue;
                    emit_general(pos, zs);
                    emit_quiet(zs, Z, 2, q->prim_index0 + k, need2);
                    pos = zs + Z;
                }
            }
            emit_general(pos, n);
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
    DevBuf<int32_t> general_ids;   // fused pipeline: the tiles of k_plan_fused
    DevBuf<DevWaveTile> wave_tiles; // ... the wave tiles of k_plan_sparse (self-contained records)
    DevBuf<DevTile> chunks;        // ... the quiet runs cut on 512-point boundaries of the batch arrays (k_plan_quiet)
    DevBuf<DevTile> span_chunks;   // ... the same for the layer-1 spans (their own kernel instance)
    DevBuf<int32_t> stat_ids;      // ... the tiles that can hold statistics (general tiles, first tile of every run), path by path
    DevBuf<int64_t> stat_first;    //     CSR offsets into stat_ids per path
    DevBuf<int64_t> stat_run;      //     per entry of stat_ids: points of the quiet run that starts there (0: not a run)
    DevBuf<int32_t> red_paths;     //     the paths by their number of entries: [<= 64 | <= 256 | <= 1024 | more] (k_reduce_stats)
    DevBuf<char> red_scratch;      //     slice results of the paths of the last class (64 x 104 bytes each)
    int64_t n_red[4] = { 0, 0, 0, 0 };
    int64_t n_tiles = 0, n_paths = 0, n_chunks = 0, n_span_chunks = 0, n_runs = 0, n_general = 0, n_wave = 0, quiet_points = 0;
    int64_t span_points = 0, chunk_points = 0, wave_points = 0;
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
        std::vector<int32_t> gv, sv;
        std::vector<int64_t> srun;
        std::vector<int64_t> sf((size_t)n_paths + 1, 0);
        wave_points = 0;
        std::vector<DevTile> cv, cs;
        std::vector<DevRun> rv;
        quiet_points = 0;
        for (size_t i = 0; i < t.tiles.size();) {
            const DevTile &t0 = t.tiles[i];
            sv.push_back((int32_t)i);                 // a general tile, a wave tile, or the first tile of a run
            srun.push_back(0);
            sf[(size_t)t0.field + 1] = (int64_t)sv.size();
            if (!t0.quiet) { gv.push_back((int32_t)i); ++i; continue; }
            if (t0.quiet == 5) { wave_points += t0.count; ++i; continue; }       // (its record: t.wtiles)
            // the run: quiet tiles that continue each other on the same straight
            int64_t cnt = t0.count;
            size_t j = i + 1;
            for (; j < t.tiles.size(); ++j) {
                const DevTile &tj = t.tiles[j];
                const bool cont = tj.quiet == t0.quiet && tj.field == t0.field && tj.start == t0.start + cnt &&
                                  (t0.quiet == 4 || (tj.idx0 == t0.idx0 && (int64_t)tj.off0 == (int64_t)t0.off0 + cnt));
                if (!cont) break;
                cnt += tj.count;
            }
            rv.push_back({ (int32_t)i, 0, cnt });
            srun.back() = cnt;
            quiet_points += cnt;
            i = j;
        }
        // Chunks: every run is cut on 512-point boundaries of the batch arrays.  Consecutive layer-1 runs (swath line, U-turn, swath
        // line, ...) are cut TOGETHER: a chunk that holds the end of one run and the start of the next is written by one wave through
        // the span decode (kind 4) instead of two partial chunks (measured 4-5 % on the streaming kernel on identical memory; the mixed
        // chunks in the same launch as the others or in the span instance's launch: no difference).
        const bool merge_runs = true;
        for (size_t r = 0; r < rv.size();) {
            const DevTile &t0 = t.tiles[(size_t)rv[r].tile];
            size_t r1 = r + 1;
            int64_t total = rv[r].count;
            if (merge_runs && (t0.quiet == 1 || t0.quiet == 3))
                for (; r1 < rv.size(); ++r1) {
                    const DevTile &tn = t.tiles[(size_t)rv[r1].tile];
                    if (!((tn.quiet == 1 || tn.quiet == 3) && tn.field == t0.field && tn.start == t0.start + total)) break;
                    total += rv[r1].count;
                }
            const int64_t g_grp = t.paths[(size_t)t0.field].off + t0.start, per = t.pass_len[(size_t)t0.field];
            size_t rc = r;                       // run that holds the current position
            int64_t rc_begin = 0;                // its first point, relative to the group
            for (int64_t done = 0; done < total;) {
                const int64_t g = g_grp + done;
                // (also for the short spans of sparse sampling, where one chunk in eight is partial: near-equal chunks from the span's
                // start, i.e. 12 % fewer waves with unaligned stores, took 1.81 instead of 1.50 ms on cfg5)
                const int64_t c = std::min<int64_t>(total - done, TILE_POINTS - (g % TILE_POINTS));
                while (done >= rc_begin + rv[rc].count) { rc_begin += rv[rc].count; ++rc; }
                const DevTile &tr = t.tiles[(size_t)rv[rc].tile];
                DevTile ch = tr;
                ch.start = t0.start + done; ch.count = (int32_t)c; ch.stat_tile = rv[rc].tile;
                const bool one_run = done + c <= rc_begin + rv[rc].count;
                if (one_run && tr.quiet != 4) ch.off0 = (int32_t)(tr.off0 + (done - rc_begin));
                else {      // a span of layer 1 (or a chunk across runs): (pass, offset in the pass) of the chunk's first point
                    ch.quiet = 4;
                    ch.idx0 = (int32_t)(ch.start / per); ch.off0 = (int32_t)(ch.start % per);
                }
                (ch.quiet == 4 ? cs : cv).push_back(ch);
                done += c;
            }
            r = r1;
        }
        n_chunks = (int64_t)cv.size(); n_span_chunks = (int64_t)cs.size(); n_runs = (int64_t)rv.size(); n_general = (int64_t)gv.size();
        n_wave = (int64_t)t.wtiles.size();
        span_points = chunk_points = 0;
        for (const DevTile &c : cs) span_points += c.count;
        for (const DevTile &c : cv) chunk_points += c.count;
        if ((e = wave_tiles.upload(t.wtiles, st)) != hipSuccess) return e;
        if ((e = chunks.upload(cv, st)) != hipSuccess) return e;
        if ((e = span_chunks.upload(cs, st)) != hipSuccess) return e;
        if ((e = general_ids.upload(gv, st)) != hipSuccess) return e;
        for (size_t p = 1; p < sf.size(); ++p) sf[p] = std::max(sf[p], sf[p - 1]);      // paths without tiles
        if ((e = stat_ids.upload(sv, st)) != hipSuccess) return e;
        if ((e = stat_first.upload(sf, st)) != hipSuccess) return e;
        if ((e = stat_run.upload(srun, st)) != hipSuccess) return e;
        {   // classes of the reduction: by the number of entries of each path (a property of the field alone)
            // (8 lanes, a wavefront, a workgroup, 64 workgroups per path: at most 8 / 4 / 4 entries per lane in the first three)
            std::vector<int32_t> cls[4];
            // (one workgroup walks up to 1024 entries; beyond that 64 workgroups + a join launch are faster: cfg3, 3900 entries, 21.8 -> 8 us)
            const int64_t wg_max = tune_int("FCPP_REDUCE_WG_MAX", 1024);
            for (int64_t p = 0; p < n_paths; ++p) {
                const int64_t ne = sf[(size_t)p + 1] - sf[(size_t)p];
                cls[ne <= 64 ? 0 : (ne <= 256 ? 1 : (ne <= wg_max ? 2 : 3))].push_back((int32_t)p);
            }
            std::vector<int32_t> all;
            for (int c = 0; c < 4; ++c) { n_red[c] = (int64_t)cls[c].size(); all.insert(all.end(), cls[c].begin(), cls[c].end()); }
            if ((e = red_paths.upload(all, st)) != hipSuccess) return e;
            if ((e = red_scratch.alloc((size_t)n_red[3] * 64 * 104)) != hipSuccess) return e;
        }
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;   // the staging vectors die here
        return hipSuccess;
    }
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
};

struct fcpp_batch {
    fcpp_ctx *ctx = nullptr;
    fcpp_vehicle veh;
    fcpp_options opt;
    int64_t n_fields = 0;
    HostPlan hp;
    DevConst cst;
    DevBuf<DevField> fields;
    DevBuf<DevPrim> prims;
    DevTiling til;             // fused pipeline (mode 1): quiet runs, wave tiles, general tiles
    DevTiling til0;            // staged pipeline (mode 0): plain near-equal tiles, built on its first run
    bool til0_built = false;
    DevBuf<int64_t> obs_off;
    DevBuf<double> obs_x, obs_y, obs_bbox;
    DevBuf<CacShape> shapes;   // [0] 180-degree, [1] 90-degree clothoid-arc-clothoid unit shapes
    DevBuf<double2> tmpl_u, tmpl_c, tmpl_u_dk;   // sampled turn templates (fcpp_fused.hip)
    DevBuf<double2> field_junc;                  // per field: line-start curvature and jump length after a U-turn
    DevBuf<double> seg;        // connector segments
    DevBuf<int32_t> seg_mask;
    // optional per-stage HIP-event timing (fcpp_batch_set_profiling)
    int profiling = 0;           // 0: off; k > 0: every k-th run carries the per-kernel events
    int64_t run_counter = 0;
    std::vector<hipEvent_t> events;   // kProfRuns x kStages x (start, stop)
    std::vector<unsigned char> ev_set; // kProfRuns x kStages: the stage launched a kernel in that run
    int prof_runs = 0;
    int last_mode = 0;
    bool partial_dirty = true;   // the fused pipeline's tile partials have not been zeroed yet
    bool two_streams = true;     // ALU-bound kernels of a step on the context's side stream (FCPP_ONE_STREAM=1 in the environment: off)
    ~fcpp_batch() { for (hipEvent_t e : events) (void)hipEventDestroy(e); }
};

namespace {
constexpr int kStages = 7;
constexpr int kProfRuns = 256;
const char *const kStageNames[2][kStages] = {
    { "k_generate", "k_curv_clamp", "k_scan_tiles", "k_scan_spine", "k_scan_apply", "k_validate", "k_reduce_stats" },
    { "k_plan_quiet_spans", "k_plan_quiet", "k_plan_sparse", "k_plan_fused", "k_reduce_stats", "", "" } };
const int kStageCount[2] = { 7, 5 };
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
        c.turn_max_kappa[v] = std::max(maxk, k_last);
        c.turn_max_jump[v] = std::max(std::max(maxj, fabs(dk[(size_t)nu - 2].y - (nu >= 3 ? dk[(size_t)nu - 3].y : 0.0))), fabs(k_last - dk[(size_t)nu - 2].y));
    }
    c.turn_len = len;
    c.turn_time = len / std::max(c.ms_turn, 0.1);
    return true;
}

static void free_paths_cache(fcpp_ctx *c);

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
    o->geofence_tol = 1e-6; o->obstacle_mode = FCPP_OBSTACLES_FLAG; o->_pad = 0;
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
    *out = c;
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

int fcpp_plan_count(const fcpp_vehicle *veh, const fcpp_options *opt, int64_t n_fields, const fcpp_field *fields,
                    const fcpp_polys *obstacles, fcpp_field_info *info_out)
{
    if (!veh || !opt || n_fields < 0 || (n_fields > 0 && (!fields || !info_out)))
        return fail(FCPP_EINVAL, "bad arguments");
    HostPlan hp;
    std::string err;
    int rc = build_host_plan(*veh, *opt, n_fields, fields, obstacles, false, hp, err);
    if (rc != FCPP_OK) return fail(rc, err);
    if (n_fields) memcpy(info_out, hp.info.data(), (size_t)n_fields * sizeof(fcpp_field_info));
    return FCPP_OK;
}

int fcpp_batch_create(fcpp_ctx *c, const fcpp_vehicle *veh, const fcpp_options *opt, int64_t n_fields,
                      const fcpp_field *fields, const fcpp_polys *obstacles, fcpp_batch **out)
{
    if (!c || !veh || !opt || !out || n_fields < 0 || (n_fields > 0 && !fields))
        return fail(FCPP_EINVAL, "bad arguments");
    if (n_fields > INT32_MAX) return fail(FCPP_ESIZE, "too many fields");
    *out = nullptr;
    HIPCHK(hipSetDevice(c->device));
    fcpp_batch *b = new (std::nothrow) fcpp_batch();
    if (!b) return fail(FCPP_ENOMEM, "out of host memory");
    b->ctx = c; b->veh = *veh; b->opt = *opt; b->n_fields = n_fields;
    b->two_streams = getenv("FCPP_ONE_STREAM") == nullptr;
    std::string err;
    int rc = build_host_plan(*veh, *opt, n_fields, fields, obstacles, true, b->hp, err);
    if (rc != FCPP_OK) { delete b; return fail(rc, err); }
    // obstacle references must stay inside the polygon table
    const int64_t n_polys = obstacles ? obstacles->n_polys : 0;
    if (n_polys < 0 || n_polys > INT32_MAX) { delete b; return fail(FCPP_ESIZE, "bad polygon count"); }
    if (n_polys > 0) {      // CSR table: offsets[0] = 0, non-decreasing, coordinates present
        if (!obstacles->offsets || obstacles->offsets[0] != 0) { delete b; return fail(FCPP_ESIZE, "polygon offsets must start at 0"); }
        for (int64_t k = 0; k < n_polys; ++k)
            if (obstacles->offsets[k + 1] < obstacles->offsets[k]) { delete b; return fail(FCPP_ESIZE, "polygon offsets must be non-decreasing"); }
        if (obstacles->offsets[n_polys] > 0 && (!obstacles->x || !obstacles->y)) { delete b; return fail(FCPP_EINVAL, "polygon coordinates are NULL"); }
    }
    for (int64_t i = 0; i < n_fields; ++i) {
        if (fields[i].n_obstacles < 0 || fields[i].obstacle_first < 0 ||
            (fields[i].n_obstacles > 0 && fields[i].obstacle_first + fields[i].n_obstacles > n_polys)) {
            delete b;
            return fail(FCPP_ESIZE, "field obstacle range outside the polygon table");
        }
    }
    b->cst = make_const(*veh, *opt);
    hipStream_t st = c->stream;
    std::vector<CacShape> shp = { make_cac_shape(kPi, opt->clothoid_frac), make_cac_shape(kHalfPi, opt->clothoid_frac) };
    // turn templates first: the tiler needs to know whether the U-turns of this batch are closed form
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return r == hipSuccess; };
    ok(b->shapes.upload(shp, st));
    b->cst.shapes = b->shapes.p;
    bool turn_quiet = false;
    std::vector<double2> h_tu, h_tc;      // host copies of the turn templates (closed-form test, halos of the wave tiles)
    if (e == hipSuccess) {
        const int nu = b->hp.tt.nu;
        ok(b->tmpl_u.alloc((size_t)nu)) && ok(b->tmpl_c.alloc((size_t)b->hp.tt.nc)) && ok(b->tmpl_u_dk.alloc((size_t)nu));
        if (e == hipSuccess) {
            int le = launch_build_templates(st, b->hp.tt, b->shapes.p, b->tmpl_u.p, b->tmpl_c.p);
            if (le == 0) le = launch_build_template_metrics(st, nu, b->tmpl_u.p, b->tmpl_u_dk.p);
            if (le != 0) e = (hipError_t)le;
        }
        b->cst.tmpl_u = b->tmpl_u.p; b->cst.tmpl_c = b->tmpl_c.p; b->cst.tmpl_u_dk = b->tmpl_u_dk.p;
        b->cst.tmpl_n = (int)nu; b->cst.tmpl_nc = std::max(1, (int)b->hp.tt.nc);
        if (e == hipSuccess) {
            const int nc = b->hp.tt.nc;
            std::vector<double2> dk((size_t)nu);
            h_tu.resize((size_t)nu); h_tc.resize((size_t)nc);
            if (nu > 0) ok(hipMemcpyAsync(h_tu.data(), b->tmpl_u.p, (size_t)nu * sizeof(double2), hipMemcpyDeviceToHost, st)) &&
                ok(hipMemcpyAsync(dk.data(), b->tmpl_u_dk.p, (size_t)nu * sizeof(double2), hipMemcpyDeviceToHost, st));
            if (nc > 0) ok(hipMemcpyAsync(h_tc.data(), b->tmpl_c.p, (size_t)nc * sizeof(double2), hipMemcpyDeviceToHost, st));
            ok(hipStreamSynchronize(st));
            if (e == hipSuccess && nu >= 3) turn_quiet = closed_form_turns(*veh, b->hp.tt, h_tu, dk, b->cst);
        }
    }
    Tiling til;
    std::vector<int64_t> offs((size_t)n_fields + 1, 0);
    for (int64_t i = 0; i < n_fields; ++i) offs[(size_t)i + 1] = offs[(size_t)i] + b->hp.fields[(size_t)i].n_total;
    std::vector<Tiling::QuietInfo> qi((size_t)n_fields);
    for (int64_t i = 0; i < n_fields; ++i) {
        const DevField &df = b->hp.fields[(size_t)i];
        Tiling::QuietInfo q;
        q.n_line = df.n_line; q.n_turn = df.n_turn; q.P = df.P; q.n_main = df.n_main; q.gen_main = df.gen_main;
        q.line_step_len = fabs(df.line_step); q.c_line = b->cst.ms_work * b->cst.ms_work;
        q.prims = b->hp.prims.data() + df.prim_first; q.prim_count = df.prim_count; q.prim_index0 = df.prim_first;
        q.two_a = 2 * b->cst.a_lon;
        q.enable = df.n_total > 0;
        // (fields narrower than 4R have line_end_x < line_start_x: their lines run against the jump from the previous turn, the
        // first point of every line is clamped -- general kernel)
        q.turn_quiet = turn_quiet && df.n_turn == b->hp.tt.nu && df.line_step > 0.0;
        // wave tiles (fcpp_sparse.hip) where eight steps of a swath line already exceed the reach of the sweeps: the reference's
        // own sampling and coarse uniform spacings; dense sampling keeps the eight-points-per-lane kernel
        q.df = &df; q.tu = h_tu.data(); q.tc = h_tc.data(); q.u_cap = b->cst.u_cap;
        q.fence_margin = 1e-3 + std::max(0.0, -opt->geofence_tol);
        q.wave_ok = e == hipSuccess && df.n_turn == b->hp.tt.nu && (int)h_tc.size() == b->hp.tt.nc &&
                    (double)tune_int("FCPP_WAVE_FACTOR", 24) * q.two_a * q.line_step_len >= b->cst.u_cap;
        // (a sweep reaches at most u_cap / (2 a step) points: up to 24 halo lanes either side still leave 14 of a wave's 64 lanes for
        // output, which beats the eight-points-per-lane kernel -- cfg2 at 0.5 m: 0.18 ms of k_plan_fused -> 0.04 ms of k_plan_sparse,
        // step 1.41 -> 1.28 ms; at 0.25 m 2.77 -> 2.68 ms; finer sampling stays with k_plan_fused)
        qi[(size_t)i] = q;
    }
    til.build(n_fields, offs.data(), qi.data());
    if (getenv("FCPP_DEBUG_TILING")) {
        size_t n_inside = 0;
        for (const DevWaveTile &w : til.wtiles) n_inside += w.inside;
        fprintf(stderr, "[fcpp] tiling: %zu tiles; wave-tile stretches refused: back halo %lld, forward halo %lld, too few outputs %lld, "
                "primitive span %lld; %zu wave tiles, %zu of them inside the geofence by the host's test\n", til.tiles.size(),
                (long long)til.wave_fail[0], (long long)til.wave_fail[1], (long long)til.wave_fail[2], (long long)til.wave_fail[3],
                til.wtiles.size(), n_inside);
    }
    ok(b->fields.upload(b->hp.fields, st)) && ok(b->prims.upload(b->hp.prims, st)) &&
        ok(b->til.upload(til, st)) &&
        ok(b->field_junc.alloc((size_t)n_fields));
    if (e == hipSuccess && n_fields > 0) {
        b->cst.field_junc = b->field_junc.p;
        const int le = launch_field_junctions(st, n_fields, b->fields.p, b->cst, b->field_junc.p);
        if (le != 0) e = (hipError_t)le;
    }
    if (e == hipSuccess && n_polys > 0) {
        std::vector<int64_t> po(obstacles->offsets, obstacles->offsets + n_polys + 1);
        const int64_t nv = po.back();
        std::vector<double> px(obstacles->x, obstacles->x + nv), py(obstacles->y, obstacles->y + nv);
        std::vector<double> bb((size_t)n_polys * 4);
        for (int64_t k = 0; k < n_polys; ++k) {
            double mnx = HUGE_VAL, mny = HUGE_VAL, mxx = -HUGE_VAL, mxy = -HUGE_VAL;
            for (int64_t q = po[(size_t)k]; q < po[(size_t)k + 1]; ++q) {
                mnx = std::min(mnx, px[(size_t)q]); mxx = std::max(mxx, px[(size_t)q]);
                mny = std::min(mny, py[(size_t)q]); mxy = std::max(mxy, py[(size_t)q]);
            }
            bb[(size_t)k * 4] = mnx; bb[(size_t)k * 4 + 1] = mny; bb[(size_t)k * 4 + 2] = mxx; bb[(size_t)k * 4 + 3] = mxy;
        }
        ok(b->obs_off.upload(po, st)) && ok(b->obs_x.upload(px, st)) && ok(b->obs_y.upload(py, st)) && ok(b->obs_bbox.upload(bb, st)) &&
            ok(hipStreamSynchronize(st));      // the staging vectors die with this block
    }
    if (e == hipSuccess) {
        std::vector<double> seg((size_t)n_fields * 8, 0.0);
        std::vector<int32_t> mask((size_t)n_fields * 2, 0);
        for (int64_t i = 0; i < n_fields; ++i) {
            const fcpp_field_info &in = b->hp.info[(size_t)i];
            const bool okf = in.status == FCPP_OK;
            double *s = &seg[(size_t)i * 4];
            s[0] = in.approach_from[0]; s[1] = in.approach_from[1]; s[2] = in.approach_to[0]; s[3] = in.approach_to[1];
            mask[(size_t)i] = okf && in.start_kept;
            double *d = &seg[(size_t)(n_fields + i) * 4];
            d[0] = in.departure_from[0]; d[1] = in.departure_from[1]; d[2] = in.departure_to[0]; d[3] = in.departure_to[1];
            mask[(size_t)(n_fields + i)] = okf && in.end_kept;
        }
        ok(b->seg.upload(seg, st)) && ok(b->seg_mask.upload(mask, st)) && ok(hipStreamSynchronize(st));
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);   // host vectors above die at scope exit
    if (e != hipSuccess) {
        delete b;
        return fail(FCPP_EHIP, std::string("batch upload: ") + hipGetErrorString(e));
    }
    *out = b;
    return FCPP_OK;
}

int fcpp_batch_info(const fcpp_batch *b, fcpp_field_info *info_out, int64_t *total_points)
{
    if (!b) return fail(FCPP_EINVAL, "batch is NULL");
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
    if (mode == 0 && !b->til0_built) {      // the staged pipeline's own tiling: plain tiles of at most TILE_POINTS points
        Tiling t0;
        std::vector<int64_t> offs((size_t)b->n_fields + 1, 0);
        for (int64_t i = 0; i < b->n_fields; ++i) offs[(size_t)i + 1] = offs[(size_t)i] + b->hp.fields[(size_t)i].n_total;
        t0.build(b->n_fields, offs.data());
        HIPCHK(b->til0.upload(t0, st));
        b->til0_built = true;
    }
    DevTiling &t = mode == 0 ? b->til0 : b->til;
    DevObstacles obs = { b->obs_off.p, b->obs_x.p, b->obs_y.p, b->obs_bbox.p };
    if (mode == 0) HIPCHK(hipMemsetAsync(t.n_adj.p, 0, (size_t)t.n_paths * sizeof(unsigned long long), st));
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
        if (b->partial_dirty) {   // slots of quiet tiles that are not the first of their run stay zero from here on
            HIPCHK(hipMemsetAsync(t.partial.p, 0, (size_t)t.n_tiles * sizeof(TilePartial), st));
            b->partial_dirty = false;
        }
        // Two streams inside the step where the general tiles are FEW and long-lived (dense sampling of a single large field: a dozen
        // tiles that walk long halos, cfg3: 5873 points in 0.37 ms): beside the HBM-bound streaming kernel they cost nothing (cfg3
        // 0.87 -> 0.51 ms).  Not when the general kernel can fill the chip itself -- side by side it takes compute units from the
        // streaming kernel (cfg2 at 0.1 m, 5950 general tiles: 6.5 vs 5.6 ms) -- and not at sparse sampling: k_plan_sparse and the
        // span kernel get in each other's way (cfg5 2.41 vs 2.29 ms, cfg1 x 4096 0.114 vs 0.098 ms; measured again with the final kernels: 2.46 vs 2.40
        // and 0.091 vs 0.085 ms, k_plan_sparse 1.07 instead of 0.69 ms; also with the span kernel held to four or five
        // waves per SIMD).
        hipStream_t sd = st;
        const bool two = b->two_streams && t.span_points + t.chunk_points > 0 && t.n_general > 0 && t.n_general <= tune_int("FCPP_TWO_STREAM_MAX", 512) &&
                         t.n_wave == 0;
        if (two) {
            sd = b->ctx->side;
            HIPCHK(hipEventRecord(b->ctx->ev_fork, st));
            HIPCHK(hipStreamWaitEvent(sd, b->ctx->ev_fork, 0));
        }
        STAGE(2, launch_plan_sparse(sd, t.n_wave, t.wave_tiles.p, b->fields.p, b->prims.p, b->cst, obs, x, y, kappa, v, fs, t.partial.p));
        STAGE(3, launch_plan_fused(sd, variant, t.n_general, t.general_ids.p, t.tiles.p, b->fields.p, b->prims.p, b->cst, obs, x,
                                   y, kappa, v, fs, t.partial.p));
        if (two) HIPCHK(hipEventRecord(b->ctx->ev_join, sd));
        STAGE(0, launch_plan_quiet(st, t.n_span_chunks, t.span_chunks.p, 16, b->fields.p, b->prims.p, b->cst, obs, x, y, kappa, v, fs, t.partial.p));
        // (straights and U-turns in ONE launch: measured 4 % faster on identical memory than an instance each, tools/ab_quiet.py)
        STAGE(1, launch_plan_quiet(st, t.n_chunks, t.chunks.p, 14, b->fields.p, b->prims.p, b->cst, obs, x, y, kappa, v, fs, t.partial.p));
        if (two) HIPCHK(hipStreamWaitEvent(st, b->ctx->ev_join, 0));
        // (four classes of paths by their number of entries; normally one of them holds every path of a batch: the stage's events
        // time the first launch)
        {
            const int groups[4] = { 8, 64, 256, 256 };     // (16 or 32 lanes for the first class: 49 -> 53 / 79 us on cfg5)
            const int32_t *pl = t.red_paths.p;
            bool first = true;
            for (int c = 0; c < 4; ++c) {
                if (t.n_red[c] == 0) continue;
                // (a class that holds every path lists them in order: no list, one dependent load less in a latency-bound kernel)
                const int32_t *list = t.n_red[c] == t.n_paths ? nullptr : pl;
                if (first)
                    STAGE(4, launch_reduce_stats(st, t.n_red[c], t.partial.p, t.stat_first.p, nullptr, stats, t.stat_ids.p, t.stat_run.p, t.tiles.p,
                                                 b->fields.p, b->prims.p, &b->cst, list, groups[c], c == 3 ? t.red_scratch.p : nullptr));
                else
                    LAUNCHCHK(launch_reduce_stats(st, t.n_red[c], t.partial.p, t.stat_first.p, nullptr, stats, t.stat_ids.p, t.stat_run.p, t.tiles.p,
                                                  b->fields.p, b->prims.p, &b->cst, list, groups[c], c == 3 ? t.red_scratch.p : nullptr));
                first = false;
                pl += t.n_red[c];
            }
        }
        if (ev) ++b->prof_runs;
        return FCPP_OK;
    }
    STAGE(0, launch_generate(st, t.n_tiles, t.tiles.p, b->fields.p, b->prims.p, b->cst, x, y, v, fs));
    STAGE(1, launch_curv_clamp(st, t.n_tiles, t.tiles.p, t.paths.p, b->cst, 1, x, y, v, v, kappa, t.n_adj.p));
    STAGE(2, launch_scan_tiles(st, t.n_tiles, t.tiles.p, t.paths.p, b->cst, x, y, v, t.agg_f.p, t.agg_b.p));
    STAGE(3, launch_scan_spine(st, t.n_tiles, t.agg_f.p, t.agg_b.p, t.carry_f.p, t.carry_b.p, t.spine.p));
    STAGE(4, launch_scan_apply(st, t.n_tiles, t.tiles.p, t.paths.p, b->cst, 3, x, y, v, v, t.carry_f.p, t.carry_b.p));
    STAGE(5, launch_validate(st, t.n_tiles, t.tiles.p, t.paths.p, b->fields.p, b->cst, obs, x, y, kappa, v, fs, t.partial.p));
    STAGE(6, launch_reduce_stats(st, t.n_paths, t.partial.p, t.tile_first.p, t.n_adj.p, stats));
#undef STAGE
    if (ev) ++b->prof_runs;
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
    const DevTiling &t = b->til;
    const int64_t all = b->hp.total_points;
    if (mode == 0) { *points = all; return FCPP_OK; }         // every staged kernel sees every point
    const int64_t per_stage[5] = { t.span_points, t.chunk_points, t.wave_points, all - t.quiet_points - t.wave_points, all };
    *points = per_stage[stage];
    return FCPP_OK;
}

int fcpp_batch_point_split(const fcpp_batch *b, int64_t *quiet_points, int64_t *general_points)
{
    if (!b) return fail(FCPP_EINVAL, "batch is NULL");
    const int64_t q = b->til.quiet_points;
    if (quiet_points) *quiet_points = q;
    if (general_points) *general_points = b->hp.total_points - q;
    return FCPP_OK;
}

int fcpp_batch_reduce_classes(const fcpp_batch *b, int64_t *classes_out)
{
    if (!b || !classes_out) return fail(FCPP_EINVAL, "bad arguments");
    for (int c = 0; c < 4; ++c) classes_out[c] = b->til.n_red[c];
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
    if (approach_xy) LAUNCHCHK(launch_straight(st, b->n_fields, b->seg.p, 50, b->seg_mask.p, approach_xy));
    if (departure_xy)
        LAUNCHCHK(launch_straight(st, b->n_fields, b->seg.p + 4 * b->n_fields, 50, b->seg_mask.p + b->n_fields, departure_xy));
    return FCPP_OK;
}

int fcpp_batch_destroy(fcpp_batch *b)
{
    if (!b) return FCPP_OK;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
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
    LAUNCHCHK(launch_reduce_stats(st, dt.n_paths, dt.partial.p, dt.tile_first.p, nullptr, stats));
    HIPCHK(hipStreamSynchronize(st));
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
    if (ga_generation_fits(n, pop)) {
        // One launch per generation (k_ga_generation): the population in buffer a -> its statistics and elites (the bookkeeping of
        // generation g - 1) and its children (generation g) at once; the launch after the last generation only evaluates the final
        // population.  cfg4: 77 -> 55 us per generation (45 us with the DPP arg-max of the elite selection).
        for (int g = 0; g <= cfg->max_generations; ++g) {
            const int a = g & 1, b = a ^ 1;
            LAUNCHCHK(launch_ga_generation(st, n, pop, D, buf[a], fit[a], dist[a], buf[b], fit[b], dist[b], *cfg, g, state.p, best_route, hist));
            if ((g & 31) == 31) {                      // the kernel is a no-op once converged; stop launching it
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
        if ((g & 31) == 31) {                          // the kernels are no-ops once converged; stop launching them
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

}  // extern "C"
