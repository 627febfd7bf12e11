// fcpp_tiler.cpp -- see fcpp_tiler.h.  The cutting rules are those of DESIGN.md section 5 and HISTORY.md (pipeline B): quiet zones of straight
// primitives, layer-1 spans, wave tiles with host-sized halos, general tiles for the rest.
#include "fcpp_tiler.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>

#include "fcpp_geom.h"
#include "fcpp_parallel.h"

namespace fcpp {

// a block's records, indices relative to the block (tile slots) until the merge
struct BlockTiles {
    std::vector<DevTile> tiles;
    std::vector<DevWaveTile> wtiles;
    std::vector<int32_t> general_ids, stat_ids;
    std::vector<int64_t> stat_run;
    std::vector<DevTile> chunks, span_chunks;        // (TileConsts.device_chunks: empty -- the device expands `groups`; n_chunk_rec / n_span_rec count the records)
    std::vector<DevChunkGroup> groups;               // e0 / chunk_base / span_base block-relative
    int64_t n_chunk_rec = 0, n_span_rec = 0, group_base = 0;
    std::vector<int32_t> cls[4];                 // fields (batch-wide indices) by reduction class
    std::vector<DevFieldWork> work[4];           // fields planned and reduced by one workgroup, by the size of that workgroup (w_first / e_first block-relative)
    std::vector<int32_t> open_wave;              // wave tiles (block-relative indices) of all other fields
    int64_t tile0[PLAN_BLOCK_FIELDS + 1], w0[PLAN_BLOCK_FIELDS + 1];      // a field's records: [tile0[k], tile0[k + 1]) ...
    int64_t stat_cnt[PLAN_BLOCK_FIELDS];         // statistic entries per field
    int64_t n_runs = 0, quiet_points = 0, wave_points = 0, work_wave_points = 0, work_span_points = 0, unfusable_work = 0, span_points = 0, chunk_points = 0, wave_inside = 0;
    int64_t wave_fail[5] = { 0, 0, 0, 0, 0 };
    // bases in the merged tables
    int64_t tile_base = 0, wave_base = 0, general_base = 0, stat_base = 0, chunk_base = 0, span_base = 0, cls_base[4] = { 0, 0, 0, 0 }, work_base[4] = { 0, 0, 0, 0 }, open_base = 0;
};

namespace {

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// tiler of one block: scratch that lives across its fields
struct FieldTiler {
    const HostPlan &hp;
    const TileConsts &tc;
    BlockTiles &out;
    std::vector<double> d, hx, hy;               // step lengths and points of the stretch being cut into wave tiles

    // one field
    int64_t p = 0;                               // batch-wide field index
    const DevField *f = nullptr;
    const DevPrim *prims = nullptr;
    int prim_count = 0, prim_index0 = 0;
    int64_t per = 0;
    bool turn_quiet = false, wave_ok = false;
    double line_step_len = 0.0;

    FieldTiler(const HostPlan &h, const TileConsts &t, BlockTiles &o) : hp(h), tc(t), out(o) {}

    int prim_of(int64_t i) const                 // index (within the field) of the primitive that holds point i
    {
        int a = 0, b = prim_count - 1;
        while (a < b) { const int m = (a + b + 1) >> 1; if (prims[m].start <= i) a = m; else b = m - 1; }
        return a;
    }

    // Path points [lo - 1, hi) of the field into hx / hy (index i - (lo - 1)): the formulas of eval_main / eval_prim (fcpp_pointfn.h) on the
    // host copies of the templates.  Only distances between consecutive points and the margin to the geofence are taken from them.
    void eval_range(int64_t first, int64_t last_excl)
    {
        const DevField &F = *f;
        int64_t i = first;
        size_t o = 0;
        // layer 1, closed form
        if (i < F.gen_main) {
            int64_t idx = i / per, off = i - idx * per;
            for (; i < last_excl && i < F.gen_main; ++i, ++o) {
                tiler_point_main(F, tc.tu, idx, off, hx[o], hy[o]);
                if (++off == per) { off = 0; ++idx; }
            }
        }
        if (i >= last_excl) return;
        // primitives: walk them in order
        int k = prim_of(i);
        while (i < last_excl) {
            const DevPrim &q = prims[k];
            const int64_t end = std::min<int64_t>(last_excl, q.start + q.n);
            for (; i < end; ++i, ++o) tiler_point_prim(q, tc.tu, tc.tc, i - q.start, hx[o], hy[o]);
            ++k;
        }
    }

    // Wave tiles of the sparse kernel for the general stretch [a, b): [ Hb halo | count outputs | Hf halo ] <= WAVE_LANES lanes.
    // The halos are sized from the path's own step lengths (see fcpp_sparse.hip): backwards from the point before the first output
    // (whose final speed the segment metrics need) until the couplings 2a|dp| add up to u_cap, a skipped step or the path's start,
    // plus one lane for the stencil of the outermost point; forwards likewise from the last output.  Every tile takes as many
    // outputs as fit.  false = some tile would hold fewer than 8 outputs (dense sampling): the stretch stays with k_plan_fused.
    bool wave_tiles(int64_t a, int64_t b)
    {
        const DevField &F = *f;
        const int WAVE_LANES = tc.wave_points;           // points of a wave tile (fcpp_sparse.hip: one wavefront per wave tile)
        const int64_t n = F.n_total;
        const double cap = tiler_halo_cap(tc.u_cap);
        // d[i - lo] = |p_i - p_(i-1)| for the stretch and WAVE_HALO_MAX + 2 points either side
        const int64_t lo = std::max<int64_t>(a - WAVE_HALO_MAX - 2, 1), hi = std::min<int64_t>(b + WAVE_HALO_MAX + 2, n);   // i in [lo, hi)
        if (hi - lo > (int64_t)1 << 22) { ++out.wave_fail[4]; return false; }
        d.resize((size_t)std::max<int64_t>(hi - lo, 0));
        hx.resize(d.size() + 1); hy.resize(d.size() + 1);
        if (hi > lo) {
            eval_range(lo - 1, hi);
            for (size_t k = 0; k < d.size(); ++k) {
                const double dx = hx[k + 1] - hx[k], dy = hy[k + 1] - hy[k];
                d[k] = sqrt(dx * dx + dy * dy);
            }
        }
        // every output point of [s, s + c) well inside the field polygon?  (host and device evaluate a point with the same formulas;
        // their roundings differ by ~1e-12 m at field-sized coordinates, the margin is 1e-7 m plus 256 ulps of the point: tiler_inside)
        auto all_inside = [&](int64_t s, int64_t c) -> bool {
            for (int64_t i = s; i < s + c; ++i) {
                if (i < lo - 1 || i >= hi) return false;
                if (!tiler_inside(F, hx[(size_t)(i - lo + 1)], hy[(size_t)(i - lo + 1)], tc.fence_margin)) return false;
            }
            return true;
        };
        auto dist = [&](int64_t i) { return d[(size_t)(i - lo)]; };     // lo <= i < hi by the halo bound below
        auto back_halo = [&](int64_t s) -> int { return tiler_back_halo(dist, s, tc.two_a, cap); };
        auto fwd_halo = [&](int64_t e) -> int { return tiler_fwd_halo(dist, e, n, tc.two_a, cap); };
        const size_t mark = out.tiles.size(), mark_w = out.wtiles.size();
        const int64_t inside_mark = out.wave_inside;
        auto refuse = [&](int why) { ++out.wave_fail[why]; out.tiles.resize(mark); out.wtiles.resize(mark_w); out.wave_inside = inside_mark; return false; };
        for (int64_t s = a; s < b;) {
            const int Hb = back_halo(s);
            if (Hb < 0) return refuse(0);
            // the largest count whose forward halo still fits -- and whose points (halos included) lie in at most nine primitives: the
            // tile record names the first one and has eight thresholds for the others
            int64_t c = std::min<int64_t>(b - s, WAVE_LANES - Hb);
            int Hf = -1;
            const int64_t first0 = s - Hb;
            const int pa0 = first0 + Hb + c - 1 >= F.gen_main ? prim_of(std::max<int64_t>(first0, F.gen_main)) : 0;
            for (; c >= 1; --c) {
                Hf = fwd_halo(s + c - 1);
                if (!(Hf >= 0 && Hb + c + Hf <= WAVE_LANES)) continue;
                const int64_t last0 = s + c - 1 + Hf;
                if (last0 >= F.gen_main && prim_of(last0) - pa0 > 8) continue;
                break;
            }
            if (c < std::min<int64_t>(8, b - s)) return refuse(Hf < 0 ? 1 : 2);
            const int64_t first = s - Hb, last = s + c - 1 + Hf;
            DevTile t;
            t.field = (int32_t)p; t.count = (int32_t)c; t.start = s; t.quiet = 5; t.stat_tile = Hb | (Hf << 16);
            if (first < F.gen_main) { t.idx0 = (int32_t)(first / per); t.off0 = (int32_t)(first % per); }
            else { t.idx0 = prim_index0 + prim_of(first); t.off0 = 0; }
            // the self-contained record: layer-1 decode of lane 0, and where the (at most 8) further primitives start among the lanes
            DevWaveTile wt;
            memset(&wt, 0, sizeof wt);
            auto clampi = [](int64_t v) { return (int32_t)std::max<int64_t>(-2, std::min<int64_t>(v, (int64_t)1 << 30)); };
            wt.out_base = F.pt_off + first; wt.field = (int32_t)p; wt.tile = (int32_t)out.tiles.size();
            wt.count = (uint8_t)c; wt.hb = (uint8_t)Hb; wt.hf = (uint8_t)Hf; wt.inside = all_inside(s, c) ? 1 : 0;
            wt.rel_main = clampi(F.gen_main - first); wt.rel_seam = clampi(F.n_main - first); wt.rel_last = clampi(n - 1 - first);
            wt.rel_zero = clampi(-first);
            wt.idx0 = t.idx0; wt.off0 = t.off0;
            for (int k = 0; k < 8; ++k) wt.thr[k] = 255;
            if (last >= F.gen_main) {
                const int64_t fl2 = std::max<int64_t>(first, F.gen_main);      // first primitive-generated point of the tile
                const int pa = prim_of(fl2), pb = prim_of(last);
                if (pb - pa > 8) return refuse(3);
                wt.p0 = prim_index0 + pa;
                wt.r0 = (int32_t)(first - prims[pa].start);
                for (int k = pa + 1; k <= pb; ++k) {
                    // (strictly ascending: the kernel counts the primitives that start at or before a point as bits of a mask; an empty
                    // primitive -- two starts on one point, no plan has one -- would be counted once)
                    if (prims[k].start <= prims[k - 1].start) return refuse(3);
                    wt.thr[k - pa - 1] = (uint8_t)(prims[k].start - first);
                }
            }
            out.wave_inside += wt.inside;
            out.wtiles.push_back(wt);
            out.tiles.push_back(t);
            s += c;
        }
        return true;
    }

    // The same stretch cut in closed form (fcpp_cutfn.h; the device planner runs the same function): the field's general stretch
    // [S, n_total) at the reference's sampling.  false: the stretch stays with the general kernel.
    std::vector<CutPrim> cprims;
    struct PrimView { const CutPrim *p; const CutPrim &operator()(int k) const { return p[k]; } };
    bool wave_tiles_closed(int64_t a)
    {
        const DevField &F = *f;
        const int64_t n = F.n_total;
        cprims.resize((size_t)prim_count);
        bool ok = true;
        double lx, ly;
        cut_main_end(F, tc.cut, lx, ly);
        for (int k = 0; k < prim_count; ++k) cprims[(size_t)k] = cut_prim_info(prims[k], F, tc.cut, lx, ly, ok);
        FieldCut fc;
        fc.n_tiles = 0; fc.status = CUT_GENERAL;
        if (ok) cut_field(F, tc.cut, PrimView{ cprims.data() }, prim_count, a, fc);
        if (fc.status != CUT_OK) { ++out.wave_fail[0]; return false; }
        for (int k = 0; k < fc.n_tiles; ++k) {
            const CutTile &ct = fc.t[k];
            const int64_t s = ct.s, c = ct.c;
            const int Hb = ct.hb, Hf = ct.hf;
            const int64_t first = s - Hb, last = s + c - 1 + Hf;
            DevTile t;
            t.field = (int32_t)p; t.count = (int32_t)c; t.start = s; t.quiet = 5; t.stat_tile = Hb | (Hf << 16);
            if (first < F.gen_main) { t.idx0 = (int32_t)(first / per); t.off0 = (int32_t)(first % per); }
            else { t.idx0 = prim_index0 + prim_of(first); t.off0 = 0; }
            DevWaveTile wt;
            memset(&wt, 0, sizeof wt);
            auto clampi = [](int64_t v) { return (int32_t)std::max<int64_t>(-2, std::min<int64_t>(v, (int64_t)1 << 30)); };
            wt.out_base = F.pt_off + first; wt.field = (int32_t)p; wt.tile = (int32_t)out.tiles.size();
            wt.count = (uint8_t)c; wt.hb = (uint8_t)Hb; wt.hf = (uint8_t)Hf; wt.inside = ct.inside;
            wt.rel_main = clampi(F.gen_main - first); wt.rel_seam = clampi(F.n_main - first); wt.rel_last = clampi(n - 1 - first);
            wt.rel_zero = clampi(-first);
            wt.idx0 = t.idx0; wt.off0 = t.off0;
            for (int q = 0; q < 8; ++q) wt.thr[q] = 255;
            if (last >= F.gen_main) {
                const int64_t fl2 = std::max<int64_t>(first, F.gen_main);
                const int pa = prim_of(fl2), pb = prim_of(last);
                wt.p0 = prim_index0 + pa;
                wt.r0 = (int32_t)(first - prims[pa].start);
                for (int q = pa + 1; q <= pb; ++q) wt.thr[q - pa - 1] = (uint8_t)(prims[q].start - first);
            }
            out.wave_inside += wt.inside;
            out.wtiles.push_back(wt);
            out.tiles.push_back(t);
        }
        return true;
    }

    void emit(int64_t s, int64_t cnt, int kind, int64_t i0, int64_t o0)
    {
        DevTile t;
        t.field = (int32_t)p; t.start = s; t.count = (int32_t)cnt; t.quiet = kind; t.stat_tile = 0;
        t.idx0 = (int32_t)i0; t.off0 = (int32_t)o0;
        out.tiles.push_back(t);
    }
    void emit_general(int64_t a, int64_t b)
    {
        const int64_t len = b - a;
        if (len <= 0) return;
        // (a field of the closed-form cut: its one general stretch [span, n_total) -- cut as the device planner cuts it, or left to the general kernel)
        const bool closed = tc.closed_cut && tc.wave_points == CUT_WAVE_LANES && b == f->n_total && a == cut_span_points(*f, tc.cut) && cut_applies(*f, tc.cut, a);
        if (closed) { if (wave_ok && wave_tiles_closed(a)) return; }
        else if (wave_ok && wave_tiles(a, b)) return;
        const int64_t k = (len + TILE_POINTS - 1) / TILE_POINTS, base = len / k, rem = len % k;
        for (int64_t i = 0; i < k; ++i) {
            const int64_t c = base + (i < rem ? 1 : 0);
            const bool in1 = per > 0 && a < f->gen_main;      // layer-1 decode of the tile start for the general kernel
            emit(a, c, 0, in1 ? a / per : 0, in1 ? a % per : 0);
            a += c;
        }
    }
    // near-equal quiet tiles of at most TILE_POINTS - 2 points (the kernel stores aligned PAIRS; a tile that starts on an
    // odd global index needs one pair more than half its points)
    // (Only the FIRST of those tiles is ever read -- by the device as the run's tile, by derive_field as the run's start -- so only it is
    // made: its count as before, the run's length in its stat_tile until the image is written (fill: zero there, as the device tiler has it).
    // At dense sampling the other tiles were most of the tiler's time.)
    void emit_quiet(int64_t zs, int64_t Z, int kind, int64_t i0, int64_t o0)
    {
        const int64_t cap = TILE_POINTS - 2, k = (Z + cap - 1) / cap, base = Z / k, rem = Z % k;
        emit(zs, base + (rem > 0 ? 1 : 0), kind, i0, o0);
        out.tiles.back().stat_tile = (int32_t)Z;
    }
    int64_t need_for(double c_nom, double step_len) const { return tiler_need_for(c_nom, step_len, tc.two_a); }

    // Tiles never straddle fields and hold at most TILE_POINTS points.  Every straight primitive -- swath lines of layer 1, headland
    // straights of layer 2 -- is cut as
    //     [ need | quiet zone ............................. | need ]
    // quiet zone = samples whose sweep neighbourhood stays on the straight: `need` samples span u_nominal / (2a) metres, the
    // farthest a slower point can lower speeds that are nominal for this straight.  Quiet zones become "quiet" tiles
    // (closed-form kernel); everything else -- turns, corner arcs, reverse fills and the margins around them -- becomes
    // wave tiles (sparse sampling) or general tiles.
    void tile_field(int64_t field)
    {
        p = field;
        f = &hp.fields[(size_t)field];
        const DevField &F = *f;
        const int64_t n = F.n_total;
        if (n <= 0) return;
        prims = hp.prims_of(field); prim_count = F.prim_count; prim_index0 = F.prim_first;
        per = (int64_t)F.n_line + F.n_turn;
        line_step_len = fabs(F.line_step);
        // (fields narrower than 4R have line_end_x < line_start_x: their lines run against the jump from the previous turn, the
        // first point of every line is clamped -- general kernel)
        turn_quiet = tc.turn_quiet && F.n_turn == tc.nu && F.line_step > 0.0;
        // wave tiles (fcpp_sparse.hip) where eight steps of a swath line already exceed the reach of the sweeps: the reference's
        // own sampling and coarse uniform spacings; dense sampling keeps the eight-points-per-lane kernel
        // (a sweep reaches at most u_cap / (2 a step) points: up to 24 halo lanes either side still leave 14 of a wave's 64 lanes for
        // output, which beats the eight-points-per-lane kernel -- cfg2 at 0.5 m: 0.18 ms of k_plan_fused -> 0.04 ms of k_plan_sparse,
        // step 1.41 -> 1.28 ms; at 0.25 m 2.77 -> 2.68 ms; finer sampling stays with k_plan_fused)
        wave_ok = tc.templates_ok && F.n_turn == tc.nu && (double)tc.wave_factor * tc.two_a * line_step_len >= tc.u_cap;
        const int64_t P = F.P, n_line = F.n_line, n_turn = F.n_turn, gen_main = F.gen_main;
        int64_t pos = 0;
        const int64_t need1 = need_for(tc.c_line, line_step_len);
        if (need1 >= 0 && per > 0 && gen_main > 0) {
            // With closed-form U-turns nothing propagates into a swath line from the turns around it (a turn starts on
            // the line's last point: a skipped step; the jump back from the turn's end is too long to bind): all
            // complete passes (line + turn) form ONE quiet span, whatever the sampling.  The last line ends at the
            // seam to the headland layer: it is cut like any other straight, without a margin at its start.
            // (Rounds 2-4 kept lines and turns as runs of their own at dense sampling -- no pass decode per point -- and the span for
            // short lines only; TileConsts.span_line_max says where that still holds.)
            int64_t first_idx = 0;
            const bool span = turn_quiet && P >= 2 && n_line - need1 < (F.obs_count > 0 ? (int64_t)64 : tc.span_line_max) && (P - 1) * per < (int64_t)0x7fffffff;
            if (span) {
                const int64_t S = (P - 1) * per;
                emit_quiet(0, S, 4, 0, 0);
                pos = S; first_idx = P - 1;
            }
            for (int64_t idx = first_idx; idx < P; ++idx) {
                // closed-form turns, dense sampling: the whole line is a quiet run, and so is the turn after it
                const bool full = turn_quiet && !span;
                const int64_t need_s = (full || (span && idx > 0)) ? 0 : need1, need_e = (full && idx < P - 1) ? 0 : need1;
                const int64_t L0 = idx * per, zs = L0 + need_s, Z = n_line - need_s - need_e;
                if (Z < 64) break;
                emit_general(pos, zs);
                emit_quiet(zs, Z, 1, idx, need_s);
                pos = zs + Z;
                if (full && idx < P - 1) { emit_quiet(L0 + n_line, n_turn, 3, idx, 0); pos = L0 + per; }
            }
        }
        for (int k = 0; k < prim_count; ++k) {
            const DevPrim &pr = prims[k];
            if (pr.kind != PRIM_LINSPACE) continue;
            const double ms = pr.v_nom / 3.6;
            const int64_t need2 = need_for(ms * ms, sqrt(pr.a[4] * pr.a[4] + pr.a[5] * pr.a[5]));
            if (need2 < 0) continue;
            const int64_t zs = pr.start + need2, Z = (int64_t)pr.n - 2 * need2;
            if (Z < 64 || zs < pos) continue;
            emit_general(pos, zs);
            emit_quiet(zs, Z, 2, prim_index0 + k, need2);
            pos = zs + Z;
        }
        emit_general(pos, n);
    }

    // an equal field of the same block (same constructor arguments => same plan, same primitives): its records with this field's indices
    void copy_field(int64_t field, int64_t proto)
    {
        const int64_t b0 = (field / PLAN_BLOCK_FIELDS) * PLAN_BLOCK_FIELDS;
        const int kp = (int)(proto - b0);
        const DevField &F = hp.fields[(size_t)field], &G = hp.fields[(size_t)proto];
        const int64_t t0 = out.tile0[kp], t1 = out.tile0[kp + 1], w0 = out.w0[kp], w1 = out.w0[kp + 1];
        const int64_t dp = F.pt_off - G.pt_off;
        for (int64_t i = t0; i < t1; ++i) { DevTile t = out.tiles[(size_t)i]; t.field = (int32_t)field; out.tiles.push_back(t); }
        for (int64_t i = w0; i < w1; ++i) {
            DevWaveTile w = out.wtiles[(size_t)i];
            w.field = (int32_t)field; w.out_base += dp;       // (w.tile, the statistics entry, is set by derive_field)
            out.wave_inside += w.inside;
            out.wtiles.push_back(w);
        }
    }

    // What the kernels' work lists need beyond the tile table, from the field's tiles [t0, t1) (block-relative slots): the tiles of
    // k_plan_fused, the entries k_reduce_stats walks (general tiles, wave tiles, the first tile of every quiet run + the run's length),
    // and the quiet runs cut into chunks on 512-point boundaries of the batch arrays.
    struct Run { int64_t tile, count; int32_t entry; };
    std::vector<Run> rv;
    void derive_field(int64_t field, int k_local, int64_t t0, int64_t t1)
    {
        const DevField &F = hp.fields[(size_t)field];
        std::vector<DevTile> &T = out.tiles;
        const size_t stat_mark = out.stat_ids.size();
        int64_t w_next = out.w0[k_local];        // the field's wave-tile records, in the order of their tiles
        rv.clear();
        for (int64_t i = t0; i < t1;) {
            DevTile &a = T[(size_t)i];
            const int32_t entry = (int32_t)out.stat_ids.size();     // the statistics entry (block-relative): the entries of a field lie side by side
            out.stat_ids.push_back((int32_t)i);                 // a general tile, a wave tile, or the first tile of a run
            out.stat_run.push_back(0);
            if (!a.quiet) { a.stat_tile = entry; out.general_ids.push_back((int32_t)i); ++i; continue; }
            if (a.quiet == 5) { out.wtiles[(size_t)w_next++].tile = entry; out.wave_points += a.count; ++i; continue; }
            // the run: one quiet zone = one tile (emit_quiet), its length in the tile's stat_tile
            const int64_t cnt = a.stat_tile, j = i + 1;
            rv.push_back({ i, cnt, entry });
            out.stat_run.back() = cnt;
            out.quiet_points += cnt;
            i = j;
        }
        const int64_t ne = (int64_t)(out.stat_ids.size() - stat_mark);
        out.stat_cnt[k_local] = ne;
        // A field whose general points are all in a few (two-point) wave tiles (fields of the reference's size have four):
        // k_plan_sparse_fields plans the tiles and reduces the field (DevFieldWork).  All others: their wave tiles go to k_plan_sparse's list, the field to a class of
        // k_reduce_stats by the number of entries of its path (a property of the field alone; 8 lanes, a wavefront, a workgroup,
        // 64 workgroups per path: at most 8 / 4 / 4 entries per lane in the first three)
        const int64_t nw = out.w0[k_local + 1] - out.w0[k_local];
        bool general = false;
        for (int64_t i = t0; i < t1 && !general; ++i) general = T[(size_t)i].quiet == 0;
        const bool is_work = tc.field_work && tc.wave_points == 128 && !general && nw >= 1 && nw <= std::min(tc.field_work_tiles, FIELD_WORK_TILES) && ne <= FIELD_WORK_ENTRIES;
        if (is_work) {
            DevFieldWork w;
            memset(&w, 0, sizeof w);
            w.field = (int32_t)field; w.n_tiles = (int32_t)nw; w.w_first = (int32_t)out.w0[k_local]; w.e_first = (int32_t)stat_mark; w.n_entries = (int32_t)ne;
            out.work[field_work_class((int)nw)].push_back(w);
            for (int64_t k = out.w0[k_local]; k < out.w0[k_local + 1]; ++k) out.work_wave_points += out.wtiles[(size_t)k].count;
        } else {
            for (int64_t k = out.w0[k_local]; k < out.w0[k_local + 1]; ++k) out.open_wave.push_back((int32_t)k);
            out.cls[tiler_reduce_class(ne, tc.reduce_wg_max)].push_back((int32_t)field);
        }
        out.n_runs += (int64_t)rv.size();
        // Chunks: every run is cut on 512-point boundaries of the batch arrays.  Consecutive layer-1 runs (swath line, U-turn, swath
        // line, ...) are cut TOGETHER: a chunk that holds the end of one run and the start of the next is written by one wave through
        // the span decode (kind 4) instead of two partial chunks (measured 4-5 % on the streaming kernel on identical memory).
        const int64_t pass = (int64_t)F.n_line + F.n_turn;
        int64_t fused_span = 0;
        for (size_t r = 0; r < rv.size();) {
            const DevTile &a = T[(size_t)rv[r].tile];
            size_t r1 = r + 1;
            int64_t total = rv[r].count;
            if (a.quiet == 1 || a.quiet == 3)
                for (; r1 < rv.size(); ++r1) {
                    const DevTile &tn = T[(size_t)rv[r1].tile];
                    if (!((tn.quiet == 1 || tn.quiet == 3) && tn.start == a.start + total)) break;
                    total += rv[r1].count;
                }
            // the layer-1 span of a field of field work is written by the field's own workgroup (k_plan_sparse_fields): no chunks
            const int64_t g_grp = F.pt_off + a.start;
            if (is_work && a.quiet == 4 && a.start == 0) {
                const bool fusable = ((g_grp % TILE_POINTS) + total + TILE_POINTS - 1) / TILE_POINTS <= FUSED_SPAN_CHUNKS;
                if (!fusable) ++out.unfusable_work;
                else if (tc.fuse_spans) { fused_span = total; out.work_span_points += total; r = r1; continue; }
            }
            // device_chunks: the records are written by the device (k_expand_chunks: the arithmetic of the loop further below); the host lists
            // the group in segments of CHUNK_SEGMENT chunks and counts the records of either list so that every segment knows where its go
            if (tc.device_chunks) {
                // ... without walking the chunks: chunk j of the group begins c_first + (j - 1) 512 points into it, and the only chunks that go
                // to the span list are those of a layer-1 span and those with a run boundary strictly inside -- found from the run
                // boundaries, O(runs) instead of O(points / 512) (a 5000 x 2000 m field at 0.05 m: 800 runs, 123 000 chunks)
                const int64_t room0 = TILE_POINTS - (g_grp % TILE_POINTS), c_first = std::min(total, room0);
                const int64_t J = total <= c_first ? 1 : 1 + (total - c_first + TILE_POINTS - 1) / TILE_POINTS;
                auto chunk_of = [&](int64_t d) { return d < c_first ? (int64_t)0 : 1 + (d - c_first) / TILE_POINTS; };
                auto chunk_start = [&](int64_t j) { return j == 0 ? (int64_t)0 : c_first + (j - 1) * TILE_POINTS; };
                auto chunk_count = [&](int64_t j) { return j == 0 ? c_first : std::min<int64_t>(TILE_POINTS, total - chunk_start(j)); };
                const bool all_span = a.quiet == 4;
                int64_t spans = 0, span_pts = 0, next_j0 = 0, last_span = -1;
                auto segments_upto = [&](int64_t j_incl) {       // the segments that begin at or before chunk j_incl: `spans` span chunks lie before them
                    for (; next_j0 <= j_incl && next_j0 < J; next_j0 += CHUNK_SEGMENT) {
                        const int64_t before = all_span ? next_j0 : spans;
                        DevChunkGroup grp;
                        memset(&grp, 0, sizeof grp);
                        grp.field = (int32_t)field; grp.e0 = rv[r].entry; grp.n_runs = (int32_t)(r1 - r); grp.g0 = g_grp; grp.total = total;
                        grp.j0 = (int32_t)next_j0; grp.n = (int32_t)std::min<int64_t>(CHUNK_SEGMENT, J - next_j0);
                        grp.chunk_base = out.n_chunk_rec + (next_j0 - before); grp.span_base = out.n_span_rec + before;
                        out.groups.push_back(grp);
                    }
                };
                if (all_span) { spans = J; span_pts = total; }
                else {
                    int64_t B = 0;
                    for (size_t k = r; k + 1 < r1; ++k) {
                        B += rv[k].count;                              // where run k + 1 begins
                        const int64_t jb = chunk_of(B);
                        if (B == chunk_start(jb) || jb == last_span) continue;
                        segments_upto(jb);
                        ++spans; span_pts += chunk_count(jb); last_span = jb;
                    }
                }
                segments_upto(J - 1);
                out.n_span_rec += spans; out.n_chunk_rec += J - spans;
                out.span_points += span_pts; out.chunk_points += total - span_pts;
                r = r1;
                continue;
            }
            size_t rc = r;                       // run that holds the current position
            int64_t rc_begin = 0;                // its first point, relative to the group
            for (int64_t done = 0; done < total;) {
                const int64_t g = g_grp + done;
                // (also for the short spans of sparse sampling, where one chunk in eight is partial: near-equal chunks from the span's
                // start, i.e. 12 % fewer waves with unaligned stores, took 1.81 instead of 1.50 ms on cfg5)
                const int64_t c = std::min<int64_t>(total - done, TILE_POINTS - (g % TILE_POINTS));
                while (done >= rc_begin + rv[rc].count) { rc_begin += rv[rc].count; ++rc; }
                const DevTile &tr = T[(size_t)rv[rc].tile];
                DevTile ch = tr;
                ch.start = a.start + done; ch.count = (int32_t)c; ch.stat_tile = rv[rc].entry;
                const bool one_run = done + c <= rc_begin + rv[rc].count;
                if (one_run && tr.quiet != 4) ch.off0 = (int32_t)(tr.off0 + (done - rc_begin));
                else {      // a span of layer 1 (or a chunk across runs): (pass, offset in the pass) of the chunk's first point
                    ch.quiet = 4;
                    ch.idx0 = (int32_t)(ch.start / pass); ch.off0 = (int32_t)(ch.start % pass);
                }
                if (ch.quiet == 4) { out.span_chunks.push_back(ch); out.span_points += c; }
                else { out.chunks.push_back(ch); out.chunk_points += c; }
                done += c;
            }
            r = r1;
        }
        if (is_work && fused_span > 0) out.work[field_work_class((int)nw)].back().fused_span = (int32_t)fused_span;
    }
};

template <class T>
T *at(unsigned char *base, size_t off) { return reinterpret_cast<T *>(base + off); }

}  // namespace

// the tables' places inside the image from their counts (n_fields, n_prims, n_tiles, ...): the same for an image built on the host and
// one built on the device (fcpp_devplan.hip)
void layout_image(ImageLayout &lay)
{
    size_t o = 0;
    auto take = [&](size_t &slot, size_t bytes) { slot = o; o = align256(o + bytes); };
    take(lay.fields, (size_t)lay.n_fields * sizeof(DevField));
    take(lay.prims, (size_t)lay.n_prims * sizeof(DevPrim));
    take(lay.tiles, (size_t)lay.n_tiles * sizeof(DevTile));
    take(lay.wtiles, (size_t)lay.n_wave * sizeof(DevWaveTile));
    take(lay.general_ids, (size_t)lay.n_general * sizeof(int32_t));
    // (chunk lists the device expands from chunk groups lie behind the uploaded part; lists the host -- or the device tiler -- writes lie here)
    if (lay.n_chunk_groups == 0) {
        take(lay.chunks, (size_t)lay.n_chunks * sizeof(DevTile));
        take(lay.span_chunks, (size_t)lay.n_span_chunks * sizeof(DevTile));
    }
    take(lay.chunk_groups, (size_t)lay.n_chunk_groups * sizeof(DevChunkGroup));
    take(lay.stat_ids, (size_t)lay.n_stat * sizeof(int32_t));
    take(lay.stat_first, (size_t)(lay.n_fields + 1) * sizeof(int64_t));
    take(lay.stat_run, (size_t)lay.n_stat * sizeof(int64_t));
    take(lay.red_paths, (size_t)lay.n_fields * sizeof(int32_t));
    take(lay.field_work, (size_t)lay.n_field_work * sizeof(DevFieldWork));
    take(lay.field_packs, (size_t)lay.n_field_work * sizeof(DevFieldPack));
    take(lay.open_wave_ids, (size_t)lay.n_open_wave * sizeof(int32_t));
    take(lay.obs_off, lay.n_polys > 0 ? (size_t)(lay.n_polys + 1) * sizeof(int64_t) : 0);
    take(lay.obs_x, (size_t)lay.n_poly_verts * sizeof(double));
    take(lay.obs_y, (size_t)lay.n_poly_verts * sizeof(double));
    take(lay.obs_bbox, (size_t)lay.n_polys * 4 * sizeof(double));
    take(lay.seg, (size_t)lay.n_fields * 8 * sizeof(double));
    take(lay.seg_mask, (size_t)lay.n_fields * 2 * sizeof(int32_t));
    lay.upload_bytes = o;
    if (lay.n_chunk_groups > 0) {
        take(lay.chunks, (size_t)lay.n_chunks * sizeof(DevTile));
        take(lay.span_chunks, (size_t)lay.n_span_chunks * sizeof(DevTile));
    }
    take(lay.partial, (size_t)lay.n_stat * sizeof(TilePartial));      // one slot per statistics entry
    take(lay.red_scratch, (size_t)lay.n_red[3] * 64 * 104);
    take(lay.field_junc, (size_t)lay.n_fields * 2 * sizeof(double));
    take(lay.work_totals, (size_t)lay.n_field_work * sizeof(TilePartial));   // per field of field_work: the statistics of its quiet runs, summed once
    take(lay.info, lay.info_on_device ? (size_t)lay.n_fields * sizeof(fcpp_field_info) : 0);   // device-side setup: fcpp_field_info, copied back on demand
    take(lay.own_stats, (size_t)lay.n_fields * sizeof(fcpp_field_stats));       // fcpp_batch_plan(stats_dev = NULL): the batch's own statistics records
    lay.total_bytes = o;
}

// the batch's obstacle polygons (CSR) and one bounding box per polygon
void fill_obstacles(const fcpp_polys *polys, const ImageLayout &lay, unsigned char *dst, size_t rebase)
{
    if (lay.n_polys > 0) {
        const int64_t np = lay.n_polys, nv = lay.n_poly_verts;
        memcpy(at<int64_t>(dst, lay.obs_off - rebase), polys->offsets, (size_t)(np + 1) * sizeof(int64_t));
        if (nv > 0) {
            memcpy(at<double>(dst, lay.obs_x - rebase), polys->x, (size_t)nv * sizeof(double));
            memcpy(at<double>(dst, lay.obs_y - rebase), polys->y, (size_t)nv * sizeof(double));
        }
        double *bb = at<double>(dst, lay.obs_bbox - rebase);
        for (int64_t k = 0; k < np; ++k) {
            double mnx = HUGE_VAL, mny = HUGE_VAL, mxx = -HUGE_VAL, mxy = -HUGE_VAL;
            for (int64_t q = polys->offsets[k]; q < polys->offsets[k + 1]; ++q) {
                mnx = std::min(mnx, polys->x[q]); mxx = std::max(mxx, polys->x[q]);
                mny = std::min(mny, polys->y[q]); mxy = std::max(mxy, polys->y[q]);
            }
            bb[k * 4] = mnx; bb[k * 4 + 1] = mny; bb[k * 4 + 2] = mxx; bb[k * 4 + 3] = mxy;
        }
    }
}

BatchTiler::BatchTiler() : blocks_(new std::vector<BlockTiles>()) {}
BatchTiler::~BatchTiler() { delete blocks_; }

int BatchTiler::plan(const HostPlan &hp, const TileConsts &tc, const fcpp_polys *polys, ImageLayout &lay, std::string &err)
{
    try { return plan_impl(hp, tc, polys, lay, err); }
    catch (const std::bad_alloc &) { err = "out of host memory"; return FCPP_ENOMEM; }
}

int BatchTiler::plan_impl(const HostPlan &hp, const TileConsts &tc, const fcpp_polys *polys, ImageLayout &lay, std::string &err)
{
    const int64_t n = (int64_t)hp.fields.size(), nb = (int64_t)hp.blocks.size();
    std::vector<BlockTiles> &B = *blocks_;
    B.clear();
    B.resize((size_t)nb);
    WorkerPool::parallel_for(nb, [&](int64_t b) {
        const PlanBlock &pb = hp.blocks[(size_t)b];
        BlockTiles &bt = B[(size_t)b];
        FieldTiler ft(hp, tc, bt);
        for (int64_t fi = pb.f0; fi < pb.f1; ++fi) {
            const int k = (int)(fi - pb.f0);
            bt.tile0[k] = (int64_t)bt.tiles.size(); bt.w0[k] = (int64_t)bt.wtiles.size();
            const int64_t proto = hp.same_as.empty() ? -1 : hp.same_as[(size_t)fi];
            // (equal fields may differ in their obstacles, which do not change the plan -- but whether the passes form a span: span_line_max)
            if (proto >= 0 && (hp.fields[(size_t)fi].obs_count > 0) == (hp.fields[(size_t)proto].obs_count > 0)) ft.copy_field(fi, proto);
            else ft.tile_field(fi);
            bt.tile0[k + 1] = (int64_t)bt.tiles.size(); bt.w0[k + 1] = (int64_t)bt.wtiles.size();
            ft.derive_field(fi, k, bt.tile0[k], bt.tile0[k + 1]);
        }
    });
    lay = ImageLayout();
    lay.n_fields = n; lay.n_prims = hp.total_prims; lay.wave_tile_points = tc.wave_points;
    for (int64_t b = 0; b < nb; ++b) {
        BlockTiles &bt = B[(size_t)b];
        bt.tile_base = lay.n_tiles; bt.wave_base = lay.n_wave; bt.general_base = lay.n_general; bt.stat_base = lay.n_stat;
        bt.chunk_base = lay.n_chunks; bt.span_base = lay.n_span_chunks;
        // (the device's tile table holds ONE tile per statistics entry -- a general tile, a wave tile, or the FIRST tile of a quiet run, which is
        // all the kernels ever read of a run: at dense sampling the other tiles of the runs were half of the image)
        lay.n_tiles += (int64_t)bt.stat_ids.size(); lay.n_wave += (int64_t)bt.wtiles.size(); lay.n_general += (int64_t)bt.general_ids.size();
        lay.n_stat += (int64_t)bt.stat_ids.size();
        lay.n_chunks += tc.device_chunks ? bt.n_chunk_rec : (int64_t)bt.chunks.size(); lay.n_span_chunks += tc.device_chunks ? bt.n_span_rec : (int64_t)bt.span_chunks.size();
        bt.group_base = lay.n_chunk_groups; lay.n_chunk_groups += (int64_t)bt.groups.size();
        for (int c = 0; c < 4; ++c) { bt.cls_base[c] = lay.n_red[c]; lay.n_red[c] += (int64_t)bt.cls[c].size(); }
        for (int c = 0; c < 4; ++c) { bt.work_base[c] = lay.n_work[c]; lay.n_work[c] += (int64_t)bt.work[c].size(); lay.n_field_work += (int64_t)bt.work[c].size(); }
        bt.open_base = lay.n_open_wave; lay.n_open_wave += (int64_t)bt.open_wave.size();
        lay.n_runs += bt.n_runs; lay.quiet_points += bt.quiet_points; lay.wave_points += bt.wave_points; lay.work_wave_points += bt.work_wave_points; lay.work_span_points += bt.work_span_points; lay.unfusable_work += bt.unfusable_work;
        lay.span_points += bt.span_points; lay.chunk_points += bt.chunk_points; lay.wave_inside += bt.wave_inside;
        for (int k = 0; k < 5; ++k) lay.wave_fail[k] += bt.wave_fail[k];
    }
    if (lay.n_tiles > INT32_MAX) { err = "too many tiles in one batch: split the batch"; return FCPP_ESIZE; }
    lay.n_polys = polys ? polys->n_polys : 0;
    lay.n_poly_verts = lay.n_polys > 0 ? polys->offsets[lay.n_polys] : 0;
    layout_image(lay);
    return FCPP_OK;
}

void BatchTiler::fill(const HostPlan &hp, const fcpp_polys *polys, const ImageLayout &lay, unsigned char *dst) const
{
    const int64_t n = lay.n_fields, nb = (int64_t)hp.blocks.size();
    const std::vector<BlockTiles> &B = *blocks_;
    // the classes of the reduction one after the other: [<= 64 entries | <= 256 | <= reduce_wg_max | more], fields in order inside each
    int64_t cls_first[4] = { 0, lay.n_red[0], lay.n_red[0] + lay.n_red[1], lay.n_red[0] + lay.n_red[1] + lay.n_red[2] };
    WorkerPool::parallel_for(nb, [&](int64_t b) {
        const PlanBlock &pb = hp.blocks[(size_t)b];
        const BlockTiles &bt = B[(size_t)b];
        const int64_t nf = pb.f1 - pb.f0;
        memcpy(at<DevField>(dst, lay.fields) + pb.f0, hp.fields.data() + pb.f0, (size_t)nf * sizeof(DevField));
        if (!pb.prims.empty()) memcpy(at<DevPrim>(dst, lay.prims) + pb.prim_base, pb.prims.data(), pb.prims.size() * sizeof(DevPrim));
        const int32_t sb = (int32_t)bt.stat_base;
        // the tile of statistics entry e lies in slot e (bt.stat_ids: the entry's tile among the block's tiles)
        DevTile *td = at<DevTile>(dst, lay.tiles) + bt.stat_base;
        for (size_t e = 0; e < bt.stat_ids.size(); ++e) {
            td[e] = bt.tiles[(size_t)bt.stat_ids[e]];
            if (td[e].quiet == 0) td[e].stat_tile += sb;            // general tiles: their statistics entry
            else if (td[e].quiet != 5) td[e].stat_tile = 0;         // quiet runs: (the host kept the run's length there)
        }
        DevWaveTile *w = at<DevWaveTile>(dst, lay.wtiles) + bt.wave_base;
        for (size_t k = 0; k < bt.wtiles.size(); ++k) { w[k] = bt.wtiles[k]; w[k].tile += sb; }
        int32_t *g = at<int32_t>(dst, lay.general_ids) + bt.general_base;
        for (size_t k = 0; k < bt.general_ids.size(); ++k) g[k] = bt.tiles[(size_t)bt.general_ids[k]].stat_tile + sb;      // (a general tile's slot = its entry)
        int32_t *si = at<int32_t>(dst, lay.stat_ids) + bt.stat_base;
        for (size_t k = 0; k < bt.stat_ids.size(); ++k) si[k] = sb + (int32_t)k;
        if (!bt.stat_run.empty()) memcpy(at<int64_t>(dst, lay.stat_run) + bt.stat_base, bt.stat_run.data(), bt.stat_run.size() * sizeof(int64_t));
        if (lay.n_chunk_groups > 0) {
            DevChunkGroup *gr = at<DevChunkGroup>(dst, lay.chunk_groups) + bt.group_base;
            for (size_t k = 0; k < bt.groups.size(); ++k) { gr[k] = bt.groups[k]; gr[k].e0 += sb; gr[k].chunk_base += bt.chunk_base; gr[k].span_base += bt.span_base; }
        } else {
            DevTile *c = at<DevTile>(dst, lay.chunks) + bt.chunk_base;
            for (size_t k = 0; k < bt.chunks.size(); ++k) { c[k] = bt.chunks[k]; c[k].stat_tile += sb; }
            DevTile *cs = at<DevTile>(dst, lay.span_chunks) + bt.span_base;
            for (size_t k = 0; k < bt.span_chunks.size(); ++k) { cs[k] = bt.span_chunks[k]; cs[k].stat_tile += sb; }
        }
        int64_t *sf = at<int64_t>(dst, lay.stat_first);
        int64_t run = bt.stat_base;
        for (int64_t k = 0; k < nf; ++k) { sf[pb.f0 + k] = run; run += bt.stat_cnt[k]; }
        if (pb.f1 == n) sf[n] = run;
        int64_t class_off = 0;                    // the records lie class by class: every class is one launch
        for (int c = 0; c < 4; ++c) {
            DevFieldWork *fw = at<DevFieldWork>(dst, lay.field_work) + class_off + bt.work_base[c];
            for (size_t k = 0; k < bt.work[c].size(); ++k) { fw[k] = bt.work[c][k]; fw[k].w_first += (int32_t)bt.wave_base; fw[k].e_first += sb; }
            class_off += lay.n_work[c];
        }
        int32_t *ow = at<int32_t>(dst, lay.open_wave_ids) + bt.open_base;
        for (size_t k = 0; k < bt.open_wave.size(); ++k) ow[k] = bt.open_wave[k] + (int32_t)bt.wave_base;
        int32_t *rp = at<int32_t>(dst, lay.red_paths);
        for (int cidx = 0; cidx < 4; ++cidx)
            if (!bt.cls[cidx].empty())
                memcpy(rp + cls_first[cidx] + bt.cls_base[cidx], bt.cls[cidx].data(), bt.cls[cidx].size() * sizeof(int32_t));
        // connector segments (MLP:1313-1355): approach rows [0, n), departure rows [n, 2n)
        double *seg = at<double>(dst, lay.seg);
        int32_t *mask = at<int32_t>(dst, lay.seg_mask);
        for (int64_t i = pb.f0; i < pb.f1; ++i) {
            const fcpp_field_info &in = hp.info[(size_t)i];
            const bool okf = in.status == FCPP_OK;
            double *s = seg + (size_t)i * 4;
            s[0] = in.approach_from[0]; s[1] = in.approach_from[1]; s[2] = in.approach_to[0]; s[3] = in.approach_to[1];
            mask[i] = okf && in.start_kept;
            double *q = seg + (size_t)(n + i) * 4;
            q[0] = in.departure_from[0]; q[1] = in.departure_from[1]; q[2] = in.departure_to[0]; q[3] = in.departure_to[1];
            mask[n + i] = okf && in.end_kept;
        }
    });
    if (n == 0) *at<int64_t>(dst, lay.stat_first) = 0;
    // the packs of k_plan_sparse_fields: per field of field work its records gathered from the tables above (fcpp_internal.h: DevFieldPack)
    {
        const DevFieldWork *fw = at<DevFieldWork>(dst, lay.field_work);
        const DevWaveTile *wt = at<DevWaveTile>(dst, lay.wtiles);
        const DevField *fd = at<DevField>(dst, lay.fields);
        const DevPrim *pr = at<DevPrim>(dst, lay.prims);
        DevFieldPack *pk = at<DevFieldPack>(dst, lay.field_packs);
        const int64_t nw = lay.n_field_work, per = 256;
        WorkerPool::parallel_for((nw + per - 1) / per, [&](int64_t blk) {
            for (int64_t i = blk * per; i < std::min(nw, (blk + 1) * per); ++i) {
                DevFieldPack &P = pk[i];
                memset(&P, 0, sizeof P);
                P.work = fw[i];
                P.field = fd[fw[i].field];
                P.span_points = fw[i].fused_span;
                for (int t = 0; t < fw[i].n_tiles && t < FIELD_WORK_TILES; ++t) {
                    const DevWaveTile &w = wt[fw[i].w_first + t];
                    P.tile[t] = w;
                    const int nl = (int)w.hb + w.count + w.hf;
                    if (w.rel_main >= nl) continue;                 // no point of layer 2 in this tile
                    int np = 1;
                    for (int k = 0; k < 8; ++k) np += w.thr[k] != 255 ? 1 : 0;
                    for (int k = 0; k < np && k < PACK_TILE_PRIMS; ++k) P.prims[t][k] = pr[w.p0 + k];
                }
            }
        });
    }
    // diagnostic (FCPP_CHUNK_SPREAD=S): the ORDER of the chunk lists permuted so that consecutive workgroups write chunks N/S apart
    // instead of neighbours -- which memory the waves in flight cover at any moment (tools/placement_probe.py)
    if (const char *e = lay.n_chunk_groups == 0 ? getenv("FCPP_CHUNK_SPREAD") : nullptr) {     // (host-written lists only: fcpp_api.cpp)
        const int64_t S = atoll(e);
        for (int pass = 0; pass < 2 && S > 1; ++pass) {
            DevTile *L = at<DevTile>(dst, pass ? lay.span_chunks : lay.chunks);
            const int64_t N = pass ? lay.n_span_chunks : lay.n_chunks;
            if (N < 2 * S) continue;
            int64_t P = N / S;
            auto gcd = [](int64_t a, int64_t b) { while (b) { const int64_t t = a % b; a = b; b = t; } return a; };
            while (gcd(P, N) != 1) ++P;
            std::vector<DevTile> tmp(L, L + N);
            for (int64_t i = 0; i < N; ++i) L[i] = tmp[(size_t)((__int128)i * P % N)];
        }
    }
    fill_obstacles(polys, lay, dst);
}

}  // namespace fcpp
