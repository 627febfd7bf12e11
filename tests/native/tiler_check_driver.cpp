// The batch builder (fcpp_host.cpp: per-field plan, fcpp_tiler.cpp: tiler + image) on the CPU, under AddressSanitizer + UBSan
// (tests/test_host_sanitizers.py builds and runs this; the tiler is pure C++).  The turn templates, which the library samples on
// the device, are sampled here with the host's libm -- the tiler only takes distances and geofence margins from them.
//
// Per round: a random batch (rectangles, parallelograms, repeated fields, hostile fields, obstacles, every sampling mode) is planned,
// tiled and written into an image; then the image is CHECKED: every table entry inside its bounds, and every path point of every
// field produced by exactly one piece of kernel work (a general tile, the output lanes of a wave tile, a chunk of a quiet run).
// Prints two checksums: `raw` over the image bytes -- must not depend on FCPP_THREADS --, and `semantic` over the per-field content
// with all batch-position indices removed -- must not depend on FCPP_NO_SHARE (equal fields planned once and copied vs. each alone).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <random>
#include <string>
#include <vector>

#include "../../field_coverage_path_planning_amd/csrc/fcpp_geom.h"
#include "../../field_coverage_path_planning_amd/csrc/fcpp_tiler.h"

using namespace fcpp;

static uint64_t fnv(uint64_t h, const void *p, size_t n)
{
    const unsigned char *b = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}
#define FAIL(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, " (round %d)\n", round); return 2; } while (0)

int main(int argc, char **argv)
{
    const uint64_t seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    const int rounds = argc > 2 ? atoi(argv[2]) : 100;
    std::mt19937_64 rng(seed);
    auto U = [&](double a, double b) { return a + (b - a) * (double)(rng() >> 11) * (1.0 / 9007199254740992.0); };
    auto pick = [&](int n) { return (int)(rng() % (uint64_t)n); };
    uint64_t raw = 1469598103934665603ull, sem = raw;
    int64_t fields_ok = 0, fields_refused = 0, tiles_total = 0, wave_total = 0, shared = 0, closed_tiles = 0, halo_diff = 0;
    for (int round = 0; round < rounds; ++round) {
        fcpp_vehicle veh = { 3.2, 8.0, 9.0, 15.0, 4.0, 2.0, 1.5, 0.85 };
        if (pick(3) == 0) { veh.working_width = U(1.0, 8.0); veh.min_turn_radius = U(3.0, 15.0); veh.max_longitudinal_accel = U(0.1, 3.0); }
        fcpp_options opt = { pick(2), 1, 0.0, 0.5, pick(4) == 0 ? -0.5 : 1e-6, pick(4) == 0, 0 };
        const int sp = pick(5);
        opt.sample_spacing = sp <= 1 ? 0.0 : (sp == 2 ? U(0.3, 3.0) : (sp == 3 ? 0.5 : U(0.08, 0.25)));
        const int n = pick(8) == 0 ? 0 : 1 + pick(pick(4) == 0 ? 300 : 40);
        std::vector<fcpp_field> fields((size_t)n);
        std::vector<int64_t> offsets(1, 0);
        std::vector<double> px, py;
        for (int i = 0; i < n; ++i) {
            fcpp_field f = {};
            if (i > 0 && pick(3) == 0) {         // a repeat of an earlier field (shared plans)
                f = fields[(size_t)pick(i)];
                f.n_obstacles = 0; f.obstacle_first = (int64_t)offsets.size() - 1;
                fields[(size_t)i] = f;
                continue;
            }
            const double big = opt.sample_spacing > 0 && opt.sample_spacing < 0.3 ? 0.35 : 1.0;   // (keep dense rounds small)
            const double L = U(60.0, 900.0 * big), H = U(40.0, 600.0 * big);
            const int kind = pick(10);
            if (kind < 5) { f.vx[1] = L; f.vx[2] = L; f.vy[2] = H; f.vy[3] = H; }
            else if (kind < 9) {
                const double ang = U(1.0, 2.1), rot = U(-0.8, 0.8), sx = H / tan(ang);
                const double qx[4] = { 0, L, L + sx, sx }, qy[4] = { 0, 0, H, H };
                for (int k = 0; k < 4; ++k) { f.vx[k] = qx[k] * cos(rot) - qy[k] * sin(rot); f.vy[k] = qx[k] * sin(rot) + qy[k] * cos(rot); }
                f.from_vertices = 1;
            } else {
                for (int k = 0; k < 4; ++k) { f.vx[k] = U(-300, 300); f.vy[k] = U(-300, 300); }
                f.from_vertices = 1;
            }
            f.has_start = pick(2); f.has_end = pick(2);
            f.start_x = U(-20, L + 20); f.start_y = U(-20, H + 20); f.end_x = U(-20, L + 20); f.end_y = U(-20, H + 20);
            f.obstacle_first = (int64_t)offsets.size() - 1;
            f.n_obstacles = pick(4) == 0 ? pick(4) : 0;
            for (int o = 0; o < f.n_obstacles; ++o) {
                const int nv = 3 + pick(6);
                const double cx = U(0.2 * L, 0.8 * L), cy = U(0.2 * H, 0.8 * H), r = U(2.0, 25.0);
                for (int k = 0; k < nv; ++k) { const double a = 6.283185307179586 * k / nv; px.push_back(cx + r * cos(a)); py.push_back(cy + r * sin(a)); }
                offsets.push_back((int64_t)px.size());
            }
            fields[(size_t)i] = f;
        }
        fcpp_polys polys = { (int64_t)offsets.size() - 1, offsets.data(), px.data(), py.data() };
        HostPlan hp;
        std::string err;
        int rc = build_host_plan(veh, opt, n, fields.data(), &polys, true, hp, err);
        if (rc != FCPP_OK) FAIL("build_host_plan: %s", err.c_str());
        for (int i = 0; i < n; ++i) shared += hp.same_as[(size_t)i] >= 0;
        // templates as k_build_templates samples them (fcpp_fused.hip)
        const TurnTemplates &tt = hp.tt;
        std::vector<Pt2> tu((size_t)tt.nu), tcn((size_t)tt.nc);
        const CacShape s_pi = make_cac_shape(kPi, opt.clothoid_frac), s_half = make_cac_shape(kHalfPi, opt.clothoid_frac);
        for (int k = 0; k < tt.nu; ++k) {
            const double sv = linspace_at(0.0, tt.u_end, tt.u_step, tt.nu, k);
            if (tt.turn_model == FCPP_TURN_ARC) tu[(size_t)k] = { tt.R * cos(sv), tt.R * sin(sv) };
            else { double X, Y; cac_unit_point(s_pi, sv / tt.u_Re, X, Y); tu[(size_t)k] = { tt.u_Re * Y, tt.u_Re * X }; }
        }
        for (int k = 0; k < tt.nc; ++k) {
            const double sv = linspace_at(0.0, tt.c_end, tt.c_step, tt.nc, k);
            if (tt.turn_model == FCPP_TURN_ARC) tcn[(size_t)k] = { tt.R * (1 - cos(sv)), tt.R * sin(sv) };
            else { double X, Y; cac_unit_point(s_half, sv / tt.c_Re, X, Y); tcn[(size_t)k] = { tt.c_Re * Y, tt.c_Re * X }; }
        }
        TileConsts tc;
        tc.tu = tu.data(); tc.tc = tcn.data(); tc.nu = tt.nu; tc.nc = tt.nc; tc.templates_ok = true;
        tc.turn_quiet = pick(4) != 0;
        tc.wave_points = pick(3) ? 128 : 64;
        tc.field_work = pick(4) != 0;
        tc.fuse_spans = tc.field_work && tt.nu <= TMPL_LDS_SAMPLES && pick(3) != 0;
        const double vm = 15.0 / 3.6;
        tc.two_a = 2 * veh.max_longitudinal_accel; tc.u_cap = vm * vm; tc.c_line = (9.0 / 3.6) * (9.0 / 3.6);
        tc.fence_margin = 1e-7 - opt.geofence_tol;
        // the closed-form cut (fcpp_cutfn.h) at the reference's sampling, as fcpp_batch_create switches it on; its constants as the library makes
        // them: chord tables of the templates (k_build_template_metrics), the jump from a turn's last sample to the next line (closed_form_turns)
        std::vector<Pt2> dku((size_t)tt.nu), dkc((size_t)tt.nc);
        for (int k = 1; k < tt.nu; ++k) { const double dx = tu[(size_t)k].x - tu[(size_t)k - 1].x, dy = tu[(size_t)k].y - tu[(size_t)k - 1].y; dku[(size_t)k] = { sqrt(dx * dx + dy * dy), 0.0 }; }
        for (int k = 1; k < tt.nc; ++k) { const double dx = tcn[(size_t)k].x - tcn[(size_t)k - 1].x, dy = tcn[(size_t)k].y - tcn[(size_t)k - 1].y; dkc[(size_t)k] = { sqrt(dx * dx + dy * dy), 0.0 }; }
        tc.closed_cut = opt.sample_spacing == 0.0 && opt.obstacle_mode == FCPP_OBSTACLES_FLAG && tc.wave_points == CUT_WAVE_LANES && pick(5) != 0;
        {
            CutConsts &cc = tc.cut;
            memset(&cc, 0, sizeof cc);
            cc.tu = tu.data(); cc.tc = tcn.data(); cc.dk_u = dku.data(); cc.dk_c = dkc.data(); cc.nu = tt.nu; cc.nc = tt.nc;
            cc.turn_quiet = tc.turn_quiet; cc.wave_factor = tc.wave_factor; cc.two_a = tc.two_a; cc.u_cap = tc.u_cap; cc.c_line = tc.c_line; cc.fence_margin = tc.fence_margin;
            if (tt.nu >= 1) {
                const bool arc = tt.turn_model == FCPP_TURN_ARC;
                const Pt2 tl = tu[(size_t)tt.nu - 1];
                for (int v = 0; v < 2; ++v) {
                    const double jx = arc ? (-tt.R + tl.x) : -tl.x, jy = (v == 0 ? veh.working_width : -veh.working_width) - tl.y;
                    cc.jump[v] = sqrt(jx * jx + jy * jy);
                }
            }
            cc.u_step_min = tt.nu >= 2 ? HUGE_VAL : 0.0; cc.c_step_min = tt.nc >= 2 ? HUGE_VAL : 0.0;
            for (int k = 1; k < tt.nu; ++k) cc.u_step_min = std::min(cc.u_step_min, dku[(size_t)k].x);
            for (int k = 1; k < tt.nc; ++k) cc.c_step_min = std::min(cc.c_step_min, dkc[(size_t)k].x);
            for (int k = 0; k < tt.nc; ++k) {
                cc.tc_lo[0] = std::min(cc.tc_lo[0], tcn[(size_t)k].x); cc.tc_hi[0] = std::max(cc.tc_hi[0], tcn[(size_t)k].x);
                cc.tc_lo[1] = std::min(cc.tc_lo[1], tcn[(size_t)k].y); cc.tc_hi[1] = std::max(cc.tc_hi[1], tcn[(size_t)k].y);
            }
        }
        BatchTiler tiler;
        ImageLayout lay;
        rc = tiler.plan(hp, tc, &polys, lay, err);
        if (rc != FCPP_OK) FAIL("tiler.plan: %s", err.c_str());
        if (tc.fuse_spans && lay.unfusable_work > 0) {          // all spans of fields of field work are fused, or none (fcpp_api.cpp)
            tc.fuse_spans = false;
            rc = tiler.plan(hp, tc, &polys, lay, err);
            if (rc != FCPP_OK) FAIL("tiler.plan (again): %s", err.c_str());
            if (lay.work_span_points != 0) FAIL("spans fused although fusing is off");
        }
        std::vector<unsigned char> img(lay.upload_bytes, 0);
        tiler.fill(hp, &polys, lay, img.data());
        raw = fnv(raw, img.data(), img.size());
        tiles_total += lay.n_tiles; wave_total += lay.n_wave;

        // ---- the image, checked
        const DevField *F = reinterpret_cast<const DevField *>(img.data() + lay.fields);
        const DevPrim *P = reinterpret_cast<const DevPrim *>(img.data() + lay.prims);
        const DevTile *T = reinterpret_cast<const DevTile *>(img.data() + lay.tiles);
        const DevWaveTile *Wt = reinterpret_cast<const DevWaveTile *>(img.data() + lay.wtiles);
        const int32_t *G = reinterpret_cast<const int32_t *>(img.data() + lay.general_ids);
        const DevTile *C = reinterpret_cast<const DevTile *>(img.data() + lay.chunks), *CS = reinterpret_cast<const DevTile *>(img.data() + lay.span_chunks);
        const int32_t *SI = reinterpret_cast<const int32_t *>(img.data() + lay.stat_ids);
        const int64_t *SF = reinterpret_cast<const int64_t *>(img.data() + lay.stat_first), *SR = reinterpret_cast<const int64_t *>(img.data() + lay.stat_run);
        const int32_t *RP = reinterpret_cast<const int32_t *>(img.data() + lay.red_paths);
        if (lay.n_fields != n) FAIL("field count");
        int64_t total = 0;
        for (int i = 0; i < n; ++i) {
            if (F[i].pt_off != total) FAIL("pt_off of field %d", i);
            if (F[i].n_total < 0 || F[i].prim_count < 0 || F[i].prim_first < 0 || F[i].prim_first + F[i].prim_count > lay.n_prims) FAIL("primitive range of field %d", i);
            total += F[i].n_total;
            (hp.info[(size_t)i].status == FCPP_OK ? fields_ok : fields_refused) += 1;
        }
        if (total != hp.total_points) FAIL("total points");
        std::vector<unsigned char> cover((size_t)total, 0);
        auto mark = [&](int64_t field, int64_t start, int64_t count, const char *what) -> bool {
            if (field < 0 || field >= n || count <= 0 || start < 0 || start + count > F[field].n_total) { fprintf(stderr, "%s outside its field\n", what); return false; }
            for (int64_t k = 0; k < count; ++k) if (cover[(size_t)(F[field].pt_off + start + k)]++) { fprintf(stderr, "%s: point planned twice\n", what); return false; }
            return true;
        };
        for (int64_t k = 0; k < lay.n_general; ++k) {
            if (G[k] < 0 || G[k] >= lay.n_tiles || T[G[k]].quiet != 0) FAIL("general id %lld", (long long)k);
            if (!mark(T[G[k]].field, T[G[k]].start, T[G[k]].count, "general tile") || T[G[k]].count > TILE_POINTS) FAIL("general tile %lld", (long long)k);
            if (T[G[k]].stat_tile < 0 || T[G[k]].stat_tile >= lay.n_stat || SI[T[G[k]].stat_tile] != G[k]) FAIL("general tile %lld: statistics entry", (long long)k);
        }
        for (int64_t k = 0; k < lay.n_wave; ++k) {
            const DevWaveTile &w = Wt[k];
            if (w.tile < 0 || w.tile >= lay.n_stat || w.field < 0 || w.field >= n || w.tile < SF[w.field] || w.tile >= SF[w.field + 1]) FAIL("wave tile %lld: statistics entry", (long long)k);
            const int32_t wtile = SI[w.tile];
            if (wtile < 0 || wtile >= lay.n_tiles || T[wtile].quiet != 5 || T[wtile].field != w.field || T[wtile].count != w.count) FAIL("wave tile %lld: slot", (long long)k);
            if (w.hb + w.count + w.hf > tc.wave_points || w.field < 0 || w.field >= n) FAIL("wave tile %lld: lanes", (long long)k);
            const int64_t first = w.out_base - F[w.field].pt_off;
            if (first < 0 || first + w.hb != T[wtile].start || first + w.hb + w.count + w.hf > F[w.field].n_total) FAIL("wave tile %lld: range", (long long)k);
            if (!mark(w.field, first + w.hb, w.count, "wave tile")) FAIL("wave tile %lld", (long long)k);
            const int64_t last = first + w.hb + w.count + w.hf - 1;
            if (last >= F[w.field].gen_main) {
                int np = 1;
                for (int q = 0; q < 8; ++q) np += w.thr[q] != 255;
                if (w.p0 < F[w.field].prim_first || w.p0 + np > F[w.field].prim_first + F[w.field].prim_count) FAIL("wave tile %lld: primitives", (long long)k);
                // lane l of layer 2 = sample (l + r0) of primitive p0, (l - thr[q]) of primitive p0 + 1 + q
                const int64_t fl2 = first > F[w.field].gen_main ? first : F[w.field].gen_main;
                if (P[w.p0].start + (fl2 - first) + w.r0 != fl2) FAIL("wave tile %lld: r0", (long long)k);
                for (int q = 0; q < 8 && w.thr[q] != 255; ++q) {
                    if (P[w.p0 + 1 + q].start != first + w.thr[q]) FAIL("wave tile %lld: threshold %d", (long long)k, q);
                    if (w.thr[q] < 1 || w.thr[q] >= 128 || (q > 0 && w.thr[q] <= w.thr[q - 1])) FAIL("wave tile %lld: thresholds not strictly ascending in [1, 128)", (long long)k);
                }
            }
        }
        if (tc.closed_cut) {
            // The halos of the closed-form cut against the path's own points: every wave tile's halos must be what the halo walks of
            // fcpp_tilefn.h give on the EVALUATED step lengths (the formulas of the kernels), and a tile that says `inside` must have every output
            // point inside the geofence with the tiler's margin.
            const double cap = tiler_halo_cap(tc.u_cap);
            for (int64_t k = 0; k < lay.n_wave; ++k) {
                const DevWaveTile &w = Wt[k];
                const DevField &f = F[w.field];
                if (!(cut_applies(f, tc.cut, cut_span_points(f, tc.cut)))) continue;
                ++closed_tiles;
                const DevPrim *fp = P + f.prim_first;
                const int64_t per = (int64_t)f.n_line + f.n_turn;
                auto point = [&](int64_t i, double &x, double &y) {
                    if (i < f.gen_main) { tiler_point_main(f, tu.data(), i / per, i % per, x, y); return; }
                    int a = 0, b = f.prim_count - 1;
                    while (a < b) { const int m = (a + b + 1) >> 1; if (fp[m].start <= i) a = m; else b = m - 1; }
                    tiler_point_prim(fp[a], tu.data(), tcn.data(), i - fp[a].start, x, y);
                };
                auto dist = [&](int64_t i) { double x0, y0, x1, y1; point(i - 1, x0, y0); point(i, x1, y1); return sqrt((x1 - x0) * (x1 - x0) + (y1 - y0) * (y1 - y0)); };
                const int64_t first = w.out_base - f.pt_off, s0 = first + w.hb, e0 = s0 + w.count - 1;
                const int hb = tiler_back_halo(dist, s0, tc.two_a, cap), hf = tiler_fwd_halo(dist, e0, f.n_total, tc.two_a, cap);
                if (hb != w.hb || hf != w.hf) {
                    ++halo_diff;
                    // a halo SHORTER than the path's own step lengths ask for would let the sweeps carry speeds into the outputs: never
                    if (hb < 0 || hf < 0 || w.hb < hb || w.hf < hf) FAIL("wave tile %lld of field %d: halos %d / %d, the evaluated path asks for %d / %d", (long long)k, w.field, w.hb, w.hf, hb, hf);
                }
                if (w.inside)
                    for (int64_t i = s0; i <= e0; ++i) { double x, y; point(i, x, y); if (!tiler_inside(f, x, y, tc.fence_margin)) FAIL("wave tile %lld says inside, point %lld is not", (long long)k, (long long)i); }
                // the closed-form step lengths themselves, point by point over the tile
                {
                    std::vector<CutPrim> cp((size_t)f.prim_count);
                    bool ok = true;
                    double lx, ly;
                    cut_main_end(f, tc.cut, lx, ly);
                    for (int q = 0; q < f.prim_count; ++q) cp[(size_t)q] = cut_prim_info(fp[q], f, tc.cut, lx, ly, ok);
                    struct PV { const CutPrim *p; const CutPrim &operator()(int q) const { return p[q]; } } pv{ cp.data() };
                    CutDist<PV> cd(f, tc.cut, pv, f.prim_count);
                    for (int64_t i = std::max<int64_t>(first, 1); i < first + w.hb + w.count + w.hf; ++i) {
                        const double a = dist(i), c = cd(i);
                        if (fabs(a - c) > 1e-9 * (1.0 + a)) FAIL("step %lld of field %d: evaluated %.17g, closed form %.17g", (long long)i, w.field, a, c);
                    }
                }
            }
        }
        for (int pass = 0; pass < 2; ++pass) {
            const DevTile *L = pass ? CS : C;
            const int64_t nl = pass ? lay.n_span_chunks : lay.n_chunks;
            for (int64_t k = 0; k < nl; ++k) {
                const DevTile &c = L[k];
                if (!mark(c.field, c.start, c.count, "chunk")) FAIL("chunk %lld/%d", (long long)k, pass);
                const int64_t g = F[c.field].pt_off + c.start;
                if (g / TILE_POINTS != (g + c.count - 1) / TILE_POINTS) FAIL("chunk %lld/%d crosses a 512-point boundary", (long long)k, pass);
                if (c.stat_tile < 0 || c.stat_tile >= lay.n_stat || SR[c.stat_tile] <= 0 || T[SI[c.stat_tile]].field != c.field || T[SI[c.stat_tile]].quiet == 0 ||
                    T[SI[c.stat_tile]].quiet == 5 || c.start < T[SI[c.stat_tile]].start || c.start >= T[SI[c.stat_tile]].start + SR[c.stat_tile])       // (a chunk may run on into the next run of its group)
                    FAIL("chunk %lld/%d: statistics entry", (long long)k, pass);
                if ((pass == 1) != (c.quiet == 4)) FAIL("chunk %lld/%d: kind", (long long)k, pass);
                if (c.quiet == 2 && (c.idx0 < F[c.field].prim_first || c.idx0 >= F[c.field].prim_first + F[c.field].prim_count)) FAIL("chunk %lld: primitive", (long long)k);
            }
        }
        {   // the packs of k_plan_sparse_fields: the field's records gathered, and the span its workgroup writes itself (no chunks for it)
            const DevFieldWork *FW = reinterpret_cast<const DevFieldWork *>(img.data() + lay.field_work);
            const DevFieldPack *PK = reinterpret_cast<const DevFieldPack *>(img.data() + lay.field_packs);
            int64_t fused_pts = 0;
            for (int64_t k = 0; k < lay.n_field_work; ++k) {
                const DevFieldPack &pk = PK[k];
                if (memcmp(&pk.work, &FW[k], sizeof(DevFieldWork)) || memcmp(&pk.field, &F[FW[k].field], sizeof(DevField))) FAIL("pack %lld: work / field record", (long long)k);
                for (int t = 0; t < FIELD_WORK_TILES; ++t) {
                    DevWaveTile zero;
                    memset(&zero, 0, sizeof zero);
                    const DevWaveTile &want = t < FW[k].n_tiles ? Wt[FW[k].w_first + t] : zero;
                    if (memcmp(&pk.tile[t], &want, sizeof want)) FAIL("pack %lld: tile %d", (long long)k, t);
                    if (t < FW[k].n_tiles && (int)want.hb + want.count + want.hf > want.rel_main) {
                        int np = 1;
                        for (int q = 0; q < 8; ++q) np += want.thr[q] != 255;
                        for (int q = 0; q < np; ++q) if (memcmp(&pk.prims[t][q], &P[want.p0 + q], sizeof(DevPrim))) FAIL("pack %lld: tile %d primitive %d", (long long)k, t, q);
                    }
                }
                if (pk.span_points != FW[k].fused_span || pk.span_points < 0) FAIL("pack %lld: span points", (long long)k);
                if (pk.span_points > 0) {
                    if (!tc.fuse_spans) FAIL("pack %lld: a fused span although fusing is off", (long long)k);
                    const int64_t e0 = FW[k].e_first, g0 = F[FW[k].field].pt_off;
                    if (SR[e0] != pk.span_points || T[SI[e0]].quiet != 4 || T[SI[e0]].start != 0) FAIL("pack %lld: the span's run entry", (long long)k);
                    if (((g0 % TILE_POINTS) + pk.span_points + TILE_POINTS - 1) / TILE_POINTS > FUSED_SPAN_CHUNKS) FAIL("pack %lld: span of too many chunks", (long long)k);
                    if (!mark(FW[k].field, 0, pk.span_points, "fused span")) FAIL("pack %lld: fused span", (long long)k);
                    fused_pts += pk.span_points;
                }
            }
            if (fused_pts != lay.work_span_points) FAIL("fused span points %lld, layout says %lld", (long long)fused_pts, (long long)lay.work_span_points);
        }
        for (int64_t g = 0; g < total; ++g) if (cover[(size_t)g] != 1) FAIL("point %lld planned %d times", (long long)g, cover[(size_t)g]);
        if (SF[0] != 0 || SF[n] != lay.n_stat) FAIL("stat_first ends");
        int64_t q_pts = 0;
        for (int i = 0; i < n; ++i) {
            if (SF[i + 1] < SF[i]) FAIL("stat_first of field %d", i);
            for (int64_t e = SF[i]; e < SF[i + 1]; ++e) {
                if (SI[e] < 0 || SI[e] >= lay.n_tiles || T[SI[e]].field != i) FAIL("stat entry %lld", (long long)e);
                if ((SR[e] > 0) != (T[SI[e]].quiet != 0 && T[SI[e]].quiet != 5)) FAIL("stat run %lld", (long long)e);
                q_pts += SR[e];
            }
        }
        if (q_pts != lay.quiet_points || lay.span_points + lay.chunk_points + lay.work_span_points != lay.quiet_points) FAIL("quiet point totals");
        {
            std::vector<unsigned char> seen((size_t)n, 0);
            // fields planned and reduced by one workgroup (DevFieldWork) are in no class; their wave tiles are not in the open list
            const DevFieldWork *FW = reinterpret_cast<const DevFieldWork *>(img.data() + lay.field_work);
            const int32_t *OW = reinterpret_cast<const int32_t *>(img.data() + lay.open_wave_ids);
            std::vector<unsigned char> wseen((size_t)lay.n_wave, 0);
            for (int64_t k = 0; k < lay.n_field_work; ++k) {
                const DevFieldWork &w = FW[k];
                if (w.field < 0 || w.field >= n || seen[(size_t)w.field]++) FAIL("field work %lld: field", (long long)k);
                if (w.e_first != SF[w.field] || w.n_entries != SF[w.field + 1] - SF[w.field] || w.n_entries > FIELD_WORK_ENTRIES) FAIL("field work %lld: entries", (long long)k);
                if (w.n_tiles < 1 || w.n_tiles > FIELD_WORK_TILES || w.w_first < 0 || w.w_first + w.n_tiles > lay.n_wave) FAIL("field work %lld: tiles", (long long)k);
                {   // the records lie class by class (one launch each), a field in the class of its tile count
                    int c = 0;
                    for (int64_t lim = lay.n_work[0]; c < 3 && k >= lim; lim += lay.n_work[++c]) {}
                    if (field_work_class(w.n_tiles) != c || w.n_tiles > FIELD_WORK_WAVES[c]) FAIL("field work %lld: class", (long long)k);
                }
                int nt = 0;
                for (int64_t e = SF[w.field]; e < SF[w.field + 1]; ++e) {
                    if (SR[e] == 0) { if (T[SI[e]].quiet != 5 || Wt[w.w_first + nt].tile != e) FAIL("field work %lld: entry %lld is not its tile %d", (long long)k, (long long)e, nt); ++nt; }
                }
                if (nt != w.n_tiles) FAIL("field work %lld: %d tiles among the entries, %d in the record", (long long)k, nt, w.n_tiles);
                for (int q = 0; q < w.n_tiles; ++q) { if (Wt[w.w_first + q].field != w.field || wseen[(size_t)(w.w_first + q)]++) FAIL("field work %lld: tile %d", (long long)k, q); }
            }
            for (int64_t k = 0; k < lay.n_open_wave; ++k) {
                if (OW[k] < 0 || OW[k] >= lay.n_wave || wseen[(size_t)OW[k]]++) FAIL("open wave tile %lld", (long long)k);
                if (k > 0 && OW[k] <= OW[k - 1]) FAIL("open wave list not ascending");
            }
            for (int64_t k = 0; k < lay.n_wave; ++k) if (wseen[(size_t)k] != 1) FAIL("wave tile %lld planned %d times", (long long)k, wseen[(size_t)k]);
            const int64_t nr = lay.n_red[0] + lay.n_red[1] + lay.n_red[2] + lay.n_red[3] + lay.n_field_work;
            if (nr != n) FAIL("reduction classes and field work hold %lld of %d fields", (long long)nr, n);
            int64_t at = 0;
            for (int c = 0; c < 4; ++c)
                for (int64_t k = 0; k < lay.n_red[c]; ++k, ++at) {
                    const int32_t f = RP[at];
                    if (f < 0 || f >= n || seen[(size_t)f]++) FAIL("reduction list entry %lld", (long long)at);
                    if (k > 0 && RP[at - 1] >= f) FAIL("reduction list not ascending at %lld", (long long)at);
                    const int64_t ne = SF[f + 1] - SF[f];
                    if ((ne <= 64 ? 0 : (ne <= 256 ? 1 : (ne <= tc.reduce_wg_max ? 2 : 3))) != c) FAIL("reduction class of field %d", f);
                }
        }
        // ---- semantic checksum: per field, everything with the batch-position indices taken out
        {
            std::vector<int64_t> tile_first_of((size_t)n + 1, lay.n_tiles);
            for (int64_t k = lay.n_tiles - 1; k >= 0; --k) tile_first_of[(size_t)T[k].field] = k;
            for (int i = n - 1; i >= 0; --i) if (tile_first_of[(size_t)i] == lay.n_tiles) tile_first_of[(size_t)i] = tile_first_of[(size_t)i + 1];
            for (int64_t k = 1; k < lay.n_tiles; ++k) if (T[k].field < T[k - 1].field) FAIL("tiles not in field order");
            for (int i = 0; i < n; ++i) {
                DevField f = F[i];
                const int64_t pf = f.prim_first, tf = tile_first_of[(size_t)i];
                f.prim_first = 0; f.pt_off = 0; f.obs_first = 0;
                sem = fnv(sem, &f, sizeof f);
                sem = fnv(sem, P + pf, (size_t)F[i].prim_count * sizeof(DevPrim));
                for (int64_t k = tf; k < lay.n_tiles && T[k].field == i; ++k) {
                    DevTile t = T[k];
                    t.field = 0;
                    // (idx0 is a pass index or a batch-wide primitive index, depending on where the tile starts: quiet runs of layer 2 are
                    // covered below through their chunks' primitive check, the wave tiles through their records)
                    sem = fnv(sem, &t.count, sizeof t.count); sem = fnv(sem, &t.start, sizeof t.start); sem = fnv(sem, &t.quiet, sizeof t.quiet);
                    if (t.quiet == 0) t.stat_tile -= (int32_t)SF[i];
                    sem = fnv(sem, &t.off0, sizeof t.off0); sem = fnv(sem, &t.stat_tile, sizeof t.stat_tile);
                }
                for (int64_t e = SF[i]; e < SF[i + 1]; ++e) { const int64_t rel = SI[e] - tf; sem = fnv(sem, &rel, 8); sem = fnv(sem, &SR[e], 8); }
            }
            for (int64_t k = 0; k < lay.n_wave; ++k) {
                DevWaveTile w = Wt[k];
                const DevField &f = F[w.field];
                w.out_base -= f.pt_off; w.tile -= (int32_t)SF[w.field];
                if (w.out_base + w.hb + w.count + w.hf - 1 >= f.gen_main) w.p0 -= f.prim_first; else w.p0 = 0;
                if (w.out_base >= f.gen_main) w.idx0 -= f.prim_first;
                w.field = 0;
                sem = fnv(sem, &w, sizeof w);
            }
        }
    }
    printf("rounds %d fields %lld refused %lld shared %lld tiles %lld wave %lld raw %llu semantic %llu closed_tiles %lld halo_diff %lld\n", rounds, (long long)fields_ok,
           (long long)fields_refused, (long long)shared, (long long)tiles_total, (long long)wave_total, (unsigned long long)raw, (unsigned long long)sem,
           (long long)closed_tiles, (long long)halo_diff);
    return 0;
}
