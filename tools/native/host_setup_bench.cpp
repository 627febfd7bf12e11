// Diagnostic: where the HOST setup of a batch spends its time (fcpp_host.cpp: per-field plan; fcpp_tiler.cpp: tiler + image), without a GPU.
//   g++ -std=c++17 -O3 -ffp-contract=off -pthread -o build/host_setup_bench tools/native/host_setup_bench.cpp \
//       field_coverage_path_planning_amd/csrc/fcpp_host.cpp field_coverage_path_planning_amd/csrc/fcpp_tiler.cpp
//   build/host_setup_bench cfg2 1024 1 0.5        (workload: cfg1 = equal 500 x 200 m fields, cfg2 = random rectangles U[100, 1000) m; fields; turn model; spacing)
// The turn templates are sampled with the host's libm here (the library samples them on the device): timings only.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <random>
#include <string>
#include <vector>

#include "../../field_coverage_path_planning_amd/csrc/fcpp_geom.h"
#include "../../field_coverage_path_planning_amd/csrc/fcpp_tiler.h"

using namespace fcpp;
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const std::string what = argc > 1 ? argv[1] : "cfg2";
    const int n = argc > 2 ? atoi(argv[2]) : 1024;
    fcpp_vehicle veh = { 3.2, 8.0, 9.0, 15.0, 4.0, 2.0, 1.5, 0.85 };
    fcpp_options opt = { argc > 3 ? atoi(argv[3]) : 1, 1, argc > 4 ? atof(argv[4]) : 0.5, 0.5, 1e-6, 0, 0 };
    std::mt19937_64 rng(1024);
    auto U = [&](double a, double b) { return a + (b - a) * (double)(rng() >> 11) * (1.0 / 9007199254740992.0); };
    std::vector<fcpp_field> fields((size_t)n);
    for (int i = 0; i < n; ++i) {
        fcpp_field f = {};
        const double L = what == "cfg1" ? 500.0 : U(100.0, 1000.0), H = what == "cfg1" ? 200.0 : U(100.0, 1000.0);
        f.vx[1] = L; f.vx[2] = L; f.vy[2] = H; f.vy[3] = H;
        fields[(size_t)i] = f;
    }
    int64_t zero = 0;
    fcpp_polys polys = { 0, &zero, nullptr, nullptr };
    // cfg3a: ONE 5000 x 2000 m field with 32 eight-gon obstacles on a jittered 8 x 4 grid, obstacle-aware swaths (cfg3_avoid of bench.py)
    std::vector<int64_t> offs(1, 0);
    std::vector<double> ox, oy;
    if (what == "cfg3a") {
        fields.resize(1);
        fcpp_field f = {};
        f.vx[1] = 5000; f.vx[2] = 5000; f.vy[2] = 2000; f.vy[3] = 2000;
        f.obstacle_first = 0; f.n_obstacles = 32;
        fields[0] = f;
        opt.obstacle_mode = FCPP_OBSTACLES_AVOID;
        for (int gy = 0; gy < 4; ++gy)
            for (int gx = 0; gx < 8; ++gx) {
                const double cx = (gx + 0.5) * 5000 / 8 + U(-100, 100), cy = (gy + 0.5) * 2000 / 4 + U(-100, 100), r = U(10, 40);
                for (int k = 0; k < 8; ++k) { ox.push_back(cx + r * cos(k * 0.7853981633974483)); oy.push_back(cy + r * sin(k * 0.7853981633974483)); }
                offs.push_back((int64_t)ox.size());
            }
        polys = { 32, offs.data(), ox.data(), oy.data() };
    }
    const int n_run = (int)fields.size();
    double best[4] = { 1e30, 1e30, 1e30, 1e30 };
    ImageLayout lay;
    const int n_reps = getenv("REPS") ? atoi(getenv("REPS")) : 9;
    for (int rep = 0; rep < n_reps; ++rep) {
        HostPlan hp;
        std::string err;
        double t0 = now_ms();
        int rc = build_host_plan(veh, opt, n_run, fields.data(), &polys, true, hp, err);
        if (rc != FCPP_OK) { fprintf(stderr, "build_host_plan: %s\n", err.c_str()); return 1; }
        double t1 = now_ms();
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
        tc.turn_quiet = true; tc.device_chunks = true;
        const double a_lon = veh.max_longitudinal_accel, vmax = 15.0 / 3.6, vw = 8.0 / 3.6;
        tc.two_a = 2 * a_lon; tc.u_cap = vmax * vmax; tc.c_line = vw * vw;
        if (getenv("FCPP_DENSE_SPAN") && atoi(getenv("FCPP_DENSE_SPAN")) <= 0) tc.span_line_max = 64;
        double t2 = now_ms();
        BatchTiler tiler;
        rc = tiler.plan(hp, tc, &polys, lay, err);
        if (rc != FCPP_OK) { fprintf(stderr, "tiler: %s\n", err.c_str()); return 1; }
        double t3 = now_ms();
        std::vector<unsigned char> img(lay.upload_bytes);
        double t4 = now_ms();
        tiler.fill(hp, &polys, lay, img.data());
        double t5 = now_ms();
        const double v[4] = { t1 - t0, t3 - t2, t5 - t4, (t1 - t0) + (t3 - t2) + (t5 - t4) };
        for (int k = 0; k < 4; ++k) best[k] = v[k] < best[k] ? v[k] : best[k];
    }
    printf("%s n=%d model=%d spacing=%g: host_plan %.3f ms  tiler %.3f ms  image %.3f ms  sum %.3f ms   image %.2f MB, %lld tiles, %lld wave tiles, %lld general, %lld chunk groups\n", what.c_str(), n,
           opt.turn_model, opt.sample_spacing, best[0], best[1], best[2], best[3], lay.upload_bytes / 1e6, (long long)lay.n_tiles, (long long)lay.n_wave, (long long)lay.n_general,
           (long long)lay.n_chunk_groups);
    return 0;
}
