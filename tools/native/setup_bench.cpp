// Host-side cost of fcpp_batch_create's setup phases (plan / tiler / image), without a GPU: the headline batch (n equal 500 x 200 m
// fields) and a cfg5-like batch (n random rotated parallelograms).  Templates sampled with the host's libm (as in
// tests/native/tiler_check_driver.cpp).  Build: g++ -O3 -std=c++17 -ffp-contract=off -pthread tools/native/setup_bench.cpp
//   field_coverage_path_planning_amd/csrc/fcpp_host.cpp field_coverage_path_planning_amd/csrc/fcpp_tiler.cpp -o build/setup_bench
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <random>
#include <string>
#include <vector>

#include "../../field_coverage_path_planning_amd/csrc/fcpp_geom.h"
#include "../../field_coverage_path_planning_amd/csrc/fcpp_parallel.h"
#include "../../field_coverage_path_planning_amd/csrc/fcpp_tiler.h"

using namespace fcpp;
static double ms(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }

int main(int argc, char **argv)
{
    const int which = argc > 1 ? atoi(argv[1]) : 0, n = argc > 2 ? atoi(argv[2]) : (which ? 65536 : 4096), reps = argc > 3 ? atoi(argv[3]) : 5;
    fcpp_vehicle veh = { 3.2, 8.0, 9.0, 15.0, 4.0, 2.0, 1.5, 0.85 };
    fcpp_options opt = { 0, 1, 0.0, 0.5, 1e-6, 0, 0 };
    if (argc > 4) opt.sample_spacing = atof(argv[4]);          // (which = 2: cfg2-like random rectangles, e.g. at 0.5 m)
    std::vector<fcpp_field> fields((size_t)n);
    std::mt19937_64 rng(65536);
    auto U = [&](double a, double b) { return a + (b - a) * (double)(rng() >> 11) * (1.0 / 9007199254740992.0); };
    for (int i = 0; i < n; ++i) {
        fcpp_field f = {};
        if (which == 0) { f.vx[1] = 500; f.vx[2] = 500; f.vy[2] = 200; f.vy[3] = 200; }
        else if (which == 2) { const double L = U(100, 1000), H = U(100, 1000); f.vx[1] = L; f.vx[2] = L; f.vy[2] = H; f.vy[3] = H; }
        else {
            const double L = U(100, 1000), H = U(100, 1000), ang = U(60, 120) * kPi / 180, rot = U(-kPi / 4, kPi / 4), sx = H / tan(ang);
            const double qx[4] = { 0, L, L + sx, sx }, qy[4] = { 0, 0, H, H };
            for (int k = 0; k < 4; ++k) { f.vx[k] = qx[k] * cos(rot) - qy[k] * sin(rot); f.vy[k] = qx[k] * sin(rot) + qy[k] * cos(rot); }
            f.from_vertices = 1;
        }
        fields[(size_t)i] = f;
    }
    printf("threads %d, %d fields (%s), sample_spacing %g\n", WorkerPool::width(), n, which == 2 ? "random rectangles" : (which ? "random parallelograms" : "equal 500 x 200 m rectangles"), opt.sample_spacing);
    {   // what a parallel_for costs when its items do nothing (the pool's wake-up and hand-back): 16 items, median of 200 calls
        std::vector<double> us;
        for (int k = 0; k < 200; ++k) {
            auto t0 = std::chrono::steady_clock::now();
            WorkerPool::parallel_for(16, [](int64_t) {});
            us.push_back(ms(t0) * 1e3);
        }
        std::sort(us.begin(), us.end());
        printf("parallel_for of 16 empty items: median %.1f us, 90 %% %.1f us\n", us[100], us[180]);
    }
    std::vector<unsigned char> img;
    for (int rep = 0; rep < reps; ++rep) {
        HostPlan hp;
        std::string err;
        auto t0 = std::chrono::steady_clock::now();
        if (build_host_plan(veh, opt, n, fields.data(), nullptr, true, hp, err) != FCPP_OK) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        const double t_plan = ms(t0);
        const TurnTemplates &tt = hp.tt;
        std::vector<Pt2> tu((size_t)tt.nu), tcn((size_t)tt.nc);
        for (int k = 0; k < tt.nu; ++k) { const double sv = linspace_at(0.0, tt.u_end, tt.u_step, tt.nu, k); tu[(size_t)k] = { tt.R * cos(sv), tt.R * sin(sv) }; }
        for (int k = 0; k < tt.nc; ++k) { const double sv = linspace_at(0.0, tt.c_end, tt.c_step, tt.nc, k); tcn[(size_t)k] = { tt.R * (1 - cos(sv)), tt.R * sin(sv) }; }
        TileConsts tc;
        tc.tu = tu.data(); tc.tc = tcn.data(); tc.nu = tt.nu; tc.nc = tt.nc; tc.templates_ok = true; tc.turn_quiet = true;
        tc.two_a = 3.0; tc.u_cap = (15 / 3.6) * (15 / 3.6); tc.c_line = 2.5 * 2.5;
        tc.device_chunks = getenv("FCPP_HOST_CHUNKS") == nullptr;       // (as fcpp_batch_create: the device expands the chunk lists)
        t0 = std::chrono::steady_clock::now();
        BatchTiler tiler;
        ImageLayout lay;
        if (tiler.plan(hp, tc, nullptr, lay, err) != FCPP_OK) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        const double t_tile = ms(t0);
        t0 = std::chrono::steady_clock::now();
        if (img.size() < lay.upload_bytes) img.resize(lay.upload_bytes);
        const double t_alloc = ms(t0);
        t0 = std::chrono::steady_clock::now();
        tiler.fill(hp, nullptr, lay, img.data());
        const double t_fill = ms(t0);
        printf("plan %.2f ms, tiler %.2f ms, image %.2f ms (+ %.2f alloc), %.1f MB; %lld points, %lld prims, %lld tiles, %lld wave tiles, %lld span chunks\n", t_plan, t_tile,
               t_fill, t_alloc, lay.upload_bytes / 1e6, (long long)hp.total_points, (long long)lay.n_prims, (long long)lay.n_tiles, (long long)lay.n_wave,
               (long long)lay.n_span_chunks);
    }
    return 0;
}
