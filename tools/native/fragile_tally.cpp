// tools/native/fragile_tally.cpp -- test infrastructure, CPU only.  The library's setup (fcpp_host.cpp: sine / cosine / atan2 from csrc/fcpp_math.h,
// so that host and device agree bit for bit) against the oracle (oracle/fcpp_oracle.c: the platform libm, as numpy uses) on the INTEGERS a
// count hinges on: the reference takes int((max_y - min_y) / W) + 1 swaths (MLP:739) and max(10, int(len / 0.5)) reverse-fill points
// (MLP:1214) from coordinates that went through a rotation into the frame of layer 1 -- where the quotient is mathematically an integer,
// the last bit of a sine decides.  Random rotated parallelograms and quadrilaterals, and rectangles whose inset height is an EXACT multiple
// of the working width (399 heights), under rotation.
//   build: g++ -O2 -std=c++17 -ffp-contract=off -pthread -o build/fragile_tally tools/native/fragile_tally.cpp field_coverage_path_planning_amd/csrc/fcpp_host.cpp oracle/fcpp_oracle.c ... (tools/fragile_tally.sh)
//   run:   build/fragile_tally <fields> <seed> <threads>  -> one summary line, then one line per mismatching field (vertices as hex floats)
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../field_coverage_path_planning_amd/csrc/fcpp_internal.h"
#include "../../oracle/fcpp_oracle.h"

using namespace fcpp;

struct Case { fcpp_field f; int cls; };

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 100000;
    const uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
    const int threads = argc > 3 ? atoi(argv[3]) : 8;
    const fcpp_vehicle veh = { 3.2, 8.0, 9.0, 15.0, 4.0, 2.0, 1.5, 0.85 };
    fcpp_options opt = { 0, 1, 0.0, 0.5, 1e-6, 0, 0 };
    const orc_vehicle oveh = { 3.2, 8.0, 9.0, 15.0, 4.0, 2.0, 1.5, 0.85 };
    orc_options oopt;
    memset(&oopt, 0, sizeof oopt);
    oopt.clothoid_fit = 1; oopt.clothoid_frac = 0.5; oopt.geofence_tol = 1e-6;

    std::vector<Case> cases((size_t)n);
    {
        std::mt19937_64 rng(seed);
        auto U = [&](double a, double b) { return a + (b - a) * (double)(rng() >> 11) * (1.0 / 9007199254740992.0); };
        for (int64_t i = 0; i < n; ++i) {
            fcpp_field f;
            memset(&f, 0, sizeof f);
            const int cls = (int)(i % 5);        // 0, 1: rotated parallelogram (cfg5's distribution); 2: quadrilateral; 3: exact-multiple height under rotation; 4: a corner of exactly 60 degrees under rotation (MLP:1043: the reverse fill of a corner of 60 degrees and more)
            double qx[4], qy[4];
            if (cls <= 1) {
                const double b = U(100, 1000), h = U(100, 1000), ang = U(M_PI / 3, 2 * M_PI / 3), sx = h / tan(ang);
                const double x[4] = { 0, b, b + sx, sx }, y[4] = { 0, 0, h, h };
                memcpy(qx, x, sizeof x); memcpy(qy, y, sizeof y);
            } else if (cls == 2) {
                const double b = U(100, 900), h = U(100, 700), m = 0.12 * (b < h ? b : h);
                const double x[4] = { U(-m, m), b + U(-m, m), b + U(-m, m), U(-m, m) }, y[4] = { U(-m, m), U(-m, m), h + U(-m, m), h + U(-m, m) };
                memcpy(qx, x, sizeof x); memcpy(qy, y, sizeof y);
            } else if (cls == 4) {
                const double b = U(100, 1000), h = U(100, 1000), sx = h / 1.7320508075688772;          // tan(60 degrees)
                const double x[4] = { 0, b, b + sx, sx }, y[4] = { 0, 0, h, h };
                memcpy(qx, x, sizeof x); memcpy(qy, y, sizeof y);
            } else {
                const int k = 1 + (int)((i / 5) % 399);              // H - 2R = k W exactly in real arithmetic
                const double b = U(100, 1000), h = 2 * veh.min_turn_radius + k * veh.working_width;
                const double x[4] = { 0, b, b, 0 }, y[4] = { 0, 0, h, h };
                memcpy(qx, x, sizeof x); memcpy(qy, y, sizeof y);
            }
            const double rot = U(-M_PI / 4, M_PI / 4), c = cos(rot), s = sin(rot), tx = cls >= 3 ? U(-50, 50) : 0.0, ty = cls >= 3 ? U(-50, 50) : 0.0;
            for (int k = 0; k < 4; ++k) { f.vx[k] = qx[k] * c - qy[k] * s + tx; f.vy[k] = qx[k] * s + qy[k] * c + ty; }
            f.from_vertices = 1;
            if (rng() & 1) { f.has_start = 1; f.start_x = U(0, 400); f.start_y = U(0, 400); }
            cases[(size_t)i] = { f, cls };
        }
    }
    // the library's decisions, in one threaded call (what fcpp_plan_count does)
    std::vector<fcpp_field> fields((size_t)n);
    for (int64_t i = 0; i < n; ++i) fields[(size_t)i] = cases[(size_t)i].f;
    HostPlan hp;
    std::string err;
    if (build_host_plan(veh, opt, n, fields.data(), nullptr, false, hp, err) != FCPP_OK) { fprintf(stderr, "build_host_plan: %s\n", err.c_str()); return 2; }
    // fragile fields by the library's own account: the swath count changes when the field is scaled by 1 +- 2^-40 (a near-integer quotient)
    std::vector<unsigned char> fragile((size_t)n, 0);
    for (int pass = 0; pass < 2; ++pass) {
        const double sc = pass ? 1.0 + 9.094947017729282e-13 : 1.0 - 9.094947017729282e-13;
        std::vector<fcpp_field> g = fields;
        for (auto &f : g) for (int k = 0; k < 4; ++k) { f.vx[k] *= sc; f.vy[k] *= sc; }
        HostPlan hq;
        if (build_host_plan(veh, opt, n, g.data(), nullptr, false, hq, err) != FCPP_OK) return 2;
        for (int64_t i = 0; i < n; ++i)
            if (hq.info[(size_t)i].n_swaths != hp.info[(size_t)i].n_swaths || memcmp(hq.info[(size_t)i].n_reverse, hp.info[(size_t)i].n_reverse, 16)) fragile[(size_t)i] = 1;
    }
    // the oracle, field by field on `threads` threads
    std::atomic<int64_t> next(0), mism(0), refused_both(0), status_diff(0), frag_mism(0);
    int64_t mism_cls[5] = { 0, 0, 0, 0, 0 }, frag_cls[5] = { 0, 0, 0, 0, 0 }, n_cls[5] = { 0, 0, 0, 0, 0 };
    std::mutex mu;
    std::vector<std::string> lines;
    auto work = [&]() {
        for (;;) {
            const int64_t i = next.fetch_add(1);
            if (i >= n) return;
            const fcpp_field &f = fields[(size_t)i];
            orc_field of;
            memset(&of, 0, sizeof of);
            memcpy(of.vx, f.vx, sizeof f.vx); memcpy(of.vy, f.vy, sizeof f.vy);
            of.from_vertices = 1; of.has_start = f.has_start; of.start_x = f.start_x; of.start_y = f.start_y;
            orc_plan p;
            memset(&p, 0, sizeof p);
            const int rc = orc_plan_field(&of, &oveh, &oopt, &p);
            const fcpp_field_info &in = hp.info[(size_t)i];
            bool bad = false;
            if ((rc != 0) != (in.status != 0)) { bad = true; ++status_diff; }
            else if (rc != 0) ++refused_both;
            else bad = p.n_swaths != in.n_swaths || p.n_main != in.n_main || p.n_head != in.n_head || p.n_loops != in.n_loops || p.start_corner != in.start_corner ||
                       p.reverse_order != in.reverse_order || p.start_from_right != in.start_from_right || memcmp(p.n_reverse, in.n_reverse, 16) != 0 || p.shape != in.shape;
            std::lock_guard<std::mutex> lk(mu);
            ++n_cls[cases[(size_t)i].cls];
            if (fragile[(size_t)i]) ++frag_cls[cases[(size_t)i].cls];
            if (bad) {
                ++mism; ++mism_cls[cases[(size_t)i].cls];
                if (fragile[(size_t)i]) ++frag_mism;
                char buf[640];
                snprintf(buf, sizeof buf, "MISMATCH field %lld class %d fragile %d: library status %d swaths %d main %lld head %lld rev %d %d %d %d | oracle rc %d swaths %d main %lld head %lld rev %d %d %d %d | verts %a %a %a %a %a %a %a %a start %d %a %a",
                         (long long)i, cases[(size_t)i].cls, (int)fragile[(size_t)i], in.status, in.n_swaths, (long long)in.n_main, (long long)in.n_head, in.n_reverse[0], in.n_reverse[1], in.n_reverse[2], in.n_reverse[3],
                         rc, rc ? 0 : p.n_swaths, rc ? 0LL : (long long)p.n_main, rc ? 0LL : (long long)p.n_head, p.n_reverse[0], p.n_reverse[1], p.n_reverse[2], p.n_reverse[3],
                         f.vx[0], f.vy[0], f.vx[1], f.vy[1], f.vx[2], f.vy[2], f.vx[3], f.vy[3], f.has_start, f.start_x, f.start_y);
                lines.push_back(buf);
            } else if (fragile[(size_t)i] && lines.size() < 4000 && cases[(size_t)i].cls == 3 && (i / 5) % 37 == 0) {
                char buf[400];
                snprintf(buf, sizeof buf, "FRAGILE-AGREE field %lld swaths %d | verts %a %a %a %a %a %a %a %a", (long long)i, in.n_swaths, f.vx[0], f.vy[0], f.vx[1], f.vy[1], f.vx[2], f.vy[2], f.vx[3], f.vy[3]);
                lines.push_back(buf);
            }
            if (!bad && cases[(size_t)i].cls == 4 && fragile[(size_t)i] && (i / 5) % 4000 == 0) {      // a sample for tests/golden (tools/gen_golden.py: the reference decides)
                char buf[400];
                snprintf(buf, sizeof buf, "CORNER60 field %lld head %lld rev %d %d %d %d | verts %a %a %a %a %a %a %a %a start %d %a %a", (long long)i, (long long)in.n_head,
                         in.n_reverse[0], in.n_reverse[1], in.n_reverse[2], in.n_reverse[3], f.vx[0], f.vy[0], f.vx[1], f.vy[1], f.vx[2], f.vy[2], f.vx[3], f.vy[3], f.has_start, f.start_x, f.start_y);
                lines.push_back(buf);
            }
            if (rc == 0) orc_plan_free(&p);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(work);
    for (auto &t : pool) t.join();
    int64_t nfrag = 0;
    for (unsigned char c : fragile) nfrag += c;
    printf("fields %lld seed %llu: mismatches %lld (status %lld) ; fragile by the library's own count (swaths or reverse-fill points change under a scaling by 1 +- 2^-40) %lld, of them mismatching %lld ; refused by both %lld\n",
           (long long)n, (unsigned long long)seed, (long long)mism.load(), (long long)status_diff.load(), (long long)nfrag, (long long)frag_mism.load(), (long long)refused_both.load());
    const char *names[5] = { "rotated parallelograms", "rotated parallelograms", "quadrilaterals", "exact-multiple heights under rotation", "a corner of exactly 60 degrees" };
    for (int c = 1; c < 5; ++c)
        printf("  %-40s fields %lld fragile %lld mismatches %lld\n", names[c], (long long)(n_cls[c] + (c == 1 ? n_cls[0] : 0)), (long long)(frag_cls[c] + (c == 1 ? frag_cls[0] : 0)), (long long)(mism_cls[c] + (c == 1 ? mism_cls[0] : 0)));
    for (const auto &l : lines) puts(l.c_str());
    return 0;
}
