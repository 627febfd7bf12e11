// Tuning microbenchmark: write streams advancing in lockstep (what k_plan_quiet does with its five output arrays).
//   A: 5 x 8-byte streams, tiles of 512 elements, every store 1 KiB-aligned
//   B: 4 x 8-byte + 1 x 4-byte stream (the real mix), aligned tiles
//   C: B with tiles of 510 elements (tile boundaries fall anywhere in a cache line; lanes beyond the count are masked)
//   S: A's bytes, one stream after the other
// hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int K8, int K4, int TILE>
__global__ __launch_bounds__(256) void k_streams(double *base, size_t dist, size_t n_tiles)
{
    const size_t tile = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    const int lane = threadIdx.x & 63;
    const size_t g0 = tile * TILE;
    const int odd = (int)(g0 & 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = 2 * (lane + 64 * k) - odd;
        const bool has0 = j >= 0 && j < TILE, has1 = j + 1 < TILE;
        const size_t idx = g0 + j;
        const double v = (double)idx;
        if (has0 && has1) {
#pragma unroll
            for (int s = 0; s < K8; ++s) *reinterpret_cast<double2 *>(base + s * dist + idx) = make_double2(v, v + 1.0);
#pragma unroll
            for (int s = 0; s < K4; ++s)
                *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned *>(base + (K8 + s) * dist) + idx) = make_uint2((unsigned)idx, 7u);
        } else if (has0 || has1) {
            const size_t i1 = has0 ? idx : idx + 1;
#pragma unroll
            for (int s = 0; s < K8; ++s) base[s * dist + i1] = v;
#pragma unroll
            for (int s = 0; s < K4; ++s) reinterpret_cast<unsigned *>(base + (K8 + s) * dist)[i1] = 7u;
        }
    }
}

// W: consecutive tiles per wavefront (fewer, longer waves)
template <int W>
__global__ __launch_bounds__(256) void k_multi(double *base, size_t dist, size_t n_tiles)
{
    const size_t t0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * W;
    const int lane = threadIdx.x & 63;
    for (int w = 0; w < W; ++w) {
        const size_t tile = t0 + w;
        if (tile >= n_tiles) return;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t idx = tile * 512 + 2 * (lane + 64 * k);
            const double v = (double)idx;
#pragma unroll
            for (int s = 0; s < 4; ++s) *reinterpret_cast<double2 *>(base + s * dist + idx) = make_double2(v, v + 1.0);
            *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned *>(base + 4 * dist) + idx) = make_uint2((unsigned)idx, 7u);
        }
    }
}

template <int W>
float run_multi(double *base, size_t dist, size_t n, int reps)
{
    const size_t n_tiles = n / 512, waves = (n_tiles + W - 1) / W;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_multi<W>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, 0, base, dist, n_tiles);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_multi<W>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, 0, base, dist, n_tiles);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

// runs: the index space is cut into alternating runs of len_a and len_b elements (a swath line and a U-turn); every run is cut into
// chunks on 512-element boundaries, one chunk per wave -- the first and the last chunk of a run are partial (masked lanes)
__global__ __launch_bounds__(256) void k_runs(double *base, size_t dist, const long long *chunk_start, const int *chunk_count, size_t n_chunks)
{
    const size_t c = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    const int lane = threadIdx.x & 63;
    const size_t g0 = (size_t)chunk_start[c];
    const int cnt = chunk_count[c], odd = (int)(g0 & 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = 2 * (lane + 64 * k) - odd;
        const bool has0 = j >= 0 && j < cnt, has1 = j + 1 < cnt;
        const size_t idx = g0 + j;
        const double v = (double)idx;
        if (has0 && has1) {
#pragma unroll
            for (int s = 0; s < 4; ++s) *reinterpret_cast<double2 *>(base + s * dist + idx) = make_double2(v, v + 1.0);
            *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned *>(base + 4 * dist) + idx) = make_uint2((unsigned)idx, 7u);
        } else if (has0 || has1) {
            const size_t i1 = has0 ? idx : idx + 1;
#pragma unroll
            for (int s = 0; s < 4; ++s) base[s * dist + i1] = v;
            reinterpret_cast<unsigned *>(base + 4 * dist)[i1] = 7u;
        }
    }
}

float run_runs(double *base, size_t dist, size_t n, size_t len_a, size_t len_b, int reps)
{
    std::vector<long long> cs;
    std::vector<int> cc;
    size_t pos = 0;
    for (int r = 0; pos < n; ++r) {
        const size_t len = std::min(n - pos, (r & 1) ? len_b : len_a);
        for (size_t done = 0; done < len;) {
            const size_t g = pos + done, c = std::min(len - done, 512 - (g % 512));
            cs.push_back((long long)g); cc.push_back((int)c);
            done += c;
        }
        pos += len;
    }
    long long *dcs; int *dcc;
    CHK(hipMalloc(&dcs, cs.size() * 8)); CHK(hipMalloc(&dcc, cc.size() * 4));
    CHK(hipMemcpy(dcs, cs.data(), cs.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(dcc, cc.data(), cc.size() * 4, hipMemcpyHostToDevice));
    const size_t nc = cs.size();
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_runs, dim3((unsigned)((nc + 3) / 4)), dim3(256), 0, 0, base, dist, dcs, dcc, nc);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_runs, dim3((unsigned)((nc + 3) / 4)), dim3(256), 0, 0, base, dist, dcs, dcc, nc);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    CHK(hipFree(dcs)); CHK(hipFree(dcc));
    return ms / reps;
}

template <int K8, int K4, int TILE>
float run(double *base, size_t dist, size_t n, int reps)
{
    const size_t n_tiles = n / TILE;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_streams<K8, K4, TILE>), dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, 0, base, dist, n_tiles);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((k_streams<K8, K4, TILE>), dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, 0, base, dist, n_tiles);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
    return ms / reps;
}

int main(int argc, char **argv)
{
    const size_t GiB = 1ull << 30;
    const size_t slab_gib = argc > 1 ? atoi(argv[1]) : 72;
    const size_t n = 1013583353ull / 512 * 512;
    char *slab;
    CHK(hipMalloc(&slab, slab_gib * GiB));
    printf("slab %zu GiB at %p\n", slab_gib, (void *)slab);
    if (argc > 2 && argv[2][0] == 'm') {   // map: a 8 GiB window (5 streams of 1.5 GiB, 1.6 GiB apart) slides over the slab
        const size_t nm = (size_t)(1.5 * GiB) / 8 / 512 * 512, dm = (size_t)(1.6 * GiB) / 4096 * 4096 / 8;
        for (int rnd = 0; rnd < 2; ++rnd) {
            printf("GB/s per 8 GiB window (B: 4x8+1x4 bytes aligned | S: one stream):");
            for (size_t b = 0; b + 8 <= slab_gib; b += 8) {
                double *base = reinterpret_cast<double *>(slab + b * GiB);
                const float tB = run<4, 1, 512>(base, dm, nm, 3);
                const float tS = run<1, 0, 512>(base, 0, nm * 4, 3);
                printf(" %.0f|%.0f", 36.0 * nm / 1e9 / tB * 1e3, 8.0 * nm * 4 / 1e9 / tS * 1e3);
            }
            printf("\n");
        }
        return 0;
    }
    const size_t dist = (size_t)(7.6 * GiB) / 4096 * 4096 / 8;   // elements
    if (argc > 2 && argv[2][0] == 'r') {   // runs with partial chunks at both ends vs one run (all chunks full)
        for (int rnd = 0; rnd < 2; ++rnd)
            for (size_t base_gib : { 0, 33 }) {
                double *base = reinterpret_cast<double *>(slab + base_gib * GiB);
                printf("base %2zu GiB | one run %.3f ms | runs 5500+398 %.3f ms | runs 1500+398 %.3f ms | runs 20000+398 %.3f ms\n", base_gib,
                       run_runs(base, dist, n, n, n, 5), run_runs(base, dist, n, 5500, 398, 5), run_runs(base, dist, n, 1500, 398, 5),
                       run_runs(base, dist, n, 20000, 398, 5));
            }
        return 0;
    }
    if (argc > 2 && argv[2][0] == 'w') {   // tiles per wave
        for (int rnd = 0; rnd < 2; ++rnd)
            for (size_t base_gib : { 0, 33 }) {
                double *base = reinterpret_cast<double *>(slab + base_gib * GiB);
                const double gB = 36.0 * n / 1e9;
                printf("base %2zu GiB | tiles per wave 1: %.3f ms  2: %.3f  4: %.3f  8: %.3f  (%.1f GB)\n", base_gib, run_multi<1>(base, dist, n, 5),
                       run_multi<2>(base, dist, n, 5), run_multi<4>(base, dist, n, 5), run_multi<8>(base, dist, n, 5), gB);
            }
        return 0;
    }
    for (int rnd = 0; rnd < 2; ++rnd)
        for (size_t base_gib : { 0, 8, 16, 24, 33 }) {
            double *base = reinterpret_cast<double *>(slab + base_gib * GiB);
            const float tA = run<5, 0, 512>(base, dist, n, 5);
            const float tB = run<4, 1, 512>(base, dist, n, 5);
            const float tC = run<4, 1, 510>(base, dist, n, 5);
            float tS = 0;
            for (int s = 0; s < 5; ++s) tS += run<1, 0, 512>(base + s * dist, 0, n, 5);
            const double gA = 5.0 * n * 8 / 1e9, gB = 36.0 * n / 1e9;
            printf("base %2zu GiB | A %.3f ms %.0f GB/s | B %.3f ms %.0f GB/s | C %.3f ms %.0f GB/s | S %.3f ms %.0f GB/s\n", base_gib, tA,
                   gA / tA * 1e3, tB, gB / tB * 1e3, tC, gB / tC * 1e3, tS, gA / tS * 1e3);
        }
    return 0;
}
