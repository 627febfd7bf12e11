// Layout microbenchmark (round 3; the review's question): the same 36 bytes per point written by the same wavefronts -- a 512-point
// chunk per wavefront, 16 bytes per lane and store, four wavefronts per workgroup, LDS-padded to four resident wavefronts per SIMD, as
// k_plan_quiet does -- into
//   soa P   : five arrays (4 x 8 bytes + 1 x 4 bytes per point), the arrays P bytes + their own size apart  (P = 0: back to back)
//   records : ONE stream of 18 KiB records, a record = the chunk's x | y | kappa | v | flags pieces (4 + 4 + 4 + 4 + 2 KiB)
// hipcc --offload-arch=gfx950 -O3 -o record_probe record_probe.hip ;  ./record_probe [points, default 240e6]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// piece k of chunk c starts at base + off_k + (c / split) * stride_k + (c % split) * part (bytes): split > 1 deals consecutive chunks
// round-robin to `split` parts of every array, `part` bytes apart
struct Layout { size_t off[5], stride[5]; unsigned split; size_t part; };

__global__ __launch_bounds__(256) void k_write(char *base, Layout L, size_t n_chunks)
{
    extern __shared__ char pad[];
    const size_t chunk = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (chunk >= n_chunks) return;
    const int lane = threadIdx.x & 63;
    const double v0 = (double)(chunk * 512 + 2 * lane);
    const size_t cq = chunk / L.split, cpart = (chunk % L.split) * L.part;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double a = v0 + 128.0 * r;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            *reinterpret_cast<double2 *>(base + L.off[k] + cpart + cq * L.stride[k] + (size_t)r * 1024 + (size_t)lane * 16) = make_double2(a, a + 1.0);
        *reinterpret_cast<uint2 *>(base + L.off[4] + cpart + cq * L.stride[4] + (size_t)r * 512 + (size_t)lane * 8) = make_uint2((unsigned)lane, 7u);
    }
}

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? (size_t)atof(argv[1]) : (size_t)240e6, n_chunks = (n + 511) / 512;
    const size_t S8 = n_chunks * 4096, S4 = n_chunks * 2048;
    size_t free_b = 0, total_b = 0;
    CHK(hipMemGetInfo(&free_b, &total_b));
    const size_t GiB = (size_t)1 << 30;
    const size_t pitches[] = { 0, 4 * GiB, 12 * GiB, 24 * GiB };
    const size_t slab = 5 * (24 * GiB + S8) + GiB;
    if (slab > free_b) { printf("not enough free memory (%zu GiB)\n", free_b >> 30); return 1; }
    char *base = nullptr;
    CHK(hipMalloc(&base, slab));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((n_chunks + 3) / 4);
    const int lds = 34 * 1024;                  // four resident wavefronts per SIMD, as the span kernel
    auto run = [&](const char *name, const Layout &L) {
        // (host-side bound check of every piece the kernel will write: the last chunk of every part of every array)
        for (int k = 0; k < 5; ++k) {
            const size_t piece = k < 4 ? 4096 : 2048, cq_max = (n_chunks - 1) / L.split;
            const size_t last = L.off[k] + (size_t)(L.split - 1) * L.part + cq_max * L.stride[k] + piece;
            if (last > slab) { printf("%-22s skipped: would write beyond the slab\n", name); return; }
        }
        std::vector<float> ms;
        for (int it = 0; it < 12; ++it) {
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), lds, 0, base, L, n_chunks);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float t; CHK(hipEventElapsedTime(&t, e0, e1));
            if (it >= 2) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        const double bytes = (double)n_chunks * 18432.0;
        printf("%-22s min %8.3f ms  median %8.3f ms   %6.2f TB/s (median)\n", name, ms.front(), ms[ms.size() / 2], bytes / (ms[ms.size() / 2] * 1e-3) / 1e12);
        fflush(stdout);
    };
    printf("%zu points, %zu chunks, %.2f GB per launch\n", n, n_chunks, (double)n_chunks * 18432.0 / 1e9);
    for (size_t P : pitches) {
        Layout L;
        L.split = 1; L.part = 0;
        for (int k = 0; k < 5; ++k) { L.off[k] = (size_t)k * (P + S8); L.stride[k] = k < 4 ? 4096 : 2048; }
        char name[64];
        snprintf(name, sizeof name, "soa, pitch %zu GiB", P >> 30);
        run(name, L);
    }
    // the arrays 24 GiB apart AND each dealt to 2 / 4 parts 12 / 6 GiB apart: ten / twenty regions written at any one time
    for (unsigned sp : { 2u, 4u }) {
        Layout L;
        L.split = sp; L.part = 24 * GiB / sp;
        if (S8 / sp + GiB > L.part) continue;
        for (int k = 0; k < 5; ++k) { L.off[k] = (size_t)k * (24 * GiB + S8); L.stride[k] = k < 4 ? 4096 : 2048; }
        char name[64];
        snprintf(name, sizeof name, "soa 24 GiB, %u parts", sp);
        run(name, L);
    }
    {
        Layout L;
        L.split = 1; L.part = 0;
        for (int k = 0; k < 5; ++k) { L.off[k] = (size_t)k * 4096; L.stride[k] = 18432; }
        run("records (18 KiB)", L);
        // the same records spread over the slab the widest soa layout used (a record every ~5x farther): does the span matter?
        for (int k = 0; k < 5; ++k) L.stride[k] = 18432 * 4;
        if ((n_chunks - 1) * L.stride[0] + 18432 <= slab) run("records, stride 72 KiB", L);
    }
    CHK(hipFree(base));
    return 0;
}
