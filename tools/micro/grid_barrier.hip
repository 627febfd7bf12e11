// What a generation boundary INSIDE a launch would cost the GA loop (round 4; the review's persistent-kernel proposal): W workgroups of 1024
// threads, one per compute unit, run G "generations"; in each, every workgroup writes a slab (a share of a 2 MB population), arrives at a
// device-scope counter (ONE atomic per workgroup), waits until all W have arrived, and reads a slab another workgroup wrote (on another
// XCD: the release / acquire must reach memory, not just the XCD's L2).  Compared with the same work as G launches of one kernel.
// Every wait is bounded (an abort flag ends the kernel if a wait exceeds ~50 ms): the grid always drains.
// hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip ;  ./grid_barrier [workgroups, default 256] [generations, default 500]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int SLAB_WORDS = 2048;      // 8 KB per workgroup and generation (256 workgroups: the 2 MB population of cfg4)

__device__ __forceinline__ void work(int *pop_w, const int *pop_r, int wg, int n_wg, int g, int *sink)
{
    // read the slab of the workgroup "opposite" (written last generation by another XCD's workgroup), write our own
    const int other = (wg + n_wg / 2 + 1) % n_wg;
    int acc = 0;
    for (int k = threadIdx.x; k < SLAB_WORDS; k += blockDim.x) acc += pop_r[other * SLAB_WORDS + k];
    for (int k = threadIdx.x; k < SLAB_WORDS; k += blockDim.x) pop_w[wg * SLAB_WORDS + k] = acc + g + k;
    if (acc == 0x7fffffff) *sink = acc;
}

__global__ __launch_bounds__(1024) void k_persistent(int *a, int *b, unsigned *counter, int *abort_flag, int G, int *sink, int *wrong)
{
    const int wg = blockIdx.x, n_wg = gridDim.x;
    __shared__ int s_abort;
    for (int g = 0; g < G; ++g) {
        int *w = (g & 1) ? a : b;
        const int *r = (g & 1) ? b : a;
        work(w, r, wg, n_wg, g, sink);
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();                                                        // release: the slab reaches memory
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(g + 1) * (unsigned)n_wg;
            int ab = 0;
            long spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1l << 21) || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ab = 1; break; }
            }
            if (ab) __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence();                                                        // acquire
            s_abort = ab;
        }
        __syncthreads();
        if (s_abort) { if (threadIdx.x == 0) *wrong = 1; return; }
    }
}

__global__ __launch_bounds__(1024) void k_generation(int *w, const int *r, int g, int *sink) { work(w, r, blockIdx.x, gridDim.x, g, sink); }

int main(int argc, char **argv)
{
    const int W = argc > 1 ? atoi(argv[1]) : 256, G = argc > 2 ? atoi(argv[2]) : 500;
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    if (W > prop.multiProcessorCount) { printf("at most one workgroup per compute unit (%d)\n", prop.multiProcessorCount); return 1; }
    int *a, *b, *abort_flag, *sink, *wrong;
    unsigned *counter;
    CHK(hipMalloc(&a, (size_t)W * SLAB_WORDS * 4)); CHK(hipMalloc(&b, (size_t)W * SLAB_WORDS * 4));
    CHK(hipMalloc(&counter, 4)); CHK(hipMalloc(&abort_flag, 4)); CHK(hipMalloc(&sink, 4)); CHK(hipMalloc(&wrong, 4));
    CHK(hipMemset(a, 0, (size_t)W * SLAB_WORDS * 4)); CHK(hipMemset(b, 0, (size_t)W * SLAB_WORDS * 4));
    hipStream_t st;
    CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int rep = 0; rep < 5; ++rep) {
        float ms_p = 0, ms_l = 0;
        CHK(hipMemsetAsync(counter, 0, 4, st)); CHK(hipMemsetAsync(abort_flag, 0, 4, st)); CHK(hipMemsetAsync(wrong, 0, 4, st));
        CHK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k_persistent, dim3(W), dim3(1024), 0, st, a, b, counter, abort_flag, G, sink, wrong);
        CHK(hipEventRecord(e1, st));
        CHK(hipStreamSynchronize(st));
        CHK(hipEventElapsedTime(&ms_p, e0, e1));
        int hw = 0;
        CHK(hipMemcpy(&hw, wrong, 4, hipMemcpyDeviceToHost));
        CHK(hipEventRecord(e0, st));
        for (int g = 0; g < G; ++g) hipLaunchKernelGGL(k_generation, dim3(W), dim3(1024), 0, st, (g & 1) ? a : b, (g & 1) ? b : a, g, sink);
        CHK(hipEventRecord(e1, st));
        CHK(hipStreamSynchronize(st));
        CHK(hipEventElapsedTime(&ms_l, e0, e1));
        printf("W %d G %d: persistent %.2f us per generation%s | one launch per generation %.2f us\n", W, G, ms_p * 1e3 / G, hw ? " (ABORTED: a wait ran out)" : "",
               ms_l * 1e3 / G);
    }
    return 0;
}
