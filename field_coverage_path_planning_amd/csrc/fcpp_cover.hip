// fcpp_cover.hip -- coverage rasterisation (include/fcpp.h: fcpp_cover_grid; MLP:1357-1371, 1426-1509).
//
// One workgroup per 64 x 64 tile of sample points: thread t owns column t & 63 and the rows (t >> 6) + 4q, q = 0..15.
// The polyline's segments are culled against the tile (bounding boxes, 256 segments per pass, one per thread), the
// survivors are compacted into LDS and every thread tests its still-open samples against them.  A tile stops as soon as
// all its samples are covered (or none is in the region: the interior of a field, for the headland ring).  No atomics on
// the grid; the three counts per job are integer atomics (order-independent).  fp64 VALU-bound: ~15 flops per test.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fcpp_cover.h"

namespace fcpp {

static constexpr int CT = 64;          // tile edge in samples
static constexpr int CQ = CT * CT / 256;   // samples per thread

__device__ __forceinline__ bool covers(double ax, double ay, double bx, double by, double X, double Y, double r2, bool strict)
{
    const double abx = bx - ax, aby = by - ay, apx = X - ax, apy = Y - ay;
    const double len2 = abx * abx + aby * aby, dot = apx * abx + apy * aby;
    double lhs, rhs = r2;
    if (dot <= 0.0) lhs = apx * apx + apy * apy;
    else if (dot >= len2) { const double bpx = X - bx, bpy = Y - by; lhs = bpx * bpx + bpy * bpy; }
    else { const double cr = abx * apy - aby * apx; lhs = cr * cr; rhs = r2 * len2; }
    return strict ? (lhs < rhs) : (lhs <= rhs);
}

__global__ __launch_bounds__(256) void k_cover(int64_t n_jobs, const DevCoverJob *__restrict__ jobs, const double *__restrict__ px,
                                               const double *__restrict__ py, uint8_t *__restrict__ grid,
                                               unsigned long long *__restrict__ counts)
{
    __shared__ double sax[256], say[256], sbx[256], sby[256];
    __shared__ int s_wave[4];
    __shared__ unsigned s_red[4][3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // which job: last one whose first tile is <= this block
    int lo = 0, hi = (int)n_jobs - 1;
    const int64_t blk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].tile_first <= blk) lo = mid; else hi = mid - 1;
    }
    const DevCoverJob &J = jobs[lo];
    const int tile = (int)(blk - J.tile_first), tx = tile % J.tiles_x, ty = tile / J.tiles_x;
    const int nx = J.nx, ny = J.ny;
    const double ox = J.ox, oy = J.oy, res = J.res, shift = J.shift, r = J.radius, r2 = r * r;
    const bool strict = J.strict != 0;
    const int i = tx * CT + lane, j0 = ty * CT + wave;
    const double X = ox + ((double)i + shift) * res;
    // region mask of this thread's samples
    unsigned open = 0;
    for (int q = 0; q < CQ; ++q) {
        const int j = j0 + 4 * q;
        bool in = i < nx && j < ny;
        if (in && J.region) {
            const double Y = oy + ((double)j + shift) * res;
            bool io = true, ii = true;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                io = io & (J.outer[3 * e] * X + J.outer[3 * e + 1] * Y + J.outer[3 * e + 2] >= 0.0);
                ii = ii & (J.inner[3 * e] * X + J.inner[3 * e + 1] * Y + J.inner[3 * e + 2] >= 0.0);
            }
            in = io && !ii;
        }
        open |= in ? (1u << q) : 0u;
    }
    const unsigned region_mask = open;
    // the tile's samples span [tx0, tx1] x [ty0, ty1]; a segment farther than r from that box cannot cover any of them
    const int i1 = min(tx * CT + CT - 1, nx - 1), j1 = min(ty * CT + CT - 1, ny - 1);
    const double bx0 = ox + ((double)(tx * CT) + shift) * res, bx1 = ox + ((double)i1 + shift) * res;
    const double by0 = oy + ((double)(ty * CT) + shift) * res, by1 = oy + ((double)j1 + shift) * res;
    const double reach = r * (1.0 + 1e-9) + 1e-9;     // conservative: culling must never drop a covering segment
    unsigned cov[2] = { 0u, 0u };
    for (int pass = 0; pass < 2; ++pass) {
        const int npts = pass == 0 ? J.n_a : J.n_b;
        const int64_t first = J.pts_first + (pass == 0 ? 0 : J.n_a);
        if (pass == 1) open = region_mask & ~cov[0];
        for (int base = 0; base < npts - 1; base += 256) {
            if (__syncthreads_or(open != 0u) == 0) break;      // nothing left to cover in this tile (also guards the LDS reuse)
            const int s = base + tid;
            bool keep = false;
            double ax = 0, ay = 0, bx = 0, by = 0;
            if (s < npts - 1) {
                ax = px[first + s]; ay = py[first + s]; bx = px[first + s + 1]; by = py[first + s + 1];
                keep = !(fmin(ax, bx) - bx1 > reach || bx0 - fmax(ax, bx) > reach || fmin(ay, by) - by1 > reach || by0 - fmax(ay, by) > reach);
            }
            const unsigned long long m = __ballot(keep);
            if (lane == 0) s_wave[wave] = __popcll(m);
            __syncthreads();
            int off = 0, total = 0;
            for (int w = 0; w < 4; ++w) { if (w < wave) off += s_wave[w]; total += s_wave[w]; }
            if (keep) {
                const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
                sax[pos] = ax; say[pos] = ay; sbx[pos] = bx; sby[pos] = by;
            }
            __syncthreads();
            if (open) {
                for (int k = 0; k < total; ++k) {
                    const double cax = sax[k], cay = say[k], cbx = sbx[k], cby = sby[k];
#pragma unroll
                    for (int q = 0; q < CQ; ++q) {
                        if ((open >> q) & 1u) {
                            const double Y = oy + ((double)(j0 + 4 * q) + shift) * res;
                            if (covers(cax, cay, cbx, cby, X, Y, r2, strict)) { open &= ~(1u << q); cov[pass] |= 1u << q; }
                        }
                    }
                    if (!open) break;
                }
            }
        }
    }
    if (J.grid_first >= 0 && i < nx) {
        for (int q = 0; q < CQ; ++q) {
            const int j = j0 + 4 * q;
            if (j < ny) grid[J.grid_first + (int64_t)j * nx + i] = (uint8_t)(((cov[0] >> q) & 1u) | (((cov[1] >> q) & 1u) << 1));
        }
    }
    // counts: region samples, covered by A, covered by A or B
    unsigned c3[3] = { (unsigned)__popc(region_mask), (unsigned)__popc(cov[0]), (unsigned)__popc(cov[0] | cov[1]) };
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        unsigned v = c3[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) s_red[wave][k] = v;
    }
    __syncthreads();
    if (tid < 3) {
        const unsigned v = s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid];
        if (v) atomicAdd(&counts[3 * (int64_t)lo + tid], (unsigned long long)v);
    }
}

int launch_cover(hipStream_t st, int64_t n_jobs, int64_t n_tiles, const DevCoverJob *jobs, const double *px, const double *py,
                 uint8_t *grid, unsigned long long *counts)
{
    if (n_jobs <= 0 || n_tiles <= 0) return 0;
    hipLaunchKernelGGL(k_cover, dim3((unsigned)n_tiles), dim3(256), 0, st, n_jobs, jobs, px, py, grid, counts);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

}  // namespace fcpp
