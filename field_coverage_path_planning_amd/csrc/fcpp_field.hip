// fcpp_field.hip -- pipeline B at sparse sampling: ONE WORKGROUP PLANS ONE FIELD, in one launch.
//
// At the reference's sampling a whole plan is a few thousand points: a closed-form span of layer 1 (every complete pass: swath line +
// U-turn, fcpp_quiet_fn.h) in a handful of 512-point chunks, and the rest (last line, seam, headland loops with their corner turns
// and reverse fills) in a handful of wave tiles (fcpp_sparse_fn.h).  As separate launches (k_quiet_run_stats, k_plan_quiet,
// k_plan_sparse, k_reduce_stats) such a batch is bound by launch boundaries, by the memory-bound and the ALU-bound kernel running
// one after the other, and by every wave tile reducing its own statistics.  Here the eight wavefronts of a workgroup share the
// field's units -- wave tiles first (ALU-bound), then the span's chunks (HBM-bound), so both kinds are in flight on a compute
// unit at once -- keep their statistics per lane, and meet ONCE: the per-lane values go through LDS, are added in wave order and
// reduced over the lanes in a fixed order (run-to-run and shard-to-shard identical sums), the span's closed-form statistics are
// added, and the field's fcpp_field_stats record is written.  No per-tile partials, no reduction launch.
// A field qualifies ("simple", decided by the host tiler) if its tiling consists of one span and wave tiles only; all other fields
// of a batch take the general launches.
#include "fcpp_quiet_fn.h"
#include "fcpp_sparse_fn.h"

namespace fcpp {

static constexpr int FW_WAVES = 8;

struct FieldShared {
    double acc[9][FW_WAVES][64];            // per-lane statistics of every wavefront (36 KB)
    double tot[9];
    unsigned long long cnt[4];              // n_viol, n_outside, n_in_obstacle, n_adjusted
};

__global__ __launch_bounds__(64 * FW_WAVES) void k_plan_field(const DevFieldWork *__restrict__ work, const DevTile *__restrict__ chunks,
                                                              const int32_t *__restrict__ wave_ids, const DevTile *__restrict__ tiles,
                                                              const DevField *__restrict__ fields, const DevPrim *__restrict__ prims,
                                                              const DevConst *__restrict__ cstp, DevObstacles obs, double *__restrict__ xo,
                                                              double *__restrict__ yo, double *__restrict__ ko, double *__restrict__ vo,
                                                              uint32_t *__restrict__ fso, fcpp_field_stats *__restrict__ stats)
{
    extern __shared__ double obs_lds[];     // batches with obstacles: FW_WAVES x 2 * OBS_LDS_VERTS doubles (else none, never touched)
    __shared__ FieldShared S;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const DevFieldWork wk = work[blockIdx.x];
    const DevField &f = fields[wk.field];
    const DevConst &cst = *cstp;            // (a device copy: its members are fetched where they are used, not held in registers)
    if (threadIdx.x < 4) S.cnt[threadIdx.x] = 0ull;
    __syncthreads();
    double *my_lds = obs_lds + wave * (2 * OBS_LDS_VERTS);
#pragma unroll
    for (int q = 0; q < 9; ++q) S.acc[q][wave][lane] = 0.0;
    int c_viol = 0, c_out = 0, c_obs = 0, c_adj = 0;
    const int n_units = wk.n_wave + wk.n_chunks;
    for (int u = wave; u < n_units; u += FW_WAVES) {        // (wave-uniform)
        if (u < wk.n_wave) {
            const DevTile tl = tiles[wave_ids[wk.wave_first + u]];
            SparseAcc acc;
            acc.clear();
            sparse_tile(tl, f, prims, cst, obs, my_lds, xo, yo, ko, vo, fso, acc);
            // the lane's running statistics live in LDS between tiles
            S.acc[0][wave][lane] += acc.s_len[0]; S.acc[1][wave][lane] += acc.s_tpre[0]; S.acc[2][wave][lane] += acc.s_t[0];
            S.acc[3][wave][lane] += acc.s_len[1]; S.acc[4][wave][lane] += acc.s_tpre[1]; S.acc[5][wave][lane] += acc.s_t[1];
            S.acc[6][wave][lane] = fmax(S.acc[6][wave][lane], acc.mk); S.acc[7][wave][lane] = fmax(S.acc[7][wave][lane], acc.ma);
            S.acc[8][wave][lane] = fmax(S.acc[8][wave][lane], acc.mj);
            c_viol += acc.c_viol; c_out += acc.c_out; c_obs += acc.c_obs; c_adj += acc.c_adj;
        } else {
            const DevTile tl = chunks[wk.chunk_first + (u - wk.n_wave)];
            quiet_tile<16>(tl, &f, prims, cst, obs, my_lds, xo, yo, ko, vo, fso, &S.cnt[1], &S.cnt[2]);
        }
    }
    if (lane == 0) {        // integer counts: the order of the additions does not matter
        if (c_viol) atomicAdd(&S.cnt[0], (unsigned long long)c_viol);
        if (c_out) atomicAdd(&S.cnt[1], (unsigned long long)c_out);
        if (c_obs) atomicAdd(&S.cnt[2], (unsigned long long)c_obs);
        if (c_adj) atomicAdd(&S.cnt[3], (unsigned long long)c_adj);
    }
    __syncthreads();
    // quantity q: per lane over the wavefronts in wave order, then over the lanes (fixed order)
    for (int q = wave; q < 9; q += FW_WAVES) {
        double v = S.acc[q][0][lane];
        if (q < 6) {
#pragma unroll
            for (int w2 = 1; w2 < FW_WAVES; ++w2) v += S.acc[q][w2][lane];
            v = wave_sum_to63(v);
        } else {
#pragma unroll
            for (int w2 = 1; w2 < FW_WAVES; ++w2) v = fmax(v, S.acc[q][w2][lane]);
            v = wave_max0_to63(v);
        }
        if (lane == 63) S.tot[q] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        TilePartial rp;
        rp.main_len = rp.main_time_pre = rp.main_time = rp.head_len = rp.head_time_pre = rp.head_time = 0.0;
        rp.max_kappa = rp.max_alat = rp.max_jump = 0.0;
        if (wk.run_tile >= 0) {
            const DevRun run = { wk.run_tile, 0, wk.run_count };
            rp = quiet_run_partial(run, tiles[wk.run_tile], fields, prims, cst);
        }
        fcpp_field_stats s;
        s.main_len_m = rp.main_len + S.tot[0]; s.main_time_pre_s = rp.main_time_pre + S.tot[1]; s.main_time_s = rp.main_time + S.tot[2];
        s.head_len_m = rp.head_len + S.tot[3]; s.head_time_pre_s = rp.head_time_pre + S.tot[4]; s.head_time_s = rp.head_time + S.tot[5];
        s.max_kappa = fmax(rp.max_kappa, S.tot[6]); s.max_alat = fmax(rp.max_alat, S.tot[7]); s.max_jump = fmax(rp.max_jump, S.tot[8]);
        s.n_viol = (int64_t)S.cnt[0]; s.n_outside = (int64_t)S.cnt[1]; s.n_in_obstacle = (int64_t)S.cnt[2]; s.n_adjusted = (int64_t)S.cnt[3];
        stats[wk.field] = s;
    }
}

int launch_plan_field(hipStream_t st, int64_t n_work, const DevFieldWork *work, const DevTile *chunks, const int32_t *wave_ids,
                      const DevTile *tiles, const DevField *fields, const DevPrim *prims, const DevConst *cst_dev, const DevObstacles &obs,
                      bool any_obstacles, double *x, double *y, double *kappa, double *v, uint32_t *fs, fcpp_field_stats *stats)
{
    if (n_work <= 0) return 0;
    const dim3 grid((unsigned)n_work), block(64 * FW_WAVES);
    const unsigned lds = any_obstacles ? (unsigned)(FW_WAVES * 2 * OBS_LDS_VERTS * sizeof(double)) : 0u;
    FCPP_LAUNCH(k_plan_field, grid, block, lds, st, work, chunks, wave_ids, tiles, fields, prims, cst_dev, obs, x, y, kappa, v, fs, stats);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

}  // namespace fcpp
