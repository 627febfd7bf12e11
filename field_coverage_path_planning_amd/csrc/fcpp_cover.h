// fcpp_cover.h -- device-side job record of the coverage rasteriser (fcpp_cover.hip) and its launcher.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace fcpp {

struct DevCoverJob {
    double ox, oy, res, shift, radius;
    int32_t nx, ny, n_a, n_b;
    int64_t pts_first, grid_first;
    int32_t strict, region;
    double outer[12], inner[12];
    int32_t tiles_x, tiles_y;      // 64 x 64-sample tiles
    int64_t tile_first;            // first workgroup of this job
};

int launch_cover(hipStream_t st, int64_t n_jobs, int64_t n_tiles, const DevCoverJob *jobs, const double *px, const double *py,
                 uint8_t *grid, unsigned long long *counts);

}  // namespace fcpp
