// The 16-bit grid of the persistent pipeline's walk nodes (rt_types.h GpuNode4Q / NodeGrid): how the grid is laid over the scene and how a
// box bound becomes a cell.  Shared by the host (rtamd_api.hip), the fold of the walk trees (rtamd_build.hip widen_nodes) and the test hooks.
#pragma once
#include <math.h>
#include <hip/hip_runtime.h>
#include "rt_types.h"

namespace rtamd {

// `lo` / `hi` bound everything the grid must hold: the root boxes of the trees and the camera.
inline NodeGrid make_node_grid(const float lo[3], const float hi[3]) {
    NodeGrid G;
    double ext[3], largest = 0.0;
    for (int k = 0; k < 3; k++) { ext[k] = (double)hi[k] - (double)lo[k]; if (!(ext[k] >= 0.0)) ext[k] = 0.0; if (ext[k] > largest) largest = ext[k]; }
    if (!(largest > 0.0)) largest = 1.0;
    for (int k = 0; k < 3; k++) {
        // a thin axis keeps cells of 1/64 of the largest axis' cells: ray origins lie as far from the grid's corner along it as the
        // other axes' scale suggests, and a cell must stay well above the rounding of those coordinates
        const double e = ext[k] > largest / 64.0 ? ext[k] : largest / 64.0;
        G.step[k] = (float)(e / RT_GRID_CELLS);
        G.lo[k] = (float)((double)lo[k] - RT_GRID_BORDER * (double)G.step[k]);
        G.istep[k] = 1.0f / G.step[k];
    }
    return G;
}

// One axis of a child box as lo | hi << 16: each bound goes to the cell below / above it and one further — the margin that covers the
// rounding of the walkers' grid-space ray (rt_device.h make_ray_grid: at most 0.02 cells for an origin inside the grid).
// `fits` is cleared when the box does not lie on the grid (the word then spans the whole axis).
__host__ __device__ inline uint32_t grid_axis_word(float lo, float hi, float grid_lo, float step, bool &fits) {
    const double a = floor(((double)lo - (double)grid_lo) / (double)step) - 1.0;
    const double b = ceil(((double)hi - (double)grid_lo) / (double)step) + 1.0;
    if (!(a >= 0.0 && b <= 65535.0 && a <= b)) { fits = false; return 0xFFFF0000u; }
    return (uint32_t)a | ((uint32_t)b << 16);
}

} // namespace rtamd
