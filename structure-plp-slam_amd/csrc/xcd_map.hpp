// Workgroups of a launch are dealt round-robin to the 8 XCDs of an MI355X, each with its own 4 MB L2.  For a grid
// (tiles, frames) that means the tiles of ONE frame land on eight different L2s and every halo row / patch row / staged
// array is fetched once per XCD.  xcd_frame_major() renumbers the workgroups so that XCD k processes whole frames
// (frames k*F/8 .. (k+1)*F/8 - 1 in order): a frame's working set then lives in one L2.  Placement only; results do not
// depend on it.  Falls back to the identity when the grid size is not a multiple of 8.
#pragma once
#include <hip/hip_runtime.h>

namespace plp {

// grid = (gx, gy): x = tile index inside a frame, y = frame.  Returns the logical (tile, frame) of this workgroup.
__device__ __forceinline__ void xcd_frame_major(unsigned& tile, unsigned& frame) {
    const unsigned gx = gridDim.x, total = gridDim.x * gridDim.y;
    tile = blockIdx.x; frame = blockIdx.y;
    if ((total & 7u) == 0u) {
        const unsigned lin = blockIdx.x + gx * blockIdx.y;
        const unsigned logical = (lin & 7u) * (total >> 3) + (lin >> 3);
        frame = logical / gx; tile = logical - frame * gx;
    }
}

}  // namespace plp
