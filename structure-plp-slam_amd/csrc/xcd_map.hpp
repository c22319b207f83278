// Workgroups of a launch are dealt round-robin to the 8 XCDs of an MI355X, each with its own 4 MB L2.  For a grid
// (tiles, frames) that means the tiles of ONE frame land on eight different L2s and every halo row / patch row / staged
// array is fetched once per XCD.  xcd_frame_major() renumbers the workgroups so that XCD k processes whole frames
// (frames k*F/8 .. (k+1)*F/8 - 1 in order): a frame's working set then lives in one L2.  Placement only; results do not
// depend on it.  Falls back to the identity when the grid size is not a multiple of 8.
#pragma once
#include <hip/hip_runtime.h>

namespace plp {

// Division of a wave-uniform index by a wave-uniform runtime divisor: there is no scalar integer division, so `n / d` costs every lane ~25 vector instructions --
// twice per workgroup in the tile kernels (this mapping, then tile -> (column, row)): 60 of k_blur7's 460.  The host knows the divisors at launch: it passes
// ceil(2^32 / d), and n / d = mulhi(n, magic) exactly while n * d < 2^32 (plp_div_magic returns 0 -- "divide" -- where that does not hold, and for d = 1).
__host__ __device__ inline uint32_t plp_div_magic(uint32_t d, uint64_t n_max) {
    if (d < 2 || n_max * d >= (1ull << 32)) return 0u;
    return (uint32_t)(((1ull << 32) + d - 1) / d);
}
__device__ __forceinline__ unsigned plp_div(unsigned n, unsigned d, uint32_t magic) { return magic ? __umulhi(n, magic) : n / d; }

// grid = (gx, gy): x = tile index inside a frame, y = frame.  Returns the logical (tile, frame) of this workgroup.  gx_magic = plp_div_magic(gx, gx * gy) or 0.
__device__ __forceinline__ void xcd_frame_major(unsigned& tile, unsigned& frame, uint32_t gx_magic = 0u) {
    const unsigned gx = gridDim.x, total = gridDim.x * gridDim.y;
    tile = blockIdx.x; frame = blockIdx.y;
    if ((total & 7u) == 0u) {
        const unsigned lin = blockIdx.x + gx * blockIdx.y;
        const unsigned logical = (lin & 7u) * (total >> 3) + (lin >> 3);
        frame = plp_div(logical, gx, gx_magic); tile = logical - frame * gx;
    }
}

}  // namespace plp
