// Structures shared by the ORB kernels and their host driver.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/plp_front.h"
#include "orb_tables.hpp"

namespace plp {

// k_quadtree keeps 26 bytes of node state per list entry (+ 8 KB of radix counters / key cache) in LDS and a level's lists never hold more than
// 3 * quota + 8 nodes: 5 888 nodes is what a CU's 160 KB hold, i.e. a per-level quota of up to 1 960 key points (K = 6 000 at 8 levels, K = 2 000 at
// 2 levels; until round 5 the bound was 2 048 nodes).  Beyond that plp_orb_create refuses (the reference's quota of a single level at K = 2 000 does).
constexpr int kQtMaxNodesLds = 5888;
// ... with the kernel's radix / key block halved (4 KB instead of 8: levels of at most 1024 cells, fewer candidates sorted with LDS counters) 6 040 nodes fit: a quota of
// up to 2 010, i.e. ONE level at K = 2 000 -- what the reference's orb_params accept and round 5 still refused (feature/orb_params.cc:40-54).
constexpr int kQtMaxNodesLdsSmallBlock = 6040;

// Per-level constants as seen by the kernels (array of n_levels in HBM + a host copy).
struct LevelDev {
    int w, h, pitch;        // level size, row pitch in the pyramid / blur planes
    int blur_tiles;         // number of 128x64 blur tiles of this level
    uint32_t blur_tiles_x_magic;   // plp_div_magic of the level's tile columns (xcd_map.hpp): a tile index -> (column, row) without a vector division
    size_t off;             // byte offset of the level inside one frame's plane set
    float scale;            // scale_factors_[level]
    int sel_base, sel_cap;  // slot range of this level in the per-frame selected list
    int cell_base, n_cells; // this level's cells in the frame's cell list
    int quota;              // num_keypts_per_level_[level]
    int n_init_x;           // quadtree initial grid width
    double delta_x, delta_y;
    size_t qt_off;          // byte offset of this level's quadtree scratch inside one frame's scratch
    int qt_cap;             // candidate capacity of that scratch (4 u32 arrays of qt_cap)
    int sort_lo, sort_hi;   // key bits that take part in the radix sort
};

// Where the pyramid of frame f lives.  Level 0 is read in place from the caller's frames
// when they are 4-byte aligned (l0 = d_imgs), otherwise from an aligned copy.
struct OrbPlanes {
    const uint8_t* l0; size_t l0_frame_stride; int l0_pitch;
    uint8_t* pyr; size_t pyr_frame_stride;      // levels >= 1 at pyr + f*stride + lv[l].off
    __host__ __device__ const uint8_t* level_ptr(int frame, int level, const LevelDev& L) const {
        return level == 0 ? l0 + (size_t)frame * l0_frame_stride : pyr + (size_t)frame * pyr_frame_stride + L.off;
    }
    __host__ __device__ int level_pitch(int level, const LevelDev& L) const { return level == 0 ? l0_pitch : L.pitch; }
};

struct BlurTaps { int k[7]; };
struct UMax { int v[kHalfPatch + 1]; };

struct ResizeDev {
    const int16_t *xofs0, *xofs1, *a0, *a1, *yofs0, *yofs1, *b0, *b1;
    int col_base[kMaxLevels], row_base[kMaxLevels];
};

struct StereoArgs {
    const plp_keypoint *kps_l, *kps_r;     // B x cap
    const uint8_t *desc_l, *desc_r;        // B x cap x 32
    const int32_t *cnt_l, *cnt_r;          // B (NULL: cap)
    int cap;
    float fxb, tb;                         // focal_x_baseline, true_baseline
    float inv_scale[kMaxLevels];
    float* x_right; float* depth;          // B x cap outputs
    int32_t* corr;                         // B x cap scratch: int-truncated best correlation, -1 = no stereo match
};
void launch_stereo(hipStream_t st, const OrbPlanes& pl_l, const OrbPlanes& pl_r, const LevelDev* d_lv, const StereoArgs& A, int B);

void launch_resize(hipStream_t st, const OrbPlanes& pl, const LevelDev* h_lv, int level, int B, const ResizeDev& rs);
void launch_fast(hipStream_t st, const OrbPlanes& pl, const CellDesc* d_cells, int n_cells, const LevelDev* d_lv, int B,
                 int ini_thr, int min_thr, const uint8_t* d_mask, size_t mask_step, size_t mask_frame_stride,
                 uint32_t* cell_cand, int32_t* cell_count);
void launch_blur(hipStream_t st, const OrbPlanes& pl, uint8_t* blur, size_t blur_frame_stride, const LevelDev* d_lv,
                 int n_levels, int total_tiles, int B, const BlurTaps& taps, const LevelDev* h_lv = nullptr);
void launch_orient_rbrief(hipStream_t st, const OrbPlanes& pl, const uint8_t* blur, size_t blur_frame_stride,
                          const LevelDev* d_lv, int n_levels, const int32_t* sel, const int32_t* sel_count,
                          int total_sel_cap, const UMax& um, plp_keypoint* kps, uint8_t* desc, int cap, int32_t* counts,
                          int32_t* status, int B);

}  // namespace plp
