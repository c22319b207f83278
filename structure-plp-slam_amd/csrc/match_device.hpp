// Problem descriptor shared by the matcher kernels and their host driver (array form of the
// reference's point matchers; see include/plp_front.h for the field meanings).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/plp_front.h"

namespace plp {

constexpr int kMatchK = 8;          // best candidates kept per query by k_match_topk

struct StagedTarget { float x, y; uint32_t packed; uint32_t t; };   // packed = octave | cell col << 8 | cell row << 16

struct MatchProblem {
    int mode;                        // plp_match_mode
    int n_cap, m_cap;                // per-problem strides of the target / query arrays
    int q_desc_stride;               // rows between two problems in q_desc (= m_cap unless the caller's queries overlap, plp_front.h)
    // targets = key points of the current frame (B x n_cap)
    const plp_keypoint* t_kps;       // undist_keypts_ (NULL in brute-force mode)
    const uint8_t* t_desc;           // descriptors_, 32 B rows
    const float* t_x_right;          // stereo_x_right_ (NULL: none)
    const uint8_t* t_occupied;       // key point already holds an observed landmark (NULL: none)
    const float* t_angle;            // brute-force mode: keypts_1[.].angle
    const int32_t* t_counts;         // per-problem n (NULL: n_cap)
    const plp_keyline* t_kl;         // line modes: key lines (replaces t_kps)
    const int32_t* t_kp_octave;      // line modes: the key-point octave read with a line index (reference quirk)
    const float* t_x_right2;         // line modes: second stereo coordinate
    const float* q_reproj2;          // line modes: reprojected end point
    const float* q_x_right2;
    int is_rgbd, num_levels_lsd;
    const int32_t* q_group; const int32_t* t_group;   // BOW mode: node id per feature
    const double* q_reproj_d;        // FUSE mode: f64 reprojection
    float inv_level_sigma_sq[16];    // FUSE mode
    int32_t* out_query_best;         // FUSE mode: B x m_cap
    int hamm_dist_thr, level_window, flags;
    const double* q_reproj2_d;       // FUSE_LINE
    const double* q_bearing; const double* t_bearing; const double* epipolar;   // TRIANGULATION
    // queries = landmarks / last-frame key points / key-frame key points (B x m_cap), reference order
    const uint8_t* q_valid;          // NULL: all valid
    const float* q_reproj;           // x, y
    const float* q_x_right;
    const int32_t* q_level;
    const float* q_angle;
    const uint8_t* q_desc;
    const uint8_t* q_has_obs;        // NULL: every accepted query blocks its key point
    const int32_t* q_counts;         // per-problem m (NULL: m_cap)
    // parameters
    float margin, lowe_ratio;
    int direction, check_orientation, num_levels;
    float scale_factors[16];
    float grid_min_x, grid_min_y;
    double inv_cell_w, inv_cell_h;
    int grid_cols, grid_rows;
    // scratch + outputs
    uint32_t* klist;                 // B x m_cap x kMatchK, packed distance<<20 | octave<<16 | target (sorted)
    uint32_t* klist2;                // B x m_cap x kMatchK: ranks kMatchK+1 .. 2 kMatchK of a query with more than kMatchK candidates (k_match_topk_cells only)
    int32_t* kcount;                 // B x m_cap
    int32_t* claim;                  // B x m_cap
    int32_t* full_list;              // B x m_cap
    StagedTarget* sorted;            // B x n_cap: free in-grid targets bucketed by grid row
    float* sorted_xr;                // B x n_cap
    uint16_t* cell_start;            // B x 4104: first position of every grid cell in `sorted`
    int sorted_valid;                // set by launch_match when k_match_prep ran
    int lds_targets;                 // targets of a frame k_match_topk_cells holds in LDS (<= n_cap; plp_match_args.t_count_hint), the rest is read from `sorted`
    int32_t* dbg;                    // 4 counters: exact rescans, resolve rounds (accumulated)
    int32_t* out_match;              // B x n_cap: query index per key point, -1 = none
    int32_t* out_num;                // B
};

struct MihRanks { uint8_t rank[256]; };   // enumeration rank of an 8-bit flip pattern inside its popcount class (Mihasher::query)
void launch_lbd_match_1nn(hipStream_t st, const uint8_t* q, const int32_t* q_counts, int nq_cap, const uint8_t* t, const int32_t* t_counts,
                          int nt_cap, const MihRanks& R, int32_t* out_idx, int32_t* out_dist, int B);
void launch_match(hipStream_t st, const MatchProblem& P, int B);
hipError_t configure_match_kernels();   // per-device kernel attributes (dynamic LDS of k_match_resolve)
struct AreaArgs {
    const plp_keypoint *kps1, *kps2; const uint8_t *desc1, *desc2; int n1, n2;
    float grid_min_x, grid_min_y; double inv_cell_w, inv_cell_h; int grid_cols, grid_rows;
    float* prev_pts; float margin, lowe_ratio; int check_orientation;
    int32_t* matched_2_in_1; int32_t* num_matches;
    uint32_t* scratch;   // n2 x 2 words: matched distance, matched idx_1
};
void launch_match_area(hipStream_t st, const AreaArgs& A);
void launch_hamming_matrix(hipStream_t st, const uint8_t* q, int nq, const uint8_t* t, int nt, uint16_t* dist);

// post-extract step (post_kernels.hip)
struct PostArgs {
    double fx, fy, cx, cy;            // camera::perspective doubles (bearings)
    double fx_f, fy_f, cx_f, cy_f;    // the float-rounded matrix cv::undistortPoints sees
    double k[5];                      // float-rounded k1, k2, p1, p2, k3
    double fxb;                       // focal_x_baseline_
    const plp_keypoint* kps; const int32_t* counts; int cap;
    const float* depth; size_t depth_step, depth_frame_stride;
    plp_keypoint* undist; double* bearings; float* x_right; float* depths;
    const plp_keyline* kl; const int32_t* kl_counts; int kl_cap; float* kl_depths; float* kl_x_right;
};
void launch_post_extract(hipStream_t st, const PostArgs& A, int B);
void launch_to_gray(hipStream_t st, const uint8_t* src, int rows, int cols, size_t src_step, size_t src_fs, int channels, int bgr, int B, uint8_t* dst,
                    size_t dst_step, size_t dst_fs);
void launch_to_depth(hipStream_t st, const void* src, int is_u16, int rows, int cols, size_t src_step, size_t src_fs, float scale, int B, float* dst,
                     size_t dst_step, size_t dst_fs);
void launch_color_vote(hipStream_t st, const uint8_t* mask, int rows, int cols, size_t step, size_t fs, const plp_keypoint* undist, const uint8_t* valid,
                       const int32_t* counts, int cap, int B, int check3, int32_t* labels);
void launch_landmark_descriptor(hipStream_t st, const uint8_t* descs, const int32_t* offsets, int L, int32_t* best_idx);
struct RectifyArgs {
    double ir[9];                     // (K_rect * R)^-1
    double d[12];                     // k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4
    double fx, fy, u0, v0;            // the unrectified camera
    int rows, cols;
    float* map_x; float* map_y; size_t map_step;
};
void launch_rectify_map(hipStream_t st, const RectifyArgs& A, bool fisheye);
void launch_remap_linear(hipStream_t st, const uint8_t* src, int rows, int cols, size_t step, size_t fs, const float* map_x, const float* map_y,
                         size_t map_step, int drows, int dcols, int B, uint8_t* dst, size_t dst_step, size_t dst_fs);

}  // namespace plp
