// Batched replay helpers: the queries of the tracker's per-frame matcher calls, built on the device from the
// features of the preceding frames of the batch.  In the reference, tracking_module reprojects the previous frame's landmarks with
// the motion model (module/frame_tracker.cc:54-101) and the local map (tracking_module.cc:908-1064); a replay without a map stands
// in a camera that pans by a fixed pixel shift per frame, so "reprojection" is the previous key point moved by that shift.
// One launch per feature kind replaces the chain of tensor slices, additions, concatenations and copies that bench.py used to issue
// (about forty small launches per step, a third of the matching stage's time).  C ABI: include/plp_front.h.
#include <hip/hip_runtime.h>

#include "plp_common.hpp"
#include "plp_barrier.hpp"

namespace plp {

// grid = (ceil(cap / 256), B), block = 256
__global__ __launch_bounds__(256) void k_replay_point_queries(const plp_keypoint* __restrict__ kps, const int32_t* __restrict__ counts, int halo, int cap, float sx,
                                                              float sy, float2* __restrict__ q1_reproj, int32_t* __restrict__ q1_level, float* __restrict__ q1_angle,
                                                              int32_t* __restrict__ q1_counts, float2* __restrict__ q2_reproj, int32_t* __restrict__ q2_level,
                                                              uint8_t* __restrict__ q2_valid) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cap) return;
    const int f1 = halo + b - 1, f2 = halo + b - 2;            // frames b-1 and b-2 in the (halo + B)-frame feature arrays
    const int c1 = min(max(counts[f1], 0), cap), c2 = min(max(counts[f2], 0), cap);
    const plp_keypoint k1 = kps[(size_t)f1 * cap + i], k2 = kps[(size_t)f2 * cap + i];
    const size_t o1 = (size_t)b * cap + i;
    q1_reproj[o1] = make_float2(__fadd_rn(k1.x, sx), __fadd_rn(k1.y, sy));
    q1_level[o1] = k1.octave; q1_angle[o1] = k1.angle;
    if (i == 0) q1_counts[b] = c1;
    // local landmarks of frame b: the key points of frame b-2 (two shifts away), then those of frame b-1
    const size_t o2 = (size_t)b * 2 * cap;
    q2_reproj[o2 + i] = make_float2(__fadd_rn(k2.x, __fmul_rn(2.f, sx)), __fadd_rn(k2.y, __fmul_rn(2.f, sy)));
    q2_level[o2 + i] = k2.octave; q2_valid[o2 + i] = i < c2;
    q2_reproj[o2 + cap + i] = make_float2(__fadd_rn(k1.x, sx), __fadd_rn(k1.y, sy));
    q2_level[o2 + cap + i] = k1.octave; q2_valid[o2 + cap + i] = i < c1;
}

// q2_* (optional): the local line landmarks of frame b = the key lines of frame b-2 (two shifts away), then those of frame b-1,
// the line counterpart of the point kernel's q2 (tracking_module.cc:975-1060 search_local_landmarks_line).  t_kp_octave (optional):
// undist_keypts_.at(i).octave of frame b for i < cap, the key POINT octave match_frame_and_landmarks_line reads with a LINE index
// (projection.cc:187,192); 0 where frame b has fewer key points than i (the reference's .at() would throw there).
__global__ __launch_bounds__(256) void k_replay_line_queries(const plp_keyline* __restrict__ kl, const int32_t* __restrict__ counts, int halo, int cap, float sx, float sy,
                                                             float2* __restrict__ q_sp, float2* __restrict__ q_ep, int32_t* __restrict__ q_level,
                                                             int32_t* __restrict__ q_counts, float2* __restrict__ q2_sp, float2* __restrict__ q2_ep,
                                                             int32_t* __restrict__ q2_level, uint8_t* __restrict__ q2_valid, const plp_keypoint* __restrict__ kps,
                                                             const int32_t* __restrict__ kp_counts, int kp_cap, int32_t* __restrict__ t_kp_octave) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cap) return;
    const int f1 = halo + b - 1;
    const plp_keyline k = kl[(size_t)f1 * cap + i];
    const size_t o = (size_t)b * cap + i;
    const float2 sp1 = make_float2(__fadd_rn(k.startPointX, sx), __fadd_rn(k.startPointY, sy));
    const float2 ep1 = make_float2(__fadd_rn(k.endPointX, sx), __fadd_rn(k.endPointY, sy));
    const int c1 = min(max(counts[f1], 0), cap);
    q_sp[o] = sp1; q_ep[o] = ep1;
    q_level[o] = k.octave;
    if (i == 0) q_counts[b] = c1;
    if (q2_sp) {
        const int f2 = halo + b - 2;
        const plp_keyline k2 = kl[(size_t)f2 * cap + i];
        const int c2 = min(max(counts[f2], 0), cap);
        const size_t o2 = (size_t)b * 2 * cap;
        const float sx2 = __fmul_rn(2.f, sx), sy2 = __fmul_rn(2.f, sy);
        q2_sp[o2 + i] = make_float2(__fadd_rn(k2.startPointX, sx2), __fadd_rn(k2.startPointY, sy2));
        q2_ep[o2 + i] = make_float2(__fadd_rn(k2.endPointX, sx2), __fadd_rn(k2.endPointY, sy2));
        q2_level[o2 + i] = k2.octave; q2_valid[o2 + i] = i < c2;
        q2_sp[o2 + cap + i] = sp1; q2_ep[o2 + cap + i] = ep1;
        q2_level[o2 + cap + i] = k.octave; q2_valid[o2 + cap + i] = i < c1;
    }
    if (t_kp_octave) {
        const int f0 = halo + b;
        const int nk = min(max(kp_counts[f0], 0), kp_cap);
        t_kp_octave[o] = i < nk ? kps[(size_t)f0 * kp_cap + i].octave : 0;
    }
}

// ---- live rows of a padded per-frame array, packed back to back (the host boundary of the batched replay: what crosses PCIe is the
// features that exist, not the capacity they were allotted; SURVEY.md 8(d) "PCIe-inclusive rate")
// one workgroup: offsets[0..B] = exclusive prefix sum of min(max(counts, 0), cap)
__global__ __launch_bounds__(1024) void k_pack_offsets(const int32_t* __restrict__ counts, int B, int cap, long long* __restrict__ offsets) {
    __shared__ long long s_wave[16];
    __shared__ long long s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_carry = 0;
    wg_barrier();
    for (int base = 0; base < B; base += 1024) {
        const int i = base + tid;
        const long long v = i < B ? (long long)min(max(counts[i], 0), cap) : 0;
        long long inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const long long up = ((long long)__shfl_up((int)(inc >> 32), o) << 32) | (unsigned)__shfl_up((int)(unsigned)inc, o);
            if (lane >= o) inc += up;
        }
        if (lane == 63) s_wave[wv] = inc;
        wg_barrier();
        long long before = s_carry;
        for (int k = 0; k < wv; ++k) before += s_wave[k];
        if (i < B) offsets[i] = before + inc - v;
        wg_barrier();
        if (tid == 1023) s_carry = before + inc;
        wg_barrier();
    }
    if (tid == 0) offsets[B] = s_carry;
}

// grid = (ceil(cap * row_words / 1024), B), block = 256: 4 dwords per thread
__global__ __launch_bounds__(256) void k_pack_rows(const uint32_t* __restrict__ src, const int32_t* __restrict__ counts, int cap, int row_words,
                                                   const long long* __restrict__ offsets, uint32_t* __restrict__ dst) {
    const int b = blockIdx.y;
    const long long words = (long long)min(max(counts[b], 0), cap) * row_words;
    const uint32_t* s = src + (size_t)b * cap * row_words;
    uint32_t* d = dst + offsets[b] * row_words;
    const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i0 + k < words) d[i0 + k] = s[i0 + k];
}

}  // namespace plp

using namespace plp;

extern "C" {

plp_status plp_replay_point_queries_device(const plp_keypoint* feat_kps, const int32_t* feat_counts, int32_t halo, int32_t B, int32_t cap, float shift_x,
                                           float shift_y, float* q1_reproj, int32_t* q1_level, float* q1_angle, int32_t* q1_counts, float* q2_reproj,
                                           int32_t* q2_level, uint8_t* q2_valid, void* hip_stream) {
    if (!feat_kps || !feat_counts || !q1_reproj || !q1_level || !q1_angle || !q1_counts || !q2_reproj || !q2_level || !q2_valid)
        return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    if (halo < 2 || B <= 0 || cap <= 0) return set_error(PLP_ERR_INVALID_ARG, "halo must be >= 2, B and cap positive");
    hipLaunchKernelGGL(k_replay_point_queries, dim3((cap + 255) / 256, B), dim3(256), 0, (hipStream_t)hip_stream, feat_kps, feat_counts, halo, cap, shift_x, shift_y,
                       reinterpret_cast<float2*>(q1_reproj), q1_level, q1_angle, q1_counts, reinterpret_cast<float2*>(q2_reproj), q2_level, q2_valid);
    PLP_HIP(hipGetLastError());
    return PLP_OK;
}

plp_status plp_replay_line_queries_device(const plp_keyline* feat_kl, const int32_t* feat_counts, int32_t halo, int32_t B, int32_t cap, float shift_x,
                                          float shift_y, float* q_sp, float* q_ep, int32_t* q_level, int32_t* q_counts, float* q2_sp, float* q2_ep,
                                          int32_t* q2_level, uint8_t* q2_valid, const plp_keypoint* feat_kps, const int32_t* feat_kp_counts, int32_t kp_cap,
                                          int32_t* t_kp_octave, void* hip_stream) {
    if (!feat_kl || !feat_counts || !q_sp || !q_ep || !q_level || !q_counts) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    const bool lm = q2_sp || q2_ep || q2_level || q2_valid;
    if (lm && !(q2_sp && q2_ep && q2_level && q2_valid)) return set_error(PLP_ERR_INVALID_ARG, "q2_sp, q2_ep, q2_level and q2_valid go together");
    if (t_kp_octave && (!feat_kps || !feat_kp_counts || kp_cap <= 0)) return set_error(PLP_ERR_INVALID_ARG, "t_kp_octave needs feat_kps, feat_kp_counts and kp_cap");
    if (halo < (lm ? 2 : 1) || B <= 0 || cap <= 0) return set_error(PLP_ERR_INVALID_ARG, "halo must be >= 1 (>= 2 with the landmark queries), B and cap positive");
    hipLaunchKernelGGL(k_replay_line_queries, dim3((cap + 255) / 256, B), dim3(256), 0, (hipStream_t)hip_stream, feat_kl, feat_counts, halo, cap, shift_x, shift_y,
                       reinterpret_cast<float2*>(q_sp), reinterpret_cast<float2*>(q_ep), q_level, q_counts, reinterpret_cast<float2*>(q2_sp),
                       reinterpret_cast<float2*>(q2_ep), q2_level, q2_valid, feat_kps, feat_kp_counts, kp_cap, t_kp_octave);
    PLP_HIP(hipGetLastError());
    return PLP_OK;
}

plp_status plp_pack_rows_device(const void* src, const int32_t* counts, int32_t B, int32_t cap, int32_t row_bytes, void* dst, int64_t* offsets,
                                int32_t compute_offsets, void* hip_stream) {
    if (!src || !counts || !dst || !offsets) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    if (B <= 0 || cap <= 0 || row_bytes <= 0 || (row_bytes & 3)) return set_error(PLP_ERR_INVALID_ARG, "B, cap positive; row_bytes a positive multiple of 4");
    hipStream_t st = (hipStream_t)hip_stream;
    if (compute_offsets) hipLaunchKernelGGL(k_pack_offsets, dim3(1), dim3(1024), 0, st, counts, B, cap, reinterpret_cast<long long*>(offsets));
    const int row_words = row_bytes / 4;
    const long long per_frame = (long long)cap * row_words;
    hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)((per_frame + 1023) / 1024), B), dim3(256), 0, st, reinterpret_cast<const uint32_t*>(src), counts, cap, row_words,
                       reinterpret_cast<const long long*>(offsets), reinterpret_cast<uint32_t*>(dst));
    PLP_HIP(hipGetLastError());
    return PLP_OK;
}

}  // extern "C"
