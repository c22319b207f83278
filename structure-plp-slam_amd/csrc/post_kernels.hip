// Post-extract per-key-point step (include/plp_front.h: plp_post_extract_*): undistortion, bearings, stereo from depth.
// Restates camera/perspective.cc:130-175 and data/frame.cc:1169-1219 of the reference; cv::undistortPoints as in
// OpenCV 3.4.16 imgproc/undistort.cpp (cvUndistortPointsInternal), f64 throughout (+, -, *, /, sqrt only: IEEE
// operations, so the results are the oracle's bit for bit; the file is compiled with -ffp-contract=off).
// One thread per key point / key line; grid = (ceil(cap / 256), B).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cfloat>

#include "match_device.hpp"

namespace plp {

__device__ __forceinline__ void undistort_point(const PostArgs& A, float u_in, float v_in, float& out_x, float& out_y) {
    const double fx = A.fx_f, fy = A.fy_f, cx = A.cx_f, cy = A.cy_f;
    const double ifx = 1. / fx, ify = 1. / fy;
    double x = u_in, y = v_in;
    const double u = x, v = y;
    x = (x - cx) * ifx;
    y = (y - cy) * ify;
    {   // tilt compensation with the identity (kept: -0.0 + 0.0 and friends must round like the reference)
        const double ux = x * 1.0 + y * 0.0 + 1.0 * 0.0, uy = x * 0.0 + y * 1.0 + 1.0 * 0.0, uz = x * 0.0 + y * 0.0 + 1.0 * 1.0;
        const double invProj = uz ? 1. / uz : 1;
        x = invProj * ux; y = invProj * uy;
    }
    const double x0 = x, y0 = y;
    double error = DBL_MAX;
    for (int j = 0;; j++) {
        if (j >= 20) break;
        if (error < 1e-6) break;
        double r2 = x * x + y * y;
        const double icdist = (1 + ((0.0 * r2 + 0.0) * r2 + 0.0) * r2) / (1 + ((A.k[4] * r2 + A.k[1]) * r2 + A.k[0]) * r2);
        if (icdist < 0) {
            x = (u - cx) * ifx;
            y = (v - cy) * ify;
            break;
        }
        const double deltaX = 2 * A.k[2] * x * y + A.k[3] * (r2 + 2 * x * x) + 0.0 * r2 + 0.0 * r2 * r2;
        const double deltaY = A.k[2] * (r2 + 2 * y * y) + 2 * A.k[3] * x * y + 0.0 * r2 + 0.0 * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
        {
            r2 = x * x + y * y;
            const double r4 = r2 * r2, r6 = r4 * r2;
            const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
            const double cdist = 1 + A.k[0] * r2 + A.k[1] * r4 + A.k[4] * r6;
            const double icdist2 = 1. / (1 + 0.0 * r2 + 0.0 * r4 + 0.0 * r6);
            const double xd0 = x * cdist * icdist2 + A.k[2] * a1 + A.k[3] * a2 + 0.0 * r2 + 0.0 * r4;
            const double yd0 = y * cdist * icdist2 + A.k[2] * a3 + A.k[3] * a1 + 0.0 * r2 + 0.0 * r4;
            const double tx = xd0 * 1.0 + yd0 * 0.0 + 1.0 * 0.0, ty = xd0 * 0.0 + yd0 * 1.0 + 1.0 * 0.0, tz = xd0 * 0.0 + yd0 * 0.0 + 1.0 * 1.0;
            const double invProj = tz ? 1. / tz : 1;
            const double xd = invProj * tx, yd = invProj * ty;
            const double x_proj = xd * fx + cx, y_proj = yd * fy + cy;
            error = sqrt((x_proj - u) * (x_proj - u) + (y_proj - v) * (y_proj - v));
        }
    }
    const double xx = fx * x + 0.0 * y + cx;
    const double yy = 0.0 * x + fy * y + cy;
    const double ww = 1. / (0.0 * x + 0.0 * y + 1.0);
    out_x = (float)(xx * ww);
    out_y = (float)(yy * ww);
}

__global__ __launch_bounds__(256) void k_post_extract(PostArgs A) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (A.kps && i < (A.counts ? min(A.counts[b], A.cap) : A.cap)) {
        const size_t o = (size_t)b * A.cap + i;
        const plp_keypoint k = A.kps[o];
        plp_keypoint un;
        undistort_point(A, k.x, k.y, un.x, un.y);
        un.size = k.size; un.angle = k.angle; un.response = 0.f; un.octave = k.octave; un.class_id = -1;
        if (A.undist) A.undist[o] = un;
        if (A.bearings) {   // perspective.cc:165-175 (true double intrinsics, float point)
            const double xn = ((double)un.x - A.cx) / A.fx, yn = ((double)un.y - A.cy) / A.fy;
            const double l2 = sqrt(xn * xn + yn * yn + 1.0);
            double* bo = A.bearings + 3 * o;
            bo[0] = xn / l2; bo[1] = yn / l2; bo[2] = 1.0 / l2;
        }
        if (A.depth && A.x_right && A.depths) {   // frame.cc:1177-1194
            const float* D = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(A.depth) + (size_t)b * A.depth_frame_stride);
            const float d = *reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(D) + (size_t)(int)k.y * A.depth_step + 4 * (size_t)(int)k.x);
            float xr = -1.f, dp = -1.f;
            if (!(d <= 0)) { dp = d; xr = (float)((double)un.x - A.fxb / (double)d); }
            A.x_right[o] = xr; A.depths[o] = dp;
        }
    }
    if (A.kl && A.depth && i < (A.kl_counts ? min(A.kl_counts[b], A.kl_cap) : A.kl_cap)) {   // frame.cc:1196-1217
        const size_t o = (size_t)b * A.kl_cap + i;
        const plp_keyline l = A.kl[o];
        const uint8_t* D = reinterpret_cast<const uint8_t*>(A.depth) + (size_t)b * A.depth_frame_stride;
        const float dsp = *reinterpret_cast<const float*>(D + (size_t)(int)l.startPointY * A.depth_step + 4 * (size_t)(int)l.startPointX);
        const float dep = *reinterpret_cast<const float*>(D + (size_t)(int)l.endPointY * A.depth_step + 4 * (size_t)(int)l.endPointX);
        if (!(dsp < 0 || dep < 0)) {
            A.kl_depths[2 * o] = dsp; A.kl_depths[2 * o + 1] = dep;
            A.kl_x_right[2 * o] = (float)((double)l.startPointX - A.fxb / (double)dsp);
            A.kl_x_right[2 * o + 1] = (float)((double)l.endPointX - A.fxb / (double)dep);
        }
    }
}

// ------------------------------------------------------------------------------------------
// landmark::compute_descriptor (data/landmark.cc:181-245): median-of-Hamming-distances selection, one wave per landmark.
// Lane = candidate row i (rows i, i + 64, ... for more than 64 observations); its median is found by counting, without a
// sort: the rank-r element of the row's distances is the smallest distance v with #(distances <= v) > r.
// grid = (ceil(L / 4)), block = 256.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_landmark_descriptor(const uint8_t* __restrict__ descs, const int32_t* __restrict__ offsets, int L,
                                                             int32_t* __restrict__ best_idx) {
    const int lane = threadIdx.x & 63, l = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (l >= L) return;
    const int o0 = offsets[l], n = offsets[l + 1] - o0;
    if (n <= 0) { if (lane == 0) best_idx[l] = -1; return; }
    const uint4* D = reinterpret_cast<const uint4*>(descs + 32 * (size_t)o0);
    const int r = (int)(unsigned)(0.5 * (double)(n - 1));
    unsigned best = 0xffffffffu;   // (median << 16 | row): smallest median, then first row
    for (int i = lane; i < n; i += 64) {
        const uint4 a0 = D[2 * i], a1 = D[2 * i + 1];
        // histogram-free selection: distances are 0..256, so binary-search the value v with count(d <= v) > r
        int lo = 0, hi = 256;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int cnt = 0;
            for (int j = 0; j < n; ++j) {
                const uint4 b0 = D[2 * j], b1 = D[2 * j + 1];
                const int d = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
                              __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
                cnt += d <= mid;
            }
            if (cnt > r) hi = mid; else lo = mid + 1;
        }
        const unsigned key = ((unsigned)lo << 16) | (unsigned)i;
        best = key < best ? key : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned w = (unsigned)__shfl_xor((int)best, o); best = w < best ? w : best; }
    if (lane == 0) best_idx[l] = (int32_t)(best & 0xffffu);
}

void launch_landmark_descriptor(hipStream_t st, const uint8_t* descs, const int32_t* offsets, int L, int32_t* best_idx) {
    hipLaunchKernelGGL(k_landmark_descriptor, dim3((L + 3) / 4), dim3(256), 0, st, descs, offsets, L, best_idx);
}

// ------------------------------------------------------------------------------------------
// util::convert_to_grayscale / convert_to_true_depth (util/image_converter.cc:33-80).  4 pixels per thread.
// grid = (ceil(cols / 1024), rows, B), block = 256.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_to_gray(const uint8_t* __restrict__ src, int cols, size_t src_step, size_t src_fs, int channels, int bgr,
                                                 uint8_t* __restrict__ dst, size_t dst_step, size_t dst_fs) {
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y, b = blockIdx.z;
    if (x0 >= cols) return;
    const uint8_t* s = src + (size_t)b * src_fs + (size_t)y * src_step + (size_t)x0 * channels;
    uint8_t* d = dst + (size_t)b * dst_fs + (size_t)y * dst_step + x0;
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (x0 + i >= cols) break;
        const uint8_t* p = s + i * channels;
        const int r = bgr ? p[2] : p[0], g = p[1], bl = bgr ? p[0] : p[2];
        packed |= (uint32_t)((bl * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14) << (8 * i);
    }
    if (x0 + 3 < cols && (((uintptr_t)d) & 3) == 0) *reinterpret_cast<uint32_t*>(d) = packed;
    else
        for (int i = 0; i < 4 && x0 + i < cols; ++i) d[i] = (uint8_t)(packed >> (8 * i));
}

__global__ __launch_bounds__(256) void k_to_depth(const void* __restrict__ src, int is_u16, int cols, size_t src_step, size_t src_fs, float scale,
                                                  float* __restrict__ dst, size_t dst_step, size_t dst_fs) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= cols) return;
    const uint8_t* row = reinterpret_cast<const uint8_t*>(src) + (size_t)b * src_fs + (size_t)y * src_step;
    const float v = is_u16 ? (float)reinterpret_cast<const uint16_t*>(row)[x] : reinterpret_cast<const float*>(row)[x];
    float* o = reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(dst) + (size_t)b * dst_fs + (size_t)y * dst_step);
    o[x] = __fadd_rn(__fmul_rn(v, scale), 0.0f);
}

void launch_to_gray(hipStream_t st, const uint8_t* src, int rows, int cols, size_t src_step, size_t src_fs, int channels, int bgr, int B, uint8_t* dst,
                    size_t dst_step, size_t dst_fs) {
    hipLaunchKernelGGL(k_to_gray, dim3((cols + 1023) / 1024, rows, B), dim3(256), 0, st, src, cols, src_step, src_fs, channels, bgr, dst, dst_step, dst_fs);
}
void launch_to_depth(hipStream_t st, const void* src, int is_u16, int rows, int cols, size_t src_step, size_t src_fs, float scale, int B, float* dst,
                     size_t dst_step, size_t dst_fs) {
    hipLaunchKernelGGL(k_to_depth, dim3((cols + 255) / 256, rows, B), dim3(256), 0, st, src, is_u16, cols, src_step, src_fs, scale, dst, dst_step, dst_fs);
}

// ------------------------------------------------------------------------------------------
// Planar_Mapping_module::create_ColorToPlane, per key point (planar_mapping_module.cc:203-330).  grid = (ceil(cap/256), B).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_color_vote(const uint8_t* __restrict__ mask, int rows, int cols, size_t step, size_t fs,
                                                    const plp_keypoint* __restrict__ undist, const uint8_t* __restrict__ valid,
                                                    const int32_t* __restrict__ counts, int cap, int check3, int32_t* __restrict__ labels) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cap) return;
    const size_t o = (size_t)b * cap + i;
    int label = 0;
    if (i < (counts ? counts[b] : cap) && !(valid && !valid[o])) {
        const uint8_t* M = mask + (size_t)b * fs;
        auto at = [&](int y, int x) -> int { const uint8_t* p = M + (size_t)y * step + 3 * (size_t)x; return p[0] + (p[1] << 8) + (p[2] << 16); };
        const float px = undist[o].x, py = undist[o].y;
        if (!(py < 0 || py > (float)rows || px < 0 || px > (float)cols)) {
            const int y = (int)py, x = (int)px;
            if (y < rows && x < cols) {
                const int center = at(y, x);
                bool ok = center != 0;
                if (ok && check3) {
                    const int dy[8] = {1, -1, 1, -1, 1, -1, 0, 0}, dx[8] = {1, -1, -1, 1, 0, 0, -1, 1};
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int ny = y + dy[k], nx = x + dx[k];
                        if (ny > 0 && ny < rows && nx > 0 && nx < cols && at(ny, nx) != center) ok = false;
                    }
                }
                if (ok) label = center;
            }
        }
    }
    labels[o] = label;
}

void launch_color_vote(hipStream_t st, const uint8_t* mask, int rows, int cols, size_t step, size_t fs, const plp_keypoint* undist, const uint8_t* valid,
                       const int32_t* counts, int cap, int B, int check3, int32_t* labels) {
    hipLaunchKernelGGL(k_color_vote, dim3((cap + 255) / 256, B), dim3(256), 0, st, mask, rows, cols, step, fs, undist, valid, counts, cap, check3, labels);
}

// ------------------------------------------------------------------------------------------
// util::stereo_rectifier (util/stereo_rectifier.cc:61-62, 83-84).
// k_rectify_map: cv::initUndistortRectifyMap(CV_32F) -- a row is one dependent chain (_x += ir[0] ...), so one thread
// walks one row; this runs once per camera.  grid = ceil(rows / 64), block = 64.
// k_remap_linear: cv::remap(INTER_LINEAR, BORDER_CONSTANT 0) on 8UC1, 4 destination pixels per thread, the map pair shared
// by the B frames of a launch.  grid = (ceil(dcols / 1024), drows, B), block = 256.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_rectify_map(RectifyArgs A) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= A.rows) return;
    const double* ir = A.ir;
    const double* d = A.d;
    double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
    float* mx = reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(A.map_x) + (size_t)i * A.map_step);
    float* my = reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(A.map_y) + (size_t)i * A.map_step);
    for (int j = 0; j < A.cols; ++j, _x += ir[0], _y += ir[3], _w += ir[6]) {
        const double w = 1. / _w, x = _x * w, y = _y * w;
        const double x2 = x * x, y2 = y * y;
        const double r2 = x2 + y2, _2xy = 2 * x * y;
        const double kr = (1 + ((d[4] * r2 + d[1]) * r2 + d[0]) * r2) / (1 + ((d[7] * r2 + d[6]) * r2 + d[5]) * r2);
        const double xd = (x * kr + d[2] * _2xy + d[3] * (r2 + 2 * x2) + d[8] * r2 + d[9] * r2 * r2);
        const double yd = (y * kr + d[2] * (r2 + 2 * y2) + d[3] * _2xy + d[10] * r2 + d[11] * r2 * r2);
        mx[j] = (float)(A.fx * xd + A.u0);
        my[j] = (float)(A.fy * yd + A.v0);
    }
}

// cv::fisheye::initUndistortRectifyMap (equidistant model), same walk along a row
__global__ __launch_bounds__(64) void k_rectify_map_fisheye(RectifyArgs A) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= A.rows) return;
    const double* ir = A.ir;
    const double* d = A.d;
    double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
    float* mx = reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(A.map_x) + (size_t)i * A.map_step);
    float* my = reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(A.map_y) + (size_t)i * A.map_step);
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    for (int j = 0; j < A.cols; ++j, _x += ir[0], _y += ir[3], _w += ir[6]) {
        double u, v;
        if (_w <= 0) { u = (_x > 0) ? -inf : inf; v = (_y > 0) ? -inf : inf; }
        else {
            const double x = _x / _w, y = _y / _w;
            const double r = sqrt(x * x + y * y);
            const double theta = atan(r);
            const double theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
            const double theta_d = theta * (1 + d[0] * theta2 + d[1] * theta4 + d[2] * theta6 + d[3] * theta8);
            const double scale = (r == 0) ? 1.0 : theta_d / r;
            u = A.fx * x * scale + A.u0;
            v = A.fy * y * scale + A.v0;
        }
        mx[j] = (float)u;
        my[j] = (float)v;
    }
}

__global__ __launch_bounds__(256) void k_remap_linear(const uint8_t* __restrict__ src, int rows, int cols, size_t step, size_t fs,
                                                      const float* __restrict__ map_x, const float* __restrict__ map_y, size_t map_step, int dcols,
                                                      uint8_t* __restrict__ dst, size_t dst_step, size_t dst_fs) {
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y, b = blockIdx.z;
    if (x0 >= dcols) return;
    const uint8_t* S = src + (size_t)b * fs;
    const float* mx = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(map_x) + (size_t)y * map_step) + x0;
    const float* my = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(map_y) + (size_t)y * map_step) + x0;
    uint8_t* d = dst + (size_t)b * dst_fs + (size_t)y * dst_step + x0;
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (x0 + i >= dcols) break;
        const int fsx = __float2int_rn(__fmul_rn(mx[i], 32.f)), fsy = __float2int_rn(__fmul_rn(my[i], 32.f));
        const int ax = fsx & 31, ay = fsy & 31;
        int sx = fsx >> 5, sy = fsy >> 5;
        sx = sx > 32767 ? 32767 : (sx < -32768 ? -32768 : sx);
        sy = sy > 32767 ? 32767 : (sy < -32768 ? -32768 : sy);
        // 15-bit weights 32 (32 - ax)(32 - ay) ...; cv's table stores 1.0 as 32767 + 1 on the opposite tap, same result
        const int w0 = 32 * (32 - ax) * (32 - ay), w1 = 32 * ax * (32 - ay), w2 = 32 * (32 - ax) * ay, w3 = 32 * ax * ay;
        const bool xin0 = sx >= 0 && sx < cols, xin1 = sx + 1 >= 0 && sx + 1 < cols, yin0 = sy >= 0 && sy < rows, yin1 = sy + 1 >= 0 && sy + 1 < rows;
        const uint8_t* r0 = S + (size_t)(yin0 ? sy : 0) * step;
        const uint8_t* r1 = S + (size_t)(yin1 ? sy + 1 : 0) * step;
        const int v0 = (xin0 && yin0) ? r0[sx] : 0, v1 = (xin1 && yin0) ? r0[sx + 1] : 0;
        const int v2 = (xin0 && yin1) ? r1[sx] : 0, v3 = (xin1 && yin1) ? r1[sx + 1] : 0;
        const int r = (v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3 + (1 << 14)) >> 15;
        packed |= (uint32_t)(r > 255 ? 255 : r) << (8 * i);
    }
    if (x0 + 3 < dcols && (((uintptr_t)d) & 3) == 0) *reinterpret_cast<uint32_t*>(d) = packed;
    else
        for (int i = 0; i < 4 && x0 + i < dcols; ++i) d[i] = (uint8_t)(packed >> (8 * i));
}

void launch_rectify_map(hipStream_t st, const RectifyArgs& A, bool fisheye) {
    if (fisheye) hipLaunchKernelGGL(k_rectify_map_fisheye, dim3((A.rows + 63) / 64), dim3(64), 0, st, A);
    else hipLaunchKernelGGL(k_rectify_map, dim3((A.rows + 63) / 64), dim3(64), 0, st, A);
}
void launch_remap_linear(hipStream_t st, const uint8_t* src, int rows, int cols, size_t step, size_t fs, const float* map_x, const float* map_y,
                         size_t map_step, int drows, int dcols, int B, uint8_t* dst, size_t dst_step, size_t dst_fs) {
    hipLaunchKernelGGL(k_remap_linear, dim3((dcols + 1023) / 1024, drows, B), dim3(256), 0, st, src, rows, cols, step, fs, map_x, map_y, map_step, dcols,
                       dst, dst_step, dst_fs);
}

void launch_post_extract(hipStream_t st, const PostArgs& A, int B) {
    const int n = A.cap > A.kl_cap ? A.cap : A.kl_cap;
    hipLaunchKernelGGL(k_post_extract, dim3((n + 255) / 256, B), dim3(256), 0, st, A);
}

}  // namespace plp
