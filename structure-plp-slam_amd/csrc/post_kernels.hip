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
    if (A.kps && i < (A.counts ? A.counts[b] : A.cap)) {
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
    if (A.kl && A.depth && i < (A.kl_counts ? A.kl_counts[b] : A.kl_cap)) {   // frame.cc:1196-1217
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

void launch_post_extract(hipStream_t st, const PostArgs& A, int B) {
    const int n = A.cap > A.kl_cap ? A.cap : A.kl_cap;
    hipLaunchKernelGGL(k_post_extract, dim3((n + 255) / 256, B), dim3(256), 0, st, A);
}

}  // namespace plp
