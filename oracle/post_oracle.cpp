// TEST INFRASTRUCTURE ONLY (oracle).  Not part of the product path.
//
// CPU restatement of the per-key-point step that follows extraction in every data::frame constructor
// (src/PLPSLAM/data/frame.cc:68-86, 110-128, ... SURVEY.md 8(f) item 1):
//   camera::perspective::undistort_keypoints          camera/perspective.cc:130-162
//   camera::perspective::convert_keypoints_to_bearings camera/perspective.cc:165-175
//   data::frame::compute_stereo_from_depth            data/frame.cc:1169-1219 (key points and key lines)
// (assign_keypoints_to_grid, data/common.cc:205-231, is restated in match_oracle.cpp.)
//
// cv::undistortPoints is third-party (OpenCV, not under /root/reference): restated from the plain-C code path of
// OpenCV 3.4.16 imgproc/undistort.cpp (cvUndistortPointsInternal) = "OpenCV-knowledge", PARITY UNPINNED -- the reference
// has no test for this step.  Facts taken from the reference: the camera matrix and the distortion vector are stored
// as cv::Mat_<float> (perspective.cc:47-48), so the solver sees FLOAT-rounded fx, fy, cx, cy, k1, k2, p1, p2, k3; R is
// empty and P is the same float matrix; TermCriteria(EPS | MAX_ITER, 20, 1e-6); input and output points are CV_32FC2.
#include <cmath>
#include <cstdint>
#include <limits>

#include "cv_restated.hpp"

namespace {

struct KeyPoint { float x, y, size, angle, response; int octave, class_id; };
struct KeyLineRec {
    float angle; int class_id; int octave; float pt_x, pt_y, response, size, startPointX, startPointY, endPointX, endPointY,
        sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY, lineLength; int numOfPixels;
};

// cvUndistortPointsInternal for one CV_32FC2 point, distortion (k1, k2, p1, p2, k3), no tilt, R = I, P = A
void undistort_point(double fx, double fy, double cx, double cy, const double k[5], float u_in, float v_in, float& out_x, float& out_y) {
    const double ifx = 1. / fx, ify = 1. / fy;
    double x = u_in, y = v_in;
    const double u = x, v = y;
    x = (x - cx) * ifx;
    y = (y - cy) * ify;
    // tilt compensation with the identity matrix: x*1 + y*0 + 1*0, invProj = 1/1
    {
        const double ux = x * 1.0 + y * 0.0 + 1.0 * 0.0, uy = x * 0.0 + y * 1.0 + 1.0 * 0.0, uz = x * 0.0 + y * 0.0 + 1.0 * 1.0;
        const double invProj = uz ? 1. / uz : 1;
        x = invProj * ux; y = invProj * uy;
    }
    const double x0 = x, y0 = y;
    double error = std::numeric_limits<double>::max();
    for (int j = 0;; j++) {
        if (j >= 20) break;
        if (error < 1e-6) break;
        double r2 = x * x + y * y;
        const double icdist = (1 + ((0.0 * r2 + 0.0) * r2 + 0.0) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0) {   // test: undistortPoints regression 14583
            x = (u - cx) * ifx;
            y = (v - cy) * ify;
            break;
        }
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + 0.0 * r2 + 0.0 * r2 * r2;
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + 0.0 * r2 + 0.0 * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
        {
            r2 = x * x + y * y;
            const double r4 = r2 * r2, r6 = r4 * r2;
            const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
            const double cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
            const double icdist2 = 1. / (1 + 0.0 * r2 + 0.0 * r4 + 0.0 * r6);
            const double xd0 = x * cdist * icdist2 + k[2] * a1 + k[3] * a2 + 0.0 * r2 + 0.0 * r4;
            const double yd0 = y * cdist * icdist2 + k[2] * a3 + k[3] * a1 + 0.0 * r2 + 0.0 * r4;
            const double tx = xd0 * 1.0 + yd0 * 0.0 + 1.0 * 0.0, ty = xd0 * 0.0 + yd0 * 1.0 + 1.0 * 0.0, tz = xd0 * 0.0 + yd0 * 0.0 + 1.0 * 1.0;
            const double invProj = tz ? 1. / tz : 1;
            const double xd = invProj * tx, yd = invProj * ty;
            const double x_proj = xd * fx + cx, y_proj = yd * fy + cy;
            error = std::sqrt((x_proj - u) * (x_proj - u) + (y_proj - v) * (y_proj - v));   // pow(d, 2) is exact squaring
        }
    }
    // RR = P * I = [fx 0 cx; 0 fy cy; 0 0 1]
    const double xx = fx * x + 0.0 * y + cx;
    const double yy = 0.0 * x + fy * y + cy;
    const double ww = 1. / (0.0 * x + 0.0 * y + 1.0);
    out_x = (float)(xx * ww);
    out_y = (float)(yy * ww);
}

}  // namespace

extern "C" {

// cam = {fx, fy, cx, cy, k1, k2, p1, p2, k3, focal_x_baseline} as the doubles of camera::perspective
void oracle_undistort_keypoints(const double* cam, const KeyPoint* dist, int n, KeyPoint* undist) {
    // cv_cam_matrix_ / cv_dist_params_ are float matrices, converted back to double by cvConvert
    const double fx = (double)(float)cam[0], fy = (double)(float)cam[1], cx = (double)(float)cam[2], cy = (double)(float)cam[3];
    const double k[5] = {(double)(float)cam[4], (double)(float)cam[5], (double)(float)cam[6], (double)(float)cam[7], (double)(float)cam[8]};
    for (int i = 0; i < n; ++i) {
        KeyPoint o{};   // undist_keypts.resize(): default key points, then pt / angle / size / octave are filled (:153-160)
        o.size = 0; o.angle = -1; o.response = 0; o.octave = 0; o.class_id = -1;
        undistort_point(fx, fy, cx, cy, k, dist[i].x, dist[i].y, o.x, o.y);
        o.angle = dist[i].angle; o.size = dist[i].size; o.octave = dist[i].octave;
        undist[i] = o;
    }
}

void oracle_bearings(const double* cam, const KeyPoint* undist, int n, double* bearings) {
    const double fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3];
    for (int i = 0; i < n; ++i) {
        const double x_normalized = (undist[i].x - cx) / fx;
        const double y_normalized = (undist[i].y - cy) / fy;
        const double l2_norm = std::sqrt(x_normalized * x_normalized + y_normalized * y_normalized + 1.0);
        bearings[3 * i] = x_normalized / l2_norm; bearings[3 * i + 1] = y_normalized / l2_norm; bearings[3 * i + 2] = 1.0 / l2_norm;
    }
}

// depth: rows x cols f32 (dense).  x_right / depths as std::vector<float>(n, -1) then overwritten.
void oracle_stereo_from_depth(const double* cam, const float* depth, int rows, int cols, const KeyPoint* kps, const KeyPoint* undist, int n,
                              float* x_right, float* depths) {
    (void)rows;
    for (int i = 0; i < n; ++i) {
        x_right[i] = -1; depths[i] = -1;
        const float x = kps[i].x, y = kps[i].y;
        const float d = depth[(size_t)(int)y * cols + (int)x];   // cv::Mat::at<float>(float, float): truncation
        if (d <= 0) continue;
        depths[i] = d;
        x_right[i] = (float)(undist[i].x - cam[9] / d);
    }
}

// key lines (frame.cc:1196-1217); the two output arrays are left untouched for skipped lines, as in the reference
void oracle_stereo_from_depth_lines(const double* cam, const float* depth, int rows, int cols, const KeyLineRec* kl, int n, float* depth_pairs,
                                    float* x_right_pairs) {
    (void)rows;
    for (int i = 0; i < n; ++i) {
        const float spx = kl[i].startPointX, spy = kl[i].startPointY, epx = kl[i].endPointX, epy = kl[i].endPointY;
        const float depth_sp = depth[(size_t)(int)spy * cols + (int)spx], depth_ep = depth[(size_t)(int)epy * cols + (int)epx];
        if (depth_sp < 0 || depth_ep < 0) continue;
        depth_pairs[2 * i] = depth_sp; depth_pairs[2 * i + 1] = depth_ep;
        x_right_pairs[2 * i] = (float)(spx - cam[9] / depth_sp);
        x_right_pairs[2 * i + 1] = (float)(epx - cam[9] / depth_ep);
    }
}

}  // extern "C"

// ---------------------------------------------------------------------------------------- input side
// util::convert_to_grayscale (util/image_converter.cc:33-75): cv::cvtColor(RGB2GRAY / BGR2GRAY / RGBA2GRAY / BGRA2GRAY) on
// CV_8U.  OpenCV-knowledge (3.4.16 color_rgb: RGB2Gray<uchar>, 14-bit fixed point R2Y = 4899, G2Y = 9617, B2Y = 1868,
// rounding term 1 << 13 folded into the red table); parity unpinned.
// util::convert_to_true_depth (:77-80): img.convertTo(img, CV_32F, 1.0 / depthmap_factor); OpenCV-knowledge: the scale is
// narrowed to float and applied in float (cvtScale work type for 16u->32f and 32f->32f is float).
extern "C" {

void oracle_convert_to_grayscale(const uint8_t* src, int rows, int cols, int channels, int bgr, uint8_t* dst) {
    for (size_t i = 0; i < (size_t)rows * cols; ++i) {
        const uint8_t* p = src + i * channels;
        const int r = bgr ? p[2] : p[0], g = p[1], b = bgr ? p[0] : p[2];
        dst[i] = (uint8_t)((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14);
    }
}

void oracle_convert_to_true_depth_u16(const uint16_t* src, size_t n, double depthmap_factor, float* dst) {
    const float scale = (float)(1.0 / depthmap_factor);
    for (size_t i = 0; i < n; ++i) dst[i] = (float)src[i] * scale + 0.0f;
}
void oracle_convert_to_true_depth_f32(const float* src, size_t n, double depthmap_factor, float* dst) {
    const float scale = (float)(1.0 / depthmap_factor);
    for (size_t i = 0; i < n; ++i) dst[i] = src[i] * scale + 0.0f;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------- plane colour vote (BASELINE config 5)
// Planar_Mapping_module::create_ColorToPlane (src/PLPSLAM/planar_mapping_module.cc:185-345), the per-key-point part: the
// colour label of the instance-segmentation mask (CV_8UC3) under an undistorted key point, kept only if it is non-zero and
// (check_3x3_window) every existing 8-neighbour has the same non-zero label.  label = c0 + (c1 << 8) + (c2 << 16); 0 = none.
// Quirks kept: the image-range test uses `>` rows / cols (:214-215), neighbours in row 0 / column 0 are never looked at
// (`> 0`, :262-293).  A point with (int)y == rows or (int)x == cols passes the reference's range test and reads out of
// bounds there; here it gets label 0.  The map mutation (new data::Plane, add_landmark) stays on the host.
extern "C" {

void oracle_color_vote(const uint8_t* mask, int rows, int cols, size_t step, const KeyPoint* undist, const uint8_t* valid, int n,
                       int check_3x3_window, int* labels) {
    auto at = [&](int y, int x) -> long { const uint8_t* p = mask + (size_t)y * step + 3 * (size_t)x; return p[0] + (p[1] << 8) + (p[2] << 16); };
    for (int i = 0; i < n; ++i) {
        labels[i] = 0;
        if (valid && !valid[i]) continue;
        const float px = undist[i].x, py = undist[i].y;
        if (py < 0 || py > rows || px < 0 || px > cols) continue;
        const int y = (int)py, x = (int)px;
        if (y >= rows || x >= cols) continue;   // reference: out-of-bounds read
        const long center = at(y, x);
        if (center == 0) continue;
        bool consistent = true;
        if (check_3x3_window) {
            static const int dy[8] = {1, -1, 1, -1, 1, -1, 0, 0}, dx[8] = {1, -1, -1, 1, 0, 0, -1, 1};
            for (int k = 0; k < 8 && consistent; ++k) {
                const int ny = y + dy[k], nx = x + dx[k];
                if (ny > 0 && ny < rows && nx > 0 && nx < cols) {
                    const long h = at(ny, nx);
                    if (h == 0 || h != center) consistent = false;
                }
            }
        }
        if (consistent) labels[i] = (int)center;
    }
}

}  // extern "C"

// ---------------------------------------------------------------------------------------- stereo rectification
// util::stereo_rectifier (src/PLPSLAM/util/stereo_rectifier.cc:38-85): the constructor builds one CV_32F map pair per eye
// with cv::initUndistortRectifyMap(K, D, R, K_rect, img_size, CV_32F, ...) (perspective model, :61-62), rectify() is
// cv::remap(in, out, map_x, map_y, cv::INTER_LINEAR) (:83-84, BORDER_CONSTANT, value 0) on the 8-bit images read by the
// EuRoC drivers (example/run_euroc_slam.cc:246).  Both are third-party (OpenCV, not under /root/reference): restated
// from the scalar code of OpenCV 3.4/4.x imgproc (undistort.cpp, imgwarp.cpp) = "OpenCV-knowledge", PARITY UNPINNED.
//   * K_rect is camera::perspective::cv_cam_matrix_, a cv::Mat_<float> (perspective.cc:47): the map sees FLOAT-rounded
//     fx, fy, cx, cy of the rectified camera; K, D, R come from the yaml as doubles (stereo_rectifier.cc:48-57).
//   * iR = (K_rect * R)^-1 by the closed-form 3x3 inverse of cv::Matx; one row walks _x, _y, _w by repeated addition.
//   * remap: float maps -> 1/32-pixel fixed point with cvRound (half to even), 15-bit bilinear weights, (sum + 2^14) >> 15.
//     The weight table is built the way initInterTab2D does, including its saturation of the weight 1.0 to 32767 and the
//     compensation that follows; the result is the same as with exact weights (tests/test_oracle_post.py checks that).
// The fisheye model (cv::fisheye::initUndistortRectifyMap, SVD inverse + atan) is not restated.
namespace {

void matx33_mul(const double a[9], const double b[9], double c[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += a[i * 3 + k] * b[k * 3 + j];
            c[i * 3 + j] = s;
        }
}

bool matx33_inv(const double a[9], double b[9]) {
    auto A = [&](int i, int j) { return a[i * 3 + j]; };
    double d = A(0, 0) * (A(1, 1) * A(2, 2) - A(2, 1) * A(1, 2)) - A(0, 1) * (A(1, 0) * A(2, 2) - A(2, 0) * A(1, 2)) +
               A(0, 2) * (A(1, 0) * A(2, 1) - A(2, 0) * A(1, 1));
    if (d == 0) return false;
    d = 1 / d;
    b[0] = (A(1, 1) * A(2, 2) - A(1, 2) * A(2, 1)) * d;
    b[1] = (A(0, 2) * A(2, 1) - A(0, 1) * A(2, 2)) * d;
    b[2] = (A(0, 1) * A(1, 2) - A(0, 2) * A(1, 1)) * d;
    b[3] = (A(1, 2) * A(2, 0) - A(1, 0) * A(2, 2)) * d;
    b[4] = (A(0, 0) * A(2, 2) - A(0, 2) * A(2, 0)) * d;
    b[5] = (A(0, 2) * A(1, 0) - A(0, 0) * A(1, 2)) * d;
    b[6] = (A(1, 0) * A(2, 1) - A(1, 1) * A(2, 0)) * d;
    b[7] = (A(0, 1) * A(2, 0) - A(0, 0) * A(2, 1)) * d;
    b[8] = (A(0, 0) * A(1, 1) - A(0, 1) * A(1, 0)) * d;
    return true;
}

using oracle::cv_round;

}  // namespace

extern "C" {

// returns 0, or -1 when K_rect * R is singular.  D: n_dist in {0, 4, 5, 8, 12} (k1 k2 p1 p2 [k3 [k4 k5 k6 [s1 s2 s3 s4]]])
int oracle_init_undistort_rectify_map(const double* K, const double* D, int n_dist, const double* R, const double* K_rect_f32, int rows,
                                      int cols, float* map_x, float* map_y) {
    double Ar[9], ArR[9], ir[9];
    for (int i = 0; i < 9; ++i) Ar[i] = (double)(float)K_rect_f32[i];
    matx33_mul(Ar, R, ArR);
    if (!matx33_inv(ArR, ir)) return -1;
    const double u0 = K[2], v0 = K[5], fx = K[0], fy = K[4];
    double d[12] = {0};
    for (int i = 0; i < n_dist && i < 12; ++i) d[i] = D[i];
    const double k1 = d[0], k2 = d[1], p1 = d[2], p2 = d[3], k3 = d[4], k4 = d[5], k5 = d[6], k6 = d[7], s1 = d[8], s2 = d[9], s3 = d[10],
                 s4 = d[11];
    for (int i = 0; i < rows; ++i) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < cols; ++j, _x += ir[0], _y += ir[3], _w += ir[6]) {
            const double w = 1. / _w, x = _x * w, y = _y * w;
            const double x2 = x * x, y2 = y * y;
            const double r2 = x2 + y2, _2xy = 2 * x * y;
            const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
            const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2);
            const double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2);
            // tilt with the identity matrix: vecTilt = (xd, yd, 1), invProj = 1
            const double vt0 = 1.0 * xd + 0.0 * yd + 0.0 * 1.0, vt1 = 0.0 * xd + 1.0 * yd + 0.0 * 1.0, vt2 = 0.0 * xd + 0.0 * yd + 1.0 * 1.0;
            const double invProj = vt2 ? 1. / vt2 : 1;
            const double u = fx * invProj * vt0 + u0;
            const double v = fy * invProj * vt1 + v0;
            map_x[(size_t)i * cols + j] = (float)u;
            map_y[(size_t)i * cols + j] = (float)v;
        }
    }
    return 0;
}

// cv::fisheye::initUndistortRectifyMap(K, D(4), R, K_rect, size, CV_32F) (util/stereo_rectifier.cc:67-68, the TUM-VI
// yaml): equidistant model theta_d = theta (1 + k1 theta^2 + k2 theta^4 + k3 theta^6 + k4 theta^8), restated from OpenCV
// 4.x fisheye.cpp (the _w <= 0 branch of 4.5+ included).  Two things keep this one from being bit-defined: OpenCV inverts
// K_rect * R with DECOMP_SVD (here: the closed-form 3x3 inverse, equal to ~1e-15 relative) and the map goes through
// atan(), whose last bit differs between libm implementations -- the map is compared with a tolerance (tests: 1e-4 px).
int oracle_fisheye_rectify_map(const double* K, const double* D4, const double* R, const double* K_rect_f32, int rows, int cols, float* map_x,
                               float* map_y) {
    double Ar[9], ArR[9], ir[9];
    for (int i = 0; i < 9; ++i) Ar[i] = (double)(float)K_rect_f32[i];
    matx33_mul(Ar, R, ArR);
    if (!matx33_inv(ArR, ir)) return -1;
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    for (int i = 0; i < rows; ++i) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < cols; ++j) {
            double u, v;
            if (_w <= 0) {
                u = (_x > 0) ? -std::numeric_limits<double>::infinity() : std::numeric_limits<double>::infinity();
                v = (_y > 0) ? -std::numeric_limits<double>::infinity() : std::numeric_limits<double>::infinity();
            } else {
                const double x = _x / _w, y = _y / _w;
                const double r = std::sqrt(x * x + y * y);
                const double theta = std::atan(r);
                const double theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
                const double theta_d = theta * (1 + D4[0] * theta2 + D4[1] * theta4 + D4[2] * theta6 + D4[3] * theta8);
                const double scale = (r == 0) ? 1.0 : theta_d / r;
                u = fx * x * scale + cx;
                v = fy * y * scale + cy;
            }
            map_x[(size_t)i * cols + j] = (float)u;
            map_y[(size_t)i * cols + j] = (float)v;
            _x += ir[0]; _y += ir[3]; _w += ir[6];
        }
    }
    return 0;
}

// cv::remap(src 8UC1, dst, map_x, map_y CV_32FC1, INTER_LINEAR, BORDER_CONSTANT, 0); dst has the maps' size
void oracle_remap_linear(const uint8_t* src, int rows, int cols, size_t step, const float* map_x, const float* map_y, int drows, int dcols,
                         uint8_t* dst) {
    oracle::remap_linear_u8(src, rows, cols, step, map_x, map_y, drows, dcols, dst);
}

}  // extern "C"
