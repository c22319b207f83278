// TEST INFRASTRUCTURE ONLY (oracle).  Not part of the product path.
//
// CPU restatement of the OpenCV primitives the reference calls on its hot
// path.  OpenCV is NOT vendored under /root/reference and is not installed in
// the build container, so every function in this file is "OpenCV-knowledge":
// it restates the published algorithm of OpenCV 3.4.16 (the version the
// reference README names, README.md:88-90) from memory of its plain-C code
// paths (no IPP, no OpenCL).  PARITY UNPINNED for this file: the reference
// holds no golden vectors for these calls; a cross-check
// against a real OpenCV build is still outstanding (tools/opencv_crosscheck.cpp).
//
// Call sites in the reference that these restate:
//   cv::resize(INTER_LINEAR)     src/PLPSLAM/feature/orb_extractor.cc:324
//   cv::FAST(..., true)          src/PLPSLAM/feature/orb_extractor.cc:404,410
//   cv::GaussianBlur (u8)        src/PLPSLAM/feature/orb_extractor.cc:149
//                                src/PLPSLAM/feature/line_descriptor/binary_descriptor_custom.cpp:355
//   cv::fastAtan2                src/PLPSLAM/feature/orb_extractor.cc:734
//   cvRound/cvFloor/cvCeil       throughout
//   cv::Sobel (u8 -> s16, k=3)   binary_descriptor_custom.cpp:392-393
//   cv::LineIterator.count       LSDDetector_custom.cpp:292-293
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace oracle {

// ---------------------------------------------------------------- images
struct Image {  // owning u8 single-channel image, dense rows
    int rows = 0, cols = 0;
    std::vector<uint8_t> data;
    Image() = default;
    Image(int r, int c) : rows(r), cols(c), data((size_t)r * c) {}
    bool empty() const { return rows == 0 || cols == 0; }
    uint8_t* row(int y) { return data.data() + (size_t)y * cols; }
    const uint8_t* row(int y) const { return data.data() + (size_t)y * cols; }
    uint8_t at(int y, int x) const { return data[(size_t)y * cols + x]; }
};

struct KeyPoint {  // same 7 fields / 28 bytes as cv::KeyPoint
    float x, y, size, angle, response;
    int octave, class_id;
};

// ---------------------------------------------------------------- rounding
// cvRound = round-half-to-even (SSE cvtsd2si / lrint under the default mode).
inline int cv_round(double v) { return (int)std::nearbyint(v); }
inline int cv_round(float v) { return (int)std::nearbyintf(v); }
inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
inline int cv_floor(float v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }
inline short sat_short(float v) {
    int i = cv_round(v);
    return (short)std::min(std::max(i, -32768), 32767);
}
inline int border101(int p, int len) {  // BORDER_REFLECT_101, any p
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

// ---------------------------------------------------------------- resize
// cv::resize(src, dst, Size(dw,dh), 0, 0, INTER_LINEAR) for CV_8UC1:
// 11-bit fixed-point coefficient tables, int32 horizontal pass, the
// (b0*(S0>>4))>>16 vertical pass.
struct ResizeTab {
    std::vector<int> xofs, yofs;
    std::vector<short> ialpha, ibeta;  // 2 per dst column / row
    int xmax = 0;
};
inline ResizeTab make_resize_tab(int sw, int sh, int dw, int dh) {
    ResizeTab t;
    t.xofs.resize(dw); t.yofs.resize(dh); t.ialpha.resize(2 * dw); t.ibeta.resize(2 * dh);
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    t.xmax = dw;
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            t.xmax = std::min(t.xmax, dx);
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        t.xofs[dx] = sx;
        t.ialpha[2 * dx] = sat_short((1.f - fx) * 2048);
        t.ialpha[2 * dx + 1] = sat_short(fx * 2048);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        t.yofs[dy] = sy;
        t.ibeta[2 * dy] = sat_short((1.f - fy) * 2048);
        t.ibeta[2 * dy + 1] = sat_short(fy * 2048);
    }
    return t;
}
inline Image resize_linear_u8(const Image& src, int dw, int dh) {
    Image dst(dh, dw);
    ResizeTab t = make_resize_tab(src.cols, src.rows, dw, dh);
    std::vector<int> r0(dw), r1(dw);
    auto hline = [&](int sy, std::vector<int>& out) {
        sy = sy < 0 ? 0 : (sy >= src.rows ? src.rows - 1 : sy);
        const uint8_t* S = src.row(sy);
        for (int dx = 0; dx < dw; ++dx) {
            int sx = t.xofs[dx];
            if (dx < t.xmax) out[dx] = S[sx] * t.ialpha[2 * dx] + S[sx + 1] * t.ialpha[2 * dx + 1];
            else out[dx] = S[sx] * 2048;
        }
    };
    for (int dy = 0; dy < dh; ++dy) {
        hline(t.yofs[dy], r0);
        hline(t.yofs[dy] + 1, r1);
        const int b0 = t.ibeta[2 * dy], b1 = t.ibeta[2 * dy + 1];
        uint8_t* D = dst.row(dy);
        for (int x = 0; x < dw; ++x)
            D[x] = (uint8_t)((((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2);
    }
    return dst;
}

// ---------------------------------------------------------------- FAST-9/16
struct FastPoint { int x, y, score; };

// cornerScore<16>: the min/max ladder over the 25-entry wrapped difference array.
inline int fast_corner_score16(const uint8_t* ptr, const int pixel[25], int threshold) {
    int d[25];
    const int v = ptr[0];
    for (int k = 0; k < 25; ++k) d[k] = v - ptr[pixel[k]];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k + 1], d[k + 2]);
        a = std::min(a, d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, d[k + 4]); a = std::min(a, d[k + 5]);
        a = std::min(a, d[k + 6]); a = std::min(a, d[k + 7]);
        a = std::min(a, d[k + 8]);
        a0 = std::max(a0, std::min(a, d[k]));
        a0 = std::max(a0, std::min(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max(d[k + 1], d[k + 2]);
        b = std::max(b, d[k + 3]); b = std::max(b, d[k + 4]); b = std::max(b, d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, d[k + 6]); b = std::max(b, d[k + 7]); b = std::max(b, d[k + 8]);
        b0 = std::min(b0, std::max(b, d[k]));
        b0 = std::min(b0, std::max(b, d[k + 9]));
    }
    return -b0 - 1;
}

// cv::FAST(roi, kps, threshold, nonmaxSuppression=true) on the w x h window whose
// top-left pixel is `base` (row stride `step`).  Output row-major, ROI coordinates.
inline void fast9_16_nms(const uint8_t* base, int step, int w, int h, int threshold,
                         std::vector<FastPoint>& out) {
    out.clear();
    static const int off[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                                   {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};
    int pixel[25];
    for (int k = 0; k < 16; ++k) pixel[k] = off[k][0] + off[k][1] * step;
    for (int k = 16; k < 25; ++k) pixel[k] = pixel[k - 16];
    threshold = std::min(std::max(threshold, 0), 255);
    uint8_t tab[512];
    for (int i = -255; i <= 255; ++i) tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);

    std::vector<uint8_t> sbuf((size_t)w * 3, 0);
    std::vector<int> cbuf((size_t)(w + 1) * 3, 0);
    uint8_t* buf[3] = {sbuf.data(), sbuf.data() + w, sbuf.data() + 2 * w};
    int* cpbuf[3] = {cbuf.data(), cbuf.data() + (w + 1), cbuf.data() + 2 * (w + 1)};

    for (int i = 3; i < h - 2; ++i) {
        const uint8_t* ptr = base + (size_t)i * step + 3;
        uint8_t* curr = buf[(i - 3) % 3];
        int* cornerpos = cpbuf[(i - 3) % 3];
        std::memset(curr, 0, w);
        int ncorners = 0;
        if (i < h - 3) {
            for (int j = 3; j < w - 3; ++j, ++ptr) {
                const int v = ptr[0];
                const uint8_t* tb = &tab[0] - v + 255;
                int d = tb[ptr[pixel[0]]] | tb[ptr[pixel[8]]];
                if (d == 0) continue;
                d &= tb[ptr[pixel[2]]] | tb[ptr[pixel[10]]];
                d &= tb[ptr[pixel[4]]] | tb[ptr[pixel[12]]];
                d &= tb[ptr[pixel[6]]] | tb[ptr[pixel[14]]];
                if (d == 0) continue;
                d &= tb[ptr[pixel[1]]] | tb[ptr[pixel[9]]];
                d &= tb[ptr[pixel[3]]] | tb[ptr[pixel[11]]];
                d &= tb[ptr[pixel[5]]] | tb[ptr[pixel[13]]];
                d &= tb[ptr[pixel[7]]] | tb[ptr[pixel[15]]];
                bool corner = false;
                if (d & 1) {
                    const int vt = v - threshold;
                    int count = 0;
                    for (int k = 0; k < 25; ++k) {
                        if (ptr[pixel[k]] < vt) { if (++count > 8) { corner = true; break; } }
                        else count = 0;
                    }
                }
                if (!corner && (d & 2)) {
                    const int vt = v + threshold;
                    int count = 0;
                    for (int k = 0; k < 25; ++k) {
                        if (ptr[pixel[k]] > vt) { if (++count > 8) { corner = true; break; } }
                        else count = 0;
                    }
                }
                if (corner) {
                    cornerpos[ncorners++] = j;
                    curr[j] = (uint8_t)fast_corner_score16(ptr, pixel, threshold);
                }
            }
        }
        cornerpos[w] = ncorners;  // slot w holds the count (as OpenCV's cornerpos[-1])
        if (i == 3) continue;
        const uint8_t* prev = buf[(i - 4 + 3) % 3];
        const uint8_t* pprev = buf[(i - 5 + 3) % 3];
        const int* ppos = cpbuf[(i - 4 + 3) % 3];
        const int np = ppos[w];
        for (int k = 0; k < np; ++k) {
            const int j = ppos[k];
            const int score = prev[j];
            if (score > prev[j + 1] && score > prev[j - 1] &&
                score > pprev[j - 1] && score > pprev[j] && score > pprev[j + 1] &&
                score > curr[j - 1] && score > curr[j] && score > curr[j + 1])
                out.push_back({j, i - 1, score});
        }
    }
}

// ---------------------------------------------------------------- Gaussian blur (u8, fixed point)
// Taps in 8.8 fixed point that sum to exactly 256: f64 Gaussian, normalised,
// quantised edge->centre with error diffusion, centre takes the remainder
// (OpenCV >= 3.4.10 "bit-exact" kernel).  KEEP ALL TAP TABLES GOING THROUGH HERE.
inline std::vector<int> gaussian_taps_q8(int n, double sigma) {
    std::vector<double> k(n);
    double sum = 0;
    const double scale2x = -0.5 / (sigma * sigma);
    for (int i = 0; i < n; ++i) {
        const double x = i - (n - 1) * 0.5;
        k[i] = std::exp(scale2x * x * x);
        sum += k[i];
    }
    for (int i = 0; i < n; ++i) k[i] /= sum;
    std::vector<int> q(n);
    const int n2 = n / 2;
    double err = 0;
    int acc = 0;
    for (int i = 0; i < n2; ++i) {
        const double adj = k[i] * 256.0 + err;
        const int v = cv_round(adj);
        err = adj - v;
        q[i] = q[n - 1 - i] = v;
        acc += v;
    }
    q[n2] = 256 - 2 * acc;
    return q;
}
// out = (sum_j ky[j] * (sum_i kx[i] * src[y+j-r][x+i-r]) + 32768) >> 16, REFLECT_101.
inline Image gaussian_blur_u8(const Image& src, int ksize, double sigma) {
    const std::vector<int> k = gaussian_taps_q8(ksize, sigma);
    const int r = ksize / 2, W = src.cols, H = src.rows;
    std::vector<uint16_t> hbuf((size_t)W * H);
    std::vector<int> xmap(W + 2 * r), ymap(H + 2 * r);
    for (int i = 0; i < W + 2 * r; ++i) xmap[i] = border101(i - r, W);
    for (int i = 0; i < H + 2 * r; ++i) ymap[i] = border101(i - r, H);
    for (int y = 0; y < H; ++y) {
        const uint8_t* S = src.row(y);
        uint16_t* hb = &hbuf[(size_t)y * W];
        for (int x = 0; x < W; ++x) {
            uint32_t a = 0;
            if (x >= r && x + r < W) for (int i = 0; i < ksize; ++i) a += (uint32_t)k[i] * S[x + i - r];
            else for (int i = 0; i < ksize; ++i) a += (uint32_t)k[i] * S[xmap[x + i]];
            hb[x] = (uint16_t)std::min<uint32_t>(a, 65535u);
        }
    }
    Image dst(H, W);
    std::vector<uint32_t> acc(W);
    for (int y = 0; y < H; ++y) {
        std::fill(acc.begin(), acc.end(), 0u);
        for (int j = 0; j < ksize; ++j) {
            const uint16_t* hb = &hbuf[(size_t)ymap[y + j] * W];
            const uint32_t kj = (uint32_t)k[j];
            for (int x = 0; x < W; ++x) acc[x] += kj * hb[x];
        }
        uint8_t* D = dst.row(y);
        for (int x = 0; x < W; ++x) D[x] = (uint8_t)std::min<uint32_t>((acc[x] + 32768u) >> 16, 255u);
    }
    return dst;
}

// ---------------------------------------------------------------- fastAtan2
// cv::fastAtan2(y, x): f32 odd polynomial, degrees in [0, 360).  No FMA.
inline float fast_atan2f_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    const float eps = (float)2.2204460492503131e-16;
    const float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// ---------------------------------------------------------------- Sobel 3x3 (u8 -> s16), REFLECT_101
inline void sobel3_s16(const Image& src, std::vector<int16_t>& dx, std::vector<int16_t>& dy) {
    const int W = src.cols, H = src.rows;
    dx.assign((size_t)W * H, 0); dy.assign((size_t)W * H, 0);
    for (int y = 0; y < H; ++y) {
        const uint8_t* r0 = src.row(border101(y - 1, H));
        const uint8_t* r1 = src.row(y);
        const uint8_t* r2 = src.row(border101(y + 1, H));
        for (int x = 0; x < W; ++x) {
            const int xm = border101(x - 1, W), xp = border101(x + 1, W);
            dx[(size_t)y * W + x] = (int16_t)((r0[xp] - r0[xm]) + 2 * (r1[xp] - r1[xm]) + (r2[xp] - r2[xm]));
            dy[(size_t)y * W + x] = (int16_t)((r2[xm] - r0[xm]) + 2 * (r2[x] - r0[x]) + (r2[xp] - r0[xp]));
        }
    }
}

// ------------------------------------------------------------------------------------------ cv::remap (INTER_LINEAR, u8)
// initInterTab2D(INTER_LINEAR, fixpt = true): 32 x 32 sub-pixel positions, 2 x 2 short weights each
inline const short* bilinear_tab_i() {
    static short tab[32 * 32 * 4];
    static bool ready = false;
    if (ready) return tab;
    float lin[32][2];
    for (int i = 0; i < 32; ++i) { const float x = i * (1.f / 32); lin[i][0] = 1.f - x; lin[i][1] = x; }
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            short* it = tab + (i * 32 + j) * 4;
            int isum = 0;
            for (int k1 = 0; k1 < 2; ++k1)
                for (int k2 = 0; k2 < 2; ++k2) {
                    const float v = lin[i][k1] * lin[j][k2] * 32768.f;
                    int r = cv_round(v);
                    r = r > 32767 ? 32767 : (r < -32768 ? -32768 : r);
                    it[k1 * 2 + k2] = (short)r;
                    isum += r;
                }
            if (isum != 32768) {
                // the compensation looks at the 2 x 2 taps starting at ksize/2 = 1; for a 2 x 2 kernel only tap (1, 1)
                // is inside the block and the entries behind it are still zero while the table is being filled
                const int diff = isum - 32768;
                it[3] = (short)(it[3] - diff);
            }
        }
    ready = true;
    return tab;
}

// cv::remap(src 8UC1, dst, map_x, map_y CV_32FC1, INTER_LINEAR, BORDER_CONSTANT, 0); dst has the maps' size
inline void remap_linear_u8(const uint8_t* src, int rows, int cols, size_t step, const float* map_x, const float* map_y, int drows, int dcols,
                         uint8_t* dst) {
    const short* wtab = bilinear_tab_i();
    const int width1 = cols - 1 > 0 ? cols - 1 : 0, height1 = rows - 1 > 0 ? rows - 1 : 0;
    auto sat_short = [](int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); };
    for (int dy = 0; dy < drows; ++dy)
        for (int dx = 0; dx < dcols; ++dx) {
            const int fsx = cv_round(map_x[(size_t)dy * dcols + dx] * 32.f), fsy = cv_round(map_y[(size_t)dy * dcols + dx] * 32.f);
            const short* w = wtab + (((fsy & 31) * 32) + (fsx & 31)) * 4;
            const int sx = sat_short(fsx >> 5), sy = sat_short(fsy >> 5);
            int val;
            if ((unsigned)sx < (unsigned)width1 && (unsigned)sy < (unsigned)height1) {
                const uint8_t* S = src + (size_t)sy * step + sx;
                val = S[0] * w[0] + S[1] * w[1] + S[step] * w[2] + S[step + 1] * w[3];
            } else if (sx >= cols || sx + 1 < 0 || sy >= rows || sy + 1 < 0) {
                dst[(size_t)dy * dcols + dx] = 0;
                continue;
            } else {
                const int v0 = (sx >= 0 && sy >= 0) ? src[(size_t)sy * step + sx] : 0;
                const int v1 = (sx + 1 < cols && sy >= 0) ? src[(size_t)sy * step + sx + 1] : 0;
                const int v2 = (sx >= 0 && sy + 1 < rows) ? src[(size_t)(sy + 1) * step + sx] : 0;
                const int v3 = (sx + 1 < cols && sy + 1 < rows) ? src[(size_t)(sy + 1) * step + sx + 1] : 0;
                val = v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3];
            }
            const int r = (val + (1 << 14)) >> 15;
            dst[(size_t)dy * dcols + dx] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
}


}  // namespace oracle
