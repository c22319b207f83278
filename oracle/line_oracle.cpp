// TEST INFRASTRUCTURE ONLY (oracle).  Not part of the product path.
//
// CPU restatement of the reference's line front-end
//   src/PLPSLAM/feature/line_extractor.cc:88-160           LineFeatureTracker::extract_LSD_LBD
//   src/PLPSLAM/feature/line_descriptor/LSDDetector_custom.cpp:216-320   LSDDetectorC::detect(opts)
//   src/PLPSLAM/feature/line_descriptor/binary_descriptor_custom.cpp:74-107, 217-258, 347-408,
//       533-679, 1018-1364                                  BinaryDescriptor (LBD)
// and of the OpenCV calls they delegate to (NOT under /root/reference; "OpenCV-knowledge" of
// OpenCV 3.4.16): cv::createLineSegmentDetector(...)->detect = lsd.cpp
// (LSD_REFINE_STD), cv::GaussianBlur (fixed-point u8), cv::resize(INTER_LINEAR_EXACT), cv::Sobel,
// cv::LineIterator.count, cv::remap with an identity map.
//
// PARITY UNPINNED: the reference has no test for any of this.
//
// Deliberate definitions (where the reference itself is implementation-defined):
//  (D1, closed in round 4) LSD visits seed pixels in the order produced by std::sort (unstable) on the
//       gradient bins (lsd.cpp ll_angle); the permutation among equal bins is whatever the C++ library's
//       algorithm leaves.  stable_order=0 (the default of the tests since round 4) calls std::sort as
//       lsd.cpp does, i.e. IS a reference built with this machine's libstdc++ -- the HIP path replays
//       that library's introsort on the device (PLP_SEED_ORDER_LIBSTDCXX, the library's default).
//       stable_order=1 is the order the LSD paper describes (bin descending, then row-major), the
//       library's cheaper PLP_SEED_ORDER_STABLE mode and the definition rounds 1-3 used.
//  (D2) Single-precision libm calls in the reference (cosf/sinf in region_grow and computeLBD,
//       atan2f for KeyLine::angle) are evaluated here as (float)f((double)x): glibc selects
//       FMA/non-FMA variants of the float routines at run time, so their last bit is not a
//       property of the reference; the double evaluation rounded to float is.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "cv_restated.hpp"
#include "lsd_restated.hpp"

namespace oracle {

struct KeyLine {   // field order of cv::line_descriptor::KeyLine (descriptor_custom.hpp:139-174), 68 bytes
    float angle;
    int class_id, octave;
    float pt_x, pt_y, response, size;
    float startPointX, startPointY, endPointX, endPointY;
    float sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
    float lineLength;
    int numOfPixels;
};
static_assert(sizeof(KeyLine) == 68, "KeyLine layout");

// ------------------------------------------------------------------------------------------ LBD
static const int kCombinations[32][2] = {{0, 1}, {0, 2}, {0, 3}, {0, 4}, {0, 5}, {0, 6}, {1, 2}, {1, 3}, {1, 4}, {1, 5}, {1, 6},
                                         {2, 3}, {2, 4}, {2, 5}, {2, 6}, {2, 7}, {2, 8}, {3, 4}, {3, 5}, {3, 6}, {3, 7}, {3, 8},
                                         {4, 5}, {4, 6}, {4, 7}, {4, 8}, {5, 6}, {5, 7}, {5, 8}, {6, 7}, {6, 8}, {7, 8}};
constexpr int kBands = 9, kBandWidth = 7;

struct LbdWeights {   // binary_descriptor_custom.cpp:217-258 (integer divisions kept)
    double local[kBandWidth * 3], global[kBands * kBandWidth];
    LbdWeights() {
        double u = (kBandWidth * 3 - 1) / 2;
        double sigma = (kBandWidth * 2 + 1) / 2;
        double inv = -1 / (2 * sigma * sigma);
        for (int i = 0; i < kBandWidth * 3; ++i) { const double d = i - u; local[i] = std::exp(d * d * inv); }
        u = (kBands * kBandWidth - 1) / 2;
        sigma = u;
        inv = -1 / (2 * sigma * sigma);
        for (int i = 0; i < kBands * kBandWidth; ++i) { const double d = i - u; global[i] = std::exp(d * d * inv); }
    }
};

// computeLBD for one line (binary_descriptor_custom.cpp:1100-1335); des = 72 floats
inline void lbd_one(const KeyLine& kl, const int16_t* dxImg, const int16_t* dyImg, int realWidth, int height, const LbdWeights& W,
                    float* des) {
    const short heightOfLSP = kBandWidth * kBands;
    float pgdLBand[kBands] = {0}, ngdLBand[kBands] = {0}, pgdL2Band[kBands] = {0}, ngdL2Band[kBands] = {0};
    float pgdOBand[kBands] = {0}, ngdOBand[kBands] = {0}, pgdO2Band[kBands] = {0}, ngdO2Band[kBands] = {0};
    const short imageWidth = (short)(realWidth - 1), imageHeight = (short)(height - 1);
    const short lengthOfLSP = (short)kl.numOfPixels;
    const short halfWidth = (short)((lengthOfLSP - 1) / 2), halfHeight = (short)((heightOfLSP - 1) / 2);
    const float midX = (float)(0.5 * (kl.sPointInOctaveX + kl.ePointInOctaveX));
    const float midY = (float)(0.5 * (kl.sPointInOctaveY + kl.ePointInOctaveY));
    float dL[2], dO[2];
    dL[0] = f_cos(kl.angle); dL[1] = f_sin(kl.angle);   // osl.direction = kl.angle  (D2)
    dO[0] = -dL[1]; dO[1] = dL[0];
    float sCorX0 = -dL[0] * halfWidth + dL[1] * halfHeight + midX;
    float sCorY0 = -dL[1] * halfWidth - dL[0] * halfHeight + midY;
    for (short hID = 0; hID < heightOfLSP; ++hID) {
        float sCorX = sCorX0, sCorY = sCorY0;
        float pgdLRow = 0, ngdLRow = 0, pgdORow = 0, ngdORow = 0;
        for (short wID = 0; wID < lengthOfLSP; ++wID) {
            short t = (short)std::round(sCorX);
            const short xCor = (t < 0) ? 0 : (t > imageWidth) ? imageWidth : t;
            t = (short)std::round(sCorY);
            const short yCor = (t < 0) ? 0 : (t > imageHeight) ? imageHeight : t;
            const short dx = dxImg[yCor * realWidth + xCor], dy = dyImg[yCor * realWidth + xCor];
            const float gDL = dx * dL[0] + dy * dL[1];
            const float gDO = dx * dO[0] + dy * dO[1];
            if (gDL > 0) pgdLRow += gDL; else ngdLRow -= gDL;
            if (gDO > 0) pgdORow += gDO; else ngdORow -= gDO;
            sCorX += dL[0];
            sCorY += dL[1];
        }
        sCorX0 -= dL[1];
        sCorY0 += dL[0];
        float coef = (float)W.global[hID];
        pgdLRow = coef * pgdLRow; ngdLRow = coef * ngdLRow;
        const float pgdL2Row = pgdLRow * pgdLRow, ngdL2Row = ngdLRow * ngdLRow;
        pgdORow = coef * pgdORow; ngdORow = coef * ngdORow;
        const float pgdO2Row = pgdORow * pgdORow, ngdO2Row = ngdORow * ngdORow;
        auto add = [&](short band, float c) {
            pgdLBand[band] += c * pgdLRow; ngdLBand[band] += c * ngdLRow;
            pgdL2Band[band] += c * c * pgdL2Row; ngdL2Band[band] += c * c * ngdL2Row;
            pgdOBand[band] += c * pgdORow; ngdOBand[band] += c * ngdORow;
            pgdO2Band[band] += c * c * pgdO2Row; ngdO2Band[band] += c * c * ngdO2Row;
        };
        short bandID = (short)(hID / kBandWidth);
        add(bandID, (float)W.local[hID % kBandWidth + kBandWidth]);
        bandID--;
        if (bandID >= 0) add(bandID, (float)W.local[hID % kBandWidth + 2 * kBandWidth]);
        bandID = (short)(bandID + 2);
        if (bandID < kBands) add(bandID, (float)W.local[hID % kBandWidth]);
    }
    const float invN2 = (float)(1.0 / (kBandWidth * 2.0)), invN3 = (float)(1.0 / (kBandWidth * 3.0));
    for (short b = 0; b < kBands; ++b) {
        const float invN = (b == 0 || b == kBands - 1) ? invN2 : invN3;
        const int d = b * 8;
        float temp = pgdLBand[b] * invN;
        des[d] = temp; des[d + 4] = std::sqrt(pgdL2Band[b] * invN - temp * temp);
        temp = ngdLBand[b] * invN;
        des[d + 1] = temp; des[d + 5] = std::sqrt(ngdL2Band[b] * invN - temp * temp);
        temp = pgdOBand[b] * invN;
        des[d + 2] = temp; des[d + 6] = std::sqrt(pgdO2Band[b] * invN - temp * temp);
        temp = ngdOBand[b] * invN;
        des[d + 3] = temp; des[d + 7] = std::sqrt(ngdO2Band[b] * invN - temp * temp);
    }
    float tempM = 0, tempS = 0;
    for (int b = 0; b < kBands; ++b) {
        const float* v = des + 8 * b;
        tempM += v[0] * v[0]; tempM += v[1] * v[1]; tempM += v[2] * v[2]; tempM += v[3] * v[3];
        tempS += v[4] * v[4]; tempS += v[5] * v[5]; tempS += v[6] * v[6]; tempS += v[7] * v[7];
    }
    tempM = 1 / std::sqrt(tempM);
    tempS = 1 / std::sqrt(tempS);
    for (int b = 0; b < kBands; ++b) {
        float* v = des + 8 * b;
        v[0] *= tempM; v[1] *= tempM; v[2] *= tempM; v[3] *= tempM;
        v[4] *= tempS; v[5] *= tempS; v[6] *= tempS; v[7] *= tempS;
    }
    for (int i = 0; i < kBands * 8; ++i)
        if (des[i] > 0.4) des[i] = (float)0.4;   // float compared with the double literal 0.4, as the reference
    float temp = 0;
    for (int i = 0; i < kBands * 8; ++i) temp += des[i] * des[i];
    temp = 1 / std::sqrt(temp);
    for (int i = 0; i < kBands * 8; ++i) des[i] = des[i] * temp;
}

inline void lbd_binary(const float* des, uint8_t* out32) {   // computeImpl :656-660 + binaryConversion :398-408
    for (int c = 0; c < 32; ++c) {
        const float* f1 = des + 8 * kCombinations[c][0];
        const float* f2 = des + 8 * kCombinations[c][1];
        uint8_t r = 0;
        for (int i = 0; i < 8; ++i)
            if (f1[i] > f2[i]) r = (uint8_t)(r + (1u << i));
        out32[c] = r;
    }
}

// ------------------------------------------------------------------------------------------ extract_LSD_LBD
struct LineResult {
    std::vector<KeyLine> all_lines;          // `lsd` (before the octave/length filter)
    std::vector<float> all_desc_f;           // 72 floats per line
    std::vector<uint8_t> all_desc;           // 32 bytes per line
    std::vector<KeyLine> keylsd;             // frame_keylsd
    std::vector<uint8_t> keylbd;             // frame_lbd_descr
    std::vector<double> linefn;              // 3 per kept line
    std::vector<std::array<float, 4>> raw;   // cv::LineSegmentDetector output
    Image scaled;
    std::vector<int> order_xy;
    std::vector<int16_t> dx, dy;
};

inline void check_extremes(std::array<float, 4>& e, int w, int h) {   // LSDDetector_custom.cpp:76-102
    if (e[0] < 0) e[0] = 0;
    if (e[0] >= w) e[0] = (float)w - 1.0f;
    if (e[2] < 0) e[2] = 0;
    if (e[2] >= w) e[2] = (float)w - 1.0f;
    if (e[1] < 0) e[1] = 0;
    if (e[1] >= h) e[1] = (float)h - 1.0f;
    if (e[3] < 0) e[3] = 0;
    if (e[3] >= h) e[3] = (float)h - 1.0f;
}

inline LineResult extract_lsd_lbd(const Image& img, bool stable_order) {
    LineResult R;
    // (i) cv::remap with the K*K^-1 map is the identity for a perspective camera
    const Image& img_temp = img;
    LsdOptions opts;
    const double min_length = 0.125 * std::min(img_temp.cols, img_temp.rows);   // line_extractor.cc:122, line_extractor.h:103
    Lsd lsd(opts, stable_order);
    R.raw = lsd.detect(img_temp);
    R.scaled = lsd.scaled;
    R.order_xy = lsd.order_xy;
    int class_counter = -1;
    for (auto extremes : R.raw) {   // LSDDetector_custom.cpp:262-303, octave 0, octaveScale 1
        check_extremes(extremes, img_temp.cols, img_temp.rows);
        const double length = (float)std::sqrt(std::pow((double)(extremes[0] - extremes[2]), 2) + std::pow((double)(extremes[1] - extremes[3]), 2));
        if (length > min_length) {
            KeyLine kl{};
            kl.startPointX = extremes[0] * 1.0f; kl.startPointY = extremes[1] * 1.0f;
            kl.endPointX = extremes[2] * 1.0f; kl.endPointY = extremes[3] * 1.0f;
            kl.sPointInOctaveX = extremes[0]; kl.sPointInOctaveY = extremes[1];
            kl.ePointInOctaveX = extremes[2]; kl.ePointInOctaveY = extremes[3];
            kl.lineLength = (float)length;
            // cv::LineIterator(img, Point2f, Point2f).count: 8-connected, integer end points = cvRound
            const int x1 = cv_round(extremes[0]), y1 = cv_round(extremes[1]), x2 = cv_round(extremes[2]), y2 = cv_round(extremes[3]);
            kl.numOfPixels = std::max(std::abs(x2 - x1), std::abs(y2 - y1)) + 1;
            kl.angle = (float)std::atan2((double)(kl.endPointY - kl.startPointY), (double)(kl.endPointX - kl.startPointX));   // (D2)
            kl.class_id = ++class_counter;
            kl.octave = 0;
            kl.size = (kl.endPointX - kl.startPointX) * (kl.endPointY - kl.startPointY);
            kl.response = kl.lineLength / std::max(img_temp.cols, img_temp.rows);
            kl.pt_x = (kl.endPointX + kl.startPointX) / 2;
            kl.pt_y = (kl.endPointY + kl.startPointY) / 2;
            R.all_lines.push_back(kl);
        }
    }
    if (!R.all_lines.empty()) {   // computeImpl returns early on an empty list (:550-554)
        const Image g = gaussian_blur_u8(img_temp, 5, 1.0);   // :355
        sobel3_s16(g, R.dx, R.dy);                            // :392-393
        static const LbdWeights W;
        R.all_desc_f.resize(R.all_lines.size() * 72);
        R.all_desc.resize(R.all_lines.size() * 32);
        for (size_t i = 0; i < R.all_lines.size(); ++i) {
            lbd_one(R.all_lines[i], R.dx.data(), R.dy.data(), img_temp.cols, img_temp.rows, W, &R.all_desc_f[i * 72]);
            lbd_binary(&R.all_desc_f[i * 72], &R.all_desc[i * 32]);
        }
    }
    for (size_t i = 0; i < R.all_lines.size(); ++i) {   // line_extractor.cc:134-141
        const KeyLine& k = R.all_lines[i];
        if (k.octave == 0 && k.lineLength >= 60) {
            R.keylsd.push_back(k);
            R.keylbd.insert(R.keylbd.end(), R.all_desc.begin() + i * 32, R.all_desc.begin() + (i + 1) * 32);
        }
    }
    for (const auto& k : R.keylsd) {   // :147-159, Eigen f64
        const double sx = k.startPointX, sy = k.startPointY, ex = k.endPointX, ey = k.endPointY;
        double a = sy * 1.0 - 1.0 * ey, b = 1.0 * ex - sx * 1.0, c = sx * ey - sy * ex;   // (sx,sy,1) x (ex,ey,1)
        const double nrm = std::sqrt(a * a + b * b);
        R.linefn.push_back(a / nrm); R.linefn.push_back(b / nrm); R.linefn.push_back(c / nrm);
    }
    return R;
}

}  // namespace oracle

using namespace oracle;

extern "C" {

struct OracleLineHandle { LineResult R; };

void* oracle_line_extract(const uint8_t* img, int rows, int cols, long step, int stable_order) {
    Image im(rows, cols);
    for (int y = 0; y < rows; ++y) std::memcpy(im.row(y), img + (size_t)y * step, cols);
    auto* h = new OracleLineHandle();
    h->R = extract_lsd_lbd(im, stable_order != 0);
    return h;
}
void oracle_line_free(void* h) { delete (OracleLineHandle*)h; }
int oracle_line_count(void* h, int which) {   // 0 kept, 1 all (before filter), 2 raw LSD segments
    auto& R = ((OracleLineHandle*)h)->R;
    return which == 0 ? (int)R.keylsd.size() : which == 1 ? (int)R.all_lines.size() : (int)R.raw.size();
}
void oracle_line_get(void* h, int which, KeyLine* kl, uint8_t* lbd, double* linefn, float* desc_f) {
    auto& R = ((OracleLineHandle*)h)->R;
    if (which == 0) {
        if (kl && !R.keylsd.empty()) std::memcpy(kl, R.keylsd.data(), R.keylsd.size() * sizeof(KeyLine));
        if (lbd && !R.keylbd.empty()) std::memcpy(lbd, R.keylbd.data(), R.keylbd.size());
        if (linefn && !R.linefn.empty()) std::memcpy(linefn, R.linefn.data(), R.linefn.size() * 8);
    } else {
        if (kl && !R.all_lines.empty()) std::memcpy(kl, R.all_lines.data(), R.all_lines.size() * sizeof(KeyLine));
        if (lbd && !R.all_desc.empty()) std::memcpy(lbd, R.all_desc.data(), R.all_desc.size());
        if (desc_f && !R.all_desc_f.empty()) std::memcpy(desc_f, R.all_desc_f.data(), R.all_desc_f.size() * 4);
    }
}
void oracle_line_raw(void* h, float* out4) {
    auto& R = ((OracleLineHandle*)h)->R;
    for (size_t i = 0; i < R.raw.size(); ++i) std::memcpy(out4 + 4 * i, R.raw[i].data(), 16);
}
void oracle_line_scaled_size(void* h, int* rows, int* cols) {
    auto& R = ((OracleLineHandle*)h)->R;
    *rows = R.scaled.rows; *cols = R.scaled.cols;
}
void oracle_line_scaled(void* h, uint8_t* dst) {
    auto& R = ((OracleLineHandle*)h)->R;
    std::memcpy(dst, R.scaled.data.data(), R.scaled.data.size());
}
void oracle_line_order(void* h, int* dst) {
    auto& R = ((OracleLineHandle*)h)->R;
    std::memcpy(dst, R.order_xy.data(), R.order_xy.size() * 4);
}
void oracle_line_sobel(void* h, int16_t* dx, int16_t* dy) {
    auto& R = ((OracleLineHandle*)h)->R;
    if (!R.dx.empty()) { std::memcpy(dx, R.dx.data(), R.dx.size() * 2); std::memcpy(dy, R.dy.data(), R.dy.size() * 2); }
}
void oracle_resize_linear_exact_u8(const uint8_t* src, int sh, int sw, double fx, double fy, uint8_t* dst) {
    Image s(sh, sw);
    std::memcpy(s.data.data(), src, (size_t)sh * sw);
    Image d = resize_linear_exact_u8(s, fx, fy);
    std::memcpy(dst, d.data.data(), d.data.size());
}
// LBD alone on given lines (tests of the descriptor with synthetic lines)
void oracle_lbd(const uint8_t* img, int rows, int cols, const KeyLine* kls, int n, uint8_t* out32, float* out72) {
    Image im(rows, cols);
    std::memcpy(im.data.data(), img, (size_t)rows * cols);
    const Image g = gaussian_blur_u8(im, 5, 1.0);
    std::vector<int16_t> dx, dy;
    sobel3_s16(g, dx, dy);
    static const LbdWeights W;
    std::vector<float> d(72);
    for (int i = 0; i < n; ++i) {
        lbd_one(kls[i], dx.data(), dy.data(), cols, rows, W, d.data());
        lbd_binary(d.data(), out32 + 32 * (size_t)i);
        if (out72) std::memcpy(out72 + 72 * (size_t)i, d.data(), 72 * 4);
    }
}

}  // extern "C"
