// TEST INFRASTRUCTURE ONLY (oracle).  Not part of the product path.
//
// cv::createLineSegmentDetector(...)->detect (OpenCV modules/imgproc/src/lsd.cpp, LSD_REFINE_STD) and
// cv::resize(INTER_LINEAR_EXACT) restated for u8 input: third-party arithmetic that is NOT under /root/reference
// ("OpenCV-knowledge" of OpenCV 3.4.16).  PARITY UNPINNED (no OpenCV in the build image); the dump
// tool tools/opencv_crosscheck.cpp produces the file that pins it wherever OpenCV 3.4.16 is installed.
// One header so that the oracle (line_oracle.cpp) and the stand-in OpenCV of oracle/_ref (ref_shim/cvshim.hpp), under
// which the reference's own LSDDetector_custom.cpp / line_extractor.cc are compiled, use the same restatement.
// Deliberate definitions D1 (seed order) and D2 (single-precision libm) are described in line_oracle.cpp.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <vector>

#include "cv_restated.hpp"

namespace oracle {

#ifdef ORACLE_D2_LIBM_FLOAT
// the reference's literal calls (cosf / sinf of this machine's glibc): liboracle_d2.so, built only to MEASURE definition D2 (tests/test_d2_libm_float.py)
inline float f_cos(float x) { return cosf(x); }
inline float f_sin(float x) { return sinf(x); }
#else
inline float f_cos(float x) { return (float)std::cos((double)x); }   // (D2)
inline float f_sin(float x) { return (float)std::sin((double)x); }
#endif

// cv::resize(src, dst, Size(), 0.5, 0.5, INTER_LINEAR_EXACT) for u8 (bit-exact 8.8 coefficients)
inline Image resize_linear_exact_u8(const Image& src, double fx, double fy) {
    const int dw = cv_round(src.cols * fx), dh = cv_round(src.rows * fy);
    Image dst(dh, dw);
    auto coeffs = [](int ssize, int dsize, std::vector<int>& ofs, std::vector<int>& c1) {
        const double inv_scale = (double)dsize / ssize;
        const double scale = 1.0 / inv_scale;
        ofs.resize(dsize); c1.resize(dsize);
        for (int d = 0; d < dsize; ++d) {
            const double val = ((double)d + 0.5) * scale - 0.5;
            const int iv = cv_floor(val);
            if (iv >= 0 && ssize > 1) {
                if (iv < ssize - 1) { ofs[d] = iv; c1[d] = cv_round((val - (double)iv) * 256.0); }
                else { ofs[d] = ssize - 1; c1[d] = -1; }   // right/bottom border: last sample
            } else { ofs[d] = 0; c1[d] = -2; }              // left/top border: first sample
        }
    };
    std::vector<int> xo, xc, yo, yc;
    coeffs(src.cols, dw, xo, xc);
    coeffs(src.rows, dh, yo, yc);
    auto hval = [&](const uint8_t* S, int d) -> uint32_t {
        if (xc[d] < 0) return (uint32_t)S[xo[d]] * 256u;
        return (uint32_t)S[xo[d]] * (uint32_t)(256 - xc[d]) + (uint32_t)S[xo[d] + 1] * (uint32_t)xc[d];
    };
    for (int y = 0; y < dh; ++y) {
        const uint8_t* S0 = src.row(yo[y]);
        const uint8_t* S1 = yc[y] < 0 ? S0 : src.row(yo[y] + 1);
        const uint32_t b1 = yc[y] < 0 ? 0u : (uint32_t)yc[y], b0 = 256u - b1;
        for (int x = 0; x < dw; ++x) {
            const uint32_t v = b0 * hval(S0, x) + b1 * hval(S1, x);
            dst.row(y)[x] = (uint8_t)std::min<uint32_t>((v + 32768u) >> 16, 255u);
        }
    }
    return dst;
}

// ------------------------------------------------------------------------------------------ LSD (lsd.cpp restated)
struct LsdOptions {   // line_extractor.cc:113-122
    int refine = 1;
    double scale = 0.5, sigma_scale = 0.6, quant = 2.0, ang_th = 22.5, log_eps = 1.0, density_th = 0.6;
    int n_bins = 1024;
};

class Lsd {
public:
    explicit Lsd(const LsdOptions& o, bool stable) : o_(o), stable_(stable) {}
    std::vector<std::array<float, 4>> detect(const Image& image);
    Image scaled;          // kept for stage parity
    std::vector<int> order_xy;   // seed order (y * w + x)

private:
    static constexpr double NOTDEF = -1024.0;
    static constexpr double M_3_2_PI_ = (3 * M_PI) / 2, M_2__PI_ = 2 * M_PI, DEG_TO_RADS = M_PI / 180;
    struct RegionPoint { int x, y; double angle, modgrad; };
    struct Rect { double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p; };

    void ll_angle(double threshold, unsigned n_bins);
    void region_grow(int sx, int sy, std::vector<RegionPoint>& reg, double& reg_angle, double prec);
    void region2rect(const std::vector<RegionPoint>& reg, double reg_angle, double prec, double p, Rect& rec) const;
    double get_theta(const std::vector<RegionPoint>& reg, double x, double y, double reg_angle, double prec) const;
    bool refine(std::vector<RegionPoint>& reg, double reg_angle, double prec, double p, Rect& rec, double density_th);
    bool reduce_region_radius(std::vector<RegionPoint>& reg, double reg_angle, double prec, double p, Rect& rec, double density,
                              double density_th);
    bool is_aligned(int x, int y, double theta, double prec) const {
        if (x < 0 || y < 0 || x >= w_ || y >= h_) return false;
        const double a = angles_[(size_t)y * w_ + x];
        if (a == NOTDEF) return false;
        double n_theta = theta - a;
        if (n_theta < 0) n_theta = -n_theta;
        if (n_theta > M_3_2_PI_) {
            n_theta -= M_2__PI_;
            if (n_theta < 0) n_theta = -n_theta;
        }
        return n_theta <= prec;
    }
    static double dist_sq(double x1, double y1, double x2, double y2) { return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1); }
    static double dist(double x1, double y1, double x2, double y2) { return std::sqrt(dist_sq(x1, y1, x2, y2)); }
    static double angle_diff_signed(double a, double b) {
        double diff = a - b;
        while (diff <= -M_PI) diff += M_2__PI_;
        while (diff > M_PI) diff -= M_2__PI_;
        return diff;
    }
    static double angle_diff(double a, double b) { return std::fabs(angle_diff_signed(a, b)); }

    LsdOptions o_;
    bool stable_;
    int w_ = 0, h_ = 0;
    std::vector<double> angles_, modgrad_;
    std::vector<uint8_t> used_;
    struct NormPoint { int x, y, norm; };
    std::vector<NormPoint> ordered_;
};

inline void Lsd::ll_angle(double threshold, unsigned n_bins) {
    w_ = scaled.cols; h_ = scaled.rows;
    angles_.assign((size_t)w_ * h_, NOTDEF);
    modgrad_.assign((size_t)w_ * h_, 0.0);   // last row/column are never read before being written in OpenCV either
    double max_grad = -1;
    for (int y = 0; y < h_ - 1; ++y) {
        const uint8_t* r0 = scaled.row(y);
        const uint8_t* r1 = scaled.row(y + 1);
        for (int x = 0; x < w_ - 1; ++x) {
            const int DA = r1[x + 1] - r0[x];
            const int BC = r0[x + 1] - r1[x];
            const int gx = DA + BC, gy = DA - BC;
            const double norm = std::sqrt((gx * gx + gy * gy) / 4.0);
            modgrad_[(size_t)y * w_ + x] = norm;
            if (norm <= threshold) angles_[(size_t)y * w_ + x] = NOTDEF;
            else {
                angles_[(size_t)y * w_ + x] = fast_atan2f_deg((float)gx, (float)-gy) * DEG_TO_RADS;
                if (norm > max_grad) max_grad = norm;
            }
        }
    }
    const double bin_coef = (max_grad > 0) ? double(n_bins - 1) / max_grad : 0;
    ordered_.clear();
    ordered_.reserve((size_t)w_ * h_);
    for (int y = 0; y < h_ - 1; ++y)
        for (int x = 0; x < w_ - 1; ++x) ordered_.push_back({x, y, int(modgrad_[(size_t)y * w_ + x] * bin_coef)});
    auto cmp = [](const NormPoint& a, const NormPoint& b) { return a.norm > b.norm; };
    if (stable_) std::stable_sort(ordered_.begin(), ordered_.end(), cmp);   // (D1)
    else std::sort(ordered_.begin(), ordered_.end(), cmp);
}

inline void Lsd::region_grow(int sx, int sy, std::vector<RegionPoint>& reg, double& reg_angle, double prec) {
    reg.clear();
    reg_angle = angles_[(size_t)sy * w_ + sx];
    reg.push_back({sx, sy, reg_angle, modgrad_[(size_t)sy * w_ + sx]});
    float sumdx = float(std::cos(reg_angle));
    float sumdy = float(std::sin(reg_angle));
    used_[(size_t)sy * w_ + sx] = 1;
    for (size_t i = 0; i < reg.size(); ++i) {
        const int px = reg[i].x, py = reg[i].y;
        const int xx_min = std::max(px - 1, 0), xx_max = std::min(px + 1, w_ - 1);
        const int yy_min = std::max(py - 1, 0), yy_max = std::min(py + 1, h_ - 1);
        for (int yy = yy_min; yy <= yy_max; ++yy)
            for (int xx = xx_min; xx <= xx_max; ++xx) {
                uint8_t& is_used = used_[(size_t)yy * w_ + xx];
                if (is_used != 1 && is_aligned(xx, yy, reg_angle, prec)) {
                    const double angle = angles_[(size_t)yy * w_ + xx];
                    is_used = 1;
                    reg.push_back({xx, yy, angle, modgrad_[(size_t)yy * w_ + xx]});
                    sumdx += f_cos(float(angle));
                    sumdy += f_sin(float(angle));
                    reg_angle = fast_atan2f_deg(sumdy, sumdx) * DEG_TO_RADS;
                }
            }
    }
}

inline double Lsd::get_theta(const std::vector<RegionPoint>& reg, double x, double y, double reg_angle, double prec) const {
    double Ixx = 0.0, Iyy = 0.0, Ixy = 0.0;
    for (const auto& r : reg) {
        const double dx = (double)r.x - x, dy = (double)r.y - y, weight = r.modgrad;
        Ixx += dy * dy * weight;
        Iyy += dx * dx * weight;
        Ixy -= dx * dy * weight;
    }
    const double lambda = 0.5 * (Ixx + Iyy - std::sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
    double theta = (std::fabs(Ixx) > std::fabs(Iyy)) ? double(fast_atan2f_deg(float(lambda - Ixx), float(Ixy)))
                                                     : double(fast_atan2f_deg(float(Ixy), float(lambda - Iyy)));
    theta *= DEG_TO_RADS;
    if (angle_diff(theta, reg_angle) > prec) theta += M_PI;
    return theta;
}

inline void Lsd::region2rect(const std::vector<RegionPoint>& reg, double reg_angle, double prec, double p, Rect& rec) const {
    double x = 0, y = 0, sum = 0;
    for (const auto& r : reg) {
        x += double(r.x) * r.modgrad;
        y += double(r.y) * r.modgrad;
        sum += r.modgrad;
    }
    x /= sum;
    y /= sum;
    const double theta = get_theta(reg, x, y, reg_angle, prec);
    const double dx = std::cos(theta), dy = std::sin(theta);
    double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
    for (const auto& r : reg) {
        const double regdx = double(r.x) - x, regdy = double(r.y) - y;
        const double l = regdx * dx + regdy * dy;
        const double w = -regdx * dy + regdy * dx;
        if (l > l_max) l_max = l; else if (l < l_min) l_min = l;
        if (w > w_max) w_max = w; else if (w < w_min) w_min = w;
    }
    rec.x1 = x + l_min * dx; rec.y1 = y + l_min * dy;
    rec.x2 = x + l_max * dx; rec.y2 = y + l_max * dy;
    rec.width = w_max - w_min;
    rec.x = x; rec.y = y; rec.theta = theta; rec.dx = dx; rec.dy = dy; rec.prec = prec; rec.p = p;
    if (rec.width < 1.0) rec.width = 1.0;
}

inline bool Lsd::reduce_region_radius(std::vector<RegionPoint>& reg, double reg_angle, double prec, double p, Rect& rec, double density,
                               double density_th) {
    const double xc = double(reg[0].x), yc = double(reg[0].y);
    const double radSq1 = dist_sq(xc, yc, rec.x1, rec.y1), radSq2 = dist_sq(xc, yc, rec.x2, rec.y2);
    double radSq = radSq1 > radSq2 ? radSq1 : radSq2;
    while (density < density_th) {
        radSq *= 0.75 * 0.75;
        for (size_t i = 0; i < reg.size(); ++i) {
            if (dist_sq(xc, yc, double(reg[i].x), double(reg[i].y)) > radSq) {
                used_[(size_t)reg[i].y * w_ + reg[i].x] = 0;
                std::swap(reg[i], reg[reg.size() - 1]);
                reg.pop_back();
                --i;
            }
        }
        if (reg.size() < 2) return false;
        region2rect(reg, reg_angle, prec, p, rec);
        density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    }
    return true;
}

inline bool Lsd::refine(std::vector<RegionPoint>& reg, double reg_angle, double prec, double p, Rect& rec, double density_th) {
    double density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    if (density >= density_th) return true;
    const double xc = double(reg[0].x), yc = double(reg[0].y);
    const double ang_c = reg[0].angle;
    double sum = 0, s_sum = 0;
    int n = 0;
    for (auto& r : reg) {
        used_[(size_t)r.y * w_ + r.x] = 0;
        if (dist(xc, yc, r.x, r.y) < rec.width) {
            const double ang_d = angle_diff_signed(r.angle, ang_c);
            sum += ang_d;
            s_sum += ang_d * ang_d;
            ++n;
        }
    }
    const double mean_angle = sum / double(n);
    const double tau = 2.0 * std::sqrt((s_sum - 2.0 * mean_angle * sum) / double(n) + mean_angle * mean_angle);
    const int sx = reg[0].x, sy = reg[0].y;
    region_grow(sx, sy, reg, reg_angle, tau);
    if (reg.size() < 2) return false;
    region2rect(reg, reg_angle, prec, p, rec);
    density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    if (density < density_th) return reduce_region_radius(reg, reg_angle, prec, p, rec, density, density_th);
    return true;
}

inline std::vector<std::array<float, 4>> Lsd::detect(const Image& image) {
    std::vector<std::array<float, 4>> lines;
    const double prec = M_PI * o_.ang_th / 180;
    const double p = o_.ang_th / 180;
    const double rho = o_.quant / std::sin(prec);
    if (o_.scale != 1) {
        const double sigma = (o_.scale < 1) ? (o_.sigma_scale / o_.scale) : o_.sigma_scale;
        const double sprec = 3;
        const unsigned h = (unsigned)(std::ceil(sigma * std::sqrt(2 * sprec * std::log(10.0))));
        Image g = gaussian_blur_u8(image, 1 + 2 * (int)h, sigma);
        scaled = resize_linear_exact_u8(g, o_.scale, o_.scale);
    } else scaled = image;
    ll_angle(rho, (unsigned)o_.n_bins);
    const double LOG_NT = 5 * (std::log10(double(w_)) + std::log10(double(h_))) / 2 + std::log10(11.0);
    const size_t min_reg_size = size_t(-LOG_NT / std::log10(p));
    used_.assign((size_t)w_ * h_, 0);
    std::vector<RegionPoint> reg;
    order_xy.clear();
    for (const auto& op : ordered_) {
        order_xy.push_back(op.y * w_ + op.x);
        if (used_[(size_t)op.y * w_ + op.x] == 0 && angles_[(size_t)op.y * w_ + op.x] != NOTDEF) {
            double reg_angle;
            region_grow(op.x, op.y, reg, reg_angle, prec);
            if (reg.size() < min_reg_size) continue;
            Rect rec;
            region2rect(reg, reg_angle, prec, p, rec);
            if (o_.refine > 0 && !refine(reg, reg_angle, prec, p, rec, o_.density_th)) continue;
            rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
            if (o_.scale != 1) { rec.x1 /= o_.scale; rec.y1 /= o_.scale; rec.x2 /= o_.scale; rec.y2 /= o_.scale; rec.width /= o_.scale; }
            lines.push_back({float(rec.x1), float(rec.y1), float(rec.x2), float(rec.y2)});
        }
    }
    return lines;
}

}  // namespace oracle
