// Drop-in replacement of the reference header src/PLPSLAM/feature/line_extractor.h (class
// PLPSLAM::feature::LineFeatureTracker, :61-104) over the C ABI of libplp_front.so.
// data/frame.cc:1143-1167 compiles against it unchanged.
#ifndef PLPSLAM_FEATURE_LINE_EXTRACTOR_H
#define PLPSLAM_FEATURE_LINE_EXTRACTOR_H

#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include <opencv2/core.hpp>

#include "PLPSLAM/camera/base.h"
#include "PLPSLAM/feature/line_descriptor/line_descriptor_custom.hpp"   // cv::line_descriptor::KeyLine
#include "PLPSLAM/type.h"                                               // Vec3_t
#include "plp_front.h"

namespace PLPSLAM {
namespace feature {

class LineFeatureTracker {
public:
    explicit LineFeatureTracker(camera::base* camera) : _camera(camera) {
        // the camera only parameterised the identity remap of line_extractor.cc:40-86,103 (elided)
        const char* e = std::getenv("PLP_DEVICE");
        check(plp_line_create(e ? std::atoi(e) : 0, &ctx_));
        // the library's default is the reference's seed order (std::sort as libstdc++ runs it); PLP_SEED_ORDER=stable selects the cheaper one
        const char* so = std::getenv("PLP_SEED_ORDER");
        if (so && std::string(so) == "stable") check(plp_line_set_seed_order(ctx_, PLP_SEED_ORDER_STABLE));
        _scale_factors.assign(1, 1.0f); _inv_scale_factors.assign(1, 1.0f);
        _level_sigma_sq.assign(1, 1.0f); _inv_level_sigma_sq.assign(1, 1.0f);
    }
    ~LineFeatureTracker() { plp_line_destroy(ctx_); }

    void extract_LSD_LBD(const cv::Mat& img, std::vector<cv::line_descriptor::KeyLine>& frame_keylsd, cv::Mat& frame_lbd_descr,
                         std::vector<Vec3_t>& keyline_functions) {
        static_assert(sizeof(cv::line_descriptor::KeyLine) == sizeof(plp_keyline), "KeyLine must be the 68-byte POD");
        constexpr int cap = 2048;
        frame_keylsd.resize(cap);
        cv::Mat lbd(cap, 32, CV_8U);
        std::vector<double> fn(3 * cap);
        int32_t n = 0;
        check(plp_line_extract(ctx_, img.data, img.rows, img.cols, img.step, reinterpret_cast<plp_keyline*>(frame_keylsd.data()), lbd.data,
                               fn.data(), cap, &n));
        frame_keylsd.resize(n);
        frame_lbd_descr = n ? lbd.rowRange(0, n).clone() : cv::Mat();
        for (int i = 0; i < n; ++i) keyline_functions.push_back(Vec3_t(fn[3 * i], fn[3 * i + 1], fn[3 * i + 2]));   // appended, as :158
    }

    unsigned int get_num_scale_levels() const { return 1; }          // _num_levels (line_extractor.cc:33-35)
    float get_scale_factor() const { return 2.0f; }                  // _scale_factor
    std::vector<float> get_scale_factors() const { return _scale_factors; }
    std::vector<float> get_inv_scale_factors() const { return _inv_scale_factors; }
    std::vector<float> get_level_sigma_sq() const { return _level_sigma_sq; }
    std::vector<float> get_inv_level_sigma_sq() const { return _inv_level_sigma_sq; }

private:
    static void check(plp_status s) {
        if (s != PLP_OK) throw std::runtime_error(std::string("plp_front: ") + plp_strerror(s) + ": " + plp_last_error());
    }
    camera::base* _camera;
    plp_line* ctx_ = nullptr;
    std::vector<float> _scale_factors, _inv_scale_factors, _level_sigma_sq, _inv_level_sigma_sq;
};

}  // namespace feature
}  // namespace PLPSLAM

#endif  // PLPSLAM_FEATURE_LINE_EXTRACTOR_H
