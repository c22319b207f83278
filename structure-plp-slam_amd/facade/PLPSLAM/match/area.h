// Drop-in replacement of the reference header src/PLPSLAM/match/area.h (class PLPSLAM::match::area; monocular
// initialisation, module/initializer.cc:191-192): same constructor and match_in_consistent_area() signature; the search
// with its "steal if strictly closer" bookkeeping (match/area.cc:33-153) runs in libplp_front.so (plp_match_area_host).
// A template on the frame type, like match/projection.h.
#ifndef PLPSLAM_MATCH_AREA_H
#define PLPSLAM_MATCH_AREA_H

#include <cstdint>
#include <vector>

#include <opencv2/core.hpp>

#include "PLPSLAM/match/projection.h"   // match::base (the reference's or its stand-in), detail::*

namespace PLPSLAM {
namespace match {

class area final : public base {
public:
    area(const float lowe_ratio, const bool check_orientation) : base(lowe_ratio, check_orientation) {}
    ~area() final = default;

    template <class Frame>
    unsigned int match_in_consistent_area(Frame& frm_1, Frame& frm_2, std::vector<cv::Point2f>& prev_matched_pts,
                                          std::vector<int>& matched_indices_2_in_frm_1, int margin = 20) {
        static_assert(sizeof(cv::Point2f) == 8 && sizeof(frm_1.undist_keypts_[0]) == sizeof(plp_keypoint), "cv::Point2f / cv::KeyPoint layout");
        const int n1 = static_cast<int>(frm_1.undist_keypts_.size()), n2 = static_cast<int>(frm_2.undist_keypts_.size());
        matched_indices_2_in_frm_1 = std::vector<int>(static_cast<size_t>(n1), -1);
        if (n1 == 0 || n2 == 0) return 0;
        auto rows32 = [](const cv::Mat& m, int n) {
            std::vector<uint8_t> d(static_cast<size_t>(n) * 32);
            for (int i = 0; i < n; ++i) { const unsigned char* p = m.ptr<unsigned char>(i); for (int k = 0; k < 32; ++k) d[static_cast<size_t>(i) * 32 + k] = p[k]; }
            return d;
        };
        const std::vector<uint8_t> d1 = rows32(frm_1.descriptors_, n1), d2 = rows32(frm_2.descriptors_, n2);
        const plp_match_grid grid = detail::grid_of(frm_2.camera_);
        static_assert(sizeof(int) == sizeof(int32_t), "int32 result array");
        int32_t num = 0;
        detail::check(plp_match_area_host(detail::shared_matcher(), reinterpret_cast<const plp_keypoint*>(frm_1.undist_keypts_.data()), d1.data(), n1,
                                          reinterpret_cast<const plp_keypoint*>(frm_2.undist_keypts_.data()), d2.data(), n2, &grid,
                                          reinterpret_cast<float*>(prev_matched_pts.data()), margin, lowe_ratio_, check_orientation_ ? 1 : 0,
                                          matched_indices_2_in_frm_1.data(), &num));
        return static_cast<unsigned int>(num);
    }
};

}  // namespace match
}  // namespace PLPSLAM

#endif  // PLPSLAM_MATCH_AREA_H
