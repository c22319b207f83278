// Drop-in replacement of the reference header src/PLPSLAM/match/robust.h for robust::match_for_triangulation
// (match/robust.cc:43-216; mapping_module::create_new_landmarks calls it for every covisible key frame of a new key
// frame): same class, constructor, method name, arguments and return value.  The node-guided search with its epipole and
// epipolar-constraint gates (:88-150, :387-405, f64 on the bearings) runs in libplp_front.so (PLP_MATCH_MODE_TRIANGULATION);
// the host lists key frame 1's features in feature-vector order, as the reference walks them.
// robust::brute_force_match (:257-385) is PLP_MATCH_MODE_BRUTE_FORCE (all-pairs Hamming with greedy exclusivity, ratio test and
// orientation check on the device); robust::match_frame_and_keyframe (:218-255) runs it and then the REFERENCE'S OWN
// solve::essential_solver on the host, exactly as robust.cc does -- the RANSAC is geometry outside this library.
// With this header robust.cc leaves the build.  Templates on the frame / key-frame types, like match/projection.h.
#ifndef PLPSLAM_MATCH_ROBUST_H
#define PLPSLAM_MATCH_ROBUST_H

#include <cstdint>
#include <type_traits>
#include <utility>
#include <vector>

#include "PLPSLAM/match/projection.h"   // match::base (the reference's or its stand-in), detail::*
#if !defined(PLP_FACADE_NO_SOLVER_INCLUDE)
#include "PLPSLAM/solve/essential_solver.h"   // the reference's RANSAC (host side), used by match_frame_and_keyframe as in robust.cc:232-239
#endif

namespace PLPSLAM {
namespace solve { class essential_solver; }
namespace match {

class robust final : public base {
public:
    explicit robust(const float lowe_ratio, const bool check_orientation) : base(lowe_ratio, check_orientation) {}
    ~robust() final = default;

    //! robust.cc:43-216
    template <class KeyFrame>
    unsigned int match_for_triangulation(KeyFrame* keyfrm_1, KeyFrame* keyfrm_2, const Mat33_t& E_12,
                                         std::vector<std::pair<unsigned int, unsigned int>>& matched_idx_pairs) {
        const Vec3_t cam_center_1 = keyfrm_1->get_cam_center();
        const Mat33_t rot_2w = keyfrm_2->get_rotation();
        const Vec3_t trans_2w = keyfrm_2->get_translation();
        Vec3_t epiplane_in_keyfrm_2;
        keyfrm_2->camera_->reproject_to_bearing(rot_2w, trans_2w, cam_center_1, epiplane_in_keyfrm_2);
        const auto assoc_lms_in_keyfrm_1 = keyfrm_1->get_landmarks();
        const auto assoc_lms_in_keyfrm_2 = keyfrm_2->get_landmarks();
        matched_idx_pairs.clear();
        const int n = static_cast<int>(keyfrm_2->num_keypts_);
        std::vector<unsigned int> q_idx;
        std::vector<int32_t> q_group, q_level;
        std::vector<uint8_t> q_valid, q_desc;
        std::vector<float> q_angle, q_xr;
        std::vector<double> q_bearing;
        for (const auto& node : keyfrm_1->bow_feat_vec_)
            for (const auto idx_1 : node.second) {
                q_idx.push_back(idx_1);
                q_group.push_back(static_cast<int32_t>(node.first));
                q_valid.push_back(assoc_lms_in_keyfrm_1.at(idx_1) ? 0 : 1);               // only features without a landmark (:80-84)
                const auto& keypt_1 = keyfrm_1->undist_keypts_.at(idx_1);
                q_level.push_back(keypt_1.octave); q_angle.push_back(keypt_1.angle);
                q_xr.push_back(keyfrm_1->stereo_x_right_.at(idx_1));
                const Vec3_t& b = keyfrm_1->bearings_.at(idx_1);
                q_bearing.push_back(b(0)); q_bearing.push_back(b(1)); q_bearing.push_back(b(2));
                const unsigned char* p = keyfrm_1->descriptors_.template ptr<unsigned char>(static_cast<int>(idx_1));
                q_desc.insert(q_desc.end(), p, p + 32);
            }
        const int m = static_cast<int>(q_idx.size());
        if (n == 0 || m == 0) return 0;
        std::vector<int32_t> t_group(static_cast<size_t>(n), -1);
        for (const auto& node : keyfrm_2->bow_feat_vec_)
            for (const auto idx_2 : node.second) t_group.at(idx_2) = static_cast<int32_t>(node.first);
        std::vector<uint8_t> t_desc(static_cast<size_t>(n) * 32), t_has_lm(static_cast<size_t>(n));
        std::vector<float> t_angle(static_cast<size_t>(n));
        std::vector<double> t_bearing(static_cast<size_t>(n) * 3);
        for (int i = 0; i < n; ++i) {
            const unsigned char* p = keyfrm_2->descriptors_.template ptr<unsigned char>(i);
            for (int k = 0; k < 32; ++k) t_desc[static_cast<size_t>(i) * 32 + k] = p[k];
            t_has_lm[i] = assoc_lms_in_keyfrm_2.at(i) ? 1 : 0;
            t_angle[i] = keyfrm_2->undist_keypts_.at(i).angle;
            const Vec3_t& b = keyfrm_2->bearings_.at(i);
            t_bearing[3 * static_cast<size_t>(i)] = b(0); t_bearing[3 * static_cast<size_t>(i) + 1] = b(1); t_bearing[3 * static_cast<size_t>(i) + 2] = b(2);
        }
        double epipolar[12];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) epipolar[3 * r + c] = E_12(r, c);
        for (int r = 0; r < 3; ++r) epipolar[9 + r] = epiplane_in_keyfrm_2(r);
        std::vector<int32_t> out(static_cast<size_t>(n), -1);
        int32_t num = 0;
        plp_match_args a{};
        a.mode = PLP_MATCH_MODE_TRIANGULATION; a.B = 1; a.n_cap = n; a.m_cap = m;
        a.t_desc = t_desc.data(); a.t_angle = t_angle.data(); a.t_group = t_group.data(); a.t_occupied = t_has_lm.data();
        a.t_x_right = keyfrm_2->stereo_x_right_.data(); a.t_bearing = t_bearing.data();
        a.q_desc = q_desc.data(); a.q_angle = q_angle.data(); a.q_group = q_group.data(); a.q_valid = q_valid.data(); a.q_x_right = q_xr.data();
        a.q_level = q_level.data(); a.q_bearing = q_bearing.data(); a.epipolar = epipolar;
        a.lowe_ratio = lowe_ratio_; a.check_orientation = check_orientation_ ? 1 : 0;
        a.num_levels = static_cast<int32_t>(keyfrm_1->scale_factors_.size()); a.scale_factors = keyfrm_1->scale_factors_.data();
        a.out_match = out.data(); a.out_num = &num;
        detail::check(plp_match_host(detail::shared_matcher(), &a));
        // pairs (idx_1, idx_2) in ascending idx_1 (:203-213)
        std::vector<int> matched_indices_2_in_keyfrm_1(static_cast<size_t>(keyfrm_1->num_keypts_), -1);
        for (int i = 0; i < n; ++i)
            if (out[i] >= 0) matched_indices_2_in_keyfrm_1.at(q_idx[static_cast<size_t>(out[i])]) = i;
        matched_idx_pairs.reserve(static_cast<size_t>(num));
        for (unsigned int idx_1 = 0; idx_1 < matched_indices_2_in_keyfrm_1.size(); ++idx_1)
            if (matched_indices_2_in_keyfrm_1[idx_1] >= 0) matched_idx_pairs.emplace_back(idx_1, static_cast<unsigned int>(matched_indices_2_in_keyfrm_1[idx_1]));
        return static_cast<unsigned int>(num);
    }

    //! robust.cc:257-385.  matches: (idx_1 in the frame, idx_2 in the key frame), ascending idx_1
    template <class Frame, class KeyFrame>
    unsigned int brute_force_match(Frame& frm, KeyFrame* keyfrm, std::vector<std::pair<int, int>>& matches) {
        matches.clear();
        const int n1 = static_cast<int>(frm.num_keypts_), n2 = static_cast<int>(keyfrm->num_keypts_);
        if (n1 == 0 || n2 == 0) return 0;
        const auto lms_2 = keyfrm->get_landmarks();
        std::vector<uint8_t> d1(static_cast<size_t>(n1) * 32), d2(static_cast<size_t>(n2) * 32), valid2(static_cast<size_t>(n2));
        std::vector<float> a1(static_cast<size_t>(n1)), a2(static_cast<size_t>(n2));
        for (int i = 0; i < n1; ++i) {
            const unsigned char* p = frm.descriptors_.template ptr<unsigned char>(i);
            for (int k = 0; k < 32; ++k) d1[static_cast<size_t>(i) * 32 + k] = p[k];
            a1[i] = frm.keypts_.at(i).angle;                                                  // keypts_, not undist_keypts_ (:262-263, :339)
        }
        for (int i = 0; i < n2; ++i) {
            const unsigned char* p = keyfrm->descriptors_.template ptr<unsigned char>(i);
            for (int k = 0; k < 32; ++k) d2[static_cast<size_t>(i) * 32 + k] = p[k];
            a2[i] = keyfrm->keypts_.at(i).angle;
            auto lm_2 = lms_2.at(i);
            valid2[i] = (lm_2 && !lm_2->will_be_erased()) ? 1 : 0;                            // (:280-291)
        }
        std::vector<int32_t> out(static_cast<size_t>(n1), -1);
        int32_t num = 0;
        plp_match_args a{};
        a.mode = PLP_MATCH_MODE_BRUTE_FORCE; a.B = 1; a.n_cap = n1; a.m_cap = n2;
        a.t_desc = d1.data(); a.t_angle = a1.data(); a.q_desc = d2.data(); a.q_angle = a2.data(); a.q_valid = valid2.data();
        a.lowe_ratio = lowe_ratio_; a.check_orientation = check_orientation_ ? 1 : 0;
        a.out_match = out.data(); a.out_num = &num;
        detail::check(plp_match_host(detail::shared_matcher(), &a));
        matches.reserve(static_cast<size_t>(num));
        for (int idx_1 = 0; idx_1 < n1; ++idx_1)
            if (out[idx_1] >= 0) matches.emplace_back(std::make_pair(idx_1, static_cast<int>(out[idx_1])));
        return static_cast<unsigned int>(num);
    }

    //! robust.cc:218-255: brute-force matches filtered by the inliers of an essential-matrix RANSAC (the reference's own solver)
    template <class Frame, class KeyFrame, class Solver = solve::essential_solver>
    unsigned int match_frame_and_keyframe(Frame& frm, KeyFrame* keyfrm, std::vector<typename std::decay<decltype(keyfrm->get_landmarks().at(0))>::type>& matched_lms_in_frm) {
        using LandmarkPtr = typename std::decay<decltype(keyfrm->get_landmarks().at(0))>::type;
        const auto num_frm_keypts = frm.num_keypts_;
        const auto keyfrm_lms = keyfrm->get_landmarks();
        unsigned int num_inlier_matches = 0;
        matched_lms_in_frm = std::vector<LandmarkPtr>(num_frm_keypts, nullptr);
        std::vector<std::pair<int, int>> matches;
        brute_force_match(frm, keyfrm, matches);
        Solver solver(frm.bearings_, keyfrm->bearings_, matches);
        solver.find_via_ransac(50, false);
        if (!solver.solution_is_valid()) return 0;
        const auto is_inlier_matches = solver.get_inlier_matches();
        for (unsigned int i = 0; i < matches.size(); ++i) {
            if (!is_inlier_matches.at(i)) continue;
            matched_lms_in_frm.at(matches.at(i).first) = keyfrm_lms.at(matches.at(i).second);
            ++num_inlier_matches;
        }
        return num_inlier_matches;
    }
};

}  // namespace match
}  // namespace PLPSLAM

#endif  // PLPSLAM_MATCH_ROBUST_H
