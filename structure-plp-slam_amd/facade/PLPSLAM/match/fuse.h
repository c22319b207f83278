// Drop-in replacement of the reference header src/PLPSLAM/match/fuse.h for fuse::replace_duplication (match/fuse.cc:169-325;
// mapping_module::fuse_landmark_duplication and the loop closer call it for every key frame of the covisibility
// neighbourhood): same class, constructor, method name, arguments and return value.  The host keeps the geometric
// pre-tests of the reference (:186-231: reprojection, distance range, viewing angle, predict_scale_level) and the map
// mutation (:300-324) in the reference's order; the windowed chi-square / Hamming search (:233-298) runs in
// libplp_front.so (PLP_MATCH_MODE_FUSE) for all landmarks of the call at once -- the pre-tests of a landmark do not depend on
// the mutations made for the landmarks before it as long as landmarks_to_check holds every landmark once, which is how the
// reference's callers build it (a set, or a vector filled through a set).  replace_duplication_line (:335-505) is the same with PLP_MATCH_MODE_FUSE_LINE.
// detect_duplication (:40-166) is the same search without the chi-square gates and with the signed level window
// (PLP_MATCH_MODE_FUSE with NO_CHI2 | SIGNED_LEVEL).  Templates on the key-frame / container types, like match/projection.h.
#ifndef PLPSLAM_MATCH_FUSE_H
#define PLPSLAM_MATCH_FUSE_H

#include <cmath>
#include <cstdint>
#include <vector>

#include "PLPSLAM/match/projection.h"   // match::base (the reference's or its stand-in), detail::*

namespace PLPSLAM {
namespace match {

class fuse final : public base {
public:
    explicit fuse(const float lowe_ratio = 0.6) : base(lowe_ratio, true) {}
    ~fuse() final = default;

    //! fuse.cc:40-166 (loop closure: the landmarks seen around the loop candidate against a key frame, through the Sim3)
    template <class KeyFrame, class Landmark>
    unsigned int detect_duplication(KeyFrame* keyfrm, const Mat44_t& Sim3_cw, const std::vector<Landmark*>& landmarks_to_check, const float margin,
                                    std::vector<Landmark*>& duplicated_lms_in_keyfrm) {
        const Mat33_t s_rot_cw = Sim3_cw.template block<3, 3>(0, 0);
        const auto s_cw = std::sqrt(s_rot_cw.template block<1, 3>(0, 0).dot(s_rot_cw.template block<1, 3>(0, 0)));
        const Mat33_t rot_cw = s_rot_cw / s_cw;
        const Vec3_t trans_cw = Sim3_cw.template block<3, 1>(0, 3) / s_cw;
        const Vec3_t cam_center = -rot_cw.transpose() * trans_cw;
        duplicated_lms_in_keyfrm = std::vector<Landmark*>(landmarks_to_check.size(), nullptr);
        const auto valid_lms_in_keyfrm = keyfrm->get_valid_landmarks();
        std::vector<unsigned int> q_i;
        std::vector<double> reproj_d;
        std::vector<int32_t> level;
        std::vector<uint8_t> desc;
        for (unsigned int i = 0; i < landmarks_to_check.size(); ++i) {
            auto* lm = landmarks_to_check.at(i);
            if (lm->will_be_erased()) continue;
            if (valid_lms_in_keyfrm.count(lm)) continue;
            const Vec3_t pos_w = lm->get_pos_in_world();
            Vec2_t reproj;
            float x_right;
            if (!keyfrm->camera_->reproject_to_image(rot_cw, trans_cw, pos_w, reproj, x_right)) continue;
            const Vec3_t cam_to_lm_vec = pos_w - cam_center;
            const auto cam_to_lm_dist = cam_to_lm_vec.norm();
            if (cam_to_lm_dist < lm->get_min_valid_distance() || lm->get_max_valid_distance() < cam_to_lm_dist) continue;
            const Vec3_t obs_mean_normal = lm->get_obs_mean_normal();
            if (cam_to_lm_vec.dot(obs_mean_normal) < 0.5 * cam_to_lm_dist) continue;
            const int pred_scale_level = lm->predict_scale_level(cam_to_lm_dist, keyfrm);      // SIGNED here (:109)
            q_i.push_back(i);
            reproj_d.push_back(reproj(0)); reproj_d.push_back(reproj(1));
            level.push_back(pred_scale_level);
            const auto lm_desc = lm->get_descriptor();
            const unsigned char* p = lm_desc.template ptr<unsigned char>(0);
            desc.insert(desc.end(), p, p + 32);
        }
        const int n = static_cast<int>(keyfrm->undist_keypts_.size()), m = static_cast<int>(q_i.size());
        if (n == 0 || m == 0) return 0;
        std::vector<uint8_t> t_desc(static_cast<size_t>(n) * 32);
        for (int k = 0; k < n; ++k) {
            const unsigned char* p = keyfrm->descriptors_.template ptr<unsigned char>(k);
            for (int b = 0; b < 32; ++b) t_desc[static_cast<size_t>(k) * 32 + b] = p[b];
        }
        std::vector<float> inv_sigma(keyfrm->scale_factors_.size(), 1.0f);      // unused without the chi-square gates
        std::vector<int32_t> best(static_cast<size_t>(m), -1);
        plp_match_args a{};
        a.mode = PLP_MATCH_MODE_FUSE; a.B = 1; a.n_cap = n; a.m_cap = m;
        a.flags = PLP_MATCH_FLAG_NO_CHI2 | PLP_MATCH_FLAG_SIGNED_LEVEL;
        a.t_kps = reinterpret_cast<const plp_keypoint*>(keyfrm->undist_keypts_.data()); a.t_desc = t_desc.data();
        a.q_reproj_d = reproj_d.data(); a.q_level = level.data(); a.q_desc = desc.data();
        a.margin = margin; a.lowe_ratio = lowe_ratio_;
        a.num_levels = static_cast<int32_t>(keyfrm->scale_factors_.size()); a.scale_factors = keyfrm->scale_factors_.data();
        a.inv_level_sigma_sq = inv_sigma.data();
        a.grid = detail::grid_of(keyfrm->camera_);
        a.out_query_best = best.data();
        detail::check(plp_match_host(detail::shared_matcher(), &a));
        unsigned int num_fused = 0;
        for (int q = 0; q < m; ++q) {      // :147-163, in the reference's order
            const int best_idx = best[static_cast<size_t>(q)];
            if (best_idx < 0) continue;
            const unsigned int i = q_i[static_cast<size_t>(q)];
            auto* lm = landmarks_to_check.at(i);
            auto* lm_in_keyfrm = keyfrm->get_landmark(best_idx);
            if (lm_in_keyfrm) {
                if (!lm_in_keyfrm->will_be_erased()) duplicated_lms_in_keyfrm.at(i) = lm_in_keyfrm;
            } else {
                lm->add_observation(keyfrm, best_idx);
                keyfrm->add_landmark(lm, best_idx);
            }
            ++num_fused;
        }
        return num_fused;
    }

    //! fuse.cc:169-325
    template <class KeyFrame, class T>
    unsigned int replace_duplication(KeyFrame* keyfrm, const T& landmarks_to_check, const float margin = 3.0) {
        const Mat33_t rot_cw = keyfrm->get_rotation();
        const Vec3_t trans_cw = keyfrm->get_translation();
        const Vec3_t cam_center = keyfrm->get_cam_center();
        using LandmarkPtr = typename std::decay<decltype(*landmarks_to_check.begin())>::type;
        std::vector<LandmarkPtr> lms;
        std::vector<double> reproj_d;
        std::vector<float> x_right_f;
        std::vector<int32_t> level;
        std::vector<uint8_t> desc;
        for (const auto lm : landmarks_to_check) {
            if (!lm) continue;
            if (lm->will_be_erased()) continue;
            if (lm->is_observed_in_keyframe(keyfrm)) continue;
            const Vec3_t pos_w = lm->get_pos_in_world();
            Vec2_t reproj;
            float x_right;
            if (!keyfrm->camera_->reproject_to_image(rot_cw, trans_cw, pos_w, reproj, x_right)) continue;
            const Vec3_t cam_to_lm_vec = pos_w - cam_center;
            const auto cam_to_lm_dist = cam_to_lm_vec.norm();
            if (cam_to_lm_dist < lm->get_min_valid_distance() || lm->get_max_valid_distance() < cam_to_lm_dist) continue;
            const Vec3_t obs_mean_normal = lm->get_obs_mean_normal();
            if (cam_to_lm_vec.dot(obs_mean_normal) < 0.5 * cam_to_lm_dist) continue;
            const auto pred_scale_level = lm->predict_scale_level(cam_to_lm_dist, keyfrm);
            lms.push_back(lm);
            reproj_d.push_back(reproj(0)); reproj_d.push_back(reproj(1));
            x_right_f.push_back(x_right);
            level.push_back(static_cast<int32_t>(pred_scale_level));
            const auto lm_desc = lm->get_descriptor();
            const unsigned char* p = lm_desc.template ptr<unsigned char>(0);
            desc.insert(desc.end(), p, p + 32);
        }
        const int n = static_cast<int>(keyfrm->undist_keypts_.size()), m = static_cast<int>(lms.size());
        if (n == 0 || m == 0) return 0;
        static_assert(sizeof(keyfrm->undist_keypts_[0]) == sizeof(plp_keypoint), "cv::KeyPoint must be the 28-byte POD");
        std::vector<uint8_t> t_desc(static_cast<size_t>(n) * 32);
        for (int i = 0; i < n; ++i) {
            const unsigned char* p = keyfrm->descriptors_.template ptr<unsigned char>(i);
            for (int k = 0; k < 32; ++k) t_desc[static_cast<size_t>(i) * 32 + k] = p[k];
        }
        std::vector<int32_t> best(static_cast<size_t>(m), -1);
        plp_match_args a{};
        a.mode = PLP_MATCH_MODE_FUSE; a.B = 1; a.n_cap = n; a.m_cap = m;
        a.t_kps = reinterpret_cast<const plp_keypoint*>(keyfrm->undist_keypts_.data()); a.t_desc = t_desc.data();
        a.t_x_right = keyfrm->stereo_x_right_.data();
        a.q_reproj_d = reproj_d.data(); a.q_x_right = x_right_f.data(); a.q_level = level.data(); a.q_desc = desc.data();
        a.margin = margin; a.lowe_ratio = lowe_ratio_;
        a.num_levels = static_cast<int32_t>(keyfrm->scale_factors_.size()); a.scale_factors = keyfrm->scale_factors_.data();
        a.inv_level_sigma_sq = keyfrm->inv_level_sigma_sq_.data();
        a.grid = detail::grid_of(keyfrm->camera_);
        a.out_query_best = best.data();
        detail::check(plp_match_host(detail::shared_matcher(), &a));
        // the map mutation, landmark by landmark in the reference's order (:300-324): a slot filled by an earlier landmark of
        // this very call is seen by the later ones
        unsigned int num_fused = 0;
        for (int q = 0; q < m; ++q) {
            const int best_idx = best[static_cast<size_t>(q)];
            if (best_idx < 0) continue;
            auto lm = lms[static_cast<size_t>(q)];
            auto* lm_in_keyfrm = keyfrm->get_landmark(best_idx);
            if (lm_in_keyfrm) {
                if (!lm_in_keyfrm->will_be_erased()) {
                    if (lm->num_observations() < lm_in_keyfrm->num_observations()) lm->replace(lm_in_keyfrm);
                    else lm_in_keyfrm->replace(lm);
                }
            } else {
                lm->add_observation(keyfrm, best_idx);
                keyfrm->add_landmark(lm, best_idx);
            }
            ++num_fused;
        }
        return num_fused;
    }

    //! fuse.cc:335-505 (3D lines against the key lines of a key frame)
    template <class KeyFrame, class T>
    unsigned int replace_duplication_line(KeyFrame* keyfrm, const T& landmarks_to_check, const float margin = 3.0) {
        const Mat33_t rot_cw = keyfrm->get_rotation();
        const Vec3_t trans_cw = keyfrm->get_translation();
        const Vec3_t cam_center = keyfrm->get_cam_center();
        using LinePtr = typename std::decay<decltype(*landmarks_to_check.begin())>::type;
        std::vector<LinePtr> lms;
        std::vector<double> sp_d, ep_d;
        std::vector<int32_t> level;
        std::vector<uint8_t> desc;
        for (const auto lm : landmarks_to_check) {
            if (!lm) continue;
            if (lm->will_be_erased()) continue;
            if (lm->is_observed_in_keyframe(keyfrm)) continue;
            const Vec6_t pos_w = lm->get_pos_in_world();
            const Vec3_t pos_w_sp = pos_w.head(3), pos_w_ep = pos_w.tail(3);
            Vec2_t reproj_sp, reproj_ep;
            float x_right_sp, x_right_ep;
            const bool in_image_sp = keyfrm->camera_->reproject_to_image(rot_cw, trans_cw, pos_w_sp, reproj_sp, x_right_sp);
            const bool in_image_ep = keyfrm->camera_->reproject_to_image(rot_cw, trans_cw, pos_w_ep, reproj_ep, x_right_ep);
            if (!in_image_sp && !in_image_ep) continue;
            if (!in_image_sp || !in_image_ep) {
                const Vec3_t pos_w_mp = 0.5 * (pos_w_sp + pos_w_ep);
                Vec2_t reproj_mp;
                float x_right_mp;
                if (!keyfrm->camera_->reproject_to_image(rot_cw, trans_cw, pos_w_mp, reproj_mp, x_right_mp)) continue;
            }
            const Vec3_t cam_to_lm_vec_sp = pos_w_sp - cam_center, cam_to_lm_vec_ep = pos_w_ep - cam_center;
            const auto cam_to_lm_dist_sp = cam_to_lm_vec_sp.norm(), cam_to_lm_dist_ep = cam_to_lm_vec_ep.norm();
            const auto max_cam_to_lm_dist = lm->get_max_valid_distance();
            const auto min_cam_to_lm_dist = lm->get_min_valid_distance();
            if (cam_to_lm_dist_sp < min_cam_to_lm_dist || max_cam_to_lm_dist < cam_to_lm_dist_sp || cam_to_lm_dist_ep < min_cam_to_lm_dist ||
                max_cam_to_lm_dist < cam_to_lm_dist_ep)
                continue;
            const Vec3_t cam_to_lm_vec_mp = 0.5 * (pos_w_sp + pos_w_ep) - cam_center;
            const auto pred_scale_level = lm->predict_scale_level(cam_to_lm_vec_mp.norm(), keyfrm->_log_scale_factor_lsd, keyfrm->_num_scale_levels_lsd);
            lms.push_back(lm);
            sp_d.push_back(reproj_sp(0)); sp_d.push_back(reproj_sp(1)); ep_d.push_back(reproj_ep(0)); ep_d.push_back(reproj_ep(1));
            level.push_back(static_cast<int32_t>(pred_scale_level));
            const auto lm_desc = lm->get_descriptor();
            const unsigned char* p = lm_desc.template ptr<unsigned char>(0);
            desc.insert(desc.end(), p, p + 32);
        }
        const int n = static_cast<int>(keyfrm->_keylsd.size()), m = static_cast<int>(lms.size());
        if (n == 0 || m == 0) return 0;
        static_assert(sizeof(keyfrm->_keylsd[0]) == sizeof(plp_keyline), "KeyLine must be the 68-byte record of descriptor_custom.hpp");
        std::vector<uint8_t> t_desc(static_cast<size_t>(n) * 32);
        for (int i = 0; i < n; ++i) {
            const unsigned char* p = keyfrm->_lbd_descr.template ptr<unsigned char>(i);
            for (int k = 0; k < 32; ++k) t_desc[static_cast<size_t>(i) * 32 + k] = p[k];
        }
        std::vector<int32_t> best(static_cast<size_t>(m), -1);
        plp_match_args a{};
        a.mode = PLP_MATCH_MODE_FUSE_LINE; a.B = 1; a.n_cap = n; a.m_cap = m;
        a.t_kl = reinterpret_cast<const plp_keyline*>(keyfrm->_keylsd.data()); a.t_desc = t_desc.data();
        a.q_reproj_d = sp_d.data(); a.q_reproj2_d = ep_d.data(); a.q_level = level.data(); a.q_desc = desc.data();
        a.margin = margin; a.lowe_ratio = lowe_ratio_;
        a.num_levels = static_cast<int32_t>(keyfrm->_scale_factors_lsd.size()); a.scale_factors = keyfrm->_scale_factors_lsd.data();
        a.inv_level_sigma_sq = keyfrm->_inv_level_sigma_sq_lsd.data();
        a.grid = detail::grid_of(keyfrm->camera_);
        a.out_query_best = best.data();
        detail::check(plp_match_host(detail::shared_matcher(), &a));
        unsigned int num_fused = 0;
        for (int q = 0; q < m; ++q) {      // :478-501, in the reference's order
            const int best_idx = best[static_cast<size_t>(q)];
            if (best_idx < 0) continue;
            auto lm = lms[static_cast<size_t>(q)];
            auto* lm_in_keyfrm = keyfrm->get_landmark_line(best_idx);
            if (lm_in_keyfrm) {
                if (!lm_in_keyfrm->will_be_erased()) {
                    if (lm->num_observations() < lm_in_keyfrm->num_observations()) lm->replace(lm_in_keyfrm);
                    else lm_in_keyfrm->replace(lm);
                }
            } else {
                lm->add_observation(keyfrm, best_idx);
                keyfrm->add_landmark_line(lm, best_idx);
            }
            ++num_fused;
        }
        return num_fused;
    }
};

}  // namespace match
}  // namespace PLPSLAM

#endif  // PLPSLAM_MATCH_FUSE_H
