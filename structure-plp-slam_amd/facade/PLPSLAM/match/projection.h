// Drop-in replacement of the reference header src/PLPSLAM/match/projection.h for the calls the tracker makes every
// frame: projection::match_frame_and_landmarks[_line] (projection.cc:37-121, 124-212, called from
// tracking_module::search_local_landmarks) and projection::match_current_and_last_frames[_line] (projection.cc:214-358,
// 361-527, called from frame_tracker::motion_based_track).  Same class name, constructor, method names, argument meaning
// and return value; the search itself runs in libplp_front.so (PLP_MATCH_MODE_LANDMARKS[_LINE] /
// PLP_MATCH_MODE_LAST_FRAME[_LINE] through plp_match_host).
//
// The methods are templates on the frame / landmark types: inside the reference tree they are instantiated with
// data::frame and data::landmark (the call sites compile unchanged), and they only touch the members the reference's own
// loops touch -- undist_keypts_, descriptors_, stereo_x_right_, landmarks_, scale_factors_, camera_ (grid, setup_type_,
// true_baseline_, reproject_to_image), cam_pose_cw_, keypts_, outlier_flags_; is_observable_in_tracking_, will_be_erased(),
// scale_level_in_tracking_, reproj_in_tracking_, x_right_in_tracking_, get_descriptor(), has_observation(),
// get_pos_in_world().  What stays on the host is exactly what the reference computes per landmark before the descriptor
// search: the skip tests and the reprojection.  The map mutation (frm.landmarks_.at(idx) = lm) is applied from the
// matcher's out_match array, which already holds the sequential loop's "last writer wins" result.
#ifndef PLPSLAM_MATCH_PROJECTION_H
#define PLPSLAM_MATCH_PROJECTION_H

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include <set>

#include "PLPSLAM/type.h"
#if defined(__has_include)
#if __has_include("PLPSLAM/match/base.h")
#include "PLPSLAM/match/base.h"     // the reference's own base class and Hamming constants (match/base.h:12-84)
#define PLP_FACADE_HAS_MATCH_BASE 1
#endif
#endif
#include "plp_front.h"

#ifndef PLP_FACADE_HAS_MATCH_BASE
namespace PLPSLAM {
namespace match {
class base {   // stand-in with the members of match/base.h:70-82, for builds outside the reference tree
public:
    base(const float lowe_ratio, const bool check_orientation) : lowe_ratio_(lowe_ratio), check_orientation_(check_orientation) {}
    virtual ~base() = default;

protected:
    const float lowe_ratio_;
    const bool check_orientation_;
};
}  // namespace match
}  // namespace PLPSLAM
#endif

namespace PLPSLAM {
namespace data {
class frame;
class keyframe;
class landmark;
class Line;
}  // namespace data
}  // namespace PLPSLAM

namespace PLPSLAM {
namespace match {

namespace detail {

// match::projection is value-constructed at every call site (tracking_module.cc, frame_tracker.cc): the device context
// behind it is shared per thread instead of being created per object.
inline plp_matcher* shared_matcher() {
    struct holder {
        plp_matcher* h = nullptr;
        ~holder() { if (h) plp_matcher_destroy(h); }
    };
    thread_local holder H;
    if (!H.h) {
        const char* e = std::getenv("PLP_DEVICE");
        if (plp_matcher_create(e ? std::atoi(e) : 0, &H.h) != PLP_OK) throw std::runtime_error(std::string("plp_matcher_create: ") + plp_last_error());
    }
    return H.h;
}

inline void check(plp_status s) {
    if (s != PLP_OK) throw std::runtime_error(std::string("plp_front: ") + plp_last_error());
}

template <class Camera>
plp_match_grid grid_of(const Camera* cam) {   // camera::base, camera/base.h:147-160
    plp_match_grid g;
    g.min_x = cam->img_bounds_.min_x_; g.min_y = cam->img_bounds_.min_y_;
    g.inv_cell_width = cam->inv_cell_width_; g.inv_cell_height = cam->inv_cell_height_;
    g.cols = static_cast<int32_t>(cam->num_grid_cols_); g.rows = static_cast<int32_t>(cam->num_grid_rows_);
    return g;
}

// the current frame as targets: undist_keypts_, descriptors_, stereo_x_right_, "has a landmark with observations"
template <class Frame>
struct frame_targets {
    std::vector<uint8_t> desc, occupied;
    const plp_keypoint* kps;
    int n;
    explicit frame_targets(const Frame& frm) {
        n = static_cast<int>(frm.undist_keypts_.size());
        static_assert(sizeof(frm.undist_keypts_[0]) == sizeof(plp_keypoint), "cv::KeyPoint must be the 28-byte POD");
        kps = reinterpret_cast<const plp_keypoint*>(frm.undist_keypts_.data());
        desc.resize(static_cast<size_t>(n) * 32);
        occupied.resize(static_cast<size_t>(n));
        for (int i = 0; i < n; ++i) {
            const unsigned char* row = frm.descriptors_.template ptr<unsigned char>(i);
            for (int k = 0; k < 32; ++k) desc[static_cast<size_t>(i) * 32 + k] = row[k];
            const auto* lm = frm.landmarks_.at(i);
            occupied[i] = (lm && lm->has_observation()) ? 1 : 0;
        }
    }
};

// the current frame's key lines as targets: _keylsd, _lbd_descr, "has a line landmark with observations"
template <class Frame>
struct frame_line_targets {
    std::vector<uint8_t> desc, occupied;
    const plp_keyline* kl;
    int n;
    explicit frame_line_targets(const Frame& frm) {
        n = static_cast<int>(frm._keylsd.size());
        static_assert(sizeof(frm._keylsd[0]) == sizeof(plp_keyline), "KeyLine must be the 68-byte record of descriptor_custom.hpp");
        kl = reinterpret_cast<const plp_keyline*>(frm._keylsd.data());
        desc.resize(static_cast<size_t>(n) * 32);
        occupied.resize(static_cast<size_t>(n));
        for (int i = 0; i < n; ++i) {
            const unsigned char* row = frm._lbd_descr.template ptr<unsigned char>(i);
            for (int k = 0; k < 32; ++k) desc[static_cast<size_t>(i) * 32 + k] = row[k];
            const auto* lm = frm._landmarks_line.at(i);
            occupied[i] = (lm && lm->has_observation()) ? 1 : 0;
        }
    }
};

}  // namespace detail

class projection final : public base {
public:
    explicit projection(const float lowe_ratio = 0.6, const bool check_orientation = true) : base(lowe_ratio, check_orientation) {}
    ~projection() final = default;

    // match_keyframes_mutually keeps its reference declaration and its body in the reference's match/projection.cc (:894-1142;
    // everything above it is deleted from that file).  Their searches are available through the
    // C ABI as well, but they are not on the per-frame path.
    //! projection.cc:529-645 (relocalisation: the key frame's landmarks reprojected with the pose PnP found)
    template <class Frame, class KeyFrame, class Landmark>
    unsigned int match_frame_and_keyframe(Frame& curr_frm, KeyFrame* keyfrm, const std::set<Landmark*>& already_matched_lms, const float margin,
                                          const unsigned int hamm_dist_thr) const {
        const Mat33_t rot_cw = curr_frm.cam_pose_cw_.template block<3, 3>(0, 0);
        const Vec3_t trans_cw = curr_frm.cam_pose_cw_.template block<3, 1>(0, 3);
        const Vec3_t cam_center = -rot_cw.transpose() * trans_cw;
        const auto landmarks = keyfrm->get_landmarks();
        std::vector<Landmark*> lms;
        std::vector<float> reproj_f, angle;
        std::vector<int32_t> level;
        std::vector<uint8_t> desc;
        for (unsigned int idx = 0; idx < landmarks.size(); idx++) {
            auto* lm = landmarks.at(idx);
            if (!lm) continue;
            if (lm->will_be_erased()) continue;
            if (already_matched_lms.count(lm)) continue;
            const Vec3_t pos_w = lm->get_pos_in_world();
            Vec2_t reproj;
            float x_right;
            if (!curr_frm.camera_->reproject_to_image(rot_cw, trans_cw, pos_w, reproj, x_right)) continue;
            const Vec3_t cam_to_lm_vec = pos_w - cam_center;
            const auto cam_to_lm_dist = cam_to_lm_vec.norm();
            if (cam_to_lm_dist < lm->get_min_valid_distance() || lm->get_max_valid_distance() < cam_to_lm_dist) continue;
            const auto pred_scale_level = lm->predict_scale_level(cam_to_lm_dist, &curr_frm);
            lms.push_back(lm);
            reproj_f.push_back(static_cast<float>(reproj(0))); reproj_f.push_back(static_cast<float>(reproj(1)));
            level.push_back(static_cast<int32_t>(pred_scale_level));
            angle.push_back(keyfrm->undist_keypts_.at(idx).angle);
            const auto lm_desc = lm->get_descriptor();
            const unsigned char* p = lm_desc.template ptr<unsigned char>(0);
            desc.insert(desc.end(), p, p + 32);
        }
        const detail::frame_targets<Frame> T(curr_frm);
        if (lms.empty() || T.n == 0) return 0;
        std::vector<uint8_t> taken(static_cast<size_t>(T.n));
        for (int i = 0; i < T.n; ++i) taken[i] = curr_frm.landmarks_.at(i) ? 1 : 0;       // any landmark blocks the slot (:602-605)
        std::vector<int32_t> out(static_cast<size_t>(T.n), -1);
        int32_t num = 0;
        plp_match_args a{};
        a.mode = PLP_MATCH_MODE_LAST_FRAME; a.B = 1; a.n_cap = T.n; a.m_cap = static_cast<int32_t>(lms.size());
        a.t_kps = T.kps; a.t_desc = T.desc.data(); a.t_occupied = taken.data();
        a.q_reproj = reproj_f.data(); a.q_level = level.data(); a.q_angle = angle.data(); a.q_desc = desc.data();
        a.margin = margin; a.lowe_ratio = lowe_ratio_; a.check_orientation = check_orientation_ ? 1 : 0;
        a.direction = 0; a.hamm_dist_thr = static_cast<int32_t>(hamm_dist_thr); a.flags = PLP_MATCH_FLAG_MARK_INVALIDATED;
        a.num_levels = static_cast<int32_t>(curr_frm.scale_factors_.size()); a.scale_factors = curr_frm.scale_factors_.data();
        a.grid = detail::grid_of(curr_frm.camera_);
        a.out_match = out.data(); a.out_num = &num;
        detail::check(plp_match_host(detail::shared_matcher(), &a));
        for (int i = 0; i < T.n; ++i) {
            if (out[i] >= 0) curr_frm.landmarks_.at(i) = lms[static_cast<size_t>(out[i])];
            else if (out[i] == -2) curr_frm.landmarks_.at(i) = nullptr;
        }
        return static_cast<unsigned int>(num);
    }
    //! projection.cc:648-779
    template <class Frame, class KeyFrame, class Line>
    unsigned int match_frame_and_keyframe_line(Frame& curr_frm, KeyFrame* keyfrm, const std::set<Line*>& already_matched_lms, const float margin,
                                               const unsigned int hamm_dist_thr) const {
        const Mat33_t rot_cw = curr_frm.cam_pose_cw_.template block<3, 3>(0, 0);
        const Vec3_t trans_cw = curr_frm.cam_pose_cw_.template block<3, 1>(0, 3);
        const Vec3_t cam_center = -rot_cw.transpose() * trans_cw;
        const auto landmarks_line = keyfrm->get_landmarks_line();
        std::vector<Line*> lms;
        std::vector<float> sp, ep;
        std::vector<int32_t> level;
        std::vector<uint8_t> desc;
        for (unsigned int idx = 0; idx < landmarks_line.size(); idx++) {
            auto* lm_line = landmarks_line.at(idx);
            if (!lm_line) continue;
            if (lm_line->will_be_erased()) continue;
            if (already_matched_lms.count(lm_line)) continue;
            const Vec6_t pos_w = lm_line->get_pos_in_world();
            const Vec3_t pos_w_sp = pos_w.template head<3>(), pos_w_ep = pos_w.template tail<3>();
            Vec2_t reproj_sp, reproj_ep;
            float x_right_sp, x_right_ep;
            const bool in_image_sp = curr_frm.camera_->reproject_to_image(rot_cw, trans_cw, pos_w_sp, reproj_sp, x_right_sp);
            const bool in_image_ep = curr_frm.camera_->reproject_to_image(rot_cw, trans_cw, pos_w_ep, reproj_ep, x_right_ep);
            if (!in_image_sp && !in_image_ep) continue;
            if (!in_image_sp || !in_image_ep) {
                const Vec3_t pos_w_mp = 0.5 * (pos_w_sp + pos_w_ep);
                Vec2_t reproj_mp;
                float x_right_mp;
                if (!curr_frm.camera_->reproject_to_image(rot_cw, trans_cw, pos_w_mp, reproj_mp, x_right_mp)) continue;
            }
            const Vec3_t cam_to_lm_vec = 0.5 * (pos_w_sp + pos_w_ep) - cam_center;
            const auto cam_to_lm_dist = cam_to_lm_vec.norm();
            if (cam_to_lm_dist < lm_line->get_min_valid_distance() || lm_line->get_max_valid_distance() < cam_to_lm_dist) continue;
            const auto pred_scale_level = lm_line->predict_scale_level(cam_to_lm_dist, curr_frm._log_scale_factor_lsd, curr_frm._num_scale_levels_lsd);
            lms.push_back(lm_line);
            sp.push_back(static_cast<float>(reproj_sp(0))); sp.push_back(static_cast<float>(reproj_sp(1)));
            ep.push_back(static_cast<float>(reproj_ep(0))); ep.push_back(static_cast<float>(reproj_ep(1)));
            level.push_back(static_cast<int32_t>(pred_scale_level));
            const auto lm_desc = lm_line->get_descriptor();
            const unsigned char* p = lm_desc.template ptr<unsigned char>(0);
            desc.insert(desc.end(), p, p + 32);
        }
        const detail::frame_line_targets<Frame> T(curr_frm);
        if (lms.empty() || T.n == 0) return 0;
        std::vector<uint8_t> taken(static_cast<size_t>(T.n));
        for (int i = 0; i < T.n; ++i) taken[i] = curr_frm._landmarks_line.at(i) ? 1 : 0;
        std::vector<int32_t> out(static_cast<size_t>(T.n), -1);
        int32_t num = 0;
        plp_match_args a{};
        a.mode = PLP_MATCH_MODE_LAST_FRAME_LINE; a.B = 1; a.n_cap = T.n; a.m_cap = static_cast<int32_t>(lms.size());
        a.t_kl = T.kl; a.t_desc = T.desc.data(); a.t_occupied = taken.data();
        a.q_reproj = sp.data(); a.q_reproj2 = ep.data(); a.q_level = level.data(); a.q_desc = desc.data();
        a.margin = margin; a.lowe_ratio = lowe_ratio_;
        a.direction = 0; a.is_rgbd = 0; a.hamm_dist_thr = static_cast<int32_t>(hamm_dist_thr);
        a.num_levels_lsd = static_cast<int32_t>(curr_frm._num_scale_levels_lsd);
        a.num_levels = static_cast<int32_t>(curr_frm._scale_factors_lsd.size()); a.scale_factors = curr_frm._scale_factors_lsd.data();
        a.grid = detail::grid_of(curr_frm.camera_);
        a.out_match = out.data(); a.out_num = &num;
        detail::check(plp_match_host(detail::shared_matcher(), &a));
        for (int i = 0; i < T.n; ++i)
            if (out[i] >= 0) curr_frm._landmarks_line.at(i) = lms[static_cast<size_t>(out[i])];
        return static_cast<unsigned int>(num);
    }
    //! projection.cc:781-892 (loop closure: more landmarks of the candidate's neighbourhood through the estimated Sim3)
    template <class KeyFrame, class Landmark>
    unsigned int match_by_Sim3_transform(KeyFrame* keyfrm, const Mat44_t& Sim3_cw, const std::vector<Landmark*>& landmarks,
                                         std::vector<Landmark*>& matched_lms_in_keyfrm, const float margin) const {
        const Mat33_t s_rot_cw = Sim3_cw.template block<3, 3>(0, 0);
        const auto s_cw = std::sqrt(s_rot_cw.template block<1, 3>(0, 0).dot(s_rot_cw.template block<1, 3>(0, 0)));
        const Mat33_t rot_cw = s_rot_cw / s_cw;
        const Vec3_t trans_cw = Sim3_cw.template block<3, 1>(0, 3) / s_cw;
        const Vec3_t cam_center = -rot_cw.transpose() * trans_cw;
        std::set<Landmark*> already_matched(matched_lms_in_keyfrm.begin(), matched_lms_in_keyfrm.end());
        already_matched.erase(static_cast<Landmark*>(nullptr));
        std::vector<Landmark*> lms;
        std::vector<float> reproj_f;
        std::vector<int32_t> level;
        std::vector<uint8_t> desc;
        for (auto lm : landmarks) {
            if (lm->will_be_erased()) continue;
            if (already_matched.count(lm)) continue;
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
            reproj_f.push_back(static_cast<float>(reproj(0))); reproj_f.push_back(static_cast<float>(reproj(1)));
            level.push_back(static_cast<int32_t>(pred_scale_level));
            const auto lm_desc = lm->get_descriptor();
            const unsigned char* p = lm_desc.template ptr<unsigned char>(0);
            desc.insert(desc.end(), p, p + 32);
        }
        const int n = static_cast<int>(keyfrm->undist_keypts_.size());
        if (lms.empty() || n == 0) return 0;
        static_assert(sizeof(keyfrm->undist_keypts_[0]) == sizeof(plp_keypoint), "cv::KeyPoint must be the 28-byte POD");
        std::vector<uint8_t> t_desc(static_cast<size_t>(n) * 32), taken(static_cast<size_t>(n));
        for (int i = 0; i < n; ++i) {
            const unsigned char* p = keyfrm->descriptors_.template ptr<unsigned char>(i);
            for (int k = 0; k < 32; ++k) t_desc[static_cast<size_t>(i) * 32 + k] = p[k];
            taken[i] = matched_lms_in_keyfrm.at(i) ? 1 : 0;
        }
        std::vector<int32_t> out(static_cast<size_t>(n), -1);
        int32_t num = 0;
        plp_match_args a{};
        a.mode = PLP_MATCH_MODE_LAST_FRAME; a.B = 1; a.n_cap = n; a.m_cap = static_cast<int32_t>(lms.size());
        a.t_kps = reinterpret_cast<const plp_keypoint*>(keyfrm->undist_keypts_.data()); a.t_desc = t_desc.data(); a.t_occupied = taken.data();
        a.q_reproj = reproj_f.data(); a.q_level = level.data(); a.q_desc = desc.data();
        a.margin = margin; a.lowe_ratio = lowe_ratio_; a.check_orientation = 0;
        a.hamm_dist_thr = 50; a.level_window = 1; a.flags = PLP_MATCH_FLAG_UNSIGNED_LEVEL;      // [pred - 1, pred] in unsigned arithmetic (:862)
        a.num_levels = static_cast<int32_t>(keyfrm->scale_factors_.size()); a.scale_factors = keyfrm->scale_factors_.data();
        a.grid = detail::grid_of(keyfrm->camera_);
        a.out_match = out.data(); a.out_num = &num;
        detail::check(plp_match_host(detail::shared_matcher(), &a));
        for (int i = 0; i < n; ++i)
            if (out[i] >= 0) matched_lms_in_keyfrm.at(i) = lms[static_cast<size_t>(out[i])];
        return static_cast<unsigned int>(num);
    }
    //! projection.cc:894-1142 (loop closure: landmarks of each key frame projected into the other through the Sim3, kept when
    //! the two best matches agree)
    template <class KeyFrame, class Landmark>
    unsigned int match_keyframes_mutually(KeyFrame* keyfrm_1, KeyFrame* keyfrm_2, std::vector<Landmark*>& matched_lms_in_keyfrm_1, const float& s_12,
                                          const Mat33_t& rot_12, const Vec3_t& trans_12, const float margin) const {
        const Mat33_t rot_1w = keyfrm_1->get_rotation();
        const Vec3_t trans_1w = keyfrm_1->get_translation();
        const Mat33_t rot_2w = keyfrm_2->get_rotation();
        const Vec3_t trans_2w = keyfrm_2->get_translation();
        const Mat33_t s_rot_12 = s_12 * rot_12;
        const Mat33_t s_rot_21 = (1.0 / s_12) * rot_12.transpose();
        const Vec3_t trans_21 = -s_rot_21 * trans_12;
        const auto landmarks_1 = keyfrm_1->get_landmarks();
        const auto landmarks_2 = keyfrm_2->get_landmarks();
        std::vector<bool> is_already_matched_in_keyfrm_1(landmarks_1.size(), false), is_already_matched_in_keyfrm_2(landmarks_2.size(), false);
        for (unsigned int idx_1 = 0; idx_1 < landmarks_1.size(); ++idx_1) {
            auto* lm = matched_lms_in_keyfrm_1.at(idx_1);
            if (!lm) continue;
            const auto idx_2 = lm->get_index_in_keyframe(keyfrm_2);
            if (0 <= idx_2 && idx_2 < static_cast<int>(landmarks_2.size())) {
                is_already_matched_in_keyfrm_1.at(idx_1) = true;
                is_already_matched_in_keyfrm_2.at(idx_2) = true;
            }
        }
        // one direction: the landmarks of `from` projected with (s_rot, trans) and searched among the key points of `to`
        // (the camera model of key frame 2 does the projection in BOTH directions, as in the reference: :957, :1040)
        auto project_and_search = [&](const std::vector<Landmark*>& lms_from, const std::vector<bool>& already, const Mat33_t& s_rot, const Vec3_t& trans,
                                      KeyFrame* to, std::vector<int>& matched_in_to) {
            std::vector<unsigned int> q_i;
            std::vector<double> reproj_d;
            std::vector<int32_t> level;
            std::vector<uint8_t> desc;
            for (unsigned int idx = 0; idx < lms_from.size(); ++idx) {
                auto* lm = lms_from.at(idx);
                if (!lm) continue;
                if (lm->will_be_erased()) continue;
                if (already.at(idx)) continue;
                const Vec3_t pos_w = lm->get_pos_in_world();
                const Vec3_t pos_to = s_rot * pos_w + trans;
                Vec2_t reproj;
                float x_right;
                if (!keyfrm_2->camera_->reproject_to_image(s_rot, trans, pos_w, reproj, x_right)) continue;
                const auto cam_to_lm_dist = pos_to.norm();
                if (cam_to_lm_dist < lm->get_min_valid_distance() || lm->get_max_valid_distance() < cam_to_lm_dist) continue;
                const auto pred_scale_level = lm->predict_scale_level(cam_to_lm_dist, to);
                q_i.push_back(idx);
                reproj_d.push_back(reproj(0)); reproj_d.push_back(reproj(1));
                level.push_back(static_cast<int32_t>(pred_scale_level));
                const auto lm_desc = lm->get_descriptor();
                const unsigned char* p = lm_desc.template ptr<unsigned char>(0);
                desc.insert(desc.end(), p, p + 32);
            }
            const int n = static_cast<int>(to->undist_keypts_.size()), m = static_cast<int>(q_i.size());
            if (n == 0 || m == 0) return;
            std::vector<uint8_t> t_desc(static_cast<size_t>(n) * 32);
            for (int k = 0; k < n; ++k) {
                const unsigned char* p = to->descriptors_.template ptr<unsigned char>(k);
                for (int b = 0; b < 32; ++b) t_desc[static_cast<size_t>(k) * 32 + b] = p[b];
            }
            std::vector<float> inv_sigma(to->scale_factors_.size(), 1.0f);
            std::vector<int32_t> best(static_cast<size_t>(m), -1);
            plp_match_args a{};
            a.mode = PLP_MATCH_MODE_FUSE; a.B = 1; a.n_cap = n; a.m_cap = m;
            a.flags = PLP_MATCH_FLAG_NO_CHI2; a.hamm_dist_thr = 100;                       // HAMMING_DIST_THR_HIGH (:1015, :1096)
            a.t_kps = reinterpret_cast<const plp_keypoint*>(to->undist_keypts_.data()); a.t_desc = t_desc.data();
            a.q_reproj_d = reproj_d.data(); a.q_level = level.data(); a.q_desc = desc.data();
            a.margin = margin; a.lowe_ratio = lowe_ratio_;
            a.num_levels = static_cast<int32_t>(to->scale_factors_.size()); a.scale_factors = to->scale_factors_.data();
            a.inv_level_sigma_sq = inv_sigma.data();
            a.grid = detail::grid_of(to->camera_);
            a.out_query_best = best.data();
            detail::check(plp_match_host(detail::shared_matcher(), &a));
            for (int q = 0; q < m; ++q) matched_in_to.at(q_i[static_cast<size_t>(q)]) = best[static_cast<size_t>(q)];
        };
        std::vector<int> matched_indices_2_in_keyfrm_1(landmarks_1.size(), -1), matched_indices_1_in_keyfrm_2(landmarks_2.size(), -1);
        {
            const Mat33_t s_rot_21w = s_rot_21 * rot_1w;
            const Vec3_t trans_21w = s_rot_21 * trans_1w + trans_21;
            project_and_search(landmarks_1, is_already_matched_in_keyfrm_1, s_rot_21w, trans_21w, keyfrm_2, matched_indices_2_in_keyfrm_1);
        }
        {
            const Mat33_t s_rot_12w = s_rot_12 * rot_2w;
            const Vec3_t trans_12w = s_rot_12 * trans_2w + trans_12;
            project_and_search(landmarks_2, is_already_matched_in_keyfrm_2, s_rot_12w, trans_12w, keyfrm_1, matched_indices_1_in_keyfrm_2);
        }
        unsigned int num_matches = 0;
        for (unsigned int i = 0; i < landmarks_1.size(); ++i) {      // the cross check (:1124-1139)
            const auto idx_2 = matched_indices_2_in_keyfrm_1.at(i);
            if (idx_2 < 0) continue;
            const auto idx_1 = matched_indices_1_in_keyfrm_2.at(idx_2);
            if (idx_1 == static_cast<int>(i)) {
                matched_lms_in_keyfrm_1.at(idx_1) = landmarks_2.at(idx_2);
                ++num_matches;
            }
        }
        return num_matches;
    }

    //! projection.cc:37-121
    template <class Frame, class Landmark>
    unsigned int match_frame_and_landmarks(Frame& frm, const std::vector<Landmark*>& local_landmarks, const float margin = 5.0) const {
        std::vector<Landmark*> lms;                    // the landmarks that pass the skip tests, in list order
        std::vector<float> reproj, x_right;
        std::vector<int32_t> level;
        std::vector<uint8_t> desc, has_obs;
        for (auto local_lm : local_landmarks) {
            if (!local_lm->is_observable_in_tracking_) continue;
            if (local_lm->will_be_erased()) continue;
            lms.push_back(local_lm);
            reproj.push_back(static_cast<float>(local_lm->reproj_in_tracking_(0)));
            reproj.push_back(static_cast<float>(local_lm->reproj_in_tracking_(1)));
            x_right.push_back(static_cast<float>(local_lm->x_right_in_tracking_));
            level.push_back(static_cast<int32_t>(local_lm->scale_level_in_tracking_));
            const auto lm_desc = local_lm->get_descriptor();
            const unsigned char* p = lm_desc.template ptr<unsigned char>(0);
            desc.insert(desc.end(), p, p + 32);
            has_obs.push_back(local_lm->has_observation() ? 1 : 0);
        }
        const detail::frame_targets<Frame> T(frm);
        if (lms.empty() || T.n == 0) return 0;
        std::vector<int32_t> out(static_cast<size_t>(T.n), -1);
        int32_t num = 0;
        plp_match_args a{};
        a.mode = PLP_MATCH_MODE_LANDMARKS; a.B = 1; a.n_cap = T.n; a.m_cap = static_cast<int32_t>(lms.size());
        a.t_kps = T.kps; a.t_desc = T.desc.data(); a.t_x_right = frm.stereo_x_right_.data(); a.t_occupied = T.occupied.data();
        a.q_reproj = reproj.data(); a.q_x_right = x_right.data(); a.q_level = level.data(); a.q_desc = desc.data(); a.q_has_obs = has_obs.data();
        a.margin = margin; a.lowe_ratio = lowe_ratio_; a.check_orientation = check_orientation_ ? 1 : 0;
        a.num_levels = static_cast<int32_t>(frm.scale_factors_.size()); a.scale_factors = frm.scale_factors_.data();
        a.grid = detail::grid_of(frm.camera_);
        a.out_match = out.data(); a.out_num = &num;
        detail::check(plp_match_host(detail::shared_matcher(), &a));
        for (int i = 0; i < T.n; ++i)
            if (out[i] >= 0) frm.landmarks_.at(i) = lms[static_cast<size_t>(out[i])];
        return static_cast<unsigned int>(num);
    }

    //! projection.cc:214-358
    template <class Frame>
    unsigned int match_current_and_last_frames(Frame& curr_frm, const Frame& last_frm, const float margin) const {
        const Mat33_t rot_cw = curr_frm.cam_pose_cw_.template block<3, 3>(0, 0);
        const Vec3_t trans_cw = curr_frm.cam_pose_cw_.template block<3, 1>(0, 3);
        const Vec3_t trans_wc = -rot_cw.transpose() * trans_cw;
        const Mat33_t rot_lw = last_frm.cam_pose_cw_.template block<3, 3>(0, 0);
        const Vec3_t trans_lw = last_frm.cam_pose_cw_.template block<3, 1>(0, 3);
        const Vec3_t trans_lc = rot_lw * trans_wc + trans_lw;
        const bool mono = static_cast<int>(curr_frm.camera_->setup_type_) == 0;   // camera::setup_type_t::Monocular
        const bool assume_forward = mono ? false : trans_lc(2) > curr_frm.camera_->true_baseline_;
        const bool assume_backward = mono ? false : -trans_lc(2) > curr_frm.camera_->true_baseline_;

        using LandmarkPtr = typename std::decay<decltype(last_frm.landmarks_.at(0))>::type;
        std::vector<LandmarkPtr> lms;
        std::vector<float> reproj_f, x_right_f, angle;
        std::vector<int32_t> level;
        std::vector<uint8_t> desc, has_obs;
        for (unsigned int idx_last = 0; idx_last < last_frm.num_keypts_; ++idx_last) {
            auto lm = last_frm.landmarks_.at(idx_last);
            if (!lm) continue;
            if (last_frm.outlier_flags_.at(idx_last)) continue;
            const Vec3_t pos_w = lm->get_pos_in_world();
            Vec2_t reproj;
            float x_right;
            if (!curr_frm.camera_->reproject_to_image(rot_cw, trans_cw, pos_w, reproj, x_right)) continue;
            lms.push_back(lm);
            reproj_f.push_back(static_cast<float>(reproj(0))); reproj_f.push_back(static_cast<float>(reproj(1)));
            x_right_f.push_back(x_right);
            level.push_back(static_cast<int32_t>(last_frm.keypts_.at(idx_last).octave));
            angle.push_back(last_frm.undist_keypts_.at(idx_last).angle);
            // a key point taken by this landmark blocks later queries only if the landmark has observations (:300)
            has_obs.push_back(lm->has_observation() ? 1 : 0);
            const auto lm_desc = lm->get_descriptor();
            const unsigned char* p = lm_desc.template ptr<unsigned char>(0);
            desc.insert(desc.end(), p, p + 32);
        }
        const detail::frame_targets<Frame> T(curr_frm);
        if (lms.empty() || T.n == 0) return 0;
        std::vector<int32_t> out(static_cast<size_t>(T.n), -1);
        int32_t num = 0;
        plp_match_args a{};
        a.mode = PLP_MATCH_MODE_LAST_FRAME; a.B = 1; a.n_cap = T.n; a.m_cap = static_cast<int32_t>(lms.size());
        a.t_kps = T.kps; a.t_desc = T.desc.data(); a.t_x_right = curr_frm.stereo_x_right_.data(); a.t_occupied = T.occupied.data();
        a.q_reproj = reproj_f.data(); a.q_x_right = x_right_f.data(); a.q_level = level.data(); a.q_angle = angle.data(); a.q_desc = desc.data();
        a.q_has_obs = has_obs.data();
        a.margin = margin; a.lowe_ratio = lowe_ratio_; a.check_orientation = check_orientation_ ? 1 : 0;
        a.direction = assume_forward ? 1 : (assume_backward ? 2 : 0);
        a.flags = PLP_MATCH_FLAG_MARK_INVALIDATED;
        // the upper octave of the assume_forward window is last_frm.num_scale_levels_ - 1 (projection.cc:280)
        a.num_levels = static_cast<int32_t>(last_frm.num_scale_levels_); a.scale_factors = curr_frm.scale_factors_.data();
        a.grid = detail::grid_of(curr_frm.camera_);
        a.out_match = out.data(); a.out_num = &num;
        detail::check(plp_match_host(detail::shared_matcher(), &a));
        for (int i = 0; i < T.n; ++i) {
            if (out[i] >= 0) curr_frm.landmarks_.at(i) = lms[static_cast<size_t>(out[i])];
            else if (out[i] == -2) curr_frm.landmarks_.at(i) = nullptr;     // matched, then removed by the orientation check (:350-354)
        }
        return static_cast<unsigned int>(num);
    }

    //! projection.cc:124-212
    template <class Frame, class Line>
    unsigned int match_frame_and_landmarks_line(Frame& frm, const std::vector<Line*>& local_landmarks_line, const float margin = 5.0) const {
        if (local_landmarks_line.empty()) return 0;
        std::vector<Line*> lms;
        std::vector<float> sp, ep;
        std::vector<int32_t> level;
        std::vector<uint8_t> desc, has_obs;
        for (auto local_lm_line : local_landmarks_line) {
            if (!local_lm_line->_is_observable_in_tracking) continue;
            if (local_lm_line->will_be_erased()) continue;
            lms.push_back(local_lm_line);
            sp.push_back(static_cast<float>(local_lm_line->_reproj_in_tracking_sp(0))); sp.push_back(static_cast<float>(local_lm_line->_reproj_in_tracking_sp(1)));
            ep.push_back(static_cast<float>(local_lm_line->_reproj_in_tracking_ep(0))); ep.push_back(static_cast<float>(local_lm_line->_reproj_in_tracking_ep(1)));
            level.push_back(static_cast<int32_t>(local_lm_line->_scale_level_in_tracking));
            const auto lm_desc = local_lm_line->get_descriptor();
            const unsigned char* p = lm_desc.template ptr<unsigned char>(0);
            desc.insert(desc.end(), p, p + 32);
            has_obs.push_back(local_lm_line->has_observation() ? 1 : 0);
        }
        const detail::frame_line_targets<Frame> T(frm);
        if (lms.empty() || T.n == 0) return 0;
        // the reference reads undist_keypts_.at(idx).octave with a LINE index for the ratio test (:187, :192)
        std::vector<int32_t> kp_octave(static_cast<size_t>(T.n));
        for (int i = 0; i < T.n; ++i) kp_octave[i] = frm.undist_keypts_.at(i).octave;
        std::vector<int32_t> out(static_cast<size_t>(T.n), -1);
        int32_t num = 0;
        plp_match_args a{};
        a.mode = PLP_MATCH_MODE_LANDMARKS_LINE; a.B = 1; a.n_cap = T.n; a.m_cap = static_cast<int32_t>(lms.size());
        a.t_kl = T.kl; a.t_desc = T.desc.data(); a.t_occupied = T.occupied.data(); a.t_kp_octave = kp_octave.data();
        a.q_reproj = sp.data(); a.q_reproj2 = ep.data(); a.q_level = level.data(); a.q_desc = desc.data(); a.q_has_obs = has_obs.data();
        a.margin = margin; a.lowe_ratio = lowe_ratio_;
        a.num_levels = static_cast<int32_t>(frm._scale_factors_lsd.size()); a.scale_factors = frm._scale_factors_lsd.data();
        a.grid = detail::grid_of(frm.camera_);
        a.out_match = out.data(); a.out_num = &num;
        detail::check(plp_match_host(detail::shared_matcher(), &a));
        for (int i = 0; i < T.n; ++i)
            if (out[i] >= 0) frm._landmarks_line.at(i) = lms[static_cast<size_t>(out[i])];
        return static_cast<unsigned int>(num);
    }

    //! projection.cc:361-527
    template <class Frame>
    unsigned int match_current_and_last_frames_line(Frame& curr_frm, const Frame& last_frm, const float margin) const {
        const Mat33_t rot_cw = curr_frm.cam_pose_cw_.template block<3, 3>(0, 0);
        const Vec3_t trans_cw = curr_frm.cam_pose_cw_.template block<3, 1>(0, 3);
        const Vec3_t trans_wc = -rot_cw.transpose() * trans_cw;
        const Mat33_t rot_lw = last_frm.cam_pose_cw_.template block<3, 3>(0, 0);
        const Vec3_t trans_lw = last_frm.cam_pose_cw_.template block<3, 1>(0, 3);
        const Vec3_t trans_lc = rot_lw * trans_wc + trans_lw;
        const int setup = static_cast<int>(curr_frm.camera_->setup_type_);        // 0 Monocular, 1 Stereo, 2 RGBD
        const bool assume_forward = setup == 0 ? false : trans_lc(2) > curr_frm.camera_->true_baseline_;
        const bool assume_backward = setup == 0 ? false : -trans_lc(2) > curr_frm.camera_->true_baseline_;

        using LinePtr = typename std::decay<decltype(last_frm._landmarks_line.at(0))>::type;
        std::vector<LinePtr> lms;
        std::vector<float> sp, ep, xr_sp, xr_ep;
        std::vector<int32_t> level;
        std::vector<uint8_t> desc, has_obs;
        for (unsigned int idx_last = 0; idx_last < last_frm._num_keylines; ++idx_last) {
            auto lm_line = last_frm._landmarks_line.at(idx_last);
            if (!lm_line) continue;
            if (last_frm._outlier_flags_line.at(idx_last)) continue;
            const Vec6_t pos_w = lm_line->get_pos_in_world();
            Vec2_t reproj_sp, reproj_ep;
            float x_right_sp, x_right_ep;
            const Vec3_t pos_sp = pos_w.template head<3>(), pos_ep = pos_w.template tail<3>();
            const bool in_image_sp = curr_frm.camera_->reproject_to_image(rot_cw, trans_cw, pos_sp, reproj_sp, x_right_sp);
            const bool in_image_ep = curr_frm.camera_->reproject_to_image(rot_cw, trans_cw, pos_ep, reproj_ep, x_right_ep);
            if (!in_image_sp && !in_image_ep) continue;
            if (!in_image_sp || !in_image_ep) {      // one end point outside: kept only if the mid point is visible (:397-412)
                const Vec3_t pos_w_mp = 0.5 * (pos_sp + pos_ep);
                Vec2_t reproj_mp;
                float x_right_mp;
                if (!curr_frm.camera_->reproject_to_image(rot_cw, trans_cw, pos_w_mp, reproj_mp, x_right_mp)) continue;
            }
            lms.push_back(lm_line);
            sp.push_back(static_cast<float>(reproj_sp(0))); sp.push_back(static_cast<float>(reproj_sp(1)));
            ep.push_back(static_cast<float>(reproj_ep(0))); ep.push_back(static_cast<float>(reproj_ep(1)));
            xr_sp.push_back(x_right_sp); xr_ep.push_back(x_right_ep);
            level.push_back(static_cast<int32_t>(last_frm._keylsd.at(idx_last).octave));
            has_obs.push_back(lm_line->has_observation() ? 1 : 0);     // (:485): only an observed landmark blocks its key line
            const auto lm_desc = lm_line->get_descriptor();
            const unsigned char* p = lm_desc.template ptr<unsigned char>(0);
            desc.insert(desc.end(), p, p + 32);
        }
        const detail::frame_line_targets<Frame> T(curr_frm);
        if (lms.empty() || T.n == 0) return 0;
        std::vector<float> t_xr(static_cast<size_t>(T.n)), t_xr2(static_cast<size_t>(T.n));
        for (int i = 0; i < T.n; ++i) {
            t_xr[i] = curr_frm._stereo_x_right_cooresponding_to_keylines.at(i).first;
            t_xr2[i] = curr_frm._stereo_x_right_cooresponding_to_keylines.at(i).second;
        }
        std::vector<int32_t> out(static_cast<size_t>(T.n), -1);
        int32_t num = 0;
        plp_match_args a{};
        a.mode = PLP_MATCH_MODE_LAST_FRAME_LINE; a.B = 1; a.n_cap = T.n; a.m_cap = static_cast<int32_t>(lms.size());
        a.t_kl = T.kl; a.t_desc = T.desc.data(); a.t_occupied = T.occupied.data(); a.t_x_right = t_xr.data(); a.t_x_right2 = t_xr2.data();
        a.q_reproj = sp.data(); a.q_reproj2 = ep.data(); a.q_x_right = xr_sp.data(); a.q_x_right2 = xr_ep.data(); a.q_level = level.data();
        a.q_desc = desc.data(); a.q_has_obs = has_obs.data();
        a.margin = margin; a.lowe_ratio = lowe_ratio_;
        a.direction = assume_forward ? 1 : (assume_backward ? 2 : 0);
        a.is_rgbd = setup == 2 ? 1 : 0;
        a.num_levels_lsd = static_cast<int32_t>(last_frm._num_scale_levels_lsd);
        a.num_levels = static_cast<int32_t>(curr_frm._scale_factors_lsd.size()); a.scale_factors = curr_frm._scale_factors_lsd.data();
        a.grid = detail::grid_of(curr_frm.camera_);
        a.out_match = out.data(); a.out_num = &num;
        detail::check(plp_match_host(detail::shared_matcher(), &a));
        for (int i = 0; i < T.n; ++i)
            if (out[i] >= 0) curr_frm._landmarks_line.at(i) = lms[static_cast<size_t>(out[i])];
        return static_cast<unsigned int>(num);
    }
};

}  // namespace match
}  // namespace PLPSLAM

#endif  // PLPSLAM_MATCH_PROJECTION_H
