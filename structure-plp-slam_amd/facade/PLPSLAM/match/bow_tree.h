// Drop-in replacement of the reference header src/PLPSLAM/match/bow_tree.h: bow_tree::match_frame_and_keyframe
// (match/bow_tree.cc:41-165; frame_tracker::bow_match_based_track and the relocalizer call it) and bow_tree::match_keyframes
// (:167-307; loop closure), same class, constructor, method names, arguments and return values.  The node-by-node search
// runs in libplp_front.so (PLP_MATCH_MODE_BOW); the host lists the first key frame's features in the order the reference
// walks its feature vector and labels every feature of the other side with its node.
// Templates on the key-frame / frame / landmark types, like match/projection.h.
#ifndef PLPSLAM_MATCH_BOW_TREE_H
#define PLPSLAM_MATCH_BOW_TREE_H

#include <cstdint>
#include <vector>

#include "PLPSLAM/match/projection.h"   // match::base (the reference's or its stand-in), detail::shared_matcher / check

namespace PLPSLAM {
namespace match {

class bow_tree final : public base {
public:
    explicit bow_tree(const float lowe_ratio = 0.6, const bool check_orientation = true) : base(lowe_ratio, check_orientation) {}
    ~bow_tree() final = default;

    //! bow_tree.cc:41-165
    template <class KeyFrame, class Frame, class Landmark>
    unsigned int match_frame_and_keyframe(KeyFrame* keyfrm, Frame& frm, std::vector<Landmark*>& matched_lms_in_frm) const {
        matched_lms_in_frm = std::vector<Landmark*>(frm.num_keypts_, nullptr);
        const auto keyfrm_lms = keyfrm->get_landmarks();
        const int n = static_cast<int>(frm.num_keypts_);
        // frame side: node of every feature (-1: in no node of the feature vector, can never be a candidate)
        std::vector<int32_t> t_group(static_cast<size_t>(n > 0 ? n : 1), -1);
        for (const auto& node : frm.bow_feat_vec_)
            for (const auto frm_idx : node.second) t_group.at(frm_idx) = static_cast<int32_t>(node.first);
        // key-frame side: features in feature-vector order (node by node, stored order inside a node)
        std::vector<unsigned int> q_idx;
        std::vector<int32_t> q_group;
        std::vector<uint8_t> q_valid, q_desc;
        std::vector<float> q_angle;
        for (const auto& node : keyfrm->bow_feat_vec_)
            for (const auto keyfrm_idx : node.second) {
                auto* lm = keyfrm_lms.at(keyfrm_idx);
                q_idx.push_back(keyfrm_idx);
                q_group.push_back(static_cast<int32_t>(node.first));
                q_valid.push_back((lm && !lm->will_be_erased()) ? 1 : 0);
                q_angle.push_back(keyfrm->keypts_.at(keyfrm_idx).angle);
                const unsigned char* p = keyfrm->descriptors_.template ptr<unsigned char>(static_cast<int>(keyfrm_idx));
                q_desc.insert(q_desc.end(), p, p + 32);
            }
        const int m = static_cast<int>(q_idx.size());
        if (n == 0 || m == 0) return 0;
        std::vector<uint8_t> t_desc(static_cast<size_t>(n) * 32);
        std::vector<float> t_angle(static_cast<size_t>(n));
        for (int i = 0; i < n; ++i) {
            const unsigned char* p = frm.descriptors_.template ptr<unsigned char>(i);
            for (int k = 0; k < 32; ++k) t_desc[static_cast<size_t>(i) * 32 + k] = p[k];
            t_angle[i] = frm.keypts_.at(i).angle;
        }
        std::vector<int32_t> out(static_cast<size_t>(n), -1);
        int32_t num = 0;
        plp_match_args a{};
        a.mode = PLP_MATCH_MODE_BOW; a.B = 1; a.n_cap = n; a.m_cap = m;
        a.t_desc = t_desc.data(); a.t_angle = t_angle.data(); a.t_group = t_group.data();
        a.q_desc = q_desc.data(); a.q_angle = q_angle.data(); a.q_group = q_group.data(); a.q_valid = q_valid.data();
        a.lowe_ratio = lowe_ratio_; a.check_orientation = check_orientation_ ? 1 : 0;
        a.out_match = out.data(); a.out_num = &num;
        detail::check(plp_match_host(detail::shared_matcher(), &a));
        for (int i = 0; i < n; ++i)
            if (out[i] >= 0) matched_lms_in_frm.at(i) = keyfrm_lms.at(q_idx[static_cast<size_t>(out[i])]);
        return static_cast<unsigned int>(num);
    }

    //! bow_tree.cc:167-307 (loop closure, Sim3 candidates): key frame 1 = queries, key frame 2 = targets that have a live landmark
    template <class KeyFrame, class Landmark>
    unsigned int match_keyframes(KeyFrame* keyfrm_1, KeyFrame* keyfrm_2, std::vector<Landmark*>& matched_lms_in_keyfrm_1) const {
        const auto keyfrm_1_lms = keyfrm_1->get_landmarks();
        const auto keyfrm_2_lms = keyfrm_2->get_landmarks();
        matched_lms_in_keyfrm_1 = std::vector<Landmark*>(keyfrm_1_lms.size(), nullptr);
        const int n = static_cast<int>(keyfrm_2_lms.size());
        std::vector<int32_t> t_group(static_cast<size_t>(n > 0 ? n : 1), -1);
        for (const auto& node : keyfrm_2->bow_feat_vec_)
            for (const auto idx_2 : node.second) t_group.at(idx_2) = static_cast<int32_t>(node.first);
        std::vector<unsigned int> q_idx;
        std::vector<int32_t> q_group;
        std::vector<uint8_t> q_valid, q_desc;
        std::vector<float> q_angle;
        for (const auto& node : keyfrm_1->bow_feat_vec_)
            for (const auto idx_1 : node.second) {
                auto* lm_1 = keyfrm_1_lms.at(idx_1);
                q_idx.push_back(idx_1);
                q_group.push_back(static_cast<int32_t>(node.first));
                q_valid.push_back((lm_1 && !lm_1->will_be_erased()) ? 1 : 0);
                q_angle.push_back(keyfrm_1->keypts_.at(idx_1).angle);
                const unsigned char* p = keyfrm_1->descriptors_.template ptr<unsigned char>(static_cast<int>(idx_1));
                q_desc.insert(q_desc.end(), p, p + 32);
            }
        const int m = static_cast<int>(q_idx.size());
        if (n == 0 || m == 0) return 0;
        std::vector<uint8_t> t_desc(static_cast<size_t>(n) * 32), t_skip(static_cast<size_t>(n));
        std::vector<float> t_angle(static_cast<size_t>(n));
        for (int i = 0; i < n; ++i) {
            const unsigned char* p = keyfrm_2->descriptors_.template ptr<unsigned char>(i);
            for (int k = 0; k < 32; ++k) t_desc[static_cast<size_t>(i) * 32 + k] = p[k];
            t_angle[i] = keyfrm_2->keypts_.at(i).angle;
            auto* lm_2 = keyfrm_2_lms.at(i);
            t_skip[i] = (!lm_2 || lm_2->will_be_erased()) ? 1 : 0;                 // :211-219
        }
        std::vector<int32_t> out(static_cast<size_t>(n), -1);
        int32_t num = 0;
        plp_match_args a{};
        a.mode = PLP_MATCH_MODE_BOW; a.B = 1; a.n_cap = n; a.m_cap = m;
        a.t_desc = t_desc.data(); a.t_angle = t_angle.data(); a.t_group = t_group.data(); a.t_occupied = t_skip.data();
        a.q_desc = q_desc.data(); a.q_angle = q_angle.data(); a.q_group = q_group.data(); a.q_valid = q_valid.data();
        a.lowe_ratio = lowe_ratio_; a.check_orientation = check_orientation_ ? 1 : 0;
        a.out_match = out.data(); a.out_num = &num;
        detail::check(plp_match_host(detail::shared_matcher(), &a));
        for (int i = 0; i < n; ++i)      // a match pairs one feature of each key frame: the target-indexed result inverts directly
            if (out[i] >= 0) matched_lms_in_keyfrm_1.at(q_idx[static_cast<size_t>(out[i])]) = keyfrm_2_lms.at(i);
        return static_cast<unsigned int>(num);
    }
};

}  // namespace match
}  // namespace PLPSLAM

#endif  // PLPSLAM_MATCH_BOW_TREE_H
