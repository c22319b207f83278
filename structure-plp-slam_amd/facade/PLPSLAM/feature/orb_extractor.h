// Drop-in replacement of the reference header src/PLPSLAM/feature/orb_extractor.h (class
// PLPSLAM::feature::orb_extractor, :38-176): same constructors, same extract() signature, same getters and
// setters, same public image_pyramid_ member — implemented over the C ABI of libplp_front.so.
// tracking_module.cc:65-97 and data/frame.cc:1125-1140 compile against it unchanged.
#ifndef PLPSLAM_FEATURE_ORB_EXTRACTOR_H
#define PLPSLAM_FEATURE_ORB_EXTRACTOR_H

#include <stdexcept>
#include <string>
#include <vector>

#include <opencv2/core.hpp>

#include "PLPSLAM/feature/orb_params.h"
#include "PLPSLAM/feature/plp_registry.h"
#include "plp_front.h"

namespace PLPSLAM {
namespace feature {

class orb_extractor {
public:
    orb_extractor() = delete;

    orb_extractor(const unsigned int max_num_keypts, const float scale_factor, const unsigned int num_levels,
                  const unsigned int ini_fast_thr, const unsigned int min_fast_thr,
                  const std::vector<std::vector<float>>& mask_rects = {})
        : orb_extractor(orb_params{max_num_keypts, scale_factor, num_levels, ini_fast_thr, min_fast_thr, mask_rects}) {}

    explicit orb_extractor(const orb_params& p) : orb_params_(p) {
        std::vector<float> rects;
        for (const auto& r : p.mask_rects_) rects.insert(rects.end(), r.begin(), r.end());
        plp_orb_params cp{p.max_num_keypts_, p.scale_factor_, p.num_levels_, p.ini_fast_thr_, p.min_fast_thr,
                          rects.empty() ? nullptr : rects.data(), static_cast<int32_t>(p.mask_rects_.size())};
        check(plp_orb_create(&cp, device_from_env(), &ctx_));
        image_pyramid_.resize(p.num_levels_);
        image_pyramid_.owner_ = this;
        plp_registry::add(&image_pyramid_, ctx_);       // match::stereo is handed image_pyramid_ and finds the context by it
    }

    virtual ~orb_extractor() { plp_registry::remove(&image_pyramid_); plp_orb_destroy(ctx_); }
    orb_extractor(const orb_extractor&) = delete;
    orb_extractor& operator=(const orb_extractor&) = delete;

    //! Extract keypoints and each descriptor of them (orb_extractor.cc:73-160)
    void extract(const cv::_InputArray& in_image, const cv::_InputArray& in_image_mask,
                 std::vector<cv::KeyPoint>& keypts, const cv::_OutputArray& out_descriptors) {
        if (in_image.empty()) return;
        const cv::Mat image = in_image.getMat();
        CV_Assert(image.type() == CV_8UC1);
        cv::Mat mask;
        if (!in_image_mask.empty()) { mask = in_image_mask.getMat(); CV_Assert(mask.type() == CV_8UC1); }
        const int cap = 2 * static_cast<int>(orb_params_.max_num_keypts_) + 64;
        static_assert(sizeof(cv::KeyPoint) == sizeof(plp_keypoint), "cv::KeyPoint must be the 28-byte POD");
        keypts.resize(cap);
        cv::Mat desc(cap, 32, CV_8U);
        int32_t n = 0;
        check(plp_orb_extract(ctx_, image.data, image.rows, image.cols, image.step, mask.empty() ? nullptr : mask.data,
                              mask.empty() ? 0 : mask.step, reinterpret_cast<plp_keypoint*>(keypts.data()), desc.data, cap, &n));
        keypts.resize(n);
        if (n == 0) out_descriptors.release();
        else desc.rowRange(0, n).copyTo(out_descriptors);
        // image_pyramid_ is public (orb_extractor.h:101); its only reader in the reference is match::stereo (data/frame.cc:277-281),
        // whose facade reads the pyramids in HBM through plp_registry.  Levels 1.. are therefore fetched from the device only when
        // host code actually looks at them (lazy_pyramid below): 0.64 MB of D2H per frame that mono / RGB-D tracking never needs.
        image_pyramid_.raw(0) = image;
        image_pyramid_.mark_stale();
        if (eager_pyramid_) image_pyramid_.front();
    }

    unsigned int get_max_num_keypoints() const { return static_cast<unsigned int>(get(PLP_ORB_MAX_NUM_KEYPOINTS)); }
    void set_max_num_keypoints(const unsigned int v) { orb_params_.max_num_keypts_ = v; set(PLP_ORB_MAX_NUM_KEYPOINTS, v); }
    float get_scale_factor() const { return static_cast<float>(get(PLP_ORB_SCALE_FACTOR)); }
    void set_scale_factor(const float v) { orb_params_.scale_factor_ = v; set(PLP_ORB_SCALE_FACTOR, v); }
    unsigned int get_num_scale_levels() const { return static_cast<unsigned int>(get(PLP_ORB_NUM_SCALE_LEVELS)); }
    void set_num_scale_levels(const unsigned int v) { orb_params_.num_levels_ = v; set(PLP_ORB_NUM_SCALE_LEVELS, v); image_pyramid_.resize(v); }
    unsigned int get_initial_fast_threshold() const { return static_cast<unsigned int>(get(PLP_ORB_INITIAL_FAST_THRESHOLD)); }
    void set_initial_fast_threshold(const unsigned int v) { orb_params_.ini_fast_thr_ = v; set(PLP_ORB_INITIAL_FAST_THRESHOLD, v); }
    unsigned int get_minimum_fast_threshold() const { return static_cast<unsigned int>(get(PLP_ORB_MINIMUM_FAST_THRESHOLD)); }
    void set_minimum_fast_threshold(const unsigned int v) { orb_params_.min_fast_thr = v; set(PLP_ORB_MINIMUM_FAST_THRESHOLD, v); }

    std::vector<float> get_scale_factors() const { return table(0); }
    std::vector<float> get_inv_scale_factors() const { return table(1); }
    std::vector<float> get_level_sigma_sq() const { return table(2); }
    std::vector<float> get_inv_level_sigma_sq() const { return table(3); }

    //! std::vector<cv::Mat> whose element accessors first bring levels 1.. over from the device, once per extracted frame.  A
    //! reader that takes it as `const std::vector<cv::Mat>&` (the reference's match::stereo constructor) sees level 0 and stale
    //! or empty upper levels -- that reader is replaced by facade/PLPSLAM/match/stereo.h, which never touches the host copy.
    class lazy_pyramid : public std::vector<cv::Mat> {
    public:
        typedef std::vector<cv::Mat> base;
        cv::Mat& at(size_t i) { fetch(); return base::at(i); }
        const cv::Mat& at(size_t i) const { fetch(); return base::at(i); }
        cv::Mat& operator[](size_t i) { fetch(); return base::operator[](i); }
        const cv::Mat& operator[](size_t i) const { fetch(); return base::operator[](i); }
        base::iterator begin() { fetch(); return base::begin(); }
        base::const_iterator begin() const { fetch(); return base::begin(); }
        base::iterator end() { fetch(); return base::end(); }
        base::const_iterator end() const { fetch(); return base::end(); }
        base::const_iterator cbegin() const { fetch(); return base::cbegin(); }
        base::const_iterator cend() const { fetch(); return base::cend(); }
        cv::Mat& front() { fetch(); return base::front(); }
        const cv::Mat& front() const { fetch(); return base::front(); }
        cv::Mat& back() { fetch(); return base::back(); }
        const cv::Mat& back() const { fetch(); return base::back(); }
        cv::Mat* data() { fetch(); return base::data(); }
        const cv::Mat* data() const { fetch(); return base::data(); }
        cv::Mat& raw(size_t i) { return base::at(i); }
        void mark_stale() { stale_ = true; }
        const orb_extractor* owner_ = nullptr;
    private:
        void fetch() const {
            if (!stale_ || !owner_) return;
            stale_ = false;
            owner_->download_pyramid(*const_cast<lazy_pyramid*>(this));
        }
        mutable bool stale_ = false;
    };
    //! Image pyramid (public in the reference, orb_extractor.h:101)
    lazy_pyramid image_pyramid_;
    //! Every level on the host, as a plain vector: the accessor for host code that binds `const std::vector<cv::Mat>&` (a reference
    //! to the base class bypasses the lazy accessors above; this one cannot be bypassed).
    const std::vector<cv::Mat>& host_pyramid() const { image_pyramid_.front(); return image_pyramid_; }
    //! Opt-in: download levels 1.. after every extract (the reference's behaviour, 0.64 MB of D2H per 640 x 480 frame), for host code
    //! that reads image_pyramid_ through a base-class reference.  Also switched on by the environment variable PLP_EAGER_PYRAMID=1.
    void set_eager_pyramid(bool on) { eager_pyramid_ = on; }

private:
    friend class lazy_pyramid;
    void download_pyramid(lazy_pyramid& pyr) const {
        for (unsigned int l = 1; l < orb_params_.num_levels_ && l < pyr.size(); ++l) {
            int32_t r = 0, c = 0;
            check(plp_orb_pyramid_level_size(ctx_, static_cast<int32_t>(l), &r, &c));
            pyr.raw(l).create(r, c, CV_8UC1);
            check(plp_orb_pyramid_host(ctx_, 0, static_cast<int32_t>(l), pyr.raw(l).data, pyr.raw(l).step));
        }
    }
    static int device_from_env() { const char* e = std::getenv("PLP_DEVICE"); return e ? std::atoi(e) : 0; }
    static void check(plp_status s) {
        if (s != PLP_OK) throw std::runtime_error(std::string("plp_front: ") + plp_strerror(s) + ": " + plp_last_error());
    }
    double get(plp_orb_param_id id) const { double v = 0; check(plp_orb_get_param(ctx_, id, &v)); return v; }
    void set(plp_orb_param_id id, double v) { check(plp_orb_set_param(ctx_, id, v)); }
    std::vector<float> table(int which) const {
        int32_t n = 0;
        check(plp_orb_get_tables(ctx_, &n, nullptr, nullptr, nullptr, nullptr, nullptr));
        std::vector<float> t[4] = {std::vector<float>(n), std::vector<float>(n), std::vector<float>(n), std::vector<float>(n)};
        check(plp_orb_get_tables(ctx_, &n, t[0].data(), t[1].data(), t[2].data(), t[3].data(), nullptr));
        return t[which];
    }
    orb_params orb_params_;
    plp_orb* ctx_ = nullptr;
    bool eager_pyramid_ = [] { const char* e = std::getenv("PLP_EAGER_PYRAMID"); return e && e[0] == '1'; }();
};

}  // namespace feature
}  // namespace PLPSLAM

#endif  // PLPSLAM_FEATURE_ORB_EXTRACTOR_H
