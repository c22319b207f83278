// TEST INFRASTRUCTURE ONLY (oracle/ref_shadow): shadows src/PLPSLAM/camera/perspective.h -- the intrinsics that
// feature/line_extractor.cc:42-48 reads, and the pinhole reprojection of camera/perspective.cc:190-209.
#ifndef PLPSLAM_CAMERA_PERSPECTIVE_H
#define PLPSLAM_CAMERA_PERSPECTIVE_H
#include "PLPSLAM/camera/base.h"
namespace PLPSLAM {
namespace camera {
class perspective : public base {
public:
    double fx_ = 500, fy_ = 500, cx_ = 320, cy_ = 240;
    bool reproject_to_image(const Mat33_t& rot_cw, const Vec3_t& trans_cw, const Vec3_t& pos_w, Vec2_t& reproj, float& x_right) const override {
        const Vec3_t pos_c = rot_cw * pos_w + trans_cw;
        if (pos_c(2) <= 0.0) return false;
        const double z_inv = 1.0 / pos_c(2);
        reproj(0) = fx_ * pos_c(0) * z_inv + cx_;
        reproj(1) = fy_ * pos_c(1) * z_inv + cy_;
        x_right = reproj(0) - focal_x_baseline_ * z_inv;
        if (reproj(0) < img_bounds_.min_x_ || reproj(0) > img_bounds_.max_x_) return false;
        if (reproj(1) < img_bounds_.min_y_ || reproj(1) > img_bounds_.max_y_) return false;
        return true;
    }
    bool reproject_to_bearing(const Mat33_t& rot_cw, const Vec3_t& trans_cw, const Vec3_t& pos_w, Vec3_t& reproj) const override {
        reproj = rot_cw * pos_w + trans_cw;
        if (reproj(2) <= 0.0) return false;
        reproj.normalize();
        return true;
    }
};
}  // namespace camera
}  // namespace PLPSLAM
#endif
