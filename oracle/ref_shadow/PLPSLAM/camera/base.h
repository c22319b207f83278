// TEST INFRASTRUCTURE ONLY (oracle/ref_shadow): shadows src/PLPSLAM/camera/base.h.  The members the reference's matcher
// and grid sources read (camera/base.h:100-160) plus the two virtual reprojections they call; the geometry behind those is
// supplied by the driver (oracle/ref_driver2.cpp), which hands the matchers the same per-landmark reprojections the
// array-form oracle receives.
#ifndef PLPSLAM_CAMERA_BASE_H
#define PLPSLAM_CAMERA_BASE_H
#include <string>

#include <opencv2/core.hpp>

#include "PLPSLAM/type.h"

namespace PLPSLAM {
namespace camera {
enum class setup_type_t { Monocular = 0, Stereo = 1, RGBD = 2 };
enum class model_type_t { Perspective = 0, Fisheye = 1, Equirectangular = 2 };
struct image_bounds { float min_x_ = 0.0, max_x_ = 0.0, min_y_ = 0.0, max_y_ = 0.0; };
class base {
public:
    virtual ~base() = default;
    setup_type_t setup_type_ = setup_type_t::Monocular;
    model_type_t model_type_ = model_type_t::Perspective;
    unsigned int cols_ = 640, rows_ = 480;
    double focal_x_baseline_ = 0.0, true_baseline_ = 0.0;
    unsigned int num_grid_cols_ = 64, num_grid_rows_ = 48;
    image_bounds img_bounds_;
    double inv_cell_width_ = 0, inv_cell_height_ = 0;
    virtual bool reproject_to_image(const Mat33_t& rot_cw, const Vec3_t& trans_cw, const Vec3_t& pos_w, Vec2_t& reproj, float& x_right) const = 0;
    virtual bool reproject_to_bearing(const Mat33_t& rot_cw, const Vec3_t& trans_cw, const Vec3_t& pos_w, Vec3_t& reproj) const = 0;
};
}  // namespace camera
}  // namespace PLPSLAM
#endif
