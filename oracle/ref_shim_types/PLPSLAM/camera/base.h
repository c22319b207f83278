// TEST INFRASTRUCTURE ONLY.  Stand-in for the reference's src/PLPSLAM/camera/base.h: the members the shipped facades and
// their check programs touch (camera/base.h:100-160), nothing else.
#pragma once
namespace PLPSLAM {
namespace camera {
enum class setup_type_t { Monocular = 0, Stereo = 1, RGBD = 2 };
struct image_bounds { float min_x_ = 0, max_x_ = 0, min_y_ = 0, max_y_ = 0; };
class base {
public:
    virtual ~base() = default;
    setup_type_t setup_type_ = setup_type_t::Monocular;
    unsigned int cols_ = 640, rows_ = 480;
    double true_baseline_ = 0.0;
    unsigned int num_grid_cols_ = 64, num_grid_rows_ = 48;
    image_bounds img_bounds_;
    double inv_cell_width_ = 0, inv_cell_height_ = 0;
};
}  // namespace camera
}  // namespace PLPSLAM
