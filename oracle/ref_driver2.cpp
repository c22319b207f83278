// TEST INFRASTRUCTURE ONLY (oracle/_ref).  C entry points around the REFERENCE'S OWN matcher / line translation units
//   src/PLPSLAM/match/{projection,bow_tree,fuse,robust,area,stereo}.cc, data/common.cc,
//   feature/line_extractor.cc, feature/line_descriptor/{LSDDetector_custom,binary_descriptor_custom,binary_descriptor_matcher}.cpp
// compiled unmodified from /root/reference against oracle/ref_shim (OpenCV stand-in) and oracle/ref_shadow (stand-ins of
// data::frame / keyframe / landmark / Line, camera::base, Eigen, DBoW2::FeatureVector, nlohmann::json).  Built by
// oracle/ref_build.sh into oracle/_ref/libplpref2.so.
//
// Every entry point has the SIGNATURE OF ITS oracle_* COUNTERPART in oracle/match_oracle.cpp / line_oracle.cpp /
// stereo_lbdmatch_oracle.cpp (array form), so tests/test_oracle_vs_ref.py runs the same random problems through both and
// tools/make_golden_ref.py writes the reference's answers to tests/golden/ref_match.npz.  An entry point turns the arrays
// into the objects the reference's function takes, calls it, and reads the result back from the objects it mutated.
//
// Host-side geometry in front of a search (reprojection, distance / normal gates, predict_scale_level) is an INPUT of the
// array form (reproj, valid, pred_level ...).  It reaches the reference code through the objects: a landmark's world
// position encodes its index ((4 j + k, 0, 1): k = 0 point or start point, 2 end point, 1 their mid point), poses are the
// identity, and `table_camera::reproject_to_image` returns the j-th row of the arrays.  What is pinned is therefore every
// line of the reference's search loops -- candidate windows (data/common.cc), skips, gates, best / second-best updates,
// thresholds, ratio tests, orientation check, the order in which earlier matches block later ones.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <set>
#include <vector>

#include "PLPSLAM/data/frame.h"
#include "PLPSLAM/data/keyframe.h"
#include "PLPSLAM/data/landmark.h"
#include "PLPSLAM/data/landmark_line.h"
#include "PLPSLAM/feature/line_extractor.h"
#include "PLPSLAM/match/area.h"
#include "PLPSLAM/match/bow_tree.h"
#include "PLPSLAM/match/fuse.h"
#include "PLPSLAM/match/projection.h"
#include "PLPSLAM/match/robust.h"
#include "PLPSLAM/match/stereo.h"

using namespace PLPSLAM;
typedef cv::line_descriptor::KeyLine KeyLine;
static_assert(sizeof(cv::KeyPoint) == 28, "cv::KeyPoint layout");
static_assert(sizeof(KeyLine) == 68, "KeyLine layout");

namespace {

// reprojection by table look-up (see the file header)
class table_camera : public camera::perspective {
public:
    std::vector<uint8_t> in_image[3];      // k = 0, 1 (mid point), 2
    std::vector<double> rx[3], ry[3];
    std::vector<float> xr[3];
    Vec3_t epipole;
    void resize(size_t m) { for (int k = 0; k < 3; ++k) { in_image[k].assign(m, 0); rx[k].assign(m, 0); ry[k].assign(m, 0); xr[k].assign(m, -1.f); } }
    bool reproject_to_image(const Mat33_t&, const Vec3_t&, const Vec3_t& pos_w, Vec2_t& reproj, float& x_right) const override {
        const long code = std::lround(pos_w(0));
        const size_t j = (size_t)(code / 4);
        const int k = (int)(code % 4);
        reproj(0) = rx[k].at(j); reproj(1) = ry[k].at(j); x_right = xr[k].at(j);
        return in_image[k].at(j) != 0;
    }
    bool reproject_to_bearing(const Mat33_t&, const Vec3_t&, const Vec3_t&, Vec3_t& reproj) const override { reproj = epipole; return true; }
};

void set_grid(camera::base& cam, const double* grid6) {
    cam.img_bounds_.min_x_ = (float)grid6[0]; cam.img_bounds_.min_y_ = (float)grid6[1];
    cam.inv_cell_width_ = grid6[2]; cam.inv_cell_height_ = grid6[3];
    cam.num_grid_cols_ = (unsigned)grid6[4]; cam.num_grid_rows_ = (unsigned)grid6[5];
    cam.img_bounds_.max_x_ = (float)(grid6[0] + grid6[4] / grid6[2]); cam.img_bounds_.max_y_ = (float)(grid6[1] + grid6[5] / grid6[3]);
}
cv::Mat desc_rows(const uint8_t* d, int n) {
    cv::Mat m(std::max(n, 1), 32, CV_8UC1);
    if (n > 0 && d) std::memcpy(m.data, d, (size_t)n * 32);
    return n > 0 ? m : m.rowRange(0, 0);
}
cv::Mat desc_row(const uint8_t* d) { cv::Mat m(1, 32, CV_8UC1); std::memcpy(m.data, d, 32); return m; }
Vec3_t coded(size_t j, int k) { return Vec3_t((double)(4 * j + k), 0.0, 1.0); }

// A caller's scale-factor array holds one entry per level it uses (at least Levels::have of them); the reference's vectors are made one entry longer (Levels::n) so that
// no `.at(level + 1)` of the reference can throw.  The extra entry is EXTENDED geometrically, not read from behind the caller's array (found by AddressSanitizer in round 6:
// tools/oracle_sanitized_tests.sh -- four bytes behind an eight-entry array were read, and never used).
struct Levels { int n, have; };
void assign_scale_factors(std::vector<float>& v, const float* sf, Levels L) {
    v.assign(sf, sf + L.have);
    while ((int)v.size() < L.n) v.push_back(v.size() >= 2 ? v.back() * (v[1] / v[0]) : v.back());
}
void assign_scale_factors(std::vector<float>& v, const float* sf, int n) { v.assign(sf, sf + n); }
inline int count_of(int n) { return n; }
inline int count_of(Levels L) { return L.n; }
// key points of a frame / key frame
template <class F, class NL> void fill_points(F& f, camera::base* cam, const cv::KeyPoint* kps, const uint8_t* desc, const float* x_right, int n, const float* sf, NL n_sf_) {
    const int n_sf = count_of(n_sf_);
    f.camera_ = cam; f.num_keypts_ = (unsigned)n;
    f.keypts_.assign(kps, kps + n); f.undist_keypts_ = f.keypts_;
    f.descriptors_ = desc_rows(desc, n);
    f.stereo_x_right_.assign(n, -1.f);
    if (x_right) f.stereo_x_right_.assign(x_right, x_right + n);
    assign_scale_factors(f.scale_factors_, sf, n_sf_);
    f.num_scale_levels_ = (unsigned)n_sf;
    f.assign_grid();
}
template <class F, class NL> void fill_lines(F& f, camera::base* cam, const KeyLine* kl, const uint8_t* lbd, int n, const float* sf_lsd, NL n_sf_) {
    f.camera_ = cam; f._num_keylines = (unsigned)n;
    f._keylsd.assign(kl, kl + n);
    f._lbd_descr = desc_rows(lbd, n);
    assign_scale_factors(f._scale_factors_lsd, sf_lsd, n_sf_);
    f._stereo_x_right_cooresponding_to_keylines.assign(n, std::make_pair(-1.f, -1.f));
}
// a frame slot that already holds a landmark WITH observations blocks (occupied = 1)
struct Pool {
    std::vector<std::unique_ptr<data::landmark>> lms;
    std::vector<std::unique_ptr<data::Line>> lines;
    data::landmark* lm() { lms.emplace_back(new data::landmark()); return lms.back().get(); }
    data::Line* line() { lines.emplace_back(new data::Line()); return lines.back().get(); }
};
void occupy(data::frame& f, Pool& P, const uint8_t* occupied, int n) {
    f.landmarks_.assign(n, nullptr); f.outlier_flags_.assign(n, false);
    for (int i = 0; i < n; ++i) if (occupied && occupied[i]) { auto* b = P.lm(); b->num_obs_ = 1; b->id_ = -2; f.landmarks_[i] = b; }
}
void occupy_lines(data::frame& f, Pool& P, const uint8_t* occupied, int n) {
    f._landmarks_line.assign(n, nullptr); f._outlier_flags_line.assign(n, false);
    for (int i = 0; i < n; ++i) if (occupied && occupied[i]) { auto* b = P.line(); b->num_obs_ = 1; b->id_ = -2; f._landmarks_line[i] = b; }
}
Levels n_levels_of(const int* lvl, const unsigned* ulvl, int m, int at_least) {
    int top = -1;
    for (int i = 0; i < m; ++i) top = std::max(top, lvl ? lvl[i] : (int)ulvl[i]);
    return Levels{std::max(at_least, top + 2), std::max(at_least, top + 1)};
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------------------------------- data/common.cc
int ref_get_cell_indices(float min_x, float min_y, double inv_w, double inv_h, int cols, int rows, float x, float y, int* cx, int* cy) {
    table_camera cam;
    const double g6[6] = {min_x, min_y, inv_w, inv_h, (double)cols, (double)rows};
    set_grid(cam, g6);
    cv::KeyPoint kp(x, y, 1.f);
    return data::get_cell_indices(&cam, kp, *cx, *cy) ? 1 : 0;
}
int ref_keypoints_in_cell(const double* grid6, const cv::KeyPoint* kps, int n, float ref_x, float ref_y, float margin, int min_level, int max_level, unsigned* out) {
    table_camera cam;
    set_grid(cam, grid6);
    data::frame f;
    const float one = 1.f;
    fill_points(f, &cam, kps, nullptr, nullptr, n, &one, 1);
    const auto v = f.get_keypoints_in_cell(ref_x, ref_y, margin, min_level, max_level);
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    return (int)v.size();
}
int ref_keylines_in_cell(const KeyLine* kl, int n, float x1, float y1, float x2, float y2, float margin, int min_level, int max_level, unsigned* out) {
    const std::vector<KeyLine> v(kl, kl + n);
    const auto r = data::get_keylines_in_cell(v, x1, y1, x2, y2, margin, min_level, max_level);
    for (size_t i = 0; i < r.size(); ++i) out[i] = r[i];
    return (int)r.size();
}

// ---------------------------------------------------------------------------------------- match/projection.cc
unsigned ref_match_frame_and_landmarks(const double* grid6, const cv::KeyPoint* kps, const uint8_t* desc, const float* x_right, const uint8_t* occupied, int n,
                                       const float* scale_factors, const uint8_t* lm_valid, const float* lm_reproj, const float* lm_x_right,
                                       const int* lm_level, const uint8_t* lm_desc, const uint8_t* lm_has_obs, int m, float margin, float lowe_ratio,
                                       int* kp_landmark) {
    table_camera cam;
    set_grid(cam, grid6);
    Pool P;
    data::frame frm;
    fill_points(frm, &cam, kps, desc, x_right, n, scale_factors, n_levels_of(lm_level, nullptr, m, 8));
    occupy(frm, P, occupied, n);
    std::vector<data::landmark*> local;
    for (int j = 0; j < m; ++j) {
        auto* lm = P.lm();
        lm->id_ = j;
        lm->is_observable_in_tracking_ = lm_valid[j] != 0;
        lm->scale_level_in_tracking_ = lm_level[j];
        lm->reproj_in_tracking_ = Vec2_t(lm_reproj[2 * j], lm_reproj[2 * j + 1]);
        lm->x_right_in_tracking_ = lm_x_right ? lm_x_right[j] : -1.f;
        lm->desc_ = desc_row(lm_desc + 32 * (size_t)j);
        lm->num_obs_ = lm_has_obs[j] ? 1 : 0;
        local.push_back(lm);
    }
    const match::projection matcher(lowe_ratio);
    const unsigned num = matcher.match_frame_and_landmarks(frm, local, margin);
    for (int i = 0; i < n; ++i) kp_landmark[i] = (frm.landmarks_[i] && frm.landmarks_[i]->id_ >= 0) ? frm.landmarks_[i]->id_ : -1;
    return num;
}

unsigned ref_match_frame_and_landmarks_line(const KeyLine* kl, const uint8_t* lbd, const int* kp_octave, const uint8_t* occupied, int n,
                                            const float* scale_factors_lsd, const uint8_t* lm_valid, const float* lm_sp, const float* lm_ep,
                                            const int* lm_level, const uint8_t* lm_desc, const uint8_t* lm_has_obs, int m, float margin, float lowe_ratio,
                                            int* line_landmark) {
    table_camera cam;
    Pool P;
    data::frame frm;
    fill_lines(frm, &cam, kl, lbd, n, scale_factors_lsd, n_levels_of(lm_level, nullptr, m, 2));
    occupy_lines(frm, P, occupied, n);
    frm.undist_keypts_.resize(n);                                    // read with the LINE index (projection.cc:187,192)
    for (int i = 0; i < n; ++i) frm.undist_keypts_[i].octave = kp_octave[i];
    std::vector<data::Line*> local;
    for (int j = 0; j < m; ++j) {
        auto* lm = P.line();
        lm->id_ = j;
        lm->_is_observable_in_tracking = lm_valid[j] != 0;
        lm->_scale_level_in_tracking = lm_level[j];
        lm->_reproj_in_tracking_sp = Vec2_t(lm_sp[2 * j], lm_sp[2 * j + 1]);
        lm->_reproj_in_tracking_ep = Vec2_t(lm_ep[2 * j], lm_ep[2 * j + 1]);
        lm->desc_ = desc_row(lm_desc + 32 * (size_t)j);
        lm->num_obs_ = lm_has_obs[j] ? 1 : 0;
        local.push_back(lm);
    }
    const match::projection matcher(lowe_ratio);
    const unsigned num = matcher.match_frame_and_landmarks_line(frm, local, margin);
    for (int i = 0; i < n; ++i) line_landmark[i] = (frm._landmarks_line[i] && frm._landmarks_line[i]->id_ >= 0) ? frm._landmarks_line[i]->id_ : -1;
    return num;
}

// direction: 0 neither, 1 assume_forward, 2 assume_backward -- produced here by the pose of the last frame (:232-239)
static void set_direction(camera::base& cam, data::frame& last, int direction, bool rgbd) {
    cam.setup_type_ = direction == 0 && !rgbd ? camera::setup_type_t::Monocular : (rgbd ? camera::setup_type_t::RGBD : camera::setup_type_t::Stereo);
    cam.true_baseline_ = 0.1;
    last.cam_pose_cw_ = Mat44_t::Identity();
    last.cam_pose_cw_(2, 3) = direction == 1 ? 1.0 : (direction == 2 ? -1.0 : 0.0);      // trans_lc(2) = trans_lw(2) with the current pose = identity
}

unsigned ref_match_current_and_last(const double* grid6, const cv::KeyPoint* kps, const uint8_t* desc, const float* x_right, const uint8_t* occupied, int n,
                                    const float* scale_factors, int num_levels, const uint8_t* valid, const float* reproj, const float* lx_right,
                                    const int* loctave, const float* langle, const uint8_t* ldesc, const uint8_t* l_has_obs, int m, float margin,
                                    int direction, int check_orientation, int* kp_last) {
    table_camera cam;
    set_grid(cam, grid6);
    cam.resize(m);
    Pool P;
    data::frame curr, last;
    fill_points(curr, &cam, kps, desc, x_right, n, scale_factors, num_levels);
    occupy(curr, P, occupied, n);
    set_direction(cam, last, direction, false);
    last.camera_ = &cam; last.num_keypts_ = (unsigned)m; last.num_scale_levels_ = (unsigned)num_levels;
    last.keypts_.resize(m); last.undist_keypts_.resize(m); last.landmarks_.assign(m, nullptr); last.outlier_flags_.assign(m, false);
    for (int j = 0; j < m; ++j) {
        auto* lm = P.lm();
        lm->id_ = j; lm->pos_w_ = coded(j, 0); lm->desc_ = desc_row(ldesc + 32 * (size_t)j); lm->num_obs_ = l_has_obs[j] ? 1 : 0;
        last.landmarks_[j] = lm;
        last.keypts_[j].octave = loctave[j]; last.undist_keypts_[j].octave = loctave[j];
        last.undist_keypts_[j].angle = langle ? langle[j] : 0.f;
        cam.in_image[0][j] = valid[j]; cam.rx[0][j] = reproj[2 * j]; cam.ry[0][j] = reproj[2 * j + 1]; cam.xr[0][j] = lx_right ? lx_right[j] : -1.f;
    }
    const match::projection matcher(0.75, check_orientation != 0);
    const unsigned num = matcher.match_current_and_last_frames(curr, last, margin);
    for (int i = 0; i < n; ++i) kp_last[i] = (curr.landmarks_[i] && curr.landmarks_[i]->id_ >= 0) ? curr.landmarks_[i]->id_ : -1;
    return num;
}

unsigned ref_match_current_and_last_line(const KeyLine* kl, const uint8_t* lbd, const float* xr_pair, const uint8_t* occupied, int n,
                                         const float* scale_factors_lsd, int num_levels_lsd, const uint8_t* valid, const float* sp, const float* ep,
                                         const float* lxr_sp, const float* lxr_ep, const int* loctave, const uint8_t* ldesc, const uint8_t* l_has_obs, int m,
                                         float margin, int direction, int is_rgbd, int* line_last) {
    table_camera cam;
    cam.resize(m);
    Pool P;
    data::frame curr, last;
    fill_lines(curr, &cam, kl, lbd, n, scale_factors_lsd, n_levels_of(loctave, nullptr, m, num_levels_lsd + 1));
    occupy_lines(curr, P, occupied, n);
    for (int i = 0; i < n; ++i) curr._stereo_x_right_cooresponding_to_keylines[i] = std::make_pair(xr_pair[2 * i], xr_pair[2 * i + 1]);
    set_direction(cam, last, direction, is_rgbd != 0);
    last.camera_ = &cam; last._num_keylines = (unsigned)m; last._num_scale_levels_lsd = (unsigned)num_levels_lsd;
    last._keylsd.resize(m); last._landmarks_line.assign(m, nullptr); last._outlier_flags_line.assign(m, false);
    for (int j = 0; j < m; ++j) {
        auto* lm = P.line();
        lm->id_ = j; lm->desc_ = desc_row(ldesc + 32 * (size_t)j); lm->num_obs_ = l_has_obs[j] ? 1 : 0;
        const Vec3_t a = coded(j, 0), b = coded(j, 2);
        lm->pos_w_ << a(0), a(1), a(2), b(0), b(1), b(2);
        last._landmarks_line[j] = lm;
        last._keylsd[j].octave = loctave[j];
        for (int k = 0; k < 3; ++k) cam.in_image[k][j] = valid[j];
        cam.rx[0][j] = sp[2 * j]; cam.ry[0][j] = sp[2 * j + 1]; cam.xr[0][j] = lxr_sp[j];
        cam.rx[2][j] = ep[2 * j]; cam.ry[2][j] = ep[2 * j + 1]; cam.xr[2][j] = lxr_ep[j];
    }
    const match::projection matcher(0.75, false);
    const unsigned num = matcher.match_current_and_last_frames_line(curr, last, margin);
    for (int i = 0; i < n; ++i) line_last[i] = (curr._landmarks_line[i] && curr._landmarks_line[i]->id_ >= 0) ? curr._landmarks_line[i]->id_ : -1;
    return num;
}

unsigned ref_match_frame_and_keyframe(const double* grid6, const cv::KeyPoint* kps, const uint8_t* desc, const uint8_t* occupied, int n, const float* scale_factors,
                                      const uint8_t* valid, const float* reproj, const unsigned* pred_level, const float* langle, const uint8_t* ldesc, int m,
                                      float margin, unsigned hamm_dist_thr, int check_orientation, int* kp_lm) {
    table_camera cam;
    set_grid(cam, grid6);
    cam.resize(m);
    Pool P;
    data::frame curr;
    fill_points(curr, &cam, kps, desc, nullptr, n, scale_factors, n_levels_of(nullptr, pred_level, m, 8));
    occupy(curr, P, occupied, n);
    data::keyframe kf;
    kf.camera_ = &cam; kf.landmarks_.assign(m, nullptr); kf.undist_keypts_.resize(m);
    for (int j = 0; j < m; ++j) {
        auto* lm = P.lm();
        lm->id_ = j; lm->pos_w_ = coded(j, 0); lm->desc_ = desc_row(ldesc + 32 * (size_t)j); lm->pred_level_ = pred_level[j];
        kf.landmarks_[j] = lm;
        kf.undist_keypts_[j].angle = langle ? langle[j] : 0.f;
        cam.in_image[0][j] = valid[j]; cam.rx[0][j] = reproj[2 * j]; cam.ry[0][j] = reproj[2 * j + 1];
    }
    const match::projection matcher(0.75, check_orientation != 0);
    const unsigned num = matcher.match_frame_and_keyframe(curr, &kf, std::set<data::landmark*>(), margin, hamm_dist_thr);
    for (int i = 0; i < n; ++i) kp_lm[i] = (curr.landmarks_[i] && curr.landmarks_[i]->id_ >= 0) ? curr.landmarks_[i]->id_ : -1;
    return num;
}

unsigned ref_match_frame_and_keyframe_line(const KeyLine* kl, const uint8_t* lbd, const uint8_t* occupied, int n, const float* scale_factors_lsd,
                                           const uint8_t* valid, const float* sp, const float* ep, const unsigned* pred_level, const uint8_t* ldesc, int m,
                                           float margin, unsigned hamm_dist_thr, int* line_lm) {
    table_camera cam;
    cam.resize(m);
    Pool P;
    data::frame curr;
    fill_lines(curr, &cam, kl, lbd, n, scale_factors_lsd, n_levels_of(nullptr, pred_level, m, 2));
    occupy_lines(curr, P, occupied, n);
    data::keyframe kf;
    kf.camera_ = &cam; kf.landmarks_line_.assign(m, nullptr);
    for (int j = 0; j < m; ++j) {
        auto* lm = P.line();
        lm->id_ = j; lm->desc_ = desc_row(ldesc + 32 * (size_t)j); lm->pred_level_ = pred_level[j];
        const Vec3_t a = coded(j, 0), b = coded(j, 2);
        lm->pos_w_ << a(0), a(1), a(2), b(0), b(1), b(2);
        kf.landmarks_line_[j] = lm;
        for (int k = 0; k < 3; ++k) cam.in_image[k][j] = valid[j];
        cam.rx[0][j] = sp[2 * j]; cam.ry[0][j] = sp[2 * j + 1]; cam.rx[2][j] = ep[2 * j]; cam.ry[2][j] = ep[2 * j + 1];
    }
    const match::projection matcher(0.75, false);
    const unsigned num = matcher.match_frame_and_keyframe_line(curr, &kf, std::set<data::Line*>(), margin, hamm_dist_thr);
    for (int i = 0; i < n; ++i) line_lm[i] = (curr._landmarks_line[i] && curr._landmarks_line[i]->id_ >= 0) ? curr._landmarks_line[i]->id_ : -1;
    return num;
}

unsigned ref_match_by_sim3(const double* grid6, const cv::KeyPoint* kps, const uint8_t* desc, const uint8_t* occupied, int n, const float* scale_factors,
                           const uint8_t* valid, const float* reproj, const unsigned* pred_level, const uint8_t* ldesc, int m, float margin, int* kp_lm) {
    table_camera cam;
    set_grid(cam, grid6);
    cam.resize(m);
    Pool P;
    data::keyframe kf;
    fill_points(kf, &cam, kps, desc, nullptr, n, scale_factors, n_levels_of(nullptr, pred_level, m, 8));
    std::vector<data::landmark*> matched(n, nullptr), lms;
    for (int i = 0; i < n; ++i) if (occupied && occupied[i]) { auto* b = P.lm(); b->id_ = -2; matched[i] = b; }
    for (int j = 0; j < m; ++j) {
        auto* lm = P.lm();
        lm->id_ = j; lm->pos_w_ = coded(j, 0); lm->mean_normal_ = lm->pos_w_; lm->desc_ = desc_row(ldesc + 32 * (size_t)j); lm->pred_level_ = pred_level[j];
        lms.push_back(lm);
        cam.in_image[0][j] = valid[j]; cam.rx[0][j] = reproj[2 * j]; cam.ry[0][j] = reproj[2 * j + 1];
    }
    const match::projection matcher(0.75, false);
    const unsigned num = matcher.match_by_Sim3_transform(&kf, Mat44_t::Identity(), lms, matched, margin);
    for (int i = 0; i < n; ++i) kp_lm[i] = (matched[i] && matched[i]->id_ >= 0) ? matched[i]->id_ : -1;
    return num;
}

// One direction of match_keyframes_mutually / the search of fuse::detect_duplication, through the WHOLE reference function:
//   signed_level != 0: fuse::detect_duplication with a sentinel landmark on every key point (duplicated_lms_in_keyfrm[i] = that
//                      key point's sentinel);  thr must be HAMMING_DIST_THR_LOW.
//   signed_level == 0: projection::match_keyframes_mutually with key frame 1 = the m query landmarks (key points without
//                      candidates) and key frame 2 = the n key points; its first pass is this search, its second pass finds
//                      nothing, so the search result is read from a landmark-side recorder: see ref_match_keyframes_mutually
//                      for the complete function.  Here the unsigned variant runs fuse::replace_duplication without chi-square
//                      effect (inv sigma = 0) and thr = HAMMING_DIST_THR_LOW only.
void ref_project_best(const double* grid6, const cv::KeyPoint* kps, const uint8_t* desc, int n, const float* scale_factors, const uint8_t* valid,
                      const double* reproj_d, const unsigned* pred_level, const uint8_t* ldesc, int m, float margin, unsigned thr, int signed_level,
                      int* best_idx_out) {
    table_camera cam;
    set_grid(cam, grid6);
    cam.resize(m);
    Pool P;
    data::keyframe kf;
    const Levels nl = n_levels_of(nullptr, pred_level, m, 8);
    fill_points(kf, &cam, kps, desc, nullptr, n, scale_factors, nl);
    kf.inv_level_sigma_sq_.assign(std::max(nl.n, 64), 0.f);
    kf.landmarks_.assign(n, nullptr);
    for (int i = 0; i < n; ++i) { auto* s = P.lm(); s->id_ = i; s->num_obs_ = 1000000; kf.landmarks_[i] = s; }
    std::vector<data::landmark*> lms;
    for (int j = 0; j < m; ++j) {
        auto* lm = P.lm();
        lm->id_ = j; lm->pos_w_ = coded(j, 0); lm->mean_normal_ = lm->pos_w_; lm->desc_ = desc_row(ldesc + 32 * (size_t)j); lm->pred_level_ = pred_level[j];
        lms.push_back(lm);
        cam.in_image[0][j] = valid[j]; cam.rx[0][j] = reproj_d[2 * j]; cam.ry[0][j] = reproj_d[2 * j + 1];
    }
    (void)thr;
    match::fuse fuser(0.6);
    if (signed_level) {
        std::vector<data::landmark*> dup;
        fuser.detect_duplication(&kf, Mat44_t::Identity(), lms, margin, dup);
        for (int j = 0; j < m; ++j) best_idx_out[j] = dup[j] ? dup[j]->id_ : -1;
    } else {
        fuser.replace_duplication(&kf, lms, margin);
        for (int j = 0; j < m; ++j) best_idx_out[j] = lms[j]->replaced_by_ ? lms[j]->replaced_by_->id_ : -1;
    }
}

// projection::match_keyframes_mutually (projection.cc:894-1142), complete: both passes and the cross check.
//   key frame 1: n1 key points, landmarks (lm1_valid) reprojected into key frame 2 at reproj_1in2 with pred_1in2; key frame 2 likewise.
//   out: matched_2_in_1[n1] = index in key frame 2 of the landmark assigned to key point idx_1 (-1 none); returns num_matches
unsigned ref_match_keyframes_mutually(const double* grid6, const cv::KeyPoint* kps1, const uint8_t* desc1, int n1, const cv::KeyPoint* kps2, const uint8_t* desc2,
                                      int n2, const float* scale_factors, int n_sf, const uint8_t* lm1_valid, const double* reproj_1in2,
                                      const unsigned* pred_1in2, const uint8_t* lm2_valid, const double* reproj_2in1, const unsigned* pred_2in1, float margin,
                                      int* matched_2_in_1) {
    table_camera cam;
    set_grid(cam, grid6);
    cam.resize((size_t)n1 + n2);
    Pool P;
    data::keyframe kf1, kf2;
    fill_points(kf1, &cam, kps1, desc1, nullptr, n1, scale_factors, n_sf);
    fill_points(kf2, &cam, kps2, desc2, nullptr, n2, scale_factors, n_sf);
    kf1.landmarks_.assign(n1, nullptr); kf2.landmarks_.assign(n2, nullptr);
    for (int j = 0; j < n1; ++j) {
        auto* lm = P.lm();
        lm->id_ = j; lm->pos_w_ = coded(j, 0); lm->desc_ = desc_row(desc1 + 32 * (size_t)j); lm->pred_level_ = pred_1in2[j];
        kf1.landmarks_[j] = lm;
        cam.in_image[0][j] = lm1_valid[j]; cam.rx[0][j] = reproj_1in2[2 * j]; cam.ry[0][j] = reproj_1in2[2 * j + 1];
    }
    for (int j = 0; j < n2; ++j) {
        auto* lm = P.lm();
        const size_t t = (size_t)n1 + j;
        lm->id_ = j; lm->pos_w_ = coded(t, 0); lm->desc_ = desc_row(desc2 + 32 * (size_t)j); lm->pred_level_ = pred_2in1[j];
        kf2.landmarks_[j] = lm;
        cam.in_image[0][t] = lm2_valid[j]; cam.rx[0][t] = reproj_2in1[2 * j]; cam.ry[0][t] = reproj_2in1[2 * j + 1];
    }
    std::vector<data::landmark*> matched(n1, nullptr);
    const match::projection matcher(0.75, false);
    const float s_12 = 1.f;
    const unsigned num = matcher.match_keyframes_mutually(&kf1, &kf2, matched, s_12, Mat33_t::Identity(), Vec3_t(0, 0, 0), margin);
    for (int i = 0; i < n1; ++i) matched_2_in_1[i] = matched[i] ? matched[i]->id_ : -1;
    return num;
}

// ---------------------------------------------------------------------------------------- match/bow_tree.cc
//   variant 0: bow_tree::match_frame_and_keyframe (queries = key-frame features, targets = frame features; t_skip unused)
//   variant 1: bow_tree::match_keyframes          (queries = key frame 1, targets = key frame 2; t_skip = no valid landmark in key frame 2)
static unsigned ref_match_bow_impl(int variant, const uint8_t* q_desc, const float* q_angle, const int* q_node, const uint8_t* q_valid, int m,
                                   const uint8_t* t_desc, const float* t_angle, const int* t_node, const uint8_t* t_skip, int n, float lowe_ratio,
                                   int check_orientation, int* t_match) {
    Pool P;
    data::keyframe kf;
    kf.num_keypts_ = (unsigned)m; kf.keypts_.resize(m); kf.undist_keypts_.resize(m); kf.descriptors_ = desc_rows(q_desc, m); kf.landmarks_.assign(m, nullptr);
    for (int q = 0; q < m; ++q) {
        kf.keypts_[q].angle = q_angle ? q_angle[q] : 0.f; kf.undist_keypts_[q].angle = kf.keypts_[q].angle;
        if (q_node[q] >= 0) kf.bow_feat_vec_[(unsigned)q_node[q]].push_back((unsigned)q);
        if (q_valid[q]) { auto* lm = P.lm(); lm->id_ = q; kf.landmarks_[q] = lm; }
    }
    const match::bow_tree matcher(lowe_ratio, check_orientation != 0);
    unsigned num = 0;
    for (int t = 0; t < n; ++t) t_match[t] = -1;
    if (variant == 0) {
        data::frame frm;
        frm.num_keypts_ = (unsigned)n; frm.keypts_.resize(n); frm.descriptors_ = desc_rows(t_desc, n);
        for (int t = 0; t < n; ++t) {
            frm.keypts_[t].angle = t_angle ? t_angle[t] : 0.f;
            if (t_node[t] >= 0) frm.bow_feat_vec_[(unsigned)t_node[t]].push_back((unsigned)t);
        }
        std::vector<data::landmark*> matched;
        num = matcher.match_frame_and_keyframe(&kf, frm, matched);
        for (int t = 0; t < n; ++t) if (matched[t]) t_match[t] = matched[t]->id_;
    } else {
        data::keyframe kf2;
        kf2.num_keypts_ = (unsigned)n; kf2.keypts_.resize(n); kf2.undist_keypts_.resize(n); kf2.descriptors_ = desc_rows(t_desc, n); kf2.landmarks_.assign(n, nullptr);
        for (int t = 0; t < n; ++t) {
            kf2.keypts_[t].angle = t_angle ? t_angle[t] : 0.f; kf2.undist_keypts_[t].angle = kf2.keypts_[t].angle;
            if (t_node[t] >= 0) kf2.bow_feat_vec_[(unsigned)t_node[t]].push_back((unsigned)t);
            if (!(t_skip && t_skip[t])) { auto* lm = P.lm(); lm->id_ = t; kf2.landmarks_[t] = lm; }
        }
        std::vector<data::landmark*> matched_in_1;
        num = matcher.match_keyframes(&kf, &kf2, matched_in_1);
        for (int q = 0; q < m; ++q) if (matched_in_1[q]) t_match[matched_in_1[q]->id_] = q;
    }
    return num;
}
unsigned ref_match_bow(const uint8_t* q_desc, const float* q_angle, const int* q_node, const uint8_t* q_valid, int m, const uint8_t* t_desc, const float* t_angle,
                       const int* t_node, const uint8_t* t_skip, int n, float lowe_ratio, int check_orientation, int* t_match) {
    return ref_match_bow_impl(t_skip ? 1 : 0, q_desc, q_angle, q_node, q_valid, m, t_desc, t_angle, t_node, t_skip, n, lowe_ratio, check_orientation, t_match);
}

// ---------------------------------------------------------------------------------------- match/robust.cc
unsigned ref_brute_force_match(const uint8_t* desc1, const float* angle1, int n1, const uint8_t* desc2, const float* angle2, const uint8_t* valid2, int n2,
                               float lowe_ratio, int check_orientation, int* match_2_in_1) {
    Pool P;
    data::frame frm;
    frm.num_keypts_ = (unsigned)n1; frm.keypts_.resize(n1); frm.descriptors_ = desc_rows(desc1, n1);
    for (int i = 0; i < n1; ++i) frm.keypts_[i].angle = angle1 ? angle1[i] : 0.f;
    data::keyframe kf;
    kf.num_keypts_ = (unsigned)n2; kf.keypts_.resize(n2); kf.descriptors_ = desc_rows(desc2, n2); kf.landmarks_.assign(n2, nullptr);
    for (int i = 0; i < n2; ++i) {
        kf.keypts_[i].angle = angle2 ? angle2[i] : 0.f;
        if (valid2[i]) { auto* lm = P.lm(); lm->id_ = i; kf.landmarks_[i] = lm; }
    }
    match::robust matcher(lowe_ratio, check_orientation != 0);
    std::vector<std::pair<int, int>> matches;
    const unsigned num = matcher.brute_force_match(frm, &kf, matches);
    for (int i = 0; i < n1; ++i) match_2_in_1[i] = -1;
    for (const auto& p : matches) match_2_in_1[p.first] = p.second;
    return num;
}
// robust::match_frame_and_keyframe (robust.cc:218-255) with every brute-force match an inlier of the essential-matrix RANSAC
// (oracle/ref_shadow/PLPSLAM/solve/essential_solver.h): out[n1] = key-frame index whose landmark the frame key point received
unsigned ref_robust_match_frame_and_keyframe(const uint8_t* desc1, const float* angle1, int n1, const uint8_t* desc2, const float* angle2, const uint8_t* valid2,
                                             int n2, float lowe_ratio, int check_orientation, int* match_2_in_1) {
    Pool P;
    data::frame frm;
    frm.num_keypts_ = (unsigned)n1; frm.keypts_.resize(n1); frm.descriptors_ = desc_rows(desc1, n1);
    for (int i = 0; i < n1; ++i) frm.keypts_[i].angle = angle1 ? angle1[i] : 0.f;
    data::keyframe kf;
    kf.num_keypts_ = (unsigned)n2; kf.keypts_.resize(n2); kf.descriptors_ = desc_rows(desc2, n2); kf.landmarks_.assign(n2, nullptr);
    for (int i = 0; i < n2; ++i) {
        kf.keypts_[i].angle = angle2 ? angle2[i] : 0.f;
        if (valid2[i]) { auto* lm = P.lm(); lm->id_ = i; kf.landmarks_[i] = lm; }
    }
    match::robust matcher(lowe_ratio, check_orientation != 0);
    std::vector<data::landmark*> matched;
    const unsigned num = matcher.match_frame_and_keyframe(frm, &kf, matched);
    for (int i = 0; i < n1; ++i) match_2_in_1[i] = matched[i] ? matched[i]->id_ : -1;
    return num;
}

unsigned ref_match_for_triangulation(const uint8_t* q_desc, const float* q_angle, const int* q_node, const uint8_t* q_has_lm, const float* q_x_right,
                                     const int* q_octave, const double* q_bearing, int m, const uint8_t* t_desc, const float* t_angle, const int* t_node,
                                     const uint8_t* t_has_lm, const float* t_x_right, const double* t_bearing, int n, const float* scale_factors,
                                     const double* E_12, const double* epipole, int check_orientation, int* match_2_of_q) {
    table_camera cam;
    cam.epipole = Vec3_t(epipole[0], epipole[1], epipole[2]);
    Pool P;
    data::keyframe k1, k2;
    auto fill = [&](data::keyframe& k, const uint8_t* desc, const float* angle, const int* node, const uint8_t* has_lm, const float* xr, const int* octave,
                    const double* bearing, int cnt) {
        k.camera_ = &cam; k.num_keypts_ = (unsigned)cnt; k.keypts_.resize(cnt); k.undist_keypts_.resize(cnt); k.descriptors_ = desc_rows(desc, cnt);
        k.landmarks_.assign(cnt, nullptr); k.stereo_x_right_.assign(cnt, -1.f); k.bearings_.resize(cnt);
        int max_oct = 7;
        for (int i = 0; i < cnt; ++i) {
            k.undist_keypts_[i].angle = angle ? angle[i] : 0.f;
            k.undist_keypts_[i].octave = octave ? octave[i] : 0;
            max_oct = std::max(max_oct, k.undist_keypts_[i].octave);
            if (node[i] >= 0) k.bow_feat_vec_[(unsigned)node[i]].push_back((unsigned)i);
            if (has_lm && has_lm[i]) { auto* lm = P.lm(); lm->id_ = i; k.landmarks_[i] = lm; }
            if (xr) k.stereo_x_right_[i] = xr[i];
            k.bearings_[i] = Vec3_t(bearing[3 * i], bearing[3 * i + 1], bearing[3 * i + 2]);
        }
        k.scale_factors_.assign(scale_factors, scale_factors + max_oct + 1);
    };
    fill(k1, q_desc, q_angle, q_node, q_has_lm, q_x_right, q_octave, q_bearing, m);
    fill(k2, t_desc, t_angle, t_node, t_has_lm, t_x_right, nullptr, t_bearing, n);
    Mat33_t E;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) E(i, j) = E_12[3 * i + j];
    match::robust matcher(0.75, check_orientation != 0);
    std::vector<std::pair<unsigned int, unsigned int>> pairs;
    const unsigned num = matcher.match_for_triangulation(&k1, &k2, E, pairs);
    for (int q = 0; q < m; ++q) match_2_of_q[q] = -1;
    for (const auto& p : pairs) match_2_of_q[p.first] = (int)p.second;
    return num;
}

// ---------------------------------------------------------------------------------------- match/area.cc
unsigned ref_match_area(const double* grid6, const cv::KeyPoint* kps1, const uint8_t* desc1, int n1, const cv::KeyPoint* kps2, const uint8_t* desc2, int n2,
                        float* prev_pts, int margin, float lowe_ratio, int check_orientation, int* matched_2_in_1) {
    table_camera cam;
    set_grid(cam, grid6);
    data::frame f1, f2;
    const float one = 1.f;
    fill_points(f1, &cam, kps1, desc1, nullptr, n1, &one, 1);
    fill_points(f2, &cam, kps2, desc2, nullptr, n2, &one, 1);
    std::vector<cv::Point2f> prev(n1);
    for (int i = 0; i < n1; ++i) prev[i] = cv::Point2f(prev_pts[2 * i], prev_pts[2 * i + 1]);
    std::vector<int> matched;
    match::area matcher(lowe_ratio, check_orientation != 0);
    const unsigned num = matcher.match_in_consistent_area(f1, f2, prev, matched, margin);
    for (int i = 0; i < n1; ++i) { matched_2_in_1[i] = matched[i]; prev_pts[2 * i] = prev[i].x; prev_pts[2 * i + 1] = prev[i].y; }
    return num;
}

// ---------------------------------------------------------------------------------------- match/fuse.cc
void ref_fuse_search(const double* grid6, const cv::KeyPoint* kps, const uint8_t* desc, const float* x_right, int n, const float* scale_factors,
                     const float* inv_level_sigma_sq, const uint8_t* lm_valid, const double* reproj_d, const float* lm_x_right, const unsigned* pred_level,
                     const uint8_t* lm_desc, int m, float margin, int* best_idx) {
    table_camera cam;
    set_grid(cam, grid6);
    cam.resize(m);
    Pool P;
    data::keyframe kf;
    const Levels nl = n_levels_of(nullptr, pred_level, m, 8);
    fill_points(kf, &cam, kps, desc, x_right, n, scale_factors, nl);
    kf.inv_level_sigma_sq_.assign(inv_level_sigma_sq, inv_level_sigma_sq + nl.have); kf.inv_level_sigma_sq_.resize(nl.n, kf.inv_level_sigma_sq_.back());
    kf.landmarks_.assign(n, nullptr);
    for (int i = 0; i < n; ++i) { auto* s = P.lm(); s->id_ = i; s->num_obs_ = 1000000; kf.landmarks_[i] = s; }   // sentinel: best_idx is read from lm->replace(sentinel)
    std::vector<data::landmark*> lms;
    for (int j = 0; j < m; ++j) {
        auto* lm = P.lm();
        lm->id_ = j; lm->pos_w_ = coded(j, 0); lm->mean_normal_ = lm->pos_w_; lm->desc_ = desc_row(lm_desc + 32 * (size_t)j); lm->pred_level_ = pred_level[j];
        lms.push_back(lm);
        cam.in_image[0][j] = lm_valid[j]; cam.rx[0][j] = reproj_d[2 * j]; cam.ry[0][j] = reproj_d[2 * j + 1]; cam.xr[0][j] = lm_x_right ? lm_x_right[j] : -1.f;
    }
    match::fuse fuser(0.6);
    fuser.replace_duplication(&kf, lms, margin);
    for (int j = 0; j < m; ++j) best_idx[j] = lms[j]->replaced_by_ ? lms[j]->replaced_by_->id_ : -1;
}

void ref_fuse_search_line(const KeyLine* kl, const uint8_t* lbd, int n, const float* scale_factors_lsd, const float* inv_level_sigma_sq_lsd, const uint8_t* valid,
                          const double* sp_d, const double* ep_d, const unsigned* pred_level, const uint8_t* ldesc, int m, float margin, int* best_idx_out) {
    table_camera cam;
    cam.resize(m);
    Pool P;
    data::keyframe kf;
    const Levels nl = n_levels_of(nullptr, pred_level, m, 2);
    fill_lines(kf, &cam, kl, lbd, n, scale_factors_lsd, nl);
    kf._inv_level_sigma_sq_lsd.assign(inv_level_sigma_sq_lsd, inv_level_sigma_sq_lsd + nl.have); kf._inv_level_sigma_sq_lsd.resize(nl.n, kf._inv_level_sigma_sq_lsd.back());
    kf.landmarks_line_.assign(n, nullptr);
    for (int i = 0; i < n; ++i) { auto* s = P.line(); s->id_ = i; s->num_obs_ = 1000000; kf.landmarks_line_[i] = s; }
    std::vector<data::Line*> lms;
    for (int j = 0; j < m; ++j) {
        auto* lm = P.line();
        lm->id_ = j; lm->desc_ = desc_row(ldesc + 32 * (size_t)j); lm->pred_level_ = pred_level[j];
        const Vec3_t a = coded(j, 0), b = coded(j, 2);
        lm->pos_w_ << a(0), a(1), a(2), b(0), b(1), b(2);
        lms.push_back(lm);
        for (int k = 0; k < 3; ++k) cam.in_image[k][j] = valid[j];
        cam.rx[0][j] = sp_d[2 * j]; cam.ry[0][j] = sp_d[2 * j + 1]; cam.rx[2][j] = ep_d[2 * j]; cam.ry[2][j] = ep_d[2 * j + 1];
    }
    match::fuse fuser(0.6);
    fuser.replace_duplication_line(&kf, lms, margin);
    for (int j = 0; j < m; ++j) best_idx_out[j] = lms[j]->replaced_by_ ? lms[j]->replaced_by_->id_ : -1;
}

// ---------------------------------------------------------------------------------------- match/stereo.cc
// pyramids as arrays of level images (level l: rows[l] x cols[l] bytes, dense)
void ref_stereo_compute(const uint8_t* const* left_levels, const uint8_t* const* right_levels, const int* rows, const int* cols, int n_levels,
                        const cv::KeyPoint* kl, int nl, const cv::KeyPoint* kr, int nr, const uint8_t* dl, const uint8_t* dr, const float* scale_factors,
                        const float* inv_scale_factors, float focal_x_baseline, float true_baseline, float* x_right_out, float* depth_out) {
    std::vector<cv::Mat> lp, rp;
    for (int l = 0; l < n_levels; ++l) {
        lp.emplace_back(rows[l], cols[l], CV_8UC1, const_cast<uint8_t*>(left_levels[l]));
        rp.emplace_back(rows[l], cols[l], CV_8UC1, const_cast<uint8_t*>(right_levels[l]));
    }
    const std::vector<cv::KeyPoint> kpl(kl, kl + nl), kpr(kr, kr + nr);
    const cv::Mat descl = desc_rows(dl, nl), descr = desc_rows(dr, nr);
    const std::vector<float> sf(scale_factors, scale_factors + n_levels), isf(inv_scale_factors, inv_scale_factors + n_levels);
    const match::stereo matcher(lp, rp, kpl, kpr, descl, descr, sf, isf, focal_x_baseline, true_baseline);
    std::vector<float> xr, dp;
    matcher.compute(xr, dp);
    for (int i = 0; i < nl; ++i) { x_right_out[i] = xr[i]; depth_out[i] = dp[i]; }
}

// ---------------------------------------------------------------------------------------- feature/line_descriptor, feature/line_extractor.cc
// BinaryDescriptorMatcher::match(query, train, matches) (binary_descriptor_matcher.cpp:197-255)
void ref_lbd_match_1nn(const uint8_t* q, int nq, const uint8_t* t, int nt, int* train_idx, int* dist) {
    const cv::Mat qm = desc_rows(q, nq), tm = desc_rows(t, nt);
    auto bdm = cv::line_descriptor::BinaryDescriptorMatcher::createBinaryDescriptorMatcher();
    std::vector<cv::DMatch> matches;
    bdm->match(qm, tm, matches);
    for (int i = 0; i < nq; ++i) { train_idx[i] = -1; dist[i] = -1; }
    for (const auto& mt : matches) { train_idx[mt.queryIdx] = mt.trainIdx; dist[mt.queryIdx] = (int)mt.distance; }
}

// BinaryDescriptor::compute on given key lines: binary (32 B) and float (72 x f32) descriptors
void ref_lbd(const uint8_t* img, int rows, int cols, const KeyLine* kls, int n, uint8_t* out32, float* out72) {
    const cv::Mat image(rows, cols, CV_8UC1, const_cast<uint8_t*>(img));
    std::vector<KeyLine> v(kls, kls + n);
    cv::Mat d8, d32;
    auto bd = cv::line_descriptor::BinaryDescriptor::createBinaryDescriptor();
    bd->compute(image, v, d8);
    for (int i = 0; i < n; ++i) std::memcpy(out32 + 32 * (size_t)i, d8.ptr(i), 32);
    if (out72) {
        auto bd2 = cv::line_descriptor::BinaryDescriptor::createBinaryDescriptor();
        bd2->compute(image, v, d32, true);
        for (int i = 0; i < n; ++i) std::memcpy(out72 + 72 * (size_t)i, d32.ptr(i), 72 * 4);
    }
}

// LineFeatureTracker(camera).extract_LSD_LBD (line_extractor.cc:88-160): returns the number of kept lines; kl / lbd / linefn hold `cap`
int ref_line_extract(const uint8_t* img, int rows, int cols, double fx, double fy, double cx, double cy, KeyLine* kl, uint8_t* lbd, double* linefn, int cap) {
    camera::perspective cam;
    cam.fx_ = fx; cam.fy_ = fy; cam.cx_ = cx; cam.cy_ = cy; cam.cols_ = (unsigned)cols; cam.rows_ = (unsigned)rows;
    feature::LineFeatureTracker tracker(&cam);
    const cv::Mat image(rows, cols, CV_8UC1, const_cast<uint8_t*>(img));
    std::vector<KeyLine> keylsd;
    cv::Mat descr;
    std::vector<Vec3_t> fn;
    tracker.extract_LSD_LBD(image, keylsd, descr, fn);
    const int n = (int)keylsd.size();
    if (n > cap) return -n;
    for (int i = 0; i < n; ++i) {
        kl[i] = keylsd[i];
        std::memcpy(lbd + 32 * (size_t)i, descr.ptr(i), 32);
        linefn[3 * i] = fn[i](0); linefn[3 * i + 1] = fn[i](1); linefn[3 * i + 2] = fn[i](2);
    }
    return n;
}

// LSDDetectorC::detect(image, keylines, scale 2, 1 octave, opts) as line_extractor.cc:113-127 calls it: all key lines before the length-60 filter
int ref_lsd_keylines(const uint8_t* img, int rows, int cols, KeyLine* kl, int cap) {
    const cv::Mat image(rows, cols, CV_8UC1, const_cast<uint8_t*>(img));
    auto det = cv::line_descriptor::LSDDetectorC::createLSDDetectorC();
    cv::line_descriptor::LSDDetectorC::LSDOptions opts;
    opts.refine = 1; opts.scale = 0.5; opts.sigma_scale = 0.6; opts.quant = 2.0; opts.ang_th = 22.5; opts.log_eps = 1.0; opts.density_th = 0.6; opts.n_bins = 1024;
    opts.min_length = 0.125 * std::min(cols, rows);
    std::vector<KeyLine> v;
    det->detect(image, v, 2, 1, opts);
    if ((int)v.size() > cap) return -(int)v.size();
    for (size_t i = 0; i < v.size(); ++i) kl[i] = v[i];
    return (int)v.size();
}

}  // extern "C"
