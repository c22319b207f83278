// TEST INFRASTRUCTURE ONLY (oracle/ref_shadow): shadows src/PLPSLAM/type.h with the same aliases over mini_eigen.hpp.
#ifndef PLPSLAM_TYPE_H
#define PLPSLAM_TYPE_H
#include <map>
#include <set>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include <Eigen/Core>
#include <opencv2/core/types.hpp>

namespace PLPSLAM {
typedef float real_t;
template <size_t R, size_t C> using MatRC_t = Eigen::Matrix<double, (int)R, (int)C>;
using Mat22_t = Eigen::Matrix2d;
using Mat33_t = Eigen::Matrix3d;
using Mat44_t = Eigen::Matrix4d;
using Mat34_t = MatRC_t<3, 4>;
template <size_t R> using VecR_t = Eigen::Matrix<double, (int)R, 1>;
using Vec2_t = Eigen::Vector2d;
using Vec3_t = Eigen::Vector3d;
using Vec4_t = Eigen::Vector4d;
using Vec6_t = VecR_t<6>;
using Quat_t = Eigen::Quaterniond;
template <typename T> using eigen_alloc_vector = std::vector<T>;
template <typename T, typename U> using eigen_alloc_map = std::map<T, U>;
template <typename T> using eigen_alloc_set = std::set<T>;
template <typename T, typename U> using eigen_alloc_unord_map = std::unordered_map<T, U>;
template <typename T> using eigen_alloc_unord_set = std::unordered_set<T>;
}  // namespace PLPSLAM
#endif
