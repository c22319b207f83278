// oracle/_ref shim of opencv2/core/eigen.hpp: the two conversions feature/line_extractor.cc:58,84 calls
#pragma once
#include "../../cvshim.hpp"
#include <Eigen/Core>
namespace cv {
template <typename S, int R, int C> inline void cv2eigen(const Mat& src, Eigen::Matrix<S, R, C>& dst) {
    for (int i = 0; i < R; ++i) for (int j = 0; j < C; ++j) dst(i, j) = (S)src.get(i, j);
}
template <typename S, int R, int C> inline void eigen2cv(const Eigen::Matrix<S, R, C>& src, Mat& dst) {
    dst.create(R, C, depth_of<S>::value);
    for (int i = 0; i < R; ++i) for (int j = 0; j < C; ++j) dst.at<S>(i, j) = src(i, j);
}
}  // namespace cv
