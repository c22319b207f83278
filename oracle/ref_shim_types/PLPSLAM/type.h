// TEST INFRASTRUCTURE ONLY.  Stand-in for the reference's src/PLPSLAM/type.h (Eigen aliases) so that the shipped matcher
// facade can be compiled and run in an image without Eigen: just enough of Matrix3d / Vector3d / Vector2d / Matrix4d for
// the expressions projection.cc:220-231 and :389-402 use (block<3,3>, block<3,1>, transpose, unary minus, M * v, v + v,
// s * v, v(i), Vec6 head<3> / tail<3>).
#pragma once
namespace PLPSLAM {
struct Vec2_t { double v[2] = {0, 0}; double& operator()(int i) { return v[i]; } double operator()(int i) const { return v[i]; } };
struct Vec3_t {
    double v[3] = {0, 0, 0};
    Vec3_t() = default;
    Vec3_t(double x, double y, double z) : v{x, y, z} {}
    double& operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
    Vec3_t operator+(const Vec3_t& o) const { Vec3_t r; for (int i = 0; i < 3; ++i) r.v[i] = v[i] + o.v[i]; return r; }
    Vec3_t operator-(const Vec3_t& o) const { Vec3_t r; for (int i = 0; i < 3; ++i) r.v[i] = v[i] - o.v[i]; return r; }
    Vec3_t operator/(double a) const { Vec3_t r; for (int i = 0; i < 3; ++i) r.v[i] = v[i] / a; return r; }
    double dot(const Vec3_t& o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
    double norm() const { return __builtin_sqrt(dot(*this)); }
};
inline Vec3_t operator*(double a, const Vec3_t& x) { Vec3_t r; for (int i = 0; i < 3; ++i) r.v[i] = a * x.v[i]; return r; }
struct Vec6_t {
    double v[6] = {0, 0, 0, 0, 0, 0};
    double& operator()(int i) { return v[i]; }
    template <int N> Vec3_t head() const { static_assert(N == 3, "head<3> only"); Vec3_t r; for (int i = 0; i < 3; ++i) r.v[i] = v[i]; return r; }
    template <int N> Vec3_t tail() const { static_assert(N == 3, "tail<3> only"); Vec3_t r; for (int i = 0; i < 3; ++i) r.v[i] = v[3 + i]; return r; }
    Vec3_t head(int) const { return head<3>(); }      // pos_w.head(3) / pos_w.tail(3) (fuse.cc:357-358)
    Vec3_t tail(int) const { return tail<3>(); }
};
struct Mat33_t {
    double m[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    double& operator()(int i, int j) { return m[i][j]; }
    double operator()(int i, int j) const { return m[i][j]; }
    Mat33_t transpose() const { Mat33_t r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = m[j][i]; return r; }
    Mat33_t operator*(const Mat33_t& o) const {
        Mat33_t r;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = m[i][0] * o.m[0][j] + m[i][1] * o.m[1][j] + m[i][2] * o.m[2][j];
        return r;
    }
    Mat33_t operator/(double a) const { Mat33_t r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = m[i][j] / a; return r; }
    template <int R, int C> Vec3_t block(int i0, int j0) const {      // block<1, 3>(i, 0): a row (projection.cc:786)
        static_assert(R == 1 && C == 3, "block<1, 3> only");
        Vec3_t r; for (int j = 0; j < 3; ++j) r.v[j] = m[i0][j0 + j]; return r;
    }
    Mat33_t operator-() const { Mat33_t r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = -m[i][j]; return r; }
    Vec3_t operator*(const Vec3_t& x) const {
        Vec3_t r;
        for (int i = 0; i < 3; ++i) r.v[i] = m[i][0] * x.v[0] + m[i][1] * x.v[1] + m[i][2] * x.v[2];
        return r;
    }
};
inline Mat33_t operator*(double a, const Mat33_t& x) { Mat33_t r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = a * x.m[i][j]; return r; }
namespace type_detail {
template <int R, int C> struct block_t;
template <> struct block_t<3, 3> { using type = Mat33_t; };
template <> struct block_t<3, 1> { using type = Vec3_t; };
}  // namespace type_detail
struct Mat44_t {
    double m[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    double& operator()(int i, int j) { return m[i][j]; }
    template <int R, int C> typename type_detail::block_t<R, C>::type block(int i0, int j0) const {
        typename type_detail::block_t<R, C>::type r;
        if constexpr (C == 3) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = m[i0 + i][j0 + j]; }
        else { for (int i = 0; i < 3; ++i) r.v[i] = m[i0 + i][j0]; }
        return r;
    }
};
}  // namespace PLPSLAM
