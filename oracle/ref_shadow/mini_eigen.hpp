// TEST INFRASTRUCTURE ONLY (oracle/_ref).  The build image has no Eigen.  This is the small part of Eigen's fixed-size
// dense API that the reference's matcher / line-extractor sources use (src/PLPSLAM/match/*.cc, feature/line_extractor.cc,
// type.h): Matrix<double, R, C> with (), <<-comma initialisation, + - * /, transpose, 3x3 inverse, dot, cross, norm,
// block<R, C>, head / tail, Identity / Zero.  Products and sums are evaluated element by element in the textbook order
// (row times column, left to right), as Eigen does for these sizes without vectorisation-dependent reassociation.
#pragma once
#include <cmath>
#include <cstddef>
#include <initializer_list>
#include <memory>
#include <stdexcept>

namespace Eigen {

template <typename S, int R, int C> class Matrix;

template <typename S, int R, int C> class CommaInit {
public:
    CommaInit(Matrix<S, R, C>& m) : m_(m), k_(0) {}
    CommaInit& operator,(S v) { m_.raw()[k_++] = v; return *this; }
    template <int R2, int C2> CommaInit& operator,(const Matrix<S, R2, C2>& o) { for (int i = 0; i < R2 * C2; ++i) m_.raw()[k_++] = o.raw()[i]; return *this; }
    CommaInit& first(S v) { return (*this, v); }
private:
    Matrix<S, R, C>& m_;
    int k_;
};

// row-major storage; vectors are C == 1
template <typename S, int R, int C> class Matrix {
public:
    Matrix() { for (int i = 0; i < R * C; ++i) d_[i] = S(0); }
    explicit Matrix(const S* p) { for (int i = 0; i < R * C; ++i) d_[i] = p[i]; }
    Matrix(S x, S y) { static_assert(R * C == 2, ""); d_[0] = x; d_[1] = y; }
    Matrix(S x, S y, S z) { static_assert(R * C == 3, ""); d_[0] = x; d_[1] = y; d_[2] = z; }
    Matrix(S x, S y, S z, S w) { static_assert(R * C == 4, ""); d_[0] = x; d_[1] = y; d_[2] = z; d_[3] = w; }
    S* raw() { return d_; }
    const S* raw() const { return d_; }
    S& operator()(int i) { static_assert(R == 1 || C == 1, "vector access"); return d_[i]; }
    const S& operator()(int i) const { static_assert(R == 1 || C == 1, "vector access"); return d_[i]; }
    S& operator[](int i) { return d_[i]; }
    const S& operator[](int i) const { return d_[i]; }
    S& operator()(int i, int j) { return d_[i * C + j]; }
    const S& operator()(int i, int j) const { return d_[i * C + j]; }
    S x() const { return d_[0]; }
    S y() const { return d_[1]; }
    S z() const { return d_[2]; }
    static Matrix Zero() { return Matrix(); }
    static Matrix Identity() { Matrix m; for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = S(1); return m; }
    void setZero() { *this = Matrix(); }
    void setIdentity() { *this = Identity(); }
    CommaInit<S, R, C> operator<<(S v) { CommaInit<S, R, C> ci(*this); ci.first(v); return ci; }
    template <int R2, int C2> CommaInit<S, R, C> operator<<(const Matrix<S, R2, C2>& o) { CommaInit<S, R, C> ci(*this); (ci, o); return ci; }
    Matrix operator+(const Matrix& o) const { Matrix r; for (int i = 0; i < R * C; ++i) r.d_[i] = d_[i] + o.d_[i]; return r; }
    Matrix operator-(const Matrix& o) const { Matrix r; for (int i = 0; i < R * C; ++i) r.d_[i] = d_[i] - o.d_[i]; return r; }
    Matrix operator-() const { Matrix r; for (int i = 0; i < R * C; ++i) r.d_[i] = -d_[i]; return r; }
    Matrix operator*(S s) const { Matrix r; for (int i = 0; i < R * C; ++i) r.d_[i] = d_[i] * s; return r; }
    Matrix operator/(S s) const { Matrix r; for (int i = 0; i < R * C; ++i) r.d_[i] = d_[i] / s; return r; }
    Matrix& operator+=(const Matrix& o) { for (int i = 0; i < R * C; ++i) d_[i] += o.d_[i]; return *this; }
    Matrix& operator-=(const Matrix& o) { for (int i = 0; i < R * C; ++i) d_[i] -= o.d_[i]; return *this; }
    Matrix& operator*=(S s) { for (int i = 0; i < R * C; ++i) d_[i] *= s; return *this; }
    Matrix& operator/=(S s) { for (int i = 0; i < R * C; ++i) d_[i] /= s; return *this; }
    template <int C2> Matrix<S, R, C2> operator*(const Matrix<S, C, C2>& o) const {
        Matrix<S, R, C2> r;
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < C2; ++j) {
                S acc = (*this)(i, 0) * o(0, j);
                for (int k = 1; k < C; ++k) acc += (*this)(i, k) * o(k, j);
                r(i, j) = acc;
            }
        return r;
    }
    Matrix<S, C, R> transpose() const { Matrix<S, C, R> r; for (int i = 0; i < R; ++i) for (int j = 0; j < C; ++j) r(j, i) = (*this)(i, j); return r; }
    S dot(const Matrix& o) const { S acc = d_[0] * o.d_[0]; for (int i = 1; i < R * C; ++i) acc += d_[i] * o.d_[i]; return acc; }
    S squaredNorm() const { return dot(*this); }
    S norm() const { return std::sqrt(squaredNorm()); }
    Matrix normalized() const { return *this / norm(); }
    void normalize() { *this = normalized(); }
    Matrix cross(const Matrix& o) const {
        static_assert(R * C == 3, "");
        return Matrix(d_[1] * o.d_[2] - d_[2] * o.d_[1], d_[2] * o.d_[0] - d_[0] * o.d_[2], d_[0] * o.d_[1] - d_[1] * o.d_[0]);
    }
    Matrix inverse() const {   // cofactors times 1 / det (3 x 3)
        static_assert(R == 3 && C == 3, "");
        const Matrix& m = *this;
        Matrix r;
        r(0, 0) = m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1); r(0, 1) = m(0, 2) * m(2, 1) - m(0, 1) * m(2, 2); r(0, 2) = m(0, 1) * m(1, 2) - m(0, 2) * m(1, 1);
        r(1, 0) = m(1, 2) * m(2, 0) - m(1, 0) * m(2, 2); r(1, 1) = m(0, 0) * m(2, 2) - m(0, 2) * m(2, 0); r(1, 2) = m(0, 2) * m(1, 0) - m(0, 0) * m(1, 2);
        r(2, 0) = m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0); r(2, 1) = m(0, 1) * m(2, 0) - m(0, 0) * m(2, 1); r(2, 2) = m(0, 0) * m(1, 1) - m(0, 1) * m(1, 0);
        const S det = m(0, 0) * r(0, 0) + m(0, 1) * r(1, 0) + m(0, 2) * r(2, 0);
        return r * (S(1) / det);
    }
    // a 1 x N row block converts to a vector where the reference dots two rows (fuse.cc:47, projection.cc:788)
    template <int R2, int C2> Matrix<S, (R2 == 1 ? C2 : R2), (R2 == 1 ? 1 : C2)> block(int i0, int j0) const {
        Matrix<S, (R2 == 1 ? C2 : R2), (R2 == 1 ? 1 : C2)> r;
        for (int i = 0; i < R2; ++i) for (int j = 0; j < C2; ++j) r.raw()[i * C2 + j] = (*this)(i0 + i, j0 + j);
        return r;
    }
    template <int R2, int C2> void set_block(int i0, int j0, const Matrix<S, R2, C2>& b) { for (int i = 0; i < R2; ++i) for (int j = 0; j < C2; ++j) (*this)(i0 + i, j0 + j) = b(i, j); }
    template <int N> Matrix<S, N, 1> head() const { Matrix<S, N, 1> r; for (int i = 0; i < N; ++i) r(i) = d_[i]; return r; }
    template <int N> Matrix<S, N, 1> tail() const { Matrix<S, N, 1> r; for (int i = 0; i < N; ++i) r(i) = d_[R * C - N + i]; return r; }
    Matrix<S, 3, 1> head(int n) const { (void)n; return head<3>(); }      // pos_w.head(3) / .tail(3) of a 6-vector (fuse.cc:360-361)
    Matrix<S, 3, 1> tail(int n) const { (void)n; return tail<3>(); }
    template <typename T> Matrix<T, R, C> cast() const { Matrix<T, R, C> r; for (int i = 0; i < R * C; ++i) r.raw()[i] = (T)d_[i]; return r; }
    bool operator==(const Matrix& o) const { for (int i = 0; i < R * C; ++i) if (d_[i] != o.d_[i]) return false; return true; }
private:
    S d_[R * C];
};
template <typename S, int R, int C> inline Matrix<S, R, C> operator*(S s, const Matrix<S, R, C>& m) { return m * s; }
template <int R, int C> inline Matrix<double, R, C> operator*(float s, const Matrix<double, R, C>& m) { return m * (double)s; }   // s_12 * rot_12 (projection.cc:906)
template <int R, int C> inline Matrix<double, R, C> operator*(int s, const Matrix<double, R, C>& m) { return m * (double)s; }

typedef Matrix<double, 2, 2> Matrix2d;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 4, 1> Vector4d;
template <typename T> using aligned_allocator = std::allocator<T>;

}  // namespace Eigen

// Quaternion: only named by the map-database JSON helpers at the top of data/common.cc, never run by oracle/_ref
namespace Eigen {
struct Quaterniond {
    double q[4] = {0, 0, 0, 1};
    Quaterniond() = default;
    explicit Quaterniond(const Matrix3d&) { throw std::logic_error("oracle/ref_shadow: Quaternion is not on the pinned path"); }
    explicit Quaterniond(const double* p) { for (int i = 0; i < 4; ++i) q[i] = p[i]; }
    double x() const { return q[0]; } double y() const { return q[1]; } double z() const { return q[2]; } double w() const { return q[3]; }
    Matrix3d toRotationMatrix() const { throw std::logic_error("oracle/ref_shadow: Quaternion is not on the pinned path"); }
};
}  // namespace Eigen
