// (float)cos((double)a), (float)sin((double)a) for a float a in [0, 8): what region growing and LBD need for every
// defined pixel / line (deliberate definition D2: single-precision libm calls are defined as the rounded f64 results).
//
// The general f64 sincos costs ~150 instructions.  Here: two-term Cody-Waite reduction to |r| <= pi/4, Taylor
// polynomials with explicit fma (relative error < 2^-44), and Ziv's rounding test: if v - |v| 2^-43 and v + |v| 2^-43 round to
// the same float, every value within the error bound -- the exact one and any libm result with < 1 ulp(f64) error -- rounds to
// that float, so it IS (float)cos((double)a).  Otherwise (about 4 arguments in a million) the caller evaluates the general
// routine.  Plain C++ (builtins only): tests/test_abi_and_model.py compiles this header for the host and checks it against
// glibc for EVERY argument the gradient kernel can produce (all (gx, gy) pairs) and for random floats.
#pragma once

#if defined(__HIPCC__)
#define PLP_HD __host__ __device__ __forceinline__
#else
#define PLP_HD static inline
#endif

namespace plp {

// returns true when both results are proven; false: use the general routine
PLP_HD bool sincos_ziv(float a, float* c_out, float* s_out) {
    const double x = (double)a;
    const double kd = __builtin_rint(x * 0.63661977236758134308);           // 2/pi
    double r = __builtin_fma(-kd, 1.57079632679489655800e+00, x);           // pi/2 hi
    r = __builtin_fma(-kd, 6.12323399573676603587e-17, r);                  // pi/2 lo
    const int k = (int)kd;
    const double r2 = r * r;
    double sp = __builtin_fma(r2, 1.0 / 6227020800.0, -1.0 / 39916800.0);
    sp = __builtin_fma(r2, sp, 1.0 / 362880.0);
    sp = __builtin_fma(r2, sp, -1.0 / 5040.0);
    sp = __builtin_fma(r2, sp, 1.0 / 120.0);
    sp = __builtin_fma(r2, sp, -1.0 / 6.0);
    sp = __builtin_fma(r * r2, sp, r);
    double cp = __builtin_fma(r2, -1.0 / 87178291200.0, 1.0 / 479001600.0);
    cp = __builtin_fma(r2, cp, -1.0 / 3628800.0);
    cp = __builtin_fma(r2, cp, 1.0 / 40320.0);
    cp = __builtin_fma(r2, cp, -1.0 / 720.0);
    cp = __builtin_fma(r2, cp, 1.0 / 24.0);
    cp = __builtin_fma(r2, cp, -0.5);
    cp = __builtin_fma(r2, cp, 1.0);
    const bool swap = k & 1;
    double c = swap ? sp : cp, s = swap ? cp : sp;
    if ((k + 1) & 2) c = -c;      // k = 1, 2 (mod 4)
    if (k & 2) s = -s;            // k = 2, 3 (mod 4)
    const double e = 1.0 / 8796093022208.0;   // 2^-43
    const float c_lo = (float)__builtin_fma(-__builtin_fabs(c), e, c), c_hi = (float)__builtin_fma(__builtin_fabs(c), e, c);
    const float s_lo = (float)__builtin_fma(-__builtin_fabs(s), e, s), s_hi = (float)__builtin_fma(__builtin_fabs(s), e, s);
    *c_out = c_lo; *s_out = s_lo;
    return c_lo == c_hi && s_lo == s_hi;
}

}  // namespace plp
