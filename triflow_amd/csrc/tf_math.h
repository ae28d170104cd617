// Scalar arithmetic helpers used by the generated stencil bodies and the kernel
// skeletons.  Included before the generated per-model code.
#pragma once
#ifndef TF_DEVICE
#error "define TF_DEVICE before including tf_math.h"
#endif
// streaming store of write-once data (Jacobian planes, F): bypasses cache
// allocation on the device, plain store in the host build
#ifndef TF_STORE_STREAM
#define TF_STORE_STREAM(ptr, val) (*(ptr) = (val))
#endif
#ifndef TF_DEVICE_M          // qualifier of in-class functions
#define TF_DEVICE_M TF_DEVICE
#endif

// ---------------------------------------------------------------- arithmetic
// The stencil bodies are compiled with -ffp-contract=off and mirror the
// operation tree NumPy evaluates, so F and J agree bit for bit with the
// reference's numpy-compiler path for + - * / sqrt and integer powers.
TF_DEVICE double tf_sq(double a) { return a * a; }

// correctly rounded (up to double rounding, ~2^-50) small integer powers via
// double-double products -- what a correctly rounded libm pow() returns
TF_DEVICE double tf_powi(double x, int n) {
    bool neg = n < 0;
    if (neg) n = -n;
    double hi = 1.0, lo = 0.0;
    for (int k = 0; k < n; ++k) {
        double p = hi * x;
        double e = __builtin_fma(hi, x, -p);
        e = __builtin_fma(lo, x, e);
        double s = p + e;
        lo = e - (s - p);
        hi = s;
    }
    if (!neg) return hi + lo;
    // 1 / (hi + lo), corrected with the exact residual of the rounded quotient
    double q = 1.0 / hi;
    double r = __builtin_fma(-hi, q, 1.0);
    r = __builtin_fma(-lo, q, r);
    return __builtin_fma(r, q, q);
}
// x / d for a divisor d that is the same for every node of a thread (dx, dx**2,
// scalar parameters): rd = RN(1/d) is computed once per thread, the quotient is
//   q = RN(x*rd);  r = x - q*d (exact, FMA);  q' = RN(q + r*rd)
// which is the correctly rounded x/d (Markstein's theorem) for finite operands in
// the normal range, the one exception being a divisor whose significand is all
// ones (probability 2^-52 for a grid spacing).  Replaces the ~35-instruction fp64
// division expansion by 3 instructions without changing a bit of the result
// (checked against NumPy on every F/J parity vector).  Non-finite inputs give
// NaN where IEEE division would give inf: the run has failed either way.
TF_DEVICE double tf_div_u(double x, double d, double rd) {
    const double q = x * rd;
    const double r = __builtin_fma(-q, d, x);
    return __builtin_fma(r, rd, q);
}
TF_DEVICE double tf_max(double a, double b) { return (a > b || a != a) ? a : b; }   // np.maximum
TF_DEVICE double tf_min(double a, double b) { return (a < b || a != a) ? a : b; }   // np.minimum
TF_DEVICE double tf_sign(double a) { return a > 0 ? 1.0 : (a < 0 ? -1.0 : a); }
TF_DEVICE double tf_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
TF_DEVICE double tf_abs(double a) { return __builtin_fabs(a); }
TF_DEVICE bool tf_finite(double a) { return (a - a) == 0.0; }

