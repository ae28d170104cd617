"""The scalar helpers behind the bit-exactness claims (csrc/tf_math.h), compiled for
the host and checked against IEEE division and exactly rounded integer powers."""
import ctypes
import os
import subprocess
from fractions import Fraction

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#define TF_DEVICE static inline
#include "tf_math.h"
extern "C" {
void div_u(const double* x, const double* d, double* out, long n) {
    for (long i = 0; i < n; ++i) out[i] = tf_div_u(x[i], d[i], 1.0 / d[i]);
}
void powi(const double* x, int e, double* out, long n) {
    for (long i = 0; i < n; ++i) out[i] = tf_powi(x[i], e);
}
}
'''


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("mathlib")
    src, so = d / "m.cpp", d / "m.so"
    src.write_text(SRC)
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off",
                    "-I", os.path.join(ROOT, "triflow_amd", "csrc"), str(src), "-o", str(so)],
                   check=True)
    return ctypes.CDLL(str(so))


def _ptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def test_uniform_division_is_ieee_division(lib):
    """tf_div_u(x, d, RN(1/d)) == x / d, bit for bit, on 4e6 random pairs over 60
    binades (the one documented exception, an all-ones significand, is excluded)."""
    rng = np.random.default_rng(0)
    n = 4_000_000
    x = rng.standard_normal(n) * 10.0 ** rng.uniform(-30, 30, n)
    d = rng.standard_normal(n) * 10.0 ** rng.uniform(-30, 30, n)
    d[d == 0] = 1.0
    out = np.empty(n)
    lib.div_u(_ptr(x), _ptr(d), _ptr(out), ctypes.c_long(n))
    assert np.array_equal(out, x / d)
    # typical stencil divisors
    for dx in (1e-4, 1 / 3, 0.1, 1e-4 ** 2, 1e-4 ** 3, 2.5e-7 ** 2, 0.5):
        dd = np.full(n, dx)
        lib.div_u(_ptr(x), _ptr(dd), _ptr(out), ctypes.c_long(n))
        assert np.array_equal(out, x / dd), dx


def test_integer_powers_are_correctly_rounded(lib):
    """tf_powi rounds the exact power once (checked with exact rational arithmetic)."""
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(0.05, 3.0, 3000), -rng.uniform(0.05, 3.0, 500),
                        np.array([1e-4, 2.5e-7, 1 / 3, 100 / 999999])])
    out = np.empty_like(x)
    for e in (3, 4, 5, 7, -2, -3):
        lib.powi(_ptr(x), ctypes.c_int(e), _ptr(out), ctypes.c_long(x.size))
        wrong = 0
        for xi, oi in zip(x, out):
            exact = Fraction(float(xi)) ** e
            # correctly rounded <=> no other double is closer to the exact value
            lo, hi = np.nextafter(oi, -np.inf), np.nextafter(oi, np.inf)
            err = abs(Fraction(float(oi)) - exact)
            if err > abs(Fraction(float(lo)) - exact) or err > abs(Fraction(float(hi)) - exact):
                wrong += 1
        assert wrong <= (2 if e < 0 else 0), (e, wrong)   # reciprocal form: double rounding at most
