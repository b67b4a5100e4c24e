"""The short elementary-function sequences of the device integrands (ssmtoybox_amd/csrc/ssmq_math.h: sincos_nr, atan2_nr)
compiled for the HOST with g++ and compared with libm in extended precision over the ranges the state-space models
produce (ssmod.py: np.sin / np.cos / np.arctan2 on sigma points) - the header is plain C++ apart from the reciprocal seed,
which the host build emulates at the accuracy measured on gfx950 (2^-24).  No GPU needed."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "ssmq_math.h"
extern "C" {
void t_sincos(const double *x, long n, double *s, double *c) { for (long i = 0; i < n; ++i) ssmq::sincos_nr(x[i], s + i, c + i); }
void t_atan2(const double *y, const double *x, long n, double *a) { for (long i = 0; i < n; ++i) a[i] = ssmq::atan2_nr(y[i], x[i]); }
}
'''


@pytest.fixture(scope='module')
def mathlib(tmp_path_factory):
    d = tmp_path_factory.mktemp('mathseq')
    (d / 't.cpp').write_text(SRC)
    so = str(d / 'libmathseq.so')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-ffp-contract=off',
                           '-I', os.path.join(ROOT, 'ssmtoybox_amd', 'csrc'), str(d / 't.cpp'), '-o', so])
    lib = ctypes.CDLL(so)
    dp = ctypes.POINTER(ctypes.c_double)
    lib.t_sincos.argtypes = [dp, ctypes.c_long, dp, dp]
    lib.t_atan2.argtypes = [dp, dp, ctypes.c_long, dp]
    return lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _ulps(got, want):
    """|got - want| in units of the last place of `want` (long double reference)."""
    want = np.asarray(want, dtype=np.longdouble)
    e = np.frexp(np.abs(want).astype(np.float64))[1]
    ulp = np.ldexp(1.0, e - 53)
    return np.abs(got.astype(np.longdouble) - want).astype(np.float64) / np.where(ulp > 0, ulp, 1.0)


def sincos(lib, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    s, c = np.empty_like(x), np.empty_like(x)
    lib.t_sincos(_p(x), x.size, _p(s), _p(c))
    return s, c


def atan2(lib, y, x):
    y, x = np.ascontiguousarray(y, dtype=np.float64), np.ascontiguousarray(x, dtype=np.float64)
    a = np.empty_like(x)
    lib.t_atan2(_p(y), _p(x), x.size, _p(a))
    return a


def test_sincos_within_a_few_ulp_of_libm(mathlib):
    rng = np.random.default_rng(0)
    for scale in (1e-3, 1.0, 10.0, 1e3, 1e5):
        x = (2 * rng.random(400000) - 1) * scale
        s, c = sincos(mathlib, x)
        xl = x.astype(np.longdouble)
        assert _ulps(s, np.sin(xl)).max() < 3.0 and _ulps(c, np.cos(xl)).max() < 3.0, scale
    # next to the multiples of pi / 2, where the reduction cancels
    k = np.arange(-3000, 3001)
    x = np.concatenate([np.nextafter(k * (np.pi / 2), np.inf), np.nextafter(k * (np.pi / 2), -np.inf), k * (np.pi / 2)])
    s, c = sincos(mathlib, x)
    xl = x.astype(np.longdouble)
    assert _ulps(s, np.sin(xl)).max() < 3.0 and _ulps(c, np.cos(xl)).max() < 3.0


def test_sincos_absolute_error_for_large_angles(mathlib):
    """Beyond 1e5 the relative accuracy near the zeros goes, the absolute one stays (the header's stated domain)."""
    rng = np.random.default_rng(1)
    x = (2 * rng.random(400000) - 1) * 1e9
    s, c = sincos(mathlib, x)
    xl = x.astype(np.longdouble)
    assert np.abs(s - np.sin(xl)).max() < 1e-12 and np.abs(c - np.cos(xl)).max() < 1e-12


def test_sincos_propagates_nan_and_inf(mathlib):
    s, c = sincos(mathlib, np.array([np.nan, np.inf, -np.inf]))
    assert np.isnan(s).all() and np.isnan(c).all()
    # out of the stated domain (|x| > 2^30: an overflowed quadrant count would be undefined behaviour and garbage): NaN, so that
    # a diverged state fails the next factorisation visibly; the largest in-domain angles still come out right
    s, c = sincos(mathlib, np.array([1.05e9, -3e9, 2.0 ** 31 * 1.6, 1e300, -1e18]))
    assert np.isnan(s[1:]).all() and np.isnan(c[1:]).all() and not np.isnan(s[0])
    edge = np.array([2.0 ** 30, -(2.0 ** 30), 2.0 ** 30 - 0.5])
    s, c = sincos(mathlib, edge)
    assert np.abs(s - np.sin(edge.astype(np.longdouble))).max() < 1e-12 and np.abs(c - np.cos(edge.astype(np.longdouble))).max() < 1e-12
    s, c = sincos(mathlib, np.array([0.0, -0.0]))
    assert np.array_equal(s, [0.0, 0.0]) and np.array_equal(c, [1.0, 1.0])


def test_atan2_within_two_ulp_of_libm(mathlib):
    rng = np.random.default_rng(2)
    for scale in (1e-6, 1.0, 6500.0, 1e8):                  # 6500: the reentry model's radar geometry (ssmod.py:1227-1252)
        for ratio in (1e-3, 1.0, 1e3):
            x = (2 * rng.random(300000) - 1) * scale
            y = (2 * rng.random(300000) - 1) * scale * ratio
            a = atan2(mathlib, y, x)
            assert _ulps(a, np.arctan2(y.astype(np.longdouble), x.astype(np.longdouble))).max() < 2.5, (scale, ratio)
    # octant boundaries and the tan(pi / 8) switch
    t = np.linspace(0, 2 * np.pi, 100001)
    y, x = np.sin(t), np.cos(t)
    assert _ulps(atan2(mathlib, y, x), np.arctan2(y.astype(np.longdouble), x.astype(np.longdouble))).max() < 2.5


def test_atan2_zeros_signs_and_nan(mathlib):
    y = np.array([0.0, -0.0, 0.0, -0.0, 0.0, -0.0, 1.0, -1.0, 2.0, -3.0])
    x = np.array([0.0, 0.0, -0.0, -0.0, -1.0, -1.0, 0.0, 0.0, -0.0, 0.0])
    got, want = atan2(mathlib, y, x), np.arctan2(y, x)
    assert np.allclose(got, want, rtol=4e-16, atol=0) and np.array_equal(np.signbit(got) | (got == 0), np.signbit(want) | (want == 0))
    assert np.isnan(atan2(mathlib, np.array([np.nan, 1.0, np.nan]), np.array([1.0, np.nan, np.nan]))).all()
    # stated limitation: an infinite operand gives NaN (libm: a multiple of pi / 4)
    assert np.isnan(atan2(mathlib, np.array([np.inf]), np.array([1.0]))).all()
