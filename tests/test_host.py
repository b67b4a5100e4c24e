"""CPU-side checks of the product package: the C-ABI library loads and exports every symbol include/ssmq.h declares, the
host logic (point sets, multi-indices, integrand descriptors, model formulas) matches the reference's golden vectors,
and every compute entry point fails loudly without a GPU (no CPU fallback).  No kernel is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import ssmq_oracle as orc
from tests._cases import MODELS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    from ssmtoybox_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'ssmq.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(ssmq_[a-z0-9_]+)\s*\(', hdr))
    assert len(declared) >= 30
    lib = _lib.load()
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    assert lib.ssmq_version() == _lib.ABI_VERSION == 102


def test_integrand_struct_layout():
    from ssmtoybox_amd import _lib
    # struct ssmq_integrand: 4 int32 + 16 doubles + 16 int32 = 208 bytes
    assert ctypes.sizeof(_lib.Integrand) == 16 + 8 * 16 + 4 * 16
    s = _lib.Integrand.make(12, (1.0, 2.0, 3.0, 4.0), (0, 2))
    assert (s.id, s.n_par, s.n_idx, s.par[3], s.idx[1]) == (12, 4, 2, 4.0, 2)
    with pytest.raises(ValueError):
        _lib.Integrand.make(1, range(17))


def test_round6_entry_points_without_a_device():
    """The round-6 entry points (ABI 102) on a host without a GPU: the job struct has the header's layout, argument errors are
    argument errors, and anything that needs the device fails loudly (SSMQ_E_HIP) - there is no CPU path behind the ABI."""
    import ssmtoybox_amd as amd
    from ssmtoybox_amd import _lib
    lib = _lib.load()
    # struct ssmq_filter_job: 4 pointers, 2 int64, 2 int32, 3 + 2 + 2 + 1 + 1 pointers, 1 double
    assert ctypes.sizeof(_lib.FilterJob) == 4 * 8 + 2 * 8 + 2 * 4 + 9 * 8 + 8
    assert lib.ssmq_filter_forward_multi_dev(0, None) == 0
    assert lib.ssmq_filter_forward_multi_dev(-1, None) == -1 and lib.ssmq_filter_forward_multi_dev(65, (_lib.FilterJob * 65)()) == -1
    assert lib.ssmq_filter_forward_multi_dev(1, (_lib.FilterJob * 1)()) == -1            # null handles
    assert lib.ssmq_filter_forward_piped(None, None, None, None, 1, 1, None, None, None, None, None, None, None, None, 0, 0) == -1
    assert lib.ssmq_pinned_is_block(ctypes.c_void_p(12345)) == 0 and lib.ssmq_pinned_free(None) == 0
    if amd.device_count() == 0:
        p = ctypes.c_void_p()
        assert lib.ssmq_pinned_alloc(ctypes.c_size_t(64), ctypes.byref(p)) == -2 and not p.value      # SSMQ_E_HIP


def test_integrand_ids_match_oracle_numbering():
    from ssmtoybox_amd import _lib
    for name in dir(orc):
        if name.startswith('F_'):
            assert getattr(_lib, name) == getattr(orc, name), name


def test_point_sets_golden(golden):
    import ssmtoybox_amd as amd
    g = golden('g1_points')
    for d in (1, 2, 3, 5, 6, 10):
        for kappa in (None, 0.0, 2.0):
            tag = 'ut_d{}_k{}'.format(d, 'none' if kappa is None else int(kappa))
            assert np.array_equal(amd.UnscentedTransform.unit_sigma_points(d, kappa=kappa), g[tag + '_pts'])
            wm, wc = amd.UnscentedTransform.weights(d, kappa=kappa)
            assert np.array_equal(wm, g[tag + '_wm']) and np.array_equal(wc, g[tag + '_wc'])
        assert np.array_equal(amd.SphericalRadialTransform.unit_sigma_points(d), g['sr_d{}_pts'.format(d)])
        assert np.array_equal(amd.SphericalRadialTransform.weights(d), g['sr_d{}_w'.format(d)])
        for deg in (3, 5):
            fs = amd.FullySymmetricStudentTransform
            assert np.array_equal(fs.unit_sigma_points(d, degree=deg), g['fs_d{}_deg{}_pts'.format(d, deg)])
            assert np.allclose(fs.weights(d, degree=deg), g['fs_d{}_deg{}_w'.format(d, deg)], rtol=1e-13, atol=0)
        assert np.array_equal(fs.unit_sigma_points(d, 5, None, 7.0), g['fs_d{}_deg5_dof7_pts'.format(d)])
        assert np.array_equal(fs.unit_sigma_points(d, 3, 1.0, 6.0), g['fs_d{}_deg3_k1_pts'.format(d)])
    for d, degs in ((1, (3, 5, 7)), (2, (3, 5, 7)), (3, (3, 5)), (5, (3,))):
        for deg in degs:
            gh = amd.GaussHermiteTransform
            assert np.array_equal(gh.unit_sigma_points(d, deg), g['gh_d{}_deg{}_pts'.format(d, deg)])
            assert np.allclose(gh.weights(d, deg), g['gh_d{}_deg{}_w'.format(d, deg)], rtol=1e-13, atol=0)
    from ssmtoybox_amd.bq.bqmod import n_sum_k
    for n, k in ((1, 0), (1, 2), (2, 2), (3, 2), (3, 3), (5, 2), (10, 2)):
        assert np.array_equal(n_sum_k(n, k), g['nsumk_{}_{}'.format(n, k)])


def test_transform_objects_have_reference_attributes():
    import ssmtoybox_amd as amd
    ut = amd.UnscentedTransform(3, kappa=1.0)
    assert isinstance(ut, amd.MomentTransform) and ut.unit_sp.shape == (3, 7) and ut.Wc.shape == (7, 7)
    assert np.array_equal(np.diag(ut.Wc), ut.wc) and np.isclose(ut.wm.sum(), 1.0)
    for cls, n in ((amd.SphericalRadialTransform, 6), (amd.FullySymmetricStudentTransform, 7)):
        t = cls(3)
        assert t.unit_sp.shape == (3, n) and t.Wc.shape == (n, n) and np.isclose(t.wm.sum(), 1.0)
    gh = amd.GaussHermiteTransform(2, degree=5)
    assert gh.unit_sp.shape == (2, 25) and np.isclose(gh.wm.sum(), 1.0)


def test_model_host_formulas_match_oracle():
    """The NumPy evaluation the model classes offer to callers is the same formula as the oracle's integrand table."""
    from tests.test_gpu_parity import make_model
    rng = np.random.default_rng(4)
    for name, (fid, p, sidx, din, dout) in MODELS.items():
        mod, f = make_model(name)
        integ, e = mod.device_integrand()
        assert integ.id == fid and e == dout
        assert [integ.par[i] for i in range(integ.n_par)] == [float(v) for v in p]
        x = rng.standard_normal(din) + (np.array([6500.0, 350.0, -1.8, -6.8, 0.7]) if 'reentry_' in name and din == 5
                                        else 0.0)
        ref = orc.integrand(fid, x if sidx is None else x[list(sidx)], 3, p)
        assert np.allclose(np.atleast_1d(f(x, 3)), ref, rtol=1e-14, atol=0), name


def test_no_cpu_fallback():
    """Without a GPU every compute entry point raises; nothing silently computes on the host."""
    import ssmtoybox_amd as amd
    from ssmtoybox_amd import ssmod, _lib
    if amd.device_count() > 0:
        pytest.skip('a GPU is present')
    with pytest.raises(amd.SsmqError):
        amd.GaussianProcessTransform(1, 1, np.array([[1.0, 3.0]]))
    with pytest.raises(amd.SsmqError):
        amd.BayesSardTransform(1, 1, np.array([[1.0, 3.0]]), np.array([[0, 1, 2]]))
    ut = amd.UnscentedTransform(1)
    with pytest.raises(amd.SsmqError):
        ut.apply(ssmod.UNGMTransition().dyn_eval, np.zeros(1), np.eye(1), np.atleast_1d(0))
    with pytest.raises(amd.SsmqError):
        ut.apply(lambda x, t: x, np.zeros(1), np.eye(1), np.atleast_1d(0))
    with pytest.raises(amd.SsmqError):
        _lib.SoA.from_host(np.zeros((4, 2)))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under ssmtoybox_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'ssmtoybox_amd')):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dirpath, fn)).read()
                assert 'ssmq_oracle' not in src and 'from oracle' not in src and 'import oracle' not in src, fn


def test_header_is_plain_c(tmp_path):
    """include/ssmq.h is the C ABI: it must compile as C99 (no C++ constructs, no torch / HIP types in signatures)."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('gcc not available')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / 't.c'
    src.write_text('#include "ssmq.h"\nint main(void) { return ssmq_version() == 0; }\n')
    res = subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-I', os.path.join(root, 'include'),
                          '-c', str(src), '-o', str(tmp_path / 't.o')], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    hdr = open(os.path.join(root, 'include', 'ssmq.h')).read()
    assert 'torch' not in hdr and 'hipStream' not in hdr and 'at::' not in hdr


def test_c_abi_point_sets_match_reference(golden):
    """ssmq_points (host code inside libssmq: runs without a GPU) against the reference's point sets and weights (G1)."""
    from ssmtoybox_amd import _lib
    lib = _lib.load()
    g = golden('g1_points')
    nan = float('nan')

    def rule(kind, d, par):
        par = np.asarray(par, dtype=np.float64)
        pp = par.ctypes.data_as(_lib.c_double_p) if par.size else None
        n = lib.ssmq_points_count(kind, d, pp, par.size)
        assert n > 0
        xi, wm, wc = np.empty((d, n)), np.empty(n), np.empty(n)
        assert lib.ssmq_points(kind, d, pp, par.size, xi.ctypes.data_as(_lib.c_double_p),
                               wm.ctypes.data_as(_lib.c_double_p), wc.ctypes.data_as(_lib.c_double_p)) == n
        return xi, wm, wc
    UT, SR, GH, FS = 0, 1, 2, 3
    for d in (1, 2, 3, 5, 6, 10):
        for kappa in (None, 0.0, 2.0):
            tag = 'ut_d{}_k{}'.format(d, 'none' if kappa is None else int(kappa))
            xi, wm, wc = rule(UT, d, [] if kappa is None else [kappa])
            assert np.array_equal(xi, g[tag + '_pts']) and np.array_equal(wm, g[tag + '_wm'])
            assert np.array_equal(wc, g[tag + '_wc'])
        xi, wm, wc = rule(UT, d, [1.0, 0.5, 1.0])
        assert np.array_equal(xi, g['ut_d{}_a05_pts'.format(d)]) and np.array_equal(wm, g['ut_d{}_a05_wm'.format(d)])
        assert np.array_equal(wc, g['ut_d{}_a05_wc'.format(d)])
        xi, wm, _ = rule(SR, d, [])
        assert np.array_equal(xi, g['sr_d{}_pts'.format(d)]) and np.array_equal(wm, g['sr_d{}_w'.format(d)])
        for deg in (3, 5):
            xi, wm, _ = rule(FS, d, [deg])
            assert np.array_equal(xi, g['fs_d{}_deg{}_pts'.format(d, deg)])
            assert np.allclose(wm, g['fs_d{}_deg{}_w'.format(d, deg)], rtol=1e-14, atol=0)
        xi, wm, _ = rule(FS, d, [5, nan, 7.0])
        assert np.array_equal(xi, g['fs_d{}_deg5_dof7_pts'.format(d)])
        assert np.allclose(wm, g['fs_d{}_deg5_dof7_w'.format(d)], rtol=1e-14, atol=0)
        xi, wm, _ = rule(FS, d, [3, 1.0, 6.0])
        assert np.array_equal(xi, g['fs_d{}_deg3_k1_pts'.format(d)]) and np.array_equal(wm, g['fs_d{}_deg3_k1_w'.format(d)])
    for d, degs in ((1, (3, 5, 7)), (2, (3, 5, 7)), (3, (3, 5)), (5, (3,))):
        for deg in degs:
            xi, wm, _ = rule(GH, d, [deg])
            # the reference takes the roots from numpy's companion-matrix eigenvalues; two root finders agree to ~1e-15
            assert np.allclose(xi, g['gh_d{}_deg{}_pts'.format(d, deg)], rtol=0, atol=4e-15)
            assert np.allclose(wm, g['gh_d{}_deg{}_w'.format(d, deg)], rtol=1e-13, atol=0)
            assert abs(wm.sum() - 1) < 1e-14
    # argument errors
    assert lib.ssmq_points_count(9, 2, None, 0) < 0 and lib.ssmq_points_count(UT, 0, None, 0) < 0
    assert lib.ssmq_points_count(GH, 10, np.array([7.0]).ctypes.data_as(_lib.c_double_p), 1) < 0     # 7^10 points


def test_polynomial_expectations_reference_known_answers():
    """ssmtoybox/tests/test_bqmod.py:262-314 of the reference (test_x_px, test_exp_x_xpx, test_exp_x_pxpx) on the
    product's methods - integer arithmetic on the multi-indices, the host part of `ssmq_bs_moments` (no device)."""
    from ssmtoybox_amd.bq.bqmod import BayesSardModel
    model = BayesSardModel.__new__(BayesSardModel)           # the three methods use no state
    mi_1d = np.array([[0, 1, 2]])
    mi_2d = np.array([[0, 1, 0, 1, 0, 2], [0, 0, 1, 1, 2, 0]])
    ke = model._exp_x_px(mi_1d)
    assert ke.shape == (3,) and np.array_equal(ke, [1, 0, 1])
    ke = model._exp_x_px(mi_2d)
    assert ke.shape == (6,) and np.array_equal(ke, [1, 0, 0, 0, 1, 1])
    ke = model._exp_x_xpx(mi_1d)
    assert ke.shape == mi_1d.shape and np.array_equal(ke, [[0, 1, 0]])
    ke = model._exp_x_xpx(mi_2d)
    assert ke.shape == mi_2d.shape and np.array_equal(ke, [[0, 1, 0, 0, 0, 0], [0, 0, 1, 0, 0, 0]])
    ke = model._exp_x_pxpx(mi_1d)
    assert np.array_equal(ke, [[1, 0, 1], [0, 1, 0], [1, 0, 3]])
    ke = model._exp_x_pxpx(mi_2d)
    assert np.array_equal(ke, [[1, 0, 0, 0, 1, 1], [0, 1, 0, 0, 0, 0], [0, 0, 1, 0, 0, 0], [0, 0, 0, 1, 0, 0],
                               [1, 0, 0, 0, 3, 1], [1, 0, 0, 0, 1, 3]])
    # and the reference's own outputs for every Bayes-Sard golden case (incl. its alpha_e factor in E[x p(x)'])
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g2_bs_weights.npz'))
    tags = sorted({k[:-3] for k in g.files if k.endswith('_mi')})
    assert len(tags) >= 10
    for t in tags:
        mi = g[t + '_mi']
        assert np.array_equal(model._exp_x_px(mi), g[t + '_px']), t
        assert np.array_equal(model._exp_x_xpx(mi), g[t + '_xpx']), t
        assert np.array_equal(model._exp_x_pxpx(mi), g[t + '_pxpx']), t
    with pytest.raises(Exception):
        model._exp_x_px(np.array([[0, -1]]))


def test_degree7_fully_symmetric_rule_is_exact_to_degree_7():
    """BASELINE configs[4] names a "fully-symmetric 7th-degree rule"; the reference has degree 3 and 5 only (mtran.py:392),
    so this rule is the build's own (parity-unpinned).  Its defining property instead: every monomial of total degree <= 7
    is integrated exactly against the multivariate-t moments E[prod x_i^a_i] = (nu/2)^k Gamma(nu/2 - k) / Gamma(nu/2)
    prod (a_i - 1)!!, k = sum(a) / 2 - the moments the reference's degree-5 rule matches up to order 5 (mtran.py:454-463).
    Python mirror and C ABI give the same points."""
    import itertools
    import math
    from ssmtoybox_amd import _lib
    from ssmtoybox_amd.mtran import FullySymmetricStudentTransform as FS
    nu = 7.0

    def dfact(m):
        return 1 if m <= 0 else m * dfact(m - 2)

    def moment(a):
        if any(v % 2 for v in a):
            return 0.0
        k = sum(a) // 2
        return (nu / 2) ** k * math.gamma(nu / 2 - k) / math.gamma(nu / 2) * np.prod([dfact(v - 1) for v in a])
    lib = _lib.load()
    for n in (1, 2, 3, 4, 10):
        x, w = FS.unit_sigma_points(n, 7), FS.weights(n, 7)
        N = 1 + 4 * n + 2 * n * (n - 1) + 4 * n * (n - 1) * (n - 2) // 3
        assert x.shape == (n, N) and w.shape == (N,) and abs(w.sum() - 1) < 1e-12
        m = min(n, 4)                       # by symmetry the first four coordinates cover every monomial type up to degree 7
        for deg in range(8):
            for combo in itertools.combinations_with_replacement(range(m), deg):
                a = [combo.count(i) for i in range(m)]
                val = float((w * np.prod(x[:m] ** np.array(a)[:, None], axis=0)).sum())
                assert abs(val - moment(a)) <= 1e-11 * max(1.0, abs(moment(a)), np.abs(w).max()), (n, a, val, moment(a))
        par = np.array([7.0, np.nan, 4.0])
        assert lib.ssmq_points_count(_lib.PTS_FS if hasattr(_lib, 'PTS_FS') else 3, n, par.ctypes.data_as(_lib.c_double_p), 3) == N
        xi, wm = np.empty((n, N)), np.empty(N)
        assert lib.ssmq_points(3, n, par.ctypes.data_as(_lib.c_double_p), 3, xi.ctypes.data_as(_lib.c_double_p),
                               wm.ctypes.data_as(_lib.c_double_p), None) == N
        assert np.allclose(xi, x, rtol=1e-14, atol=0) and np.allclose(wm, w, rtol=1e-10, atol=1e-12)
    assert FS(10, degree=7).unit_sp.shape == (10, 1181)


def test_cxx_consumer_of_the_c_abi_builds(tmp_path):
    """tools/micro/threads_rate.cpp - a C++ program that drives whole filters through include/ssmq.h from several threads (run on the
    GPU box by tools/micro/run_threads_rate.sh, from a fresh shell: a process that has initialised the GPU must not start it) - compiles
    against the header and links against libssmq.so: every entry point it uses is declared and exported."""
    import shutil
    import subprocess
    if not shutil.which('g++'):
        pytest.skip('no g++ here')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, 'ssmtoybox_amd')
    if not os.path.exists(os.path.join(lib, 'libssmq.so')):
        pytest.skip('libssmq.so not built')
    res = subprocess.run(['g++', '-O1', '-std=c++17', '-Wall', '-I' + os.path.join(root, 'include'),
                          os.path.join(root, 'tools', 'micro', 'threads_rate.cpp'), '-o', str(tmp_path / 'threads_rate'), '-L' + lib, '-lssmq',
                          '-lpthread', '-Wl,-rpath,' + lib], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0, res.stdout
