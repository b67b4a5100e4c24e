"""The single-GPU legs of bench.py beside the headline: the north-star transform (D = E = 6), the other BASELINE configs,
the theta-batched step and the linearisation transform.  Each returns a detail record (bench_detail.json); the scalars the
result line keeps are picked in benchlib/record.py."""
import ctypes
import time

import numpy as np

from .common import (HBM_PEAK_GBS, CLOCK_HZ, F64_MFMA_PEAK_TF, pmc_traffic, pmc_traffic_named, pmc_issue, issue_block, settle,
                     timed_passes, cpu_baseline_filter, cpu_baseline_apply)
from .workloads import FilterBench, simulate_ungm, synthetic_reentry6


def measure_linearize(B=1000000, iters=20):
    """The linearisation transform of the extended Kalman filter (mtran.py:49-59; csrc/ssmq_linear.hip: k_linearize) on the
    pendulum dynamics, B = 1e6 trajectories resident in HBM: an HBM-bound map, 8 (D + D^2 + E + E^2 + E D) = 128 algorithmic
    bytes per trajectory.  Checked against the oracle on a few trajectories."""
    import ssmtoybox_amd as amd
    from ssmtoybox_amd import _lib, ssmod
    from oracle import ssmq_oracle as orc
    mod = ssmod.Pendulum2DTransition(ssmod.GaussRV(2), ssmod.GaussRV(2), dt=0.01)
    D = E = 2
    tf = amd.LinearizationTransform(D)
    rng = np.random.default_rng(2)
    means = rng.standard_normal((B, D))
    a = rng.standard_normal((B, D, D))
    covs = np.einsum('bij,bkj->bik', a, a) + 0.2 * np.eye(D)
    mean, cov = _lib.SoA.from_host(means), _lib.SoA.from_host(covs)
    mf, cf, cfx = _lib.SoA(E, B), _lib.SoA(E * E, B), _lib.SoA(E * D, B)
    st = _lib.DeviceBuffer(4 * mean.ld)
    tbuf = _lib.DeviceBuffer(8)
    tbuf.upload(np.zeros(1))
    f = mod.dyn_eval
    settle(lambda: tf.apply_batch_dev(f, mean, cov, tbuf, mf, cf, cfx, st, 0), _lib.sync)
    e0, e1 = _lib.Event(), _lib.Event()
    e0.record()
    for _ in range(iters):
        tf.apply_batch_dev(f, mean, cov, tbuf, mf, cf, cfx, st, 0)
    e1.record()
    ms = e0.elapsed_ms(e1) / iters
    g_mf, g_cf, g_cfx = mf.to_host(), cf.to_host((E, E)), cfx.to_host((E, D))
    err = 0.0
    for i in (0, B // 2, B - 1):
        r = orc.apply_linear(orc.F_PENDULUM_DYN, means[i], covs[i], 0.0, (0.01,))
        err = max(err, float(np.abs(g_mf[i] - r[0]).max() / np.abs(r[0]).max()), float(np.abs(g_cf[i] - r[1]).max() / np.abs(r[1]).max()),
                  float(np.abs(g_cfx[i] - r[2]).max() / np.abs(r[2]).max()))
    name = tf.kernel_name(f)
    for buf in (mean, cov, mf, cf, cfx):
        buf.buf.free()
    st.free()
    tbuf.free()
    nbytes = 8.0 * B * (D + D * D + E + E * E + E * D)
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {'kernel': name, 'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
            'ms_per_launch': ms, 'bytes_per_launch': nbytes, 'transforms_per_s': B / (ms * 1e-3), 'max_rel_err_vs_oracle': err,
            'workload': 'LinearizationTransform (the transform of ExtendedKalman), pendulum dynamics D=E=2, B=1e6'}


def measure_theta_step(calls=1000):
    """Latency of the theta-batched step of the marginalised GPQ filter (`ssmq_gp_theta_step`, one call =
    weights of both transforms, time update, measurement transform, Kalman update and log-likelihood for every parameter
    item) at the item counts the filter sends: param_dim + 1 (gradient) on the pendulum model.  Wall clock through the
    Python wrapper, inputs and outputs on the host."""
    from ssmtoybox_amd import ssinf, ssmod
    dyn = ssmod.Pendulum2DTransition(ssmod.GaussRV(2, mean=np.array([1.5, 0.0]), cov=0.01 * np.eye(2)),
                                     ssmod.GaussRV(2, cov=0.01 * np.eye(2)), 0.01)
    obs = ssmod.Pendulum2DMeasurement(ssmod.GaussRV(1, cov=np.array([[0.1]])), 2)
    alg = ssinf.MarginalizedGaussianProcessKalman(dyn, obs, 'rbf', 'sr')
    rng = np.random.default_rng(0)
    P = alg.param_dim + 1
    theta = 0.1 * rng.standard_normal((P, alg.param_dim))
    y = rng.standard_normal(1)
    m0, P0 = np.zeros(2), np.eye(2)
    for _ in range(200):
        alg.theta_step(theta, m0, P0, y, 1)
    # a host / device ping-pong of 12-20 us kernels: the device idles most of the time and its power state moves between
    # blocks of calls (63 us and 133 us per call were both seen for whole blocks inside this script, 60-65 us in a fresh
    # process), so five blocks are timed and the median and the best are reported
    blocks = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(calls // 5):
            alg.theta_step(theta, m0, P0, y, 1)
        blocks.append((time.perf_counter() - t0) / (calls // 5) * 1e6)
    us = float(np.median(blocks))
    # the same entry point at the item count of the batched marginalised filter: 8 192 items with a state of their own each
    # (B (param_dim + 1) objective points of one optimiser round)
    n_big = 8192
    th_big = 0.1 * rng.standard_normal((n_big, alg.param_dim))
    m_big, P_big = np.tile(m0, (n_big, 1)), np.tile(P0, (n_big, 1, 1))
    y_big = rng.standard_normal((n_big, 1))
    for _ in range(3):
        alg.theta_step(th_big, m_big, P_big, y_big, 1)
    t0 = time.perf_counter()
    for _ in range(10):
        alg.theta_step(th_big, m_big, P_big, y_big, 1)
    big_s = (time.perf_counter() - t0) / 10
    # ... and the batched filter itself: UNGM, B = 1024 trajectories x T = 10 steps, every trajectory at its own pace
    du = ssmod.UNGMTransition(ssmod.GaussRV(1), ssmod.GaussRV(1, cov=np.array([[10.0]])))
    ou = ssmod.UNGMMeasurement(ssmod.GaussRV(1), 1)
    mg = ssinf.MarginalizedGaussianProcessKalman(du, ou, 'rbf', 'sr')
    _, yu = simulate_ungm(1024, 10, 5)
    du_data = np.ascontiguousarray(yu[None])
    mg.forward_pass_batch(du_data[:, :, :64])
    t0 = time.perf_counter()
    mg.forward_pass_batch(du_data)
    mg_s = time.perf_counter() - t0
    batch = {'us_per_trajectory_step': 1e6 * mg_s / (1024 * 10), 'ms_per_time_step': 1e3 * mg_s / 10, 'trajectories': 1024, 'time_steps': 10,
             'device_rounds': mg.batch_stats['rounds'], 'bfgs_iterations': mg.batch_stats['iterations'], 'theta_items': mg.batch_stats['items'],
             'failed_trajectories': int((mg.batch_failed > 0).sum()),
             'workload': 'MarginalizedGaussianProcessKalman.forward_pass_batch on UNGM (ssmq_gp_marginal_filter_batch: B BFGS runs, one theta '
                         'step per round; the reference: one scipy BFGS per trajectory and step, ~1.5 ms per trajectory-step here)'}
    return {'items_8192_ms_per_call': 1e3 * big_s, 'items_8192_per_s': n_big / big_s, 'marginal_filter_batch': batch, 'us_per_call': us, 'us_per_call_best_block': float(min(blocks)), 'items': P, 'theta_steps_per_s': P / (us * 1e-6), 'launches_per_call': 2,
            'kernels': ['k_theta_weights', 'k_theta_chain'],
            'workload': 'MarginalizedGaussianProcessKalman.theta_step, pendulum 2-D + 1-D measurement, spherical-radial points, '
                        '%d parameter items (param_dim + 1), host arrays in and out' % P}


class C5GemmBench:
    """The GEMM-shaped stage of the Bayes-Sard transform at D = E = 10 with the fully-symmetric degree-5 rule (N = 201,
    BASELINE config C5): T = FX Wc for B = 1e4 trajectories, (B E) x 208 x 208 on the matrix cores, integrand values
    resident in HBM (synthetic, the reference has no 10-D model)."""

    def __init__(self, amd, B, seed):
        from ssmtoybox_amd import _lib
        from ssmtoybox_amd.bq.bqmod import n_sum_k
        self._lib = _lib
        lib = _lib.load()
        mi = np.hstack([n_sum_k(10, k) for k in range(3)])
        self.tf = amd.BayesSardTransform(10, 10, np.array([[1.0] + [3.0] * 10]), mi, 'fs', {'degree': 5})
        self.h = self.tf._handle_for(10)
        npad = ctypes.c_int(0)
        lib.ssmq_fxwc_batch_dev(ctypes.c_void_p(self.h), 0, None, 0, None, 0, ctypes.byref(npad))
        self.N, self.NP, self.M = self.tf.wm.shape[0], npad.value, B * 10
        if not self.NP:
            raise RuntimeError('no matrix-core instantiation for N = {}'.format(self.N))
        rng = np.random.default_rng(seed)
        self.fx = np.zeros((self.M, self.NP))
        self.fx[:, :self.N] = rng.standard_normal((self.M, self.N))
        self.d_fx, self.d_t = _lib.DeviceBuffer(self.fx.nbytes), _lib.DeviceBuffer(self.fx.nbytes)
        self.d_fx.upload(self.fx)
        self.gemm_kernel = 'k_fxwc_mfma<13,1>'

    def launch(self):
        self._lib.check(self._lib.load().ssmq_fxwc_batch_dev(ctypes.c_void_p(self.h), self.M, ctypes.c_void_p(self.d_fx.ptr),
                                                             self.NP, ctypes.c_void_p(self.d_t.ptr), self.NP, None),
                        'ssmq_fxwc_batch_dev')

    def check(self):
        """Sampled rows against the NumPy product (a check, not the oracle: the oracle covers the whole transform)."""
        self.launch()
        self._lib.sync()
        t = self.d_t.download((self.M, self.NP))
        rows = np.arange(0, self.M, max(1, self.M // 257))
        ref = self.fx[rows, :self.N].dot(self.tf.Wc)
        scale = np.abs(self.fx[rows, :self.N]).dot(np.abs(self.tf.Wc)).max()
        return float(np.abs(t[rows, :self.N] - ref).max() / scale)

    def measure(self, warmup=5, iters=50):
        settle(self.launch, self._lib.sync)
        for _ in range(warmup):
            self.launch()
        self._lib.sync()
        e0, e1 = self._lib.Event(), self._lib.Event()
        e0.record()
        for _ in range(iters):
            self.launch()
        e1.record()
        ms = e0.elapsed_ms(e1) / iters
        return ms, 2.0 * self.M * self.NP * self.NP

    def measure_full_transform(self, B, with_cpu=True, warmup=3, iters=20):
        """The whole D = 10 transform with the device-evaluated synthetic model (ssmod.Smooth10DTransition): Cholesky +
        points + integrand pass, the GEMM, the per-trajectory rest - three launches, moments resident in HBM."""
        from ssmtoybox_amd import ssmod
        _lib = self._lib
        rng = np.random.default_rng(6)
        means = rng.standard_normal((B, 10))
        a = rng.standard_normal((B, 10, 10)) / np.sqrt(10)
        covs = np.einsum('bij,bkj->bik', a, a) + 0.1 * np.eye(10)
        f = ssmod.Smooth10DTransition().dyn_eval
        mean, cov = _lib.SoA.from_host(means), _lib.SoA.from_host(covs)
        mf, cf, cfx = _lib.SoA(10, B), _lib.SoA(100, B), _lib.SoA(100, B)
        st = _lib.DeviceBuffer(4 * mean.ld)
        tbuf = _lib.DeviceBuffer(8)
        tbuf.upload(np.zeros(1))
        settle(lambda: self.tf.apply_batch_dev(f, mean, cov, tbuf, mf, cf, cfx, st, 0), _lib.sync)
        for _ in range(warmup):
            self.tf.apply_batch_dev(f, mean, cov, tbuf, mf, cf, cfx, st, 0)
        _lib.sync()
        e0, e1 = _lib.Event(), _lib.Event()
        e0.record()
        for _ in range(iters):
            self.tf.apply_batch_dev(f, mean, cov, tbuf, mf, cf, cfx, st, 0)
        e1.record()
        ms = e0.elapsed_ms(e1) / iters
        for buf in (mean, cov, mf, cf, cfx):
            buf.buf.free()
        cb = None
        if with_cpu:
            ns = 512          # ~0.3 ms per transform and core at N = 201: a bounded sample of the same inputs
            cb = cpu_baseline_apply(self.tf, _lib.F_SMOOTH10D_DYN, (), 10, 10, means[:ns], covs[:ns], 4.0,
                                    'the D=E=10, N=201 Bayes-Sard transform (whole transform, not only the GEMM)')
        return ms, cb


def measure_c5_unisolvent(amd, B=100000, iters=20):
    """The other half of BASELINE configs[4] as SURVEY 8d restates it: Bayes-Sard transform at D = E = 10 with the
    unscented point set, N = 21 = number of basis functions (unisolvent case), device-resident moments, device integrand
    (k_apply_tile: generic shapes of 9-64 points, every product on the matrix cores)."""
    from ssmtoybox_amd import _lib, ssmod
    D = 10
    mi = np.hstack((np.zeros((D, 1), dtype=int), np.eye(D, dtype=int), 2 * np.eye(D, dtype=int)))
    tf = amd.BayesSardTransform(D, D, np.array([[1.0] + [3.0] * D]), multi_ind=mi, point_str='ut')
    f = ssmod.Smooth10DTransition().dyn_eval
    rng = np.random.default_rng(6)
    means = rng.standard_normal((B, D))
    a = rng.standard_normal((B, D, D)) / np.sqrt(D)
    covs = np.einsum('bij,bkj->bik', a, a) + 0.1 * np.eye(D)
    mean, cov = _lib.SoA.from_host(means), _lib.SoA.from_host(covs)
    mf, cf, cfx = _lib.SoA(D, B), _lib.SoA(D * D, B), _lib.SoA(D * D, B)
    st = _lib.DeviceBuffer(4 * mean.ld)
    tbuf = _lib.DeviceBuffer(8)
    tbuf.upload(np.zeros(1))
    settle(lambda: tf.apply_batch_dev(f, mean, cov, tbuf, mf, cf, cfx, st, 0), _lib.sync)
    for _ in range(3):
        tf.apply_batch_dev(f, mean, cov, tbuf, mf, cf, cfx, st, 0)
    _lib.sync()
    e0, e1 = _lib.Event(), _lib.Event()
    e0.record()
    for _ in range(iters):
        tf.apply_batch_dev(f, mean, cov, tbuf, mf, cf, cfx, st, 0)
    e1.record()
    ms = e0.elapsed_ms(e1) / iters
    name = tf.kernel_name(f)
    for buf in (mean, cov, mf, cf, cfx):
        buf.buf.free()
    st.free()
    tbuf.free()
    alg = 8 * (D + D * D + D + D * D + D * D) * B          # 2480 B per transform at D = E = 10
    gbs = alg / (ms * 1e-3) / 1e9
    return {'kernel': name, 'ms_per_launch': ms, 'transforms_per_s': B / (ms * 1e-3), 'bound': 'hbm', 'achieved': gbs,
            'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS, 'bytes_per_launch': alg,
            'workload': 'Bayes-Sard transform, D=E=10, unscented points N=21 = basis functions (unisolvent), B=1e5',
            'note': 'latency / issue-bound shape (Cholesky chain, 28 dependent matrix steps per trajectory): DESIGN.md 3.7'}


def measure_c5_degree7(amd, B=10000, iters=5, with_cpu=True):
    """BASELINE configs[4] AS WORDED: Bayes-Sard transform at D = E = 10 with a fully-symmetric rule of degree 7.  The
    reference has degree 3 and 5 only (mtran.py:392); the rule is this build's own (1181 points, exact to degree 7:
    tests/test_host.py), so the POINTS are parity-unpinned; weights and transform on them are pinned to the reference run on
    the injected set (tests/golden/g12_large_weights.npz, tests/test_gpu_parity.py::test_config4_as_worded_degree7_full_batch).  Route: two launches -
    k_eval_wave (factor, points, integrand values FX to memory in fragment order) and k_bq_stream (csrc/ssmq_bq_stream.hip: the
    product with Wc = S + S', panels of 16 column tiles, no LDS staging and no barrier); `ms_per_launch` is both together."""
    from ssmtoybox_amd import _lib, ssmod
    from ssmtoybox_amd.bq.bqmod import n_sum_k
    from oracle import ssmq_oracle as orc
    D = 10
    mi = np.hstack([n_sum_k(D, k) for k in range(3)])
    t0 = time.perf_counter()
    tf = amd.BayesSardTransform(D, D, np.array([[1.0] + [3.0] * D]), mi, 'fs', {'degree': 7})
    t_weights = time.perf_counter() - t0
    N = tf.wm.shape[0]
    f = ssmod.Smooth10DTransition().dyn_eval
    rng = np.random.default_rng(6)
    means = rng.standard_normal((B, D))
    a = rng.standard_normal((B, D, D)) / np.sqrt(D)
    covs = np.einsum('bij,bkj->bik', a, a) + 0.1 * np.eye(D)
    mean, cov = _lib.SoA.from_host(means), _lib.SoA.from_host(covs)
    mf, cf, cfx = _lib.SoA(D, B), _lib.SoA(D * D, B), _lib.SoA(D * D, B)
    st = _lib.DeviceBuffer(4 * mean.ld)
    tbuf = _lib.DeviceBuffer(8)
    tbuf.upload(np.zeros(1))
    settle(lambda: tf.apply_batch_dev(f, mean, cov, tbuf, mf, cf, cfx, st, 0), _lib.sync)
    for _ in range(2):
        tf.apply_batch_dev(f, mean, cov, tbuf, mf, cf, cfx, st, 0)
    _lib.sync()
    e0, e1 = _lib.Event(), _lib.Event()
    e0.record()
    for _ in range(iters):
        tf.apply_batch_dev(f, mean, cov, tbuf, mf, cf, cfx, st, 0)
    e1.record()
    ms = e0.elapsed_ms(e1) / iters
    g_mf, g_cf = mf.to_host(), cf.to_host((D, D))
    # the check: the ORACLE's weights on this point set (oracle/ssmq_oracle.py: bs_weights, pinned to the reference's
    # weights on the same 1181 points by tests/golden/g12_large_weights.npz) and the oracle's transform with them, against
    # the device's transform with the DEVICE's weights - both halves of the route are compared, not the apply alone
    w = orc.bs_weights(np.array([[1.0] + [3.0] * D]), tf.model.points, mi)
    w_err = max(float(np.max(np.abs(tf.wm - w['wm'])) / np.max(np.abs(w['wm']))),
                float(np.max(np.abs(tf.Wc - w['Wc'])) / np.max(np.abs(w['Wc']))),
                float(np.max(np.abs(tf.Wcc - w['Wcc'])) / np.max(np.abs(w['Wcc']))))
    err = 0.0
    for i in (0, B // 3, B // 2, B - 1):
        r = orc.apply_bq(orc.F_SMOOTH10D_DYN, means[i], covs[i], 0.0, tf.model.points, w)
        sc = float(np.max(np.abs(r[0])))
        err = max(err, float(np.max(np.abs(g_mf[i] - r[0])) / sc), float(np.max(np.abs(g_cf[i] - r[1])) / max(sc ** 2, np.abs(r[1]).max())))
    name = tf.kernel_name(f)
    for buf in (mean, cov, mf, cf, cfx):
        buf.buf.free()
    st.free()
    tbuf.free()
    flop = 2.0 * B * D * float(N) * N + 2.0 * B * D * D * N + 2.0 * B * D * N * D      # algorithmic, as for N = 201
    tfs = flop / (ms * 1e-3) / 1e12
    nkb = (N + 15) // 16
    # executed by k_bq_stream: per 16-row tile nkb (nkb + 1) / 2 + nkb tile steps x 4 instructions + nkb x 8 in C = T fx'
    flop_exec = ((B + 5) // 6) * 4 * ((nkb * (nkb + 1) // 2 + nkb) * 4 + nkb * 8) * 2048.0 if name == 'k_bq_stream' else None
    # `frac` counts what the matrix cores EXECUTE (the kernel forms fx Wc fx' as C + C': half the dense product); the dense
    # (algorithmic) count divided by the same time is reported beside it and may exceed the peak
    tfe = (flop_exec / (ms * 1e-3) / 1e12) if flop_exec else tfs
    tr_s, tr_e, tr_f = pmc_traffic_named('k_bq_stream'), pmc_traffic_named('k_eval_wave'), pmc_traffic_named('k_bq_stream_finish')
    rec = {'kernel': name, 'points': int(N), 'ms_per_launch': ms, 'transforms_per_s': B / (ms * 1e-3), 'bound': 'mfma',
           'achieved': tfe, 'peak': F64_MFMA_PEAK_TF, 'unit': 'TFLOP/s', 'frac': tfe / F64_MFMA_PEAK_TF,
           'flop_per_launch': flop_exec if flop_exec else flop, 'executed_flop_per_launch': flop_exec,
           'executed_frac': (flop_exec / (ms * 1e-3) / 1e12 / F64_MFMA_PEAK_TF) if flop_exec else None,
           'algorithmic_flop_per_launch': flop, 'algorithmic_tflops': tfs, 'algorithmic_over_peak': tfs / F64_MFMA_PEAK_TF,
           'launches': ['k_eval_wave', 'k_bq_stream', 'k_bq_stream_finish'] if name == 'k_bq_stream' else None,
           'traffic': (tr_s + tr_e + (tr_f or 0.0)) if (name == 'k_bq_stream' and tr_s and tr_e) else None,
           'traffic_by_launch': {'k_eval_wave': tr_e, 'k_bq_stream': tr_s, 'k_bq_stream_finish': tr_f} if name == 'k_bq_stream' else None,
           'algorithmic_bytes': 8.0 * B * (D + D * D + D + D * D + D * D),
           'weights_s': t_weights, 'max_scaled_err_vs_oracle': err, 'weights_rel_err_vs_oracle': w_err,
           'check': 'device weights + device transform against ORACLE weights + oracle transform (the oracle weights are pinned to '
                    'the reference on this point set: tests/golden/g12_large_weights.npz); cond(K) = 8.3e5, so 64 cond eps = 1.2e-8',
           'workload': 'BASELINE configs[4] as worded: Bayes-Sard, D=E=10, fully-symmetric DEGREE-7 rule (this build\'s own: '
                       '1181 points; the rule is not in the reference, weights and transform on it are pinned by golden g12), 66 basis functions, B=1e4; frac on the executed flop (C + C^T form), algorithmic_* = the dense products 2 B E N^2 + 2 B E^2 N + 2 B E N D'}
    if with_cpu:
        rec['cpu_baseline'] = cpu_baseline_apply(tf, _lib.F_SMOOTH10D_DYN, (), D, D, means[:64], covs[:64], 4.0,
                                                 'the D=E=10, N=1181 degree-7 Bayes-Sard transform')
    return rec


class Study6Bench:
    """The six filters of the reference's UNGM studies (UKF, CKF, GHKF-5, GPQKF, TPQKF, BSQKF; research/bsq/bsq_ungm.py:132-137,
    research/tpq/tpq_base.py:175-192) over the SAME B x T measurements, device-resident: one after the other
    (ssmq_filter_forward_dev six times) and as one launch graph (ssmq_filter_forward_multi_dev)."""

    def __init__(self, amd, B, T, seed):
        from ssmtoybox_amd import _lib, ssmod as sm, ssinf
        from ssmtoybox_amd.mtran import resolve_integrand
        from benchlib.workloads import simulate_ungm
        self._lib, self.B, self.T = _lib, B, T
        self.ld = ld = (B + 63) // 64 * 64
        dyn = sm.UNGMTransition(sm.GaussRV(1), sm.GaussRV(1, cov=np.array([[10.0]])))
        obs = sm.UNGMMeasurement(sm.GaussRV(1), 1)
        par, mi = np.array([[1.0, 3.0]]), np.array([[0, 1, 2]])
        self.names = ['ukf', 'ckf', 'ghkf5', 'gpqkf', 'tpqkf', 'bsqkf']
        self.algs = [ssinf.UnscentedKalman(dyn, obs), ssinf.CubatureKalman(dyn, obs), ssinf.GaussHermiteKalman(dyn, obs, deg=5),
                     ssinf.GaussianProcessKalman(dyn, obs, par, par, 'rbf', 'ut'),
                     ssinf.StudentProcessKalman(dyn, obs, par, par, 'rbf', 'ut'), ssinf.BayesSardKalman(dyn, obs, par, par, mi, mi, 'ut')]
        _, y = simulate_ungm(B, T, seed)
        ybuf = np.zeros((T, 1, ld))
        ybuf[:, 0, :B] = y
        self.d_y = _lib.DeviceBuffer(ybuf.nbytes)
        self.d_y.upload(ybuf)
        mb, Pb = np.zeros((1, ld)), np.ones((1, ld))
        self.d_m0, self.d_P0 = _lib.DeviceBuffer(mb.nbytes), _lib.DeviceBuffer(Pb.nbytes)
        self.d_m0.upload(mb)
        self.d_P0.upload(Pb)
        self.f_dyn, _ = resolve_integrand(dyn.dyn_eval)
        self.f_obs, _ = resolve_integrand(obs.meas_eval)
        self.gqg, self.pg = _lib.as_c(np.array([[10.0]]))
        self.rr, self.pr = _lib.as_c(np.array([[1.0]]))
        self.out, self.out2 = [], []
        self.jobs = (_lib.FilterJob * len(self.algs))()
        for i, a in enumerate(self.algs):
            bufs = [(_lib.DeviceBuffer(8 * T * ld), _lib.DeviceBuffer(8 * T * ld), _lib.DeviceBuffer(4 * ld)) for _ in range(2)]
            self.out.append(bufs[0])
            self.out2.append(bufs[1])
            j = self.jobs[i]
            j.h_dyn, j.f_dyn = a.tf_dyn._handle_for(1), ctypes.pointer(self.f_dyn)
            j.h_obs, j.f_obs = a.tf_obs._handle_for(1), ctypes.pointer(self.f_obs)
            j.B, j.ld, j.T = B, ld, T
            j.d_y, j.d_m0, j.d_P0 = self.d_y.ptr, self.d_m0.ptr, self.d_P0.ptr
            j.GQG, j.R = self.pg, self.pr
            j.d_fm, j.d_fP, j.d_status = bufs[0][0].ptr, bufs[0][1].ptr, bufs[0][2].ptr

    def _single(self, i, bufs):
        j = self.jobs[i]
        self._lib.check(self._lib.load().ssmq_filter_forward_dev(
            ctypes.c_void_p(j.h_dyn), j.f_dyn, ctypes.c_void_p(j.h_obs), j.f_obs, self.B, self.ld, self.T, ctypes.c_void_p(self.d_y.ptr),
            ctypes.c_void_p(self.d_m0.ptr), ctypes.c_void_p(self.d_P0.ptr), self.pg, self.pr, ctypes.c_void_p(bufs[0].ptr),
            ctypes.c_void_p(bufs[1].ptr), ctypes.c_void_p(bufs[2].ptr)), 'ssmq_filter_forward_dev')

    def _timed(self, step, blocks=5, per=20):
        settle(step, self._lib.sync)
        times = []
        for _ in range(blocks):
            e0, e1 = self._lib.Event(), self._lib.Event()
            e0.record()
            for _ in range(per):
                step()
            e1.record()
            times.append(e0.elapsed_ms(e1) / per)
        return float(np.median(times))

    def time_single(self, i):
        return self._timed(lambda: self._single(i, self.out2[i]))

    def time_serial(self):
        def step():
            for i in range(len(self.algs)):
                self._single(i, self.out2[i])
        return self._timed(step)

    def time_multi(self):
        lib = self._lib.load()
        return self._timed(lambda: self._lib.check(lib.ssmq_filter_forward_multi_dev(len(self.algs), self.jobs),
                                                   'ssmq_filter_forward_multi_dev'))

    def check(self):
        """The multi-launch's outputs against the single calls', bit for bit."""
        lib = self._lib.load()
        for i in range(len(self.algs)):
            self._single(i, self.out2[i])
        self._lib.check(lib.ssmq_filter_forward_multi_dev(len(self.algs), self.jobs), 'ssmq_filter_forward_multi_dev')
        self._lib.sync()
        T, ld = self.T, self.ld
        ok = True
        for a, b in zip(self.out, self.out2):
            ok = ok and np.array_equal(a[0].download((T, ld)), b[0].download((T, ld)), equal_nan=True)
            ok = ok and np.array_equal(a[1].download((T, ld)), b[1].download((T, ld)), equal_nan=True)
            ok = ok and np.array_equal(a[2].download((ld,), dtype=np.int32), b[2].download((ld,), dtype=np.int32))
        return bool(ok)

    def free(self):
        for bufs in self.out + self.out2:
            for b in bufs:
                b.free()
        for b in (self.d_y, self.d_m0, self.d_P0):
            b.free()


def measure_study6(amd, B=10000, T=100):
    """Six configs[1]-sized filters over the same measurements (the loop of the reference's UNGM studies) as one launch."""
    st = Study6Bench(amd, B, T, seed=1)
    one = st.time_single(3)                     # the headline filter (GPQKF) alone
    ser = st.time_serial()
    mul = st.time_multi()
    ok = st.check()
    st.free()
    return {'filters': st.names, 'mc': B, 'time_steps': T, 'one_pass_ms': one, 'serial_ms': ser, 'multi_ms': mul, 'x_one_pass': mul / one,
            'serial_x_one_pass': ser / one, 'steps_per_s': len(st.names) * B * T / (mul * 1e-3), 'equal_to_serial_calls': ok,
            'note': 'ssmq_filter_forward_multi_dev: the six filters of the UNGM family as ONE kernel (k_filter_multi_ungm), device-resident; '
                    'x_one_pass = time of the six together / time of the GPQ-Kalman pass alone'}


class Mt6Bench:
    """Batched GPQ moment transform, D = E = 6, N = 13, B = 1e5, rotating buffer sets (> 256 MB in total so that the
    Infinity Cache cannot hold the working set between launches)."""

    def __init__(self, amd, B, seed, nsets=4):
        from ssmtoybox_amd import _lib, ssmod
        self._lib = _lib
        self.B = B
        self.ld = (B + 63) // 64 * 64
        par = np.array([[1.0] + [3.0] * 6])
        self.tf = amd.GaussianProcessTransform(6, 6, par, 'rbf', 'ut')
        self.model = ssmod.ReentryVehicle2DBiasTransition(dt=0.1)
        self.f = self.model.dyn_eval
        self.sets = []
        self.host = []
        for i in range(nsets):
            means, covs = synthetic_reentry6(B, seed + i)
            mean, cov = _lib.SoA.from_host(means), _lib.SoA.from_host(covs)
            mf, cf, cfx = _lib.SoA(6, B), _lib.SoA(36, B), _lib.SoA(36, B)
            st = _lib.DeviceBuffer(4 * mean.ld)
            self.sets.append((mean, cov, mf, cf, cfx, st))
            if i == 0:
                self.host = (means, covs)
        self.time = _lib.DeviceBuffer(8)
        self.time.upload(np.zeros(1))
        self.kernel = self.tf.kernel_name(self.f)
        self.i = 0

    def launch(self):
        mean, cov, mf, cf, cfx, st = self.sets[self.i % len(self.sets)]
        self.i += 1
        self.tf.apply_batch_dev(self.f, mean, cov, self.time, mf, cf, cfx, st, 0)

    def measure(self, warmup=10, iters=100, blocks=5):
        """Median over `blocks` blocks of iters / blocks launches each (HIP events around a block; SURVEY 8d protocol)."""
        settle(self.launch, self._lib.sync)
        for _ in range(warmup):
            self.launch()
        self._lib.sync()
        per = max(1, iters // blocks)
        times = []
        for _ in range(blocks):
            e0, e1 = self._lib.Event(), self._lib.Event()
            e0.record()
            for _ in range(per):
                self.launch()
            e1.record()
            times.append(e0.elapsed_ms(e1) / per)
        ms = float(np.median(times))
        self.block_ms = [float(t) for t in times]
        bytes_alg = 8 * (6 + 36 + 6 + 36 + 36) * self.B          # 960 B per transform at D = E = 6
        bytes_moved = 8 * (6 + 21 + 6 + 36 + 36) * self.B        # what the kernel actually reads + writes (lower tri. in)
        return ms, bytes_alg, bytes_moved

    def free(self):
        for s_ in self.sets:
            for b in s_[:5]:
                b.buf.free()
            s_[5].free()
        self.time.free()

    def check(self):
        """Parity of set 0 against the oracle on a sample (bench is not a test, but never report an unchecked number)."""
        from oracle import ssmq_oracle as orc
        mean, cov, mf, cf, cfx, st = self.sets[0]
        self.i = 0
        self.launch()
        self._lib.sync()
        g_mf, g_cf, g_cfx = mf.to_host(), cf.to_host((6, 6)), cfx.to_host((6, 6))
        w = dict(wm=self.tf.wm, Wc=self.tf.Wc, Wcc=self.tf.Wcc, model_var=self.tf.model.model_var)
        means, covs = self.host
        worst = 0.0
        for i in range(0, self.B, max(1, self.B // 64)):
            r = orc.apply_bq(orc.F_REENTRY2D_BIAS_DYN, means[i], covs[i], 0.0, orc.points_ut(6), w, (0.1,))
            s = float(np.max(np.abs(r[0])))
            worst = max(worst, np.max(np.abs(g_mf[i] - r[0])) / s, np.max(np.abs(g_cf[i] - r[1])) / s ** 2,
                        np.max(np.abs(g_cfx[i] - r[2])) / (s * np.sqrt(np.max(np.abs(covs[i])))))
        return float(worst)



def register_kernel_ms(amd, workload, filt, B, T, seed):
    """The same device-resident pass with the one-trajectory-per-lane kernel forced (SSMQ_FUSED_QUAD=0, SSMQ_FUSED_WSPLIT=0): what a
    leg that now takes k_filter_quad / k_filter_wsplit is compared with."""
    import os
    old = {k: os.environ.get(k) for k in ('SSMQ_FUSED_QUAD', 'SSMQ_FUSED_WSPLIT')}
    os.environ['SSMQ_FUSED_QUAD'], os.environ['SSMQ_FUSED_WSPLIT'] = '0', '0'
    try:
        wl = FilterBench(amd, B, T, seed, workload, filt)
        settle(wl.step, wl._lib.sync)
        ms = timed_passes(wl, 3, 20)
        wl.free()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return ms


def filter_leg(amd, workload, filt, B, T, seed, cpu_sample, cpu_budget, what, with_cpu=True):
    """One extra filter workload: device-resident passes timed with HIP events, algorithmic bytes 8 (Y + D + D^2) per
    filter step, trajectories that fail are counted; the C port timed beside it on a sample and used to
    cross-check the device result on the same trajectories."""
    wl = FilterBench(amd, B, T, seed, workload, filt)
    settle(wl.step, wl._lib.sync)
    ms = timed_passes(wl, 3, 20)
    fm, fP, st = wl.results()
    ach = wl.bytes_per_pass() / (ms * 1e-3) / 1e9
    rec = {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
           'traffic': pmc_traffic(wl.kernel, wl.ld), 'kernel': wl.kernel, 'bytes_per_launch': wl.bytes_per_pass(),
           'ms_per_launch': ms, 'filter_steps_per_s': B * T / (ms * 1e-3), 'failed_trajectories': int((st != 0).sum()),
           'workload': what}
    ib = issue_block(wl.kernel, T, ms)
    if ib:
        rec['issue'] = ib
    if with_cpu:
        cb, cfm, cfP, cst = cpu_baseline_filter(wl, cpu_sample, cpu_budget, what)
        rec['cpu_baseline'] = cb
        good = (st[:cpu_sample] == 0) & (cst == 0)
        rec['status_equal_vs_cpu_port'] = float(np.mean((st[:cpu_sample] == 0) == (cst == 0)))
        if good.any():
            # filtered means of the same trajectories, device vs C port, in standard deviations of the filter's own
            # covariance (|dm_i| / sqrt(P_ii)): scale-free, and meaningful for states whose mean is zero
            D = wl.D
            sd = np.sqrt(np.abs(cfP[np.arange(D), np.arange(D)][:, :, good]))
            rel = np.max(np.abs(fm[:, :, :cpu_sample][:, :, good] - cfm[:, :, good]) / sd, axis=0)
            rec['mean_diff_vs_cpu_port_in_sigmas'] = {'median': float(np.median(rel)), 'p99': float(np.quantile(rel, 0.99)),
                                                      'first_step_max': float(rel[0].max())}
    wl.free()
    return rec


def saturated_sweep(amd, T, batches, base_kernel, base_ms, base_B):
    """The headline filter pass (UNGM GPQ-Kalman) at growing batch sizes: BASELINE's B = 1e4 is 157 waves on 1024 SIMDs;
    this shows what the same kernel does on a full chip.  Trajectories and measurements come from the device simulator.
    Per entry: HBM fraction (24 algorithmic bytes per filter step) and the chip-wide fp64 issue fraction (VALU
    instructions per wave and step from the committed SQ counters x 4 cycles, over all SIMDs)."""
    pm = pmc_issue(base_kernel)
    valu_ws = pm['SQ_INSTS_VALU'] / pm['SQ_WAVES'] / 100.0 if pm else None      # counters were taken at T = 100
    rows = []
    for B in batches:
        if B == base_B:
            ms, kernel, failed = base_ms, base_kernel, None
        else:
            wl = FilterBench(amd, B, T, seed=41, workload='ungm', filt='gpqkf', device_data=True)
            settle(wl.step, wl._lib.sync)
            ms = timed_passes(wl, 2, 10)
            st = wl.d_st.download((wl.ld,), dtype=np.int32)[:B]
            failed, kernel = int((st != 0).sum()), wl.kernel
            wl.free()
        ach = 24.0 * B * T / (ms * 1e-3) / 1e9
        row = {'mc': B, 'ms_per_launch': ms, 'filter_steps_per_s': B * T / (ms * 1e-3), 'achieved': ach, 'unit': 'GB/s',
               'frac': ach / HBM_PEAK_GBS, 'waves_per_simd': (B + 63) // 64 / 1024.0}
        if failed is not None:
            row['failed_trajectories'] = failed
        if valu_ws:
            row['issue_frac_chip'] = (B + 63) // 64 * T * valu_ws * 4.0 / CLOCK_HZ / (1024.0 * ms * 1e-3)
        rows.append(row)
    return rows



def c5_full_record(c5, with_cpu):
    """BASELINE configs[4] as SURVEY 8d restates it (fully-symmetric degree-5 rule, N = 201): the GEMM-shaped stage alone
    (k_fxwc_mfma) and the whole transform in one launch (k_bq_fused), with the flop counted both ways."""
    err = c5.check()
    ms, flop = c5.measure()
    ms_full, cb5 = c5.measure_full_transform(10000, with_cpu)
    tf_s = flop / (ms * 1e-3) / 1e12
    rec = {'bound': 'mfma', 'achieved': tf_s, 'peak': F64_MFMA_PEAK_TF, 'unit': 'TFLOP/s',
           'frac': tf_s / F64_MFMA_PEAK_TF, 'traffic': None, 'kernel': c5.gemm_kernel,
           'flop_per_launch': flop, 'ms_per_launch': ms, 'max_scaled_err_vs_numpy': err,
           'full_transform_ms': ms_full, 'full_transforms_per_s': 10000 / (ms_full * 1e-3),
           'workload': 'Bayes-Sard transform, D=E=10, fully-symmetric DEGREE-5 rule N=201 (padded 208) '
                       'standing in for BASELINE configs[4]\'s 7th-degree rule - the reference has '
                       'degree 3 and 5 only (mtran.py:392) - B=1e4: (1e5 x 208) x (208 x 208) on '
                       'v_mfma_f64_16x16x4_f64'}
    # the whole transform is ONE launch since round 3 (k_bq_fused: factor, points, integrand values into an LDS tile,
    # both matrix-core products and the covariance epilogue; FX never reaches HBM): its matrix-core arithmetic is the
    # main product on 16-row tiles of 224 columns plus the second product of the covariance epilogue
    from ssmtoybox_amd import ssmod
    name_full = c5.tf.kernel_name(ssmod.Smooth10DTransition().dyn_eval)
    # flop, both ways (N = 201 points, E = D = 10, B = 1e4):
    #   algorithmic (the dense products as the reference forms them): 2 B E N^2 (fx Wc) + 2 B E^2 N ((fx Wc) fx')
    #     + 2 B E N D (fx Wcc')
    #   executed on the matrix cores by k_bq_fused since round 4 (Wc = S + S': the zero k-blocks of the triangle are skipped):
    #     per 16-row tile 13 14 / 2 + 13 = 104 tile steps x 4 instructions in the main product + 13 x 8 in C = T fx'; 2048 flop each
    Nn, Ee, Bb = 201, 10, 10000
    flop_alg = 2.0 * Bb * Ee * Nn * Nn + 2.0 * Bb * Ee * Ee * Nn + 2.0 * Bb * Ee * Nn * 10
    tiles = (Bb + 5) // 6
    flop_exec = tiles * 4 * (104 * 4 + 13 * 8) * 2048.0 if name_full == 'k_bq_fused' else 2.0 * c5.M * c5.NP * (c5.NP + 16) + 2.0 * c5.M * c5.NP * 32
    alg_bytes = 10000 * 8.0 * (10 + 100 + 10 + 100 + 100) + 4.0 * 10000
    tr = pmc_traffic_named('k_bq_fused') if name_full == 'k_bq_fused' else None
    rec['full_transform'] = {
        'kernel': name_full, 'ms_per_launch': ms_full, 'bound': 'mfma', 'flop_per_launch': flop_exec,
        'achieved': flop_exec / (ms_full * 1e-3) / 1e12, 'peak': F64_MFMA_PEAK_TF, 'unit': 'TFLOP/s',
        'frac': flop_exec / (ms_full * 1e-3) / 1e12 / F64_MFMA_PEAK_TF,
        'executed_flop_per_launch': flop_exec, 'executed_frac': flop_exec / (ms_full * 1e-3) / 1e12 / F64_MFMA_PEAK_TF,
        'algorithmic_flop_per_launch': flop_alg, 'algorithmic_tflops': flop_alg / (ms_full * 1e-3) / 1e12,
        'algorithmic_over_peak': flop_alg / (ms_full * 1e-3) / 1e12 / F64_MFMA_PEAK_TF,
        'algorithmic_bytes': alg_bytes, 'traffic': tr,
        'traffic_over_algorithmic': (tr / alg_bytes) if tr else None,
        'note': 'frac counts the flop the matrix cores EXECUTE: the kernel forms fx Wc fx\' as C + C\' with C = (fx tril(Wc)) fx\', '
                'fewer than the dense products as the reference forms them (algorithmic_*).  Round 3 (full product): 0.290-0.293 ms'}
    if cb5:
        rec['cpu_baseline'] = cb5
    return rec


def measure_api_rate(B=10000, T=100, reps=5):
    """The headline workload through the drop-in entry point `GaussianProcessKalman.forward_pass_batch`: measurements (1, T, B)
    as a host array in, filtered means (D, T, B) and covariances (D, D, T, B) as host arrays out - upload, layout conversion,
    the fused pass and the download (`forward_pass` returns host arrays: ssinf.py:118).  Wall clock, best of `reps` calls after
    one untimed call; `api_steps_per_s` is NOT the contract's `value` (inputs there are resident in HBM)."""
    from ssmtoybox_amd import ssinf, ssmod
    _, y = simulate_ungm(B, T, 1)
    dyn = ssmod.UNGMTransition(ssmod.GaussRV(1), ssmod.GaussRV(1, cov=np.array([[10.0]])))
    obs = ssmod.UNGMMeasurement(ssmod.GaussRV(1), 1)
    par = np.array([[1.0, 3.0]])
    alg = ssinf.GaussianProcessKalman(dyn, obs, par, par, 'rbf', 'ut')
    yy = np.ascontiguousarray(y[None])                       # (1, T, B)
    alg.forward_pass_batch(yy, raise_on_failure=False)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        alg.forward_pass_batch(yy, raise_on_failure=False)
        ts.append(time.perf_counter() - t0)
    t = min(ts)
    return {'api_steps_per_s': B * T / t, 'api_ms_per_call': 1e3 * t,
            'api_note': 'forward_pass_batch, host arrays in and out (PCIe + layout conversion inclusive), best of %d calls' % reps}
