#!/usr/bin/env python3
"""
bench.py - headline measurement: filter steps/s of the batched Monte-Carlo GPQ-Kalman filter on the UNGM model
(BASELINE.json configs[1]: GaussianProcessTransform (RBF, UT points), D = 1, N = 3, 1e4 MC trajectories per GPU,
T = 100 time steps), plus the achieved-bandwidth figure of the batched GPQ moment transform at state-dim 6 / 1e5
trajectories (BASELINE.json north_star target).

One "step" of this bench = one forward pass of the filter over the whole batch = B * T filter steps (each: two moment
transforms + one measurement update).  Inputs (measurements, initial moments, weights) are resident in HBM before the
timed region; filtered means / covariances of every step are written to HBM (forward_pass returns all of them).

Launch:  python bench.py [--gpus N --steps K --warmup W]          (N = 1)
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU)
Independent MC trajectories shard across ranks with no data-path collective (weak scaling: B per GPU fixed); the only
collective is the final all-reduce of the per-step error sums: RCCL behind the C ABI (ssmq_allreduce_sum), rendezvous
from the launcher's environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT).  No PyTorch in this process; compute
goes python -> ctypes -> libssmq.so (HIP).  (SSMQ_BENCH_BACKEND=gloo swaps in a torch.distributed gloo group for
rehearsals with several ranks on one GPU.)

The N = 1 run also carries the other BASELINE configs as extra blocks of the same JSON line (roofline_mt6, roofline_c3,
roofline_c4, roofline_c5), each with the C oracle timed beside it (cpu_baseline, a bounded sample).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
CLOCK_HZ = 2.4e9          # MI355X peak engine clock
F64_MFMA_PEAK_TF = 78.6  # MI355X fp64 matrix peak = 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz (equals the fp64 vector peak)


def pmc_traffic(kernel_name, grid=None):
    """HBM bytes per launch of `kernel_name` at grid size `grid` (threads) from the committed PMC summary
    (profiles/pmc_traffic.json, produced by profiles/collect_pmc.sh + profiles/pmc_summary.py from separate rocprofv3
    --pmc passes, FETCH_SIZE doubled as the gfx950 note in MI355X_MICROARCH.md prescribes).  Entries are keyed on
    (kernel, grid): the same kernel launched at two batch sizes has two entries.  None if there is no entry for this
    pair - never the figure of another grid."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        table = json.load(open(path))
    except (OSError, ValueError):
        return None
    import re
    base = kernel_name.split('<')[0]
    nums = [int(v) for v in re.findall(r'=(\d+)', kernel_name)]
    if 'fused' in base:
        # reported name: <D=,Y=,ND=,NO=,F_DYN,F_OBS,FORM,TP=,SELO=,OPT=>; profile: <D,Y,ND,NO,FD,FO,FORM,TP,SELO,OPT,STU>
        want = nums[:4] + [1 if 'SSMQ_FORM_SIGMA' in kernel_name else 0] + nums[4:7]
        pick = lambda t: t[:4] + t[6:10]
    else:
        want = nums[:3]                                # (D, E, N) identify the shape
        pick = lambda t: t[:3]
    hits = []
    for key, rec in table.items():
        if key.startswith('_') or key.split('<')[0] != base:
            continue
        have = [int(v) for v in re.findall(r'-?\d+', key.split('<', 1)[1].split('>')[0])]
        if pick(have) == want:
            hits.append(rec)
    if grid is not None:
        hits = [r for r in hits if int(r.get('grid', -1)) == int(grid)]
    return hits[0].get('hbm_bytes_per_launch') if len(hits) == 1 else None


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


def pmc_traffic_named(prefix):
    """HBM bytes per launch of the ONE entry of profiles/pmc_traffic.json whose kernel name (what stands before its template
    arguments and the grid) is `prefix`."""
    try:
        table = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
    except (OSError, ValueError):
        return None
    hits = [rec for key, rec in table.items() if key.split('@')[0].split('<')[0] == prefix]
    return hits[0].get('hbm_bytes_per_launch') if len(hits) == 1 else None


def simulate_ungm(B, T, seed):
    """Synthetic UNGM trajectories + measurements (x0 ~ N(0,1), q ~ N(0,10), r ~ N(0,1): tests/test_ssinf.py:23-30 of the
    reference), vectorised over the batch.  Returns x (T, B), y (T, B)."""
    rng = np.random.default_rng(seed)
    x = np.zeros((T + 1, B))
    x[0] = rng.standard_normal(B)
    q = rng.standard_normal((T, B)) * np.sqrt(10.0)
    r = rng.standard_normal((T, B))
    for k in range(1, T + 1):
        xp = x[k - 1]
        x[k] = 0.5 * xp + 25 * (xp / (1 + xp ** 2)) + 8 * np.cos(1.2 * (k - 1)) + q[k - 1]
    y = 0.05 * x[1:] ** 2 + r
    return x[1:], y


def simulate_reentry(B, T, seed, bias_state=False):
    """Synthetic reentry-vehicle trajectories + radar measurements (tests/test_ssinf.py:53-63 setup of the reference),
    vectorised over the batch; `bias_state` appends the pass-through sixth state of this build's 6-D variant.
    Returns x (D, T, B), y (2, T, B), m0, P0, Q (noise cov), G (noise gain), R."""
    rng = np.random.default_rng(seed)
    m0 = np.array([6500.4, 349.14, -1.8093, -6.7967, 0.6932] + ([0.0] if bias_state else []))
    p0 = np.array([1e-6, 1e-6, 1e-6, 1e-6, 1.0] + ([1e-2] if bias_state else []))
    qd = np.array([2.4064e-5, 2.4064e-5, 1e-6] + ([1e-6] if bias_state else []))
    rd = np.array([1e-6, 0.17e-6])
    D, nq = m0.size, qd.size
    G = np.vstack((np.zeros((2, nq)), np.eye(nq)))
    x = m0[:, None] + np.sqrt(p0)[:, None] * rng.standard_normal((D, B))
    xs, ys = np.zeros((D, T, B)), np.zeros((2, T, B))
    dt, r0, h0, gm0, b0 = 0.1, 6374.0, 13.406, 3.9860e5, -0.59783
    for k in range(T):
        b = b0 * np.exp(x[4])
        rr, vv = np.hypot(x[0], x[1]), np.hypot(x[2], x[3])
        dr = b * np.exp((r0 - rr) / h0) * vv
        gr = -gm0 / rr ** 3
        xn = x.copy()
        xn[0], xn[1] = x[0] + dt * x[2], x[1] + dt * x[3]
        xn[2], xn[3] = x[2] + dt * (dr * x[2] + gr * x[0]), x[3] + dt * (dr * x[3] + gr * x[1])
        x = xn + G.dot(np.sqrt(qd)[:, None] * rng.standard_normal((nq, B)))
        xs[:, k] = x
        ys[:, k] = np.stack((np.hypot(x[0], x[1]), np.arctan2(x[1], x[0]))) + np.sqrt(rd)[:, None] * rng.standard_normal((2, B))
    return xs, ys, m0, np.diag(p0), np.diag(qd), G, np.diag(rd)


def synthetic_reentry6(B, seed):
    """reentry-shaped 6-D batch of means / covariances."""
    rng = np.random.default_rng(seed)
    m0 = np.array([6500.4, 349.14, -1.8093, -6.7967, 0.6932, 0.0])
    p0 = np.array([1e-6, 1e-6, 1e-6, 1e-6, 1.0, 1e-2])
    means = m0 + rng.standard_normal((B, 6)) * np.sqrt(p0)
    a = rng.standard_normal((B, 6, 6)) / np.sqrt(6)
    s = np.sqrt(p0)
    covs = np.einsum('i,bij,bkj,k->bik', s, a, a, s) + 1e-6 * np.diag(p0)
    return means, 0.5 * (covs + covs.transpose(0, 2, 1))


class FilterBench:
    """A sigma-point / BQ Kalman filter on B trajectories, T steps, everything resident on the device.
    workload: 'ungm' (BASELINE configs[1]: GPQ-Kalman, D = 1, N = 3) | 'reentry5' (configs[2] with the reference's 5-D
    model, N = 11) | 'reentry6' (the synthetic 6-D variant, N = 13); filt: 'gpqkf' | 'ukf'."""

    def __init__(self, amd, B, T, seed, workload='ungm', filt='gpqkf', device_data=False):
        from ssmtoybox_amd import _lib, ssmod, ssinf
        from ssmtoybox_amd.mtran import resolve_integrand
        self._lib = _lib
        self.B, self.T = B, T
        self.ld = ld = (B + 63) // 64 * 64
        d_xy = None
        if workload == 'ungm':
            m0, P0 = np.zeros(1), np.eye(1)
            dyn = ssmod.UNGMTransition(ssmod.GaussRV(1), ssmod.GaussRV(1, cov=np.array([[10.0]])))
            obs = ssmod.UNGMMeasurement(ssmod.GaussRV(1), 1)
            if device_data:          # large batches: trajectories and measurements from the device simulator, never on the host
                d_xy = ssmod.simulate_dev(dyn, obs, T, B, seed=seed)[:2]
                self.x_true = y = None
            else:
                self.x_true, y = simulate_ungm(B, T, seed)
                self.x_true, y = self.x_true[None], y[None]
            ell = 3.0
        elif workload == 'ct':
            # BASELINE configs[3]: coordinated-turn dynamics (5 states), four bearing sensors (tests/test_ssinf.py:66-82
            # of the reference); data from the device simulator with HEAVY-TAILED measurement noise: Student-t, 3 degrees
            # of freedom, the covariance the filter is told (scale = (nu - 2) / nu R, research/tpq/tpq_ungm.py:60-63)
            m0 = np.array([1000, 300, 1000, 0, np.deg2rad(-3.0)])
            P0 = np.diag([100, 10, 100, 10, 0.1])
            dt, r1, r2 = 0.1, 0.1, 1.75e-4
            A = np.array([[dt ** 3 / 3, dt ** 2 / 2], [dt ** 2 / 2, dt]])
            Q = np.zeros((5, 5))
            Q[:2, :2], Q[2:4, 2:4], Q[4, 4] = r1 * A, r1 * A, r2 * dt
            sensors = np.vstack((1000 * np.eye(2), -1000 * np.eye(2))).astype(float)
            dyn = ssmod.CoordinatedTurnTransition(ssmod.GaussRV(5, m0, P0), ssmod.GaussRV(5, cov=Q), dt=dt)
            obs = ssmod.BearingMeasurement(ssmod.GaussRV(4, cov=10e-3 * np.eye(4)), 5, state_index=[0, 2],
                                           sensor_pos=sensors)
            sim_obs = ssmod.BearingMeasurement(ssmod.StudentRV(4, scale=(1.0 / 3.0) * 10e-3 * np.eye(4), dof=3.0), 5,
                                               state_index=[0, 2], sensor_pos=sensors)
            d_x, d_y, _ = ssmod.simulate_dev(dyn, sim_obs, T, B, seed=seed)
            self.x_true = d_x.download((T, 5, ld))[:, :, :B].transpose(1, 0, 2)
            y = d_y.download((T, 4, ld))[:, :, :B].transpose(1, 0, 2)
            d_x.free()
            d_y.free()
            ell = 100.0
        else:
            bias = workload == 'reentry6'
            self.x_true, y, m0, P0, Q, G, R = simulate_reentry(B, T, seed, bias)
            cls = ssmod.ReentryVehicle2DBiasTransition if bias else ssmod.ReentryVehicle2DTransition
            dyn = cls(ssmod.GaussRV(m0.size, m0, P0), ssmod.GaussRV(Q.shape[0], cov=Q))
            obs = ssmod.Radar2DMeasurement(ssmod.GaussRV(2, cov=R), m0.size)
            ell = 3.0
        self.D, self.Y = dyn.dim_state, obs.dim_out
        D, Y = self.D, self.Y
        self.y_host, self.m0, self.P0 = y, np.asarray(m0, dtype=float), np.asarray(P0, dtype=float)
        if filt == 'ukf':
            self.alg = ssinf.UnscentedKalman(dyn, obs)
        elif filt == 'bsqkf':
            # the reference's reentry study (research/bsq/bsq_tracking.py:263-281): unisolvent multi-index [0 | I | 2I],
            # model variances overwritten
            mi = np.hstack((np.zeros((D, 1)), np.eye(D), 2 * np.eye(D))).astype(int)
            self.alg = ssinf.BayesSardKalman(dyn, obs, np.array([[1.0] + [1.0] * D]),
                                             np.array([[1.0, 0.9, 0.9] + [1e4] * (D - 2)]), mi, mi, 'ut')
            self.alg.tf_dyn.model.model_var = 2e-6 * np.eye(D)
            self.alg.tf_obs.model.model_var = 0 * np.eye(Y)
        elif filt == 'tpqkf':
            par = np.array([[1.0] + [ell] * (D - 1) + [1.0]]) if workload == 'ct' else np.array([[1.0] + [ell] * D])
            self.alg = ssinf.StudentProcessKalman(dyn, obs, par, par)
        else:
            par = np.array([[1.0] + [ell] * D])
            self.alg = ssinf.GaussianProcessKalman(dyn, obs, par, par, 'rbf', 'ut')
        if d_xy is not None:
            self.d_x, self.d_y = d_xy
        else:
            ybuf = np.zeros((T, Y, ld))
            ybuf[:, :, :B] = y.transpose(1, 0, 2)
            self.d_y = _lib.DeviceBuffer(ybuf.nbytes)
            self.d_y.upload(ybuf)
            xbuf = np.zeros((T, D, ld))                 # true states, same planes as the filter output (error sums)
            xbuf[:, :, :B] = self.x_true.transpose(1, 0, 2)
            self.d_x = _lib.DeviceBuffer(xbuf.nbytes)
            self.d_x.upload(xbuf)
        mb = np.zeros((D, ld))
        mb[:] = m0[:, None]
        Pb = np.zeros((D * D, ld))
        Pb[:] = P0.reshape(-1, 1)
        self.d_m0, self.d_P0 = _lib.DeviceBuffer(mb.nbytes), _lib.DeviceBuffer(Pb.nbytes)
        self.d_m0.upload(mb)
        self.d_P0.upload(Pb)
        self.d_fm = _lib.DeviceBuffer(8 * T * D * ld)
        self.d_fP = _lib.DeviceBuffer(8 * T * D * D * ld)
        self.d_st = _lib.DeviceBuffer(4 * ld)
        self.f_dyn, _ = resolve_integrand(dyn.dyn_eval)
        self.f_obs, _ = resolve_integrand(obs.meas_eval)
        self.h_dyn = self.alg.tf_dyn._handle_for(D)
        self.h_obs = self.alg.tf_obs._handle_for(Y)
        self.gqg, self.pg = _lib.as_c(self.alg.G.dot(self.alg.q_cov).dot(self.alg.G.T))
        self.rr, self.pr = _lib.as_c(self.alg.r_cov)
        self.kernel = self.alg.kernel_name()

    def step(self):
        lib = self._lib.load()
        self._lib.check(lib.ssmq_filter_forward_dev(
            ctypes.c_void_p(self.h_dyn), ctypes.byref(self.f_dyn), ctypes.c_void_p(self.h_obs),
            ctypes.byref(self.f_obs), self.B, self.ld, self.T, ctypes.c_void_p(self.d_y.ptr),
            ctypes.c_void_p(self.d_m0.ptr), ctypes.c_void_p(self.d_P0.ptr), self.pg, self.pr,
            ctypes.c_void_p(self.d_fm.ptr), ctypes.c_void_p(self.d_fP.ptr), ctypes.c_void_p(self.d_st.ptr)),
            'ssmq_filter_forward_dev')

    def results(self):
        """Filtered means (D, T, B), covariances (D, D, T, B), status (B,)."""
        T, D, ld, B = self.T, self.D, self.ld, self.B
        fm = self.d_fm.download((T, D, ld))[:, :, :B].transpose(1, 0, 2)
        fP = self.d_fP.download((T, D, D, ld))[:, :, :, :B].transpose(1, 2, 0, 3)
        st = self.d_st.download((ld,), dtype=np.int32)[:B]
        return fm, fP, st

    def bytes_per_pass(self):
        # bytes_step = 8 (dim_y + D + D^2) per filter step, filter outputs stored every step
        return 8 * (self.Y + self.D + self.D * self.D) * self.B * self.T

    def free(self):
        for b in (self.d_y, self.d_x, self.d_m0, self.d_P0, self.d_fm, self.d_fP, self.d_st):
            b.free()


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


_CPU_PORT = {}


def cpu_port_info():
    """Switch the C port to its -O3 -march=native build, compiled on THIS host when the first baseline leg runs
    (oracle/Makefile: native), and name the host: every cpu_baseline record carries `cpu_model` and `flags`."""
    if not _CPU_PORT:
        from oracle import c_oracle as co
        _CPU_PORT['flags'] = co.use_native()
        _CPU_PORT['cpu_model'] = co.cpu_model()
    return dict(_CPU_PORT)


def host_cores(max_threads=16):
    """Host threads the CPU baseline may use: this process's CPU share, at most 16 (a 1-GPU box's share)."""
    from oracle import c_oracle as co
    cpu_port_info()
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return min(cores, co.max_threads(), max_threads)


def c_port_transforms(wl):
    """The filter of a FilterBench as transform blocks of the C oracle (oracle/ssmq_oracle.c), with the very weights
    the device run uses (BQ: tf.wm / Wc / Wcc / model_var as the HIP weights kernel produced them): what is compared and
    timed is the filter arithmetic, not two evaluations of an ill-conditioned inverse."""
    from oracle import c_oracle as co
    from ssmtoybox_amd.mtran import SigmaPointTransform
    out = []
    for tf, integ, E in ((wl.alg.tf_dyn, wl.f_dyn, wl.D), (wl.alg.tf_obs, wl.f_obs, wl.Y)):
        ci = co.Integrand.make(integ.id, [integ.par[i] for i in range(integ.n_par)],
                               [integ.idx[i] for i in range(integ.n_idx)] if integ.n_idx else None)
        if isinstance(tf, SigmaPointTransform):
            out.append(co.make_transform(1, tf.unit_sp.shape[0], E, tf.unit_sp, tf.wm, np.diag(tf.Wc).copy(),
                                         integrand=ci))
        else:
            mv = tf.model.model_var
            bc = 1 if tf.I_out.shape[0] != E else 0            # dim_out = 1 transforms broadcast the model variance
            emv = (np.asarray(mv, dtype=float) * np.ones((E, E))) if np.ndim(mv) == 0 else np.asarray(mv, dtype=float)
            nu = float(getattr(tf.model, 'nu', 0.0) or 0.0) if type(tf).__name__.startswith('StudentT') else 0.0
            out.append(co.make_transform(0, tf.model.points.shape[0], E, tf.model.points, tf.wm, tf.Wc, tf.Wcc, emv, bc,
                                         nu, tf.model.iK if nu > 0 else None, ci))
    return out


def cpu_baseline_filter(wl, B_sample, budget_s, what):
    """The C oracle's restatement of the same filter pass on the host cores (kind "port"), OpenMP over trajectories, on
    the first B_sample trajectories of the device run, repeated for ~budget_s.  Returns (record, fm (D, T, b), status)."""
    from oracle import c_oracle as co
    (td, k1), (to, k2) = c_port_transforms(wl)
    cores = host_cores()
    T = wl.T
    yb = np.ascontiguousarray(wl.y_host[:, :, :B_sample].transpose(2, 1, 0))
    GQG = wl.alg.G.dot(wl.alg.q_cov).dot(wl.alg.G.T)
    t0 = time.perf_counter()
    fm, fP, st = co.filter_forward(td, to, yb, wl.m0, wl.P0, GQG, wl.alg.r_cov, threads=cores)
    dt = time.perf_counter() - t0
    passes, total = 1, dt
    while total + dt < budget_s and passes < 2000:
        t0 = time.perf_counter()
        co.filter_forward(td, to, yb, wl.m0, wl.P0, GQG, wl.alg.r_cov, threads=cores)
        total += time.perf_counter() - t0
        passes += 1
    rec = {'value': passes * B_sample * T / total, 'unit': 'filter steps/s', 'cores': cores, 'kind': 'port', **cpu_port_info(),
           'sample': '{} passes of the first {} trajectories x T={} of {}, oracle/ssmq_oracle.c, OpenMP over '
                     'trajectories, {:.1f} s'.format(passes, B_sample, T, what, total)}
    return rec, fm.transpose(2, 1, 0), fP.transpose(2, 3, 1, 0), st


def cpu_baseline_apply(tf, integ_id, integ_par, D, E, means, covs, budget_s, what):
    """One batched moment transform in the C oracle (same weights as the device handle), on the host cores."""
    from oracle import c_oracle as co
    cores = host_cores()
    mv = tf.model.model_var
    emv = (np.asarray(mv, dtype=float) * np.ones((E, E))) if np.ndim(mv) == 0 else np.asarray(mv, dtype=float)
    t, keep = co.make_transform(0, D, E, tf.model.points, tf.wm, tf.Wc, tf.Wcc, emv,
                                integrand=co.Integrand.make(integ_id, integ_par))
    t0 = time.perf_counter()
    co.apply_batch(t, means, covs, 0.0, threads=cores)
    dt = time.perf_counter() - t0
    passes, total = 1, dt
    while total + dt < budget_s and passes < 2000:
        t0 = time.perf_counter()
        co.apply_batch(t, means, covs, 0.0, threads=cores)
        total += time.perf_counter() - t0
        passes += 1
    return {'value': passes * means.shape[0] / total, 'unit': 'transforms/s', 'cores': cores, 'kind': 'port', **cpu_port_info(),
            'sample': '{} passes of {} transforms of {}, oracle/ssmq_oracle.c, OpenMP over trajectories, {:.1f} s'.format(
                passes, means.shape[0], what, total)}


def pmc_issue(kernel):
    """SQ counters of a fused filter kernel from the committed summary (profiles/r02_fused_sq.csv: the rocprofv3 --pmc passes
    of tools/pmc_fused.sh over this bench; SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles,
    MI355X_MICROARCH.md).  `kernel`: the name bench.py reports (k_filter_fused<D=..,Y=..,ND=..,NO=..,..,FORM,TP=..,SELO=..,
    OPT=..>); matched against the template arguments <D, Y, ND, NO, FD, FO, FORM, TP, SELO, OPT, STU> of the profile."""
    import csv
    import re
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r0*_fused_sq.csv')))      # the latest round's summary
    path = found[-1] if found else os.path.join(ROOT, 'profiles', 'r02_fused_sq.csv')
    if not kernel.startswith('k_filter_fused<'):
        return None
    nums = [int(v) for v in re.findall(r'=(\d+)', kernel)]
    if len(nums) < 7:
        return None
    want = nums[:4] + [1 if 'SSMQ_FORM_SIGMA' in kernel else 0] + nums[4:7]      # D Y ND NO | FORM | TP SELO OPT
    rows = {}
    try:
        for r in csv.DictReader(open(path)):
            if 'k_filter_fused<' not in r['kernel']:
                continue
            t = [int(v) for v in re.findall(r'-?\d+', r['kernel'].split('<', 1)[1].split('>')[0])]
            if len(t) >= 10 and t[:4] + t[6:10] == want:
                rows[r['counter']] = float(r['mean_per_launch'])
    except (OSError, KeyError, ValueError):
        return None
    need = ('SQ_INSTS_VALU', 'SQ_WAVE_CYCLES', 'SQ_WAVES', 'SQ_ACTIVE_INST_VALU', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY',
            'SQ_INSTS_SALU')
    return rows if all(k in rows for k in need) else None


def issue_block(kernel, T, ms_per_launch):
    """The fused time loops are bound by fp64 VALU issue, not by HBM: instructions of the committed PMC pass against this
    run's HIP-event launch time, per wave."""
    pm = pmc_issue(kernel)
    if not pm:
        return None
    waves = pm['SQ_WAVES']
    valu_wave = pm['SQ_INSTS_VALU'] / waves
    peak = CLOCK_HZ / 4.0       # one fp64 VALU instruction per 4 cycles per SIMD
    achieved = valu_wave / (ms_per_launch * 1e-3)
    return {'bound': 'fp64-issue', 'unit': 'VALU instructions/s per wave', 'achieved': achieved, 'peak': peak,
            'frac': achieved / peak, 'kernel': kernel,
            'valu_instructions_per_wave_per_step': valu_wave / T,
            'salu_instructions_per_wave_per_step': pm['SQ_INSTS_SALU'] / waves / T,
            'pmc': {'source': 'profiles/r0*_fused_sq.csv, latest (rocprofv3 --pmc, tools/pmc_fused.sh)',
                    'frac_valu_x4_over_wave_cycles': pm['SQ_INSTS_VALU'] / pm['SQ_WAVE_CYCLES'],
                    'active_inst_valu_over_wave_cycles': pm['SQ_ACTIVE_INST_VALU'] / pm['SQ_WAVE_CYCLES'],
                    'wait_any_over_wave_cycles': pm['SQ_WAIT_ANY'] / pm['SQ_WAVE_CYCLES'],
                    'wait_inst_any_over_wave_cycles': pm['SQ_WAIT_INST_ANY'] / pm['SQ_WAVE_CYCLES'],
                    'waves': waves, 'simds': 1024},
            'note': 'a wave issues one fp64 VALU instruction per 4 cycles at best; frac = this kernel\'s instructions per '
                    'wave x 4 cycles / its launch time.  Waves beyond one per SIMD share the issue slots: with 1563 waves '
                    'on 1024 SIMDs (B = 1e5) the SIMDs that host two set the time, frac per wave is then at most 0.5'}


def timed_passes(wl, warmup, iters):
    for _ in range(warmup):
        wl.step()
    wl._lib.sync()
    e0, e1 = wl._lib.Event(), wl._lib.Event()
    e0.record()
    for _ in range(iters):
        wl.step()
    e1.record()
    return e0.elapsed_ms(e1) / iters


def settle(step, sync, seconds=0.06):
    """Run `step` untimed for about `seconds`: after the idle gaps between the legs of this script (set-up, host-side
    checks, the CPU baselines) the device needs some 20-50 ms of continuous work before its clocks are back up - a 0.5 ms
    kernel timed right after three warm-up launches read 15-25 % slow (tools/thermal_check.py: 577 / 512 / 482 us for
    consecutive groups of ten passes from idle, 455 us once warm, 572 us again after 2 s of idle).  The headline pass is
    not affected (32.1-32.4 us with 10, 500 or 3000 warm-up steps) and keeps exactly the --warmup it is given."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(5):
            step()
        sync()


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


def make_comm():
    """Communicator from the launcher's environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*): RCCL behind the C ABI
    (default; no PyTorch), or SSMQ_BENCH_BACKEND=gloo - a torch.distributed gloo group, for rehearsals with several ranks
    on one GPU or none.  Returns (comm, rank, world, local_rank)."""
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    backend = os.environ.get('SSMQ_BENCH_BACKEND', 'rccl')
    from ssmtoybox_amd import mcshard, _lib
    have = _lib.device_count()
    ndev = max(have, 1)
    local_rank %= ndev
    if have > 0:                                 # (none: the stand-in ranks of tests/test_rccl_stub.py on a machine without a GPU)
        _lib.set_device(local_rank)
    launched = world > 1 or ('RANK' in os.environ and 'MASTER_PORT' in os.environ)
    if not launched:
        return mcshard.SingleComm(), 0, 1, local_rank
    why = ''
    if backend != 'gloo' and world > ndev and os.environ.get('SSMQ_BENCH_FORCE_RCCL') != '1':
        # RCCL refuses two ranks on one device; every rank sees the same device count, so all of them take this branch
        backend, why = 'gloo', '{} ranks on {} device(s)'.format(world, ndev)
        if rank == 0:
            sys.stderr.write('bench.py: {} - RCCL needs one device per rank, all-reduce over gloo\n'.format(why))
    if backend == 'gloo':
        import torch.distributed as dist
        dist.init_process_group('gloo')
        comm = mcshard.TorchComm(dist)
        comm.fallback_reason = why
        return comm, rank, world, local_rank
    comm = mcshard.open_comm(rank, world, force_rccl=os.environ.get('SSMQ_BENCH_FORCE_RCCL') == '1',
                             log=lambda m: sys.stderr.write(m + '\n'))
    return comm, rank, world, local_rank


def final_aggregation(comm, rank, world, loc, lcr_sums_of, pass_ms_dev, B):
    """What every rank does after its timed passes - the path's only collectives:
    phase 1: this rank's per-time-step error sums `loc` (mcshard.device_error_sums: reduced on the device from the filter's
    output buffers), ONE all-reduce of the packed buffer; phase 2: log credibility ratio against the GLOBAL per-step MSE matrix
    (`lcr_sums_of(mse)` -> this rank's sums), a second all-reduce; then the per-rank launch times and trajectory counts (one
    slot per rank, summed) and what the final collective costs: the packed phase-1 buffer all-reduced 20 times after a common
    start (every rank takes part: collective calls).  Shared by main() and the stand-in ranks of tests/test_rccl_stub.py."""
    from ssmtoybox_amd import mcshard
    agg = mcshard.finalize(mcshard.allreduce_sums(loc, comm))
    lcr = mcshard.finalize_lcr(mcshard.allreduce_sums(lcr_sums_of(agg['mse']), comm))
    slot = np.zeros(2 * world)
    slot[rank], slot[world + rank] = pass_ms_dev, B
    slot = comm.allreduce_sum(slot)
    n_packed = sum(int(np.asarray(v).size) for v in loc.values())
    lat = []
    comm.barrier()
    for _ in range(20):
        t1 = time.perf_counter()
        comm.allreduce_sum(np.zeros(n_packed))
        lat.append(time.perf_counter() - t1)
    return dict(agg=agg, lcr=lcr, slot=slot, allreduce_us=float(np.median(lat)) * 1e6, n_packed=n_packed)


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


def free_port():
    import socket
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def child_env(rank, world, port, id_file, base=None):
    """Environment of rank `rank` of a self-spawned launch: what torch.distributed.run would export (RANK, LOCAL_RANK,
    WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR, MASTER_PORT) plus the explicit rendezvous file of the RCCL id, so the
    ranks do not depend on sharing a parent pid."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), SSMQ_RCCL_ID_FILE=id_file, SSMQ_BENCH_CHILD='1')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC only on this pool (RCCL across processes)
    return env


def needs_launcher(gpus, env=None):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: this process starts the N ranks itself."""
    env = os.environ if env is None else env
    return gpus > 1 and 'WORLD_SIZE' not in env and 'RANK' not in env


def launch_ranks(gpus, argv, timeout_s=1500.0, script=None):
    """Start `gpus` fresh processes of this file (one rank per GPU), relay rank 0's JSON line, return the exit code.
    This process never touches the GPU (children are started with subprocess, not exec)."""
    import subprocess
    import tempfile
    tmp = tempfile.mkdtemp(prefix='ssmq_bench_')
    id_file = os.path.join(tmp, 'rccl.id')
    port = free_port()
    me = os.path.abspath(script or __file__)
    procs = []
    for r in range(gpus):
        procs.append(subprocess.Popen([sys.executable, me] + list(argv), env=child_env(r, gpus, port, id_file),
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    rc, line = 0, None
    t_end = time.time() + timeout_s
    try:
        text, _ = procs[0].communicate(timeout=max(1.0, t_end - time.time()))
        for ln in text.decode('utf-8', 'replace').splitlines():
            if ln.startswith('{'):
                line = ln
            elif ln.strip():
                sys.stderr.write(ln + '\n')
        for pr in procs:
            pr.wait(timeout=max(1.0, t_end - time.time()))
    except subprocess.TimeoutExpired:
        sys.stderr.write('bench.py: ranks did not finish within {:.0f} s\n'.format(timeout_s))
        rc = 124
    for r, pr in enumerate(procs):
        if pr.poll() is None:
            pr.kill()
            pr.wait()
        if pr.returncode and not rc:
            sys.stderr.write('bench.py: rank {} exited with code {}\n'.format(r, pr.returncode))
            rc = pr.returncode if pr.returncode > 0 else 1
    for name in os.listdir(tmp):
        try:
            os.unlink(os.path.join(tmp, name))
        except OSError:
            pass
    try:
        os.rmdir(tmp)
    except OSError:
        pass
    if line is None:
        sys.stderr.write('bench.py: rank 0 printed no result line\n')
        return rc or 1
    out = json.loads(line)
    out.setdefault('config', {})['launcher'] = 'bench.py --gpus {}: {} child processes, one rank per GPU'.format(gpus, gpus)
    print(json.dumps(out))
    if out.get('n_gpus') != gpus:
        sys.stderr.write('bench.py: result line reports n_gpus = {} for --gpus {}\n'.format(out.get('n_gpus'), gpus))
        return rc or 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=10000, help='MC trajectories per GPU (weak scaling, the default)')
    ap.add_argument('--total-batch', type=int, default=0,
                    help='strong scaling: this many MC trajectories in total, split over the ranks in contiguous slices '
                         '(mcshard.shard_bounds); e.g. BASELINE configs[2]: --workload reentry6 --filter ukf '
                         '--total-batch 100000 --time-steps 50')
    ap.add_argument('--time-steps', type=int, default=100)
    ap.add_argument('--workload', default='ungm', choices=['ungm', 'reentry5', 'reentry6', 'ct'],
                    help="'ungm' is the headline (BASELINE configs[1]); the others are extra measurements")
    ap.add_argument('--filter', default='gpqkf', choices=['gpqkf', 'ukf', 'tpqkf', 'bsqkf'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-mt6', action='store_true', help='skip the single-kernel / other-config legs of the N = 1 run')
    args = ap.parse_args()

    if needs_launcher(args.gpus):
        # one rank per GPU, started from here; nothing above or in this branch initialises the GPU
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    # the result line goes to the descriptor stdout had at start-up, whatever a library does to fd 1 later
    result_out = os.fdopen(os.dup(1), 'w')
    import ssmtoybox_amd as amd
    if amd.device_count() < 1:
        raise SystemExit('bench.py needs a GPU: the HIP path has no CPU fallback')
    from ssmtoybox_amd import _lib, mcshard
    comm, rank, world, local_rank = make_comm()
    if world != args.gpus and rank == 0:
        sys.stderr.write('bench.py: --gpus {} but the launcher started {} rank(s); reporting n_gpus = {}\n'.format(
            args.gpus, world, world))

    B, T = args.batch, args.time_steps
    strong = args.total_batch > 0
    if strong:
        lo, hi = mcshard.shard_bounds(args.total_batch, rank, world)
        B = hi - lo
        if B < 1:
            raise SystemExit('bench.py: --total-batch {} leaves rank {} of {} without trajectories'.format(
                args.total_batch, rank, world))
    wl = FilterBench(amd, B, T, seed=1 + rank, workload=args.workload, filt=args.filter)
    for _ in range(args.warmup):
        wl.step()
    comm.barrier()                      # common start: device synchronisation + barrier on every rank
    ev0, ev1 = _lib.Event(), _lib.Event()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        wl.step()
    ev1.record()
    _lib.sync()                         # this rank's K steps are complete ...
    elapsed = time.perf_counter() - t0
    pass_ms_dev = ev0.elapsed_ms(ev1) / max(args.steps, 1)
    comm.barrier()                      # ... closing barrier; the job's time is the slowest rank's
    elapsed = float(comm.allreduce_max(np.array([elapsed]))[0])

    # final aggregation: per-time-step error sums -> RMSE / NLL (the path's only collective)
    fa = final_aggregation(
        comm, rank, world, mcshard.device_error_sums(wl.D, B, wl.ld, T, wl.d_x, wl.d_fm, wl.d_fP, wl.d_st),
        lambda mse: mcshard.device_lcr_sums(wl.D, B, wl.ld, T, wl.d_x, wl.d_fm, wl.d_fP, mse, wl.d_st), pass_ms_dev, B)
    agg, lcr, slot, allreduce_us, n_packed = fa['agg'], fa['lcr'], fa['slot'], fa['allreduce_us'], fa['n_packed']
    rmse, nll = agg['rmse_total'], float(agg['nll_avg'].mean())

    out = None
    headline = args.workload == 'ungm' and args.filter == 'gpqkf'
    if rank == 0:
        b_total = int(round(slot[world:].sum()))          # trajectories of all ranks (world x B when weak)
        steps_total = b_total * T * args.steps
        value = steps_total / elapsed
        bytes_pass = wl.bytes_per_pass()
        ach = bytes_pass / (pass_ms_dev * 1e-3) / 1e9
        out = {
            'metric': 'filter steps/sec (batched MC) for GPQ-Kalman UNGM' if headline else
            'filter steps/sec (batched MC), {} {}'.format(args.filter, args.workload),
            'value': value, 'unit': 'filter steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': ('GaussianProcessTransform (RBF, UT points) GPQ-Kalman on UNGM, D=1, N=3, '
                                    '{} MC trajectories per GPU x T={} time steps per pass (BASELINE configs[1])'.format(B, T))
                       if headline else
                       '{} on {} (D={}, Y={}), {} MC trajectories per GPU x T={}'.format(args.filter, args.workload, wl.D,
                                                                                       wl.Y, B, T),
                       'mc_per_gpu': B, 'mc_total': b_total, 'time_steps': T, 'parallelism': 'mc-shard x{}'.format(world),
                       'per_rank_kernel_ms': [float(v) for v in slot[:world]],
                       'per_rank_trajectories': [int(round(v)) for v in slot[world:]],
                       'allreduce_us': allreduce_us, 'allreduce_bytes': 8 * n_packed,
                       'collective': type(comm).__name__ + (
                           ' (gloo fallback: ' + comm.fallback_reason + ')' if getattr(comm, 'fallback_reason', '') else '')},
            'roofline': {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': ach / HBM_PEAK_GBS, 'traffic': pmc_traffic(wl.kernel, wl.ld),
                         'kernel': wl.kernel,
                         'bytes_per_launch': bytes_pass, 'ms_per_launch': pass_ms_dev,
                         'note': 'the contract\'s HBM figure; this kernel is a serial recursion per trajectory and is '
                                 'bound by what ONE wave per SIMD can issue (157 waves for 1024 SIMDs at B=1e4), see '
                                 'roofline_issue and DESIGN.md 3.4'},
            'rmse': rmse, 'nll': nll, 'inclination_indicator': float(np.mean(lcr)),
            'trajectories_aggregated': int(agg['count']),
            # what the averages above leave out (summed over ranks, worst time step): failed filters / singular covariances
            'excluded_failed_trajectories': int(agg['excluded_failed'].max()) if T else 0,
            'excluded_singular_covariances': int(agg['excluded_not_pd'].max()) if T else 0,
        }
        ib = issue_block(wl.kernel, T, pass_ms_dev)
        if ib:
            out['roofline_issue'] = ib
    single = world == 1      # the single-kernel legs and the CPU baselines belong to the N = 1 run only
    with_cpu = not args.no_cpu_baseline
    if rank == 0 and single and with_cpu and headline:
        cb, cpu_fm, _, cpu_st = cpu_baseline_filter(wl, B, 8.0, 'UNGM GPQ-Kalman (configs[1])')
        out['cpu_baseline'] = cb
        # the GPU pass and the CPU port ran the same trajectories: cross-check them
        fm, _, st = wl.results()
        good = (st == 0) & (cpu_st == 0)
        rel = np.abs(fm[0][:, good] - cpu_fm[0][:, good]) / np.max(np.abs(cpu_fm[0][:, good]))
        # identical weights and measurements; the UNGM recursion amplifies rounding differences along a trajectory
        # (uncentred covariance, bq/bqmtran.py:199), hence median and max over the 1e6 filtered means
        out['rel_diff_vs_cpu_port'] = {'median': float(np.median(rel)), 'p99': float(np.quantile(rel, 0.99)),
                                       'max': float(rel.max())}
    if rank == 0 and single and not args.no_mt6:
        mt = Mt6Bench(amd, 100000, seed=2)
        err = mt.check()
        ms, b_alg, b_mov = mt.measure()
        ach = b_alg / (ms * 1e-3) / 1e9
        out['roofline_mt6'] = {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                               'frac': ach / HBM_PEAK_GBS, 'traffic': pmc_traffic(mt.kernel, mt.ld), 'kernel': mt.kernel,
                               'bytes_per_launch': b_alg, 'bytes_moved_per_launch': b_mov, 'ms_per_launch': ms,
                               'transforms_per_s': mt.B / (ms * 1e-3), 'max_scaled_err_vs_oracle': err,
                               'workload': 'batched GPQ moment transform, D=E=6, N=13, B=1e5, 4 rotating buffer sets '
                                           '(north_star target; BASELINE configs[2] transform shape)'}
        # the same figures inside `roofline`, the block the driver's record keeps (north_star: >= 0.40 on this kernel)
        out['roofline']['target'] = {k: out['roofline_mt6'][k] for k in (
            'kernel', 'frac', 'achieved', 'unit', 'ms_per_launch', 'bytes_per_launch', 'traffic', 'max_scaled_err_vs_oracle')}
        out['roofline']['target']['workload'] = 'north_star: batched GPQ moment transform D=E=6, N=13, B=1e5 (>= 0.40 asked)'
        # ... and once more as SCALARS of `roofline`: the driver's record keeps scalar fields only
        r6 = out['roofline_mt6']
        out['roofline'].update({
            'target_kernel': r6['kernel'], 'target_frac': r6['frac'], 'target_achieved_gbs': r6['achieved'],
            'target_ms_per_launch': r6['ms_per_launch'], 'target_bytes_per_launch': r6['bytes_per_launch'],
            'target_bytes_moved_per_launch': r6['bytes_moved_per_launch'], 'target_traffic': r6['traffic'],
            'target_max_scaled_err_vs_oracle': r6['max_scaled_err_vs_oracle'],
            'target_timing': 'median of {} blocks of {} launches, HIP events'.format(len(mt.block_ms), max(1, 100 // len(mt.block_ms))),
            'target_ms_min_block': min(mt.block_ms), 'target_ms_max_block': max(mt.block_ms)})
        out['roofline_mt6']['block_ms'] = mt.block_ms
        if with_cpu:
            means, covs = mt.host
            out['roofline_mt6']['cpu_baseline'] = cpu_baseline_apply(
                mt.tf, mt.model._fid, (0.1,), 6, 6, means[:50000], covs[:50000], 3.0, 'the D=E=6 GPQ transform')
        mt.free()
        # the same kernel with ten generations of waves instead of one (B = 1e6): how close it gets to HBM when the
        # load / compute / store phases of different waves overlap
        mt = Mt6Bench(amd, 1000000, seed=12, nsets=2)
        ms, b_alg, _ = mt.measure(warmup=3, iters=30)
        out['roofline_mt6']['at_1e6_trajectories'] = {'ms_per_launch': ms, 'achieved': b_alg / (ms * 1e-3) / 1e9,
                                                      'unit': 'GB/s', 'frac': b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                      'bytes_per_launch': b_alg}
        mt.free()
        out['roofline']['target']['frac_at_1e6'] = out['roofline_mt6']['at_1e6_trajectories']['frac']
        out['roofline']['target_frac_at_1e6'] = out['roofline_mt6']['at_1e6_trajectories']['frac']
        if headline:
            out['roofline']['saturated'] = saturated_sweep(amd, T, (10000, 100000, 1000000), wl.kernel, pass_ms_dev, B)
            for row in out['roofline']['saturated']:
                tag = {10000: '1e4', 100000: '1e5', 1000000: '1e6'}.get(row['mc'], str(row['mc']))
                out['roofline']['saturated_frac_' + tag] = row['frac']
                if 'issue_frac_chip' in row:
                    out['roofline']['saturated_issue_frac_chip_' + tag] = row['issue_frac_chip']
    if rank == 0 and single and not args.no_mt6:
        # BASELINE configs[2]: the filters that are stable on the reentry model (the GPQ-Kalman recursion itself fails
        # within three steps on every trajectory, in the reference as here: tests/test_gpu_parity.py::test_config3_gpqkf_*)
        out['roofline_c3'] = {
            'ukf_reentry5': filter_leg(amd, 'reentry5', 'ukf', 100000, 50, 31, 4000, 3.0,
                                       'UKF, reentry 5-D + radar (the reference\'s model), B=1e5 x T=50', with_cpu),
            'bsqkf_reentry5': filter_leg(amd, 'reentry5', 'bsqkf', 100000, 50, 32, 4000, 3.0,
                                         'Bayes-Sard Kalman (research/bsq/bsq_tracking.py set-up), reentry 5-D + radar, '
                                         'B=1e5 x T=50', with_cpu),
            'ukf_reentry6': filter_leg(amd, 'reentry6', 'ukf', 100000, 50, 33, 4000, 3.0,
                                       'UKF, reentry-shaped 6-D + radar (BASELINE state-dim 6), B=1e5 x T=50', with_cpu),
        }
        # BASELINE configs[3]: t-process quadrature Kalman filter, 5-D coordinated turn + four bearing sensors
        out['roofline_c4'] = filter_leg(amd, 'ct', 'tpqkf', 10000, 20, 34, 2000, 3.0,
                                        'TPQ-Kalman (StudentProcessKalman), coordinated turn 5-D + 4 bearings, B=1e4 x T=20',
                                        with_cpu)
    if rank == 0 and single and not args.no_mt6:
        c5 = C5GemmBench(amd, 10000, seed=5)
        err = c5.check()
        ms, flop = c5.measure()
        ms_full, cb5 = c5.measure_full_transform(10000, with_cpu)
        tf_s = flop / (ms * 1e-3) / 1e12
        out['roofline_c5'] = {'bound': 'mfma', 'achieved': tf_s, 'peak': F64_MFMA_PEAK_TF, 'unit': 'TFLOP/s',
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
        name_full = c5.tf.kernel_name(__import__('ssmtoybox_amd').ssmod.Smooth10DTransition().dyn_eval)
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
        out['roofline_c5']['full_transform'] = {
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
            out['roofline_c5']['cpu_baseline'] = cb5
        out['roofline_c5']['unisolvent_n21'] = measure_c5_unisolvent(amd)
        out['roofline_c5']['degree7_as_worded'] = measure_c5_degree7(amd, with_cpu=with_cpu)
    if rank == 0 and single and not args.no_mt6:
        out['theta_step'] = measure_theta_step()
        out['roofline_linear'] = measure_linearize()
    if rank == 0:
        result_out.write(json.dumps(out) + '\n')
        result_out.flush()
    wl.free()
    comm.close()
    if getattr(comm, 'abandoned_rccl_thread', False):
        sys.stdout.flush()
        os._exit(0)          # a helper thread is still inside ncclCommInitRank: no atexit handler may wait for it


if __name__ == '__main__':
    main()
