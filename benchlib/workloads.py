"""Synthetic workloads of bench.py: UNGM / reentry / coordinated-turn trajectories and the device-resident filter pass."""
import ctypes

import numpy as np

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
        self.kernel = self.alg.kernel_name(B)

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

