"""
State-space models: the closed-form dynamics / measurement functions of the reference's `ssmtoybox/ssmod.py`, each tied
to its device integrand (include/ssmq.h `enum ssmq_integrand_id`) so that a moment transform given `model.dyn_eval` /
`model.meas_eval` evaluates it on the GPU and sigma points never leave HBM.

Only what the moment-transform path needs is here: dimensions, noise additivity, noise gain, the integrand descriptor and
a NumPy evaluation of the same formula for callers that want function values on the host (it is never used by
`apply()`).  `simulate_discrete` / `simulate_continuous` / `simulate_measurements` (ssmod.py:168-244, 1011-1039) run on
the device (`ssmq_simulate_rv_dev`: counter-based Philox generator, so results match the reference's np.random streams
only statistically) for Gaussian, Student-t and Gaussian-mixture random variables; Jacobians stay in the reference.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import Integrand


class GaussRV:
    """Mean / covariance carrier with the reference's `get_stats()` protocol (utils.py:580-625); sampling happens on the
    device inside the simulators."""

    def __init__(self, dim, mean=None, cov=None):
        self.dim = dim
        self.mean = np.zeros(dim) if mean is None else np.atleast_1d(np.asarray(mean, dtype=float))
        self.cov = np.eye(dim) if cov is None else np.atleast_2d(np.asarray(cov, dtype=float))

    def get_stats(self):
        return self.mean, self.cov


class StudentRV:
    """Mean / scale matrix / degrees of freedom with the reference's `get_stats()` protocol (utils.py:628-674)."""

    def __init__(self, dim, mean=None, scale=None, dof=3.0):
        self.dim = dim
        self.mean = np.zeros(dim) if mean is None else np.atleast_1d(np.asarray(mean, dtype=float))
        self.scale = np.eye(dim) if scale is None else np.atleast_2d(np.asarray(scale, dtype=float))
        self.dof = 3.0 if dof <= 2.0 else dof

    def get_stats(self):
        return self.mean, self.scale, self.dof


class GaussianMixtureRV:
    """Gaussian mixture (research/tpq/tpq_base.py:13-32, sampled by utils.gauss_mixture utils.py:254-299): tuples of
    means and covariances, mixing proportions `alphas`."""

    def __init__(self, dim, means=None, covs=None, alphas=None):
        if covs is None or alphas is None or len(covs) != len(alphas):
            raise ValueError('Same number of means, covariances and mixture weights needs to be supplied!')
        self.dim = dim
        self.covs = tuple(np.atleast_2d(np.asarray(c, dtype=float)) for c in covs)
        self.means = tuple(np.zeros(dim) for _ in covs) if means is None else tuple(
            np.atleast_1d(np.asarray(m, dtype=float)) for m in means)
        if len(self.means) != len(self.covs):
            raise ValueError('Same number of means, covariances and mixture weights needs to be supplied!')
        self.alphas = np.asarray(alphas, dtype=float)

    def get_stats(self):
        return self.means, self.covs, self.alphas


def _lower_factor(cov):
    cov = np.atleast_2d(np.asarray(cov, dtype=np.float64))
    try:
        L = np.linalg.cholesky(cov)
    except np.linalg.LinAlgError:
        # positive SEMI-definite (e.g. a noise-free model): any A with A A' = cov serves; make it lower triangular
        lam, V = np.linalg.eigh(0.5 * (cov + cov.T))
        if lam.min() < -1e-12 * max(1.0, abs(lam).max()):
            raise
        A = V * np.sqrt(np.clip(lam, 0.0, None))
        L = np.linalg.qr(A.T)[1].T
    return np.ascontiguousarray(L)


def _gauss_stats(rv, what):
    """Mean and lower Cholesky factor of a Gaussian random variable (the device simulator draws mean + L z)."""
    if not isinstance(rv, GaussRV):
        raise NotImplementedError('a GaussRV is needed for ' + what)
    return np.ascontiguousarray(rv.mean, dtype=np.float64), _lower_factor(rv.cov)


def _rv_desc(rv, what):
    """struct ssmq_rv for one of the reference's random variables (+ the arrays it points to, to be kept alive)."""
    p = lambda a: a.ctypes.data_as(_lib.c_double_p)
    d = _lib.Rv()
    if isinstance(rv, GaussRV):
        mean, chol = np.ascontiguousarray(rv.mean, dtype=np.float64), _lower_factor(rv.cov)
        d.kind, d.dim, d.n_comp, d.dof, alpha = _lib.RV_GAUSS, mean.size, 1, 0.0, None
    elif isinstance(rv, StudentRV):
        mean, chol = np.ascontiguousarray(rv.mean, dtype=np.float64), _lower_factor(rv.scale)
        d.kind, d.dim, d.n_comp, d.dof, alpha = _lib.RV_STUDENT, mean.size, 1, float(rv.dof), None
    elif isinstance(rv, GaussianMixtureRV):
        mean = np.ascontiguousarray(np.stack(rv.means), dtype=np.float64)
        chol = np.ascontiguousarray(np.stack([_lower_factor(c) for c in rv.covs]))
        alpha = np.ascontiguousarray(rv.alphas / rv.alphas.sum(), dtype=np.float64)
        d.kind, d.dim, d.n_comp, d.dof = _lib.RV_MIXTURE, rv.dim, len(rv.covs), 0.0
    else:
        raise NotImplementedError('{}: GaussRV, StudentRV or GaussianMixtureRV expected'.format(what))
    d.mean, d.chol = p(mean), p(chol)
    d.alpha = p(alpha) if alpha is not None else None
    return d, (mean, chol, alpha)


def simulate_dev(dyn, obs, steps, mc_sims, seed=0, traj_offset=0, continuous_dt=None):
    """States and measurements of `mc_sims` trajectories generated on the device, left there in the filter's layout:
    returns (d_x, d_y, ld) with d_x planes [steps][D][ld], d_y [steps][Y][ld] (DeviceBuffers; caller frees).
    x[0] ~ init_rv, x[k] = dyn_fcn(x[k-1], q[k-1], k-1), y[k] = meas_fcn(x[k], r[k], k+1) (ssmod.py:168-199, 1011-1039);
    with `continuous_dt` the states are the Euler-Maruyama steps of dyn_fcn_cont instead (ssmod.py:201-244).
    Trajectory b uses the random stream of global index traj_offset + b.  obs = None: states only (d_y is None)."""
    lib = _lib.load()
    D = dyn.dim_state
    Y = obs.dim_out if obs is not None else 0
    ld = (mc_sims + 63) // 64 * 64
    x0, k0 = _rv_desc(dyn.init_rv, 'the initial state')
    q, k1 = _rv_desc(dyn.noise_rv, 'the process noise')
    G = np.ascontiguousarray(dyn.noise_gain, dtype=np.float64)
    f_dyn, _ = dyn.device_integrand()
    d_x = _lib.DeviceBuffer(8 * steps * D * ld)
    d_y = f_obs = r = None
    if obs is not None:
        r, k2 = _rv_desc(obs.noise_rv, 'the measurement noise')
        f_obs, _ = obs.device_integrand()
        d_y = _lib.DeviceBuffer(8 * steps * Y * ld)
    _lib.check(lib.ssmq_simulate_rv_dev(ctypes.byref(f_dyn), ctypes.byref(f_obs) if obs is not None else None, D, Y,
                                        ctypes.byref(x0), ctypes.byref(q), ctypes.byref(r) if obs is not None else None,
                                        G.ctypes.data_as(_lib.c_double_p), 1 if dyn.noise_additive else 0,
                                        1 if (obs is None or obs.noise_additive) else 0, mc_sims, ld, steps,
                                        0 if continuous_dt is None else 1, 0.0 if continuous_dt is None else float(continuous_dt),
                                        seed, traj_offset, ctypes.c_void_p(d_x.ptr),
                                        ctypes.c_void_p(d_y.ptr) if d_y is not None else None), 'ssmq_simulate_rv_dev')
    return d_x, d_y, ld


class TransitionModel:
    """x_{k+1} = f(x_k, q_k, k)   (ssmod.py:10-244)."""
    dim_state = None
    dim_noise = None
    noise_additive = True
    _fid = None

    def __init__(self, init_rv=None, noise_rv=None, noise_gain=None):
        self.dim_in = self.dim_state if self.noise_additive else self.dim_state + self.dim_noise
        self.init_rv, self.noise_rv = init_rv, noise_rv
        self.zero_q = np.zeros(self.dim_noise)
        self.noise_gain = np.eye(self.dim_state, self.dim_noise) if noise_gain is None else noise_gain

    def _par(self):
        return ()

    def device_integrand(self):
        """(ssmq_integrand, dim_out) for the C ABI."""
        return Integrand.make(self._fid, self._par()), self.dim_state

    def dyn_fcn(self, x, q, time):
        raise NotImplementedError

    def simulate_discrete(self, steps, mc_sims=1, seed=0, traj_offset=0):
        """(dim_state, steps, mc_sims) state trajectories (ssmod.py:168-199), generated on the device.  The reference
        draws from the global np.random state; here the stream is named by `seed` (and the trajectory's global index)."""
        d_x, _, ld = simulate_dev(self, None, steps, mc_sims, seed, traj_offset)
        x = d_x.download((steps, self.dim_state, ld))[:, :, :mc_sims].transpose(1, 0, 2)
        d_x.free()
        return np.ascontiguousarray(x)

    def dyn_fcn_cont(self, x, q, time):
        """Continuous-time dynamics dx/dt (ssmod.py:81-104): defined by the reentry and constant-turn-rate models only."""
        return None

    def simulate_continuous(self, duration, dt=0.1, mc_sims=1, seed=0, traj_offset=0):
        """(dim_state, floor(duration / dt), mc_sims) Euler-Maruyama trajectories of the continuous-time dynamics
        (ssmod.py:201-244: x[k] = x[k-1] + dt dyn_fcn_cont(x[k-1], (sqrt(dt) / dt) q[k-1], k-1), the initial state is
        not returned), generated on the device."""
        steps = int(np.floor(duration / dt))
        d_x, _, ld = simulate_dev(self, None, steps, mc_sims, seed, traj_offset, continuous_dt=dt)
        x = d_x.download((steps, self.dim_state, ld))[:, :, :mc_sims].transpose(1, 0, 2)
        d_x.free()
        return np.ascontiguousarray(x)

    def dyn_eval(self, xq, time, dx=False):
        """Noise-additivity-aware evaluation (ssmod.py:129-166): additive models are evaluated at zero noise."""
        if dx:                          # ssmod.py:153-165; None for the models without dyn_fcn_dx, as in the reference
            if self.noise_additive:
                return self.dyn_fcn_dx(xq, self.zero_q, time)
            return self.dyn_fcn_dx(xq[:self.dim_state], xq[-self.dim_noise:], time)
        if self.noise_additive:
            return self.dyn_fcn(xq, self.zero_q, time)
        return self.dyn_fcn(xq[:self.dim_state], xq[-self.dim_noise:], time)

    def dyn_fcn_dx(self, x, q, time):
        """Jacobian of the dynamics (ssmod.py:105-127): implemented by UNGM, UNGM-NA, pendulum and constant velocity only."""
        return None


class UNGMTransition(TransitionModel):
    """ssmod.py:247-275."""
    dim_state, dim_noise, noise_additive, _fid = 1, 1, True, _lib.F_UNGM_DYN

    def dyn_fcn(self, x, q, time):
        return np.asarray(0.5 * x[0] + 25 * (x[0] / (1 + x[0] ** 2)) + 8 * np.cos(1.2 * time)) + q

    def dyn_fcn_dx(self, x, q, time):
        return np.asarray([[0.5 + 25 * (1 - x[0] ** 2) / (1 + x[0] ** 2) ** 2]])          # ssmod.py:271-272


class UNGMNATransition(TransitionModel):
    """ssmod.py:278-306 (non-additive noise: input [x, q])."""
    dim_state, dim_noise, noise_additive, _fid = 1, 1, False, _lib.F_UNGMNA_DYN

    def dyn_fcn(self, x, q, time):
        return np.asarray(0.5 * x[0] + 25 * (x[0] / (1 + x[0] ** 2)) + 8 * q[0] * np.cos(1.2 * time))

    def dyn_fcn_dx(self, x, q, time):
        return np.asarray([[0.5 + 25 * (1 - x[0] ** 2) / (1 + x[0] ** 2) ** 2, 8 * np.cos(1.2 * time)]])   # ssmod.py:305-306


class Pendulum2DTransition(TransitionModel):
    """ssmod.py:309-365."""
    dim_state, dim_noise, noise_additive, _fid = 2, 2, True, _lib.F_PENDULUM_DYN
    g = 9.81

    def __init__(self, init_rv=None, noise_rv=None, dt=0.01):
        super().__init__(init_rv, noise_rv)
        self.dt = dt

    def _par(self):
        return (self.dt,)

    def dyn_fcn(self, x, q, time):
        return np.array([x[0] + x[1] * self.dt, x[1] - self.g * self.dt * np.sin(x[0])]) + q

    def dyn_fcn_dx(self, x, r, time):
        return np.array([[1.0, self.dt], [-self.g * self.dt * np.cos(x[0]), 1.0]])           # ssmod.py:363-365


class ReentryVehicle1DTransition(TransitionModel):
    """ssmod.py:368-435."""
    dim_state, dim_noise, noise_additive, _fid = 3, 3, True, _lib.F_REENTRY1D_DYN

    def __init__(self, init_rv=None, noise_rv=None, dt=0.1):
        super().__init__(init_rv, noise_rv)
        self.dt = dt
        self.Gamma = 1 / 6.096

    def _par(self):
        return (self.dt,)

    def dyn_fcn(self, x, q, time):
        return np.array([x[0] - self.dt * x[1] + q[0],
                         x[1] - self.dt * np.exp(-self.Gamma * x[0]) * x[1] ** 2 * x[2] + q[1], x[2] + q[2]])

    def dyn_fcn_cont(self, x, q, time):
        """ssmod.py:429-432."""
        return np.array([-x[1] + q[0], -np.exp(-self.Gamma * x[0]) * x[1] ** 2 * x[2] + q[1], q[2]])


class ReentryVehicle2DTransition(TransitionModel):
    """ssmod.py:438-584 (5-D state [x, y, vx, vy, omega]; noise enters the last three states)."""
    dim_state, dim_noise, noise_additive, _fid = 5, 3, True, _lib.F_REENTRY2D_DYN

    def __init__(self, init_rv=None, noise_rv=None, dt=0.1):
        self.dt = dt
        self.R0, self.H0, self.Gm0, self.b0 = 6374, 13.406, 3.9860e5, -0.59783
        super().__init__(init_rv, noise_rv, np.vstack((np.zeros((2, 3)), np.eye(3))))

    def _par(self):
        return (self.dt,)

    def _core(self, x):
        b = self.b0 * np.exp(x[4])
        R = np.sqrt(x[0] ** 2 + x[1] ** 2)
        V = np.sqrt(x[2] ** 2 + x[3] ** 2)
        D = b * np.exp((self.R0 - R) / self.H0) * V
        G = -self.Gm0 / R ** 3
        return [x[0] + self.dt * x[2], x[1] + self.dt * x[3], x[2] + self.dt * (D * x[2] + G * x[0]),
                x[3] + self.dt * (D * x[3] + G * x[1]), x[4]]

    def dyn_fcn(self, x, q, time):
        return np.array(self._core(x)) + self.noise_gain.dot(q)

    def dyn_fcn_cont(self, x, q, time):
        """ssmod.py:569-585."""
        b = self.b0 * np.exp(x[4])
        R = np.sqrt(x[0] ** 2 + x[1] ** 2)
        V = np.sqrt(x[2] ** 2 + x[3] ** 2)
        D = b * np.exp((self.R0 - R) / self.H0) * V
        G = -self.Gm0 / R ** 3
        return np.array([x[2], x[3], D * x[2] + G * x[0] + q[0], D * x[3] + G * x[1] + q[1], q[2]])


class ReentryVehicle2DBiasTransition(ReentryVehicle2DTransition):
    """This build's synthetic 6-D benchmark model: reentry-2D on states 0..4 plus a
    pass-through sixth state.  Not part of the reference."""
    dim_state, dim_noise, noise_additive, _fid = 6, 4, True, _lib.F_REENTRY2D_BIAS_DYN

    def __init__(self, init_rv=None, noise_rv=None, dt=0.1):
        self.dt = dt
        self.R0, self.H0, self.Gm0, self.b0 = 6374, 13.406, 3.9860e5, -0.59783
        TransitionModel.__init__(self, init_rv, noise_rv, np.vstack((np.zeros((2, 4)), np.eye(4))))

    def dyn_fcn(self, x, q, time):
        return np.array(self._core(x) + [x[5]]) + self.noise_gain.dot(q)


class Smooth10DTransition(TransitionModel):
    """This build's synthetic 10-D model (not in the reference, whose models stop at 7 inputs): a smooth map for the
    Bayes-Sard quadrature configuration at D = 10 that the device can evaluate itself,
    out[i] = sin(x[i]) + x[5+i]^2, out[5+i] = x[5+i] cos(x[i]), i = 0..4."""
    dim_state, dim_noise, noise_additive, _fid = 10, 10, True, _lib.F_SMOOTH10D_DYN

    def dyn_fcn(self, x, q, time):
        return np.concatenate((np.sin(x[:5]) + x[5:10] ** 2, x[5:10] * np.cos(x[:5]))) + self.noise_gain.dot(q)


class CoordinatedTurnTransition(TransitionModel):
    """ssmod.py:587-696."""
    dim_state, dim_noise, noise_additive, _fid = 5, 5, True, _lib.F_CT_DYN

    def __init__(self, init_rv=None, noise_rv=None, dt=0.1):
        super().__init__(init_rv, noise_rv)
        self.dt = dt

    def _par(self):
        return (self.dt,)

    def dyn_fcn(self, x, q, *args):
        om = x[4]
        a, b = np.sin(om * self.dt), np.cos(om * self.dt)
        c, d = np.sin(om * self.dt) / om, (1 - np.cos(om * self.dt)) / om
        return np.array([x[0] + c * x[1] - d * x[3], b * x[1] - a * x[3], d * x[1] + x[2] + c * x[3],
                         a * x[1] + b * x[3], x[4]]) + q


class ConstantTurnRateSpeed(TransitionModel):
    """ssmod.py:699-780 (non-additive noise: input [x(5), q(2)])."""
    dim_state, dim_noise, noise_additive, _fid = 5, 2, False, _lib.F_CTRS_DYN

    def __init__(self, init_rv=None, noise_rv=None, dt=0.05):
        super().__init__(init_rv, noise_rv)
        self.dt = dt

    def _par(self):
        return (self.dt,)

    def dyn_fcn(self, x, q, time):
        dt = self.dt
        if x[4] == 0:
            f = np.array([dt * x[2] * np.cos(x[3]), dt * x[2] * np.sin(x[3]), dt * q[0],
                          dt * x[3] + 0.5 * dt ** 2 * q[1], dt * q[1]])
        else:
            c = x[2] / x[4]
            f = np.array([c * (np.sin(x[3] + x[4] * dt) - np.sin(x[3])) + 0.5 * dt ** 2 * np.cos(x[3]) * q[0],
                          c * (-np.cos(x[3] + x[4] * dt) + np.cos(x[3])) + 0.5 * dt ** 2 * np.sin(x[3]) * q[0],
                          dt * q[0], dt * x[3] + 0.5 * dt ** 2 * q[1], dt * q[1]])
        return x + f

    def dyn_fcn_cont(self, x, q, time):
        """ssmod.py:779-780."""
        return np.array([x[2] * np.cos(x[3]), x[2] * np.sin(x[3]), 0, x[4], 0])


class ConstantVelocity(TransitionModel):
    """ssmod.py:783-855."""
    dim_state, dim_noise, noise_additive, _fid = 4, 2, True, _lib.F_CV_DYN

    def __init__(self, init_rv=None, noise_rv=None, dt=0.1):
        self.dt = dt
        gain = np.array([[dt ** 2 / 2, 0], [dt, 0], [0, dt ** 2 / 2], [0, dt]])
        super().__init__(init_rv, noise_rv, gain)

    def _par(self):
        return (self.dt,)

    def dyn_fcn(self, x, q, time):
        return np.array([x[0] + self.dt * x[1], x[1], x[2] + self.dt * x[3], x[3]]) + self.noise_gain.dot(q)

    def dyn_fcn_dx(self, x, q, time):
        # ssmod.py:848-852 returns the TRANSPOSE of the transition matrix; kept as written
        return np.array([[1, self.dt, 0, 0], [0, 1, 0, 0], [0, 0, 1, self.dt], [0, 0, 0, 1]]).T


class MeasurementModel:
    """y_k = h(x_k, r_k, k)   (ssmod.py:863-1039)."""
    dim_in = None
    dim_out = None
    dim_noise = None
    dim_substate = None
    noise_additive = True
    _fid = None

    def __init__(self, noise_rv, dim_state, state_index=None):
        self.noise_rv = noise_rv
        self.zero_r = np.zeros(self.dim_noise)
        self.state_index = state_index
        self.dim_in = dim_state if self.noise_additive else dim_state + self.dim_noise
        self.dim_state = dim_state

    def _par(self):
        return ()

    def device_integrand(self):
        idx = None
        if self.state_index is not None:
            idx = list(self.state_index)
            if not self.noise_additive:   # the noise components follow the selected sub-state
                idx += list(range(self.dim_state, self.dim_state + self.dim_noise))
        return Integrand.make(self._fid, self._par(), idx), self.dim_out

    def meas_fcn(self, x, r, time):
        raise NotImplementedError

    def simulate_measurements(self, x, seed=0, traj_offset=0):
        """(dim_out, steps, mc_sims) measurements of the given states x (dim_state, steps, mc_sims), y[k] taken at time
        k + 1 (ssmod.py:1011-1039), generated on the device."""
        lib = _lib.load()
        x = np.asarray(x, dtype=np.float64)
        D, steps, mc_sims = x.shape
        Y = self.dim_out
        ld = (mc_sims + 63) // 64 * 64
        r, keep = _rv_desc(self.noise_rv, 'the measurement noise')
        f_obs, _ = self.device_integrand()
        xb = np.zeros((steps, D, ld))
        xb[:, :, :mc_sims] = x.transpose(1, 0, 2)
        d_x, d_y = _lib.DeviceBuffer(xb.nbytes), _lib.DeviceBuffer(8 * steps * Y * ld)
        d_x.upload(xb)
        _lib.check(lib.ssmq_simulate_rv_dev(None, ctypes.byref(f_obs), D, Y, None, None, ctypes.byref(r), None, 1,
                                            1 if self.noise_additive else 0, mc_sims, ld, steps, 0, 0.0, seed, traj_offset,
                                            ctypes.c_void_p(d_x.ptr), ctypes.c_void_p(d_y.ptr)), 'ssmq_simulate_rv_dev')
        y = d_y.download((steps, Y, ld))[:, :, :mc_sims].transpose(1, 0, 2)
        d_x.free()
        d_y.free()
        return np.ascontiguousarray(y)

    def meas_eval(self, xr, time, dx=False):
        """ssmod.py:960-1009."""
        if dx:
            # ssmod.py:985-1009, as written there: with state_index = None the assignment `out[:, None] = jac` broadcasts a
            # one-column Jacobian into every state column (the device kernel does the same: csrc/ssmq_linear.hip)
            if self.state_index is not None:
                xr = xr[self.state_index]
            if self.noise_additive:
                out = np.zeros((self.dim_out, self.dim_state))
                out[:, self.state_index] = self.meas_fcn_dx(xr, self.zero_r, time)
                return out
            x, r = xr[:self.dim_substate], xr[-self.dim_noise:]
            out = np.zeros((self.dim_out, self.dim_state + self.dim_noise))
            jac = self.meas_fcn_dx(x, r, time)
            out[:, self.state_index] = jac[:, :self.dim_substate]
            out[:, self.dim_state:] = jac[:, self.dim_substate:]
            return out
        if self.noise_additive:
            if self.state_index is not None:
                xr = xr[self.state_index]
            return self.meas_fcn(xr, self.zero_r, time)
        x, r = xr[:self.dim_state], xr[-self.dim_noise:]
        if self.state_index is not None:
            x = x[self.state_index]
        return self.meas_fcn(x, r, time)

    def meas_fcn_dx(self, x, r, time):
        """Jacobian of the measurement function (ssmod.py:937-958): UNGM, UNGM-NA and pendulum only."""
        return None


class UNGMMeasurement(MeasurementModel):
    """ssmod.py:1042-1064."""
    dim_out, dim_substate, dim_noise, noise_additive, _fid = 1, 1, 1, True, _lib.F_UNGM_MEAS

    def meas_fcn(self, x, r, time):
        return np.asarray([0.05 * x[0] ** 2]) + r

    def meas_fcn_dx(self, x, r, time):
        return np.asarray([0.1 * x[0]])                                                    # ssmod.py:1063-1064


class UNGMNAMeasurement(MeasurementModel):
    """ssmod.py:1067-1089 (non-additive noise: input [x, r])."""
    dim_out, dim_substate, dim_noise, noise_additive, _fid = 1, 1, 1, False, _lib.F_UNGMNA_MEAS

    def meas_fcn(self, x, r, time):
        return np.asarray([0.05 * r[0] * x[0] ** 2])

    def meas_fcn_dx(self, x, r, time):
        return np.asarray([[0.1 * r[0] * x[0], 0.05 * x[0] ** 2]])                          # ssmod.py:1088-1089


class Pendulum2DMeasurement(MeasurementModel):
    """ssmod.py:1092-1118."""
    dim_out, dim_substate, dim_noise, noise_additive, _fid = 1, 1, 1, True, _lib.F_PENDULUM_MEAS

    def meas_fcn(self, x, r, time):
        return np.array([np.sin(x[0])]) + r

    def meas_fcn_dx(self, x, r, time):
        return np.array([[np.cos(x[0])]])                                                  # ssmod.py:1117-1118


class RangeMeasurement(MeasurementModel):
    """ssmod.py:1121-1152."""
    dim_out, dim_substate, dim_noise, noise_additive, _fid = 1, 1, 1, True, _lib.F_RANGE_MEAS

    def meas_fcn(self, x, r, time):
        return np.array([np.sqrt(30.0 ** 2 + (x[0] - 30.0) ** 2)]) + r


class BearingMeasurement(MeasurementModel):
    """ssmod.py:1155-1198 (one bearing per sensor; default 4 sensors)."""
    dim_substate, noise_additive, _fid = 2, True, _lib.F_BEARING_MEAS

    def __init__(self, noise_rv, dim_state, state_index=None, sensor_pos=None):
        self.sensor_pos = np.vstack((np.eye(2), -np.eye(2))) if sensor_pos is None else np.asarray(sensor_pos, float)
        self.dim_out = len(self.sensor_pos)
        self.dim_noise = self.dim_out
        super().__init__(noise_rv, dim_state, state_index)

    def _par(self):
        return tuple(self.sensor_pos.reshape(-1))

    def meas_fcn(self, x, r, time):
        return np.arctan2(x[1] - self.sensor_pos[:, 1], x[0] - self.sensor_pos[:, 0]) + r


class Radar2DMeasurement(MeasurementModel):
    """ssmod.py:1201-1255 (range and bearing from the radar location)."""
    dim_out, dim_substate, dim_noise, noise_additive, _fid = 2, 2, 2, True, _lib.F_RADAR2D_MEAS

    def __init__(self, noise_rv, dim_state, state_index=None, radar_loc=None):
        super().__init__(noise_rv, dim_state, state_index)
        self.radar_loc = np.array([0.0, 0.0]) if radar_loc is None else np.asarray(radar_loc, dtype=float)

    def _par(self):
        return (self.radar_loc[0], self.radar_loc[1])

    def meas_fcn(self, x, r, time):
        dx, dy = x[0] - self.radar_loc[0], x[1] - self.radar_loc[1]
        return np.array([np.sqrt(dx ** 2 + dy ** 2), np.arctan2(dy, dx)]) + r
