"""
Callers of the moment-transform path kept on the device: Gaussian filters (reference:
ssmtoybox/ssinf.py:215-323, 347-552) running B independent trajectories per kernel launch.

`forward_pass(data)` keeps the reference's one-trajectory signature; `forward_pass_batch(data)` takes
data of shape (dim_y, T, B) - the reference's Monte-Carlo axis (`for imc in range(mc)` in its research scripts,
e.g. research/tpq/tpq_base.py:175-192) - and returns (D, T, B) / (D, D, T, B).

Per time step k = 1..T (both transforms use time index k - 1: ssinf.py:104, 276-288):
    dyn transform, + G Q G'   ->   obs transform, + R   ->   Kalman update (ssinf.py:297-323)
all inside `ssmq_filter_forward_dev` (ssmtoybox_amd/csrc); nothing is computed in NumPy.
`StudentianInference` (ssinf.py:555-736) runs the same loop with the reference's scale-matrix bookkeeping
(`ssmq_student_filter_forward_dev`); `backward_pass*` is the RTS smoother (`ssmq_filter_smooth_dev`).  Models that take
their noise as an argument (UNGMNA, CTRS; ssinf.py:271-295) go through `ssmq_filter_forward_aug_dev` (forward pass only).
`MarginalInference` (ssinf.py:1034-1292) evaluates its theta-conditioned steps on the device (`ssmq_gp_theta_step`); the
BFGS optimiser stays on the host as in the reference.
"""
import ctypes

import numpy as np

from . import _lib
from .mtran import (MomentTransform, LinearizationTransform, UnscentedTransform, SphericalRadialTransform, GaussHermiteTransform,
                    FullySymmetricStudentTransform, resolve_integrand)
from .bq.bqmtran import GaussianProcessTransform, BayesSardTransform, StudentTProcessTransform
from .ssmod import TransitionModel, MeasurementModel


class GaussianInference:
    """Gaussian filter (ssinf.py:215-323)."""

    def __init__(self, mod_dyn, mod_obs, tf_dyn, tf_obs):
        assert isinstance(mod_dyn, TransitionModel) and isinstance(mod_obs, MeasurementModel)
        assert isinstance(tf_dyn, MomentTransform) and isinstance(tf_obs, MomentTransform)
        self.mod_dyn, self.mod_obs, self.tf_dyn, self.tf_obs = mod_dyn, mod_obs, tf_dyn, tf_obs
        self.x0_mean, self.x0_cov = mod_dyn.init_rv.get_stats()
        self.q_mean, self.q_cov = mod_dyn.noise_rv.get_stats()
        self.r_mean, self.r_cov = mod_obs.noise_rv.get_stats()
        self.G = mod_dyn.noise_gain
        self.fi_mean = self.fi_cov = None
        self.sm_mean = self.sm_cov = None
        self.status = None
        self._data = None

    def reset(self):
        self.fi_mean = self.fi_cov = None
        self.sm_mean = self.sm_cov = None
        self.status = None
        self._data = None

    @property
    def _additive(self):
        return self.mod_dyn.noise_additive and self.mod_obs.noise_additive

    def kernel_name(self, batch=0):
        """Which kernel(s) the device filter loop runs for this filter (one fused kernel, or a replayed hipGraph) on a batch of
        `batch` trajectories (0: a batch that fills the device; small batches of some systems run k_filter_wsplit)."""
        if not self._additive:
            return ('k_filter_fused_aug (one kernel for the time loop) where instantiated, else a launch loop of 5 T '
                    'launches (k_augment | apply dyn | k_augment | apply obs | k_kalman_update)')
        f_dyn, e_dyn = resolve_integrand(self.mod_dyn.dyn_eval)
        f_obs, e_obs = resolve_integrand(self.mod_obs.meas_eval)
        buf = ctypes.create_string_buffer(512)
        _lib.check(_lib.load().ssmq_filter_kernel_name_batch(ctypes.c_void_p(self.tf_dyn._handle_for(e_dyn)),
                                                             ctypes.byref(f_dyn),
                                                             ctypes.c_void_p(self.tf_obs._handle_for(e_obs)),
                                                             ctypes.byref(f_obs), int(batch), buf, 512), 'ssmq_filter_kernel_name_batch')
        return buf.value.decode()

    def forward_pass(self, data):
        """data (dim_y, T) -> filtered means (D, T), covariances (D, D, T)  (ssinf.py:66-118)."""
        fm, fP = self.forward_pass_batch(np.asarray(data)[..., None])
        return fm[..., 0], fP[..., 0]

    def _launch(self, lib, h_dyn, f_dyn, h_obs, f_obs, B, ld, T, d_y, d_m0, d_P0, d_fm, d_fP, d_st):
        if not self._additive:
            return self._launch_aug(lib, h_dyn, f_dyn, h_obs, f_obs, B, ld, T, d_y, d_m0, d_P0, d_fm, d_fP, d_st)
        gqg, pg = _lib.as_c(self.G.dot(self.q_cov).dot(self.G.T))
        rr, pr = _lib.as_c(self.r_cov)
        _lib.check(lib.ssmq_filter_forward_dev(ctypes.c_void_p(h_dyn), ctypes.byref(f_dyn), ctypes.c_void_p(h_obs),
                                               ctypes.byref(f_obs), B, ld, T, ctypes.c_void_p(d_y.ptr),
                                               ctypes.c_void_p(d_m0.ptr), ctypes.c_void_p(d_P0.ptr), pg, pr,
                                               ctypes.c_void_p(d_fm.ptr), ctypes.c_void_p(d_fP.ptr),
                                               ctypes.c_void_p(d_st.ptr)), 'ssmq_filter_forward_dev')

    def _launch_aug(self, lib, h_dyn, f_dyn, h_obs, f_obs, B, ld, T, d_y, d_m0, d_P0, d_fm, d_fP, d_st, d_sm=None,
                    d_sP=None):
        """Noise enters the model functions: augmented moments per transform (ssinf.py:271-272, 282-283, 294-295).
        With d_sm / d_sP the RTS smoother runs as well (`ssmq_filter_smooth_aug_dev`)."""
        if self.mod_dyn.noise_additive:
            dq, (qm, pqm) = 0, (None, None)
            qc, pqc = _lib.as_c(self.G.dot(self.q_cov).dot(self.G.T))
        else:
            dq = int(np.atleast_1d(self.q_mean).shape[0])
            qm, pqm = _lib.as_c(np.atleast_1d(self.q_mean))
            qc, pqc = _lib.as_c(np.atleast_2d(self.q_cov))
        if self.mod_obs.noise_additive:
            dr, (rm, prm) = 0, (None, None)
            rc, prc = _lib.as_c(self.r_cov)
        else:
            dr = int(np.atleast_1d(self.r_mean).shape[0])
            rm, prm = _lib.as_c(np.atleast_1d(self.r_mean))
            rc, prc = _lib.as_c(np.atleast_2d(self.r_cov))
        if d_sm is not None:
            _lib.check(lib.ssmq_filter_smooth_aug_dev(ctypes.c_void_p(h_dyn), ctypes.byref(f_dyn), ctypes.c_void_p(h_obs),
                                                      ctypes.byref(f_obs), self.mod_dyn.dim_state, B, ld, T,
                                                      ctypes.c_void_p(d_y.ptr), ctypes.c_void_p(d_m0.ptr),
                                                      ctypes.c_void_p(d_P0.ptr), pqm, pqc, dq, prm, prc, dr,
                                                      ctypes.c_void_p(d_fm.ptr), ctypes.c_void_p(d_fP.ptr),
                                                      ctypes.c_void_p(d_sm.ptr), ctypes.c_void_p(d_sP.ptr),
                                                      ctypes.c_void_p(d_st.ptr)), 'ssmq_filter_smooth_aug_dev')
            return
        _lib.check(lib.ssmq_filter_forward_aug_dev(ctypes.c_void_p(h_dyn), ctypes.byref(f_dyn), ctypes.c_void_p(h_obs),
                                                   ctypes.byref(f_obs), self.mod_dyn.dim_state, B, ld, T,
                                                   ctypes.c_void_p(d_y.ptr), ctypes.c_void_p(d_m0.ptr),
                                                   ctypes.c_void_p(d_P0.ptr), pqm, pqc, dq, prm, prc, dr,
                                                   ctypes.c_void_p(d_fm.ptr), ctypes.c_void_p(d_fP.ptr),
                                                   ctypes.c_void_p(d_st.ptr)), 'ssmq_filter_forward_aug_dev')

    def _initial_cov(self):
        return self.x0_cov

    def backward_pass(self):
        """RTS smoothing of the trajectory given to the last forward_pass (ssinf.py:120-147)."""
        sm, sP = self.backward_pass_batch()
        return sm[..., 0], sP[..., 0]

    def backward_pass_batch(self):
        """RTS smoothing of the batch given to the last forward_pass_batch: (D, T, B), (D, D, T, B).  The reference's
        indexing is kept (the last two smoothed steps equal the filtered ones)."""
        assert self._data is not None, 'run forward_pass first'     # the reference asserts its 'filtered' flag
        self.forward_pass_batch(self._data, smooth=True)
        return self.sm_mean, self.sm_cov

    def forward_pass_dev(self, d_y, B, ld, T):
        """Filter measurements that are already on the device (planes [T][dim_y][ld], e.g. from `ssmod.simulate_dev`)
        and leave the results there: returns DeviceBuffers (d_fm [T][D][ld], d_fP [T][D*D][ld], d_status [ld]) for
        `mcshard.device_error_sums`; the caller frees them.  Every trajectory starts from the model's initial moments."""
        lib = _lib.load()
        D = self.mod_dyn.dim_state
        mrow = np.ascontiguousarray(np.repeat(np.asarray(self.x0_mean, dtype=np.float64).reshape(D, 1), 64, axis=1))
        Prow = np.ascontiguousarray(np.repeat(np.asarray(self._initial_cov(), dtype=np.float64).reshape(D * D, 1), 64,
                                              axis=1))
        d_m0, d_P0 = _lib.DeviceBuffer(8 * D * ld), _lib.DeviceBuffer(8 * D * D * ld)
        for host, dev, n in ((mrow, d_m0, D), (Prow, d_P0, D * D)):       # one 64-lane tile per plane, doubled on the device
            for e in range(n):
                dev.upload(host[e], byte_offset=8 * e * ld)
                done = 64
                while done < ld:
                    step = min(done, ld - done)
                    _lib.check(lib.ssmq_memcpy_d2d(dev.at(8 * (e * ld + done)), dev.at(8 * e * ld),
                                                   ctypes.c_size_t(8 * step)), 'ssmq_memcpy_d2d')
                    done += step
        d_fm, d_fP = _lib.DeviceBuffer(8 * T * D * ld), _lib.DeviceBuffer(8 * T * D * D * ld)
        d_st = _lib.DeviceBuffer(4 * ld)
        f_dyn, e_dyn = resolve_integrand(self.mod_dyn.dyn_eval)
        f_obs, e_obs = resolve_integrand(self.mod_obs.meas_eval)
        h_dyn, h_obs = self.tf_dyn._handle_for(e_dyn), self.tf_obs._handle_for(e_obs)
        self._launch(lib, h_dyn, f_dyn, h_obs, f_obs, B, ld, T, d_y, d_m0, d_P0, d_fm, d_fP, d_st)
        _lib.sync()
        d_m0.free()
        d_P0.free()
        return d_fm, d_fP, d_st

    def _forward_pass_piped(self, lib, data, x0_mean, x0_cov, raise_on_failure):
        """The forward pass with its transfers overlapped (`ssmq_filter_forward_piped`: time blocks; the measurements of the next
        block go up and the moments of the previous one come down while a block runs).  None when the filter has no time-block
        kernel - the caller then takes the upload / pass / download path.  Results: the same bits either way."""
        if type(self)._launch is not GaussianInference._launch:        # Studentian recursion: not pipelined
            return None
        Y, T, B = data.shape
        D = self.mod_dyn.dim_state
        f_dyn, e_dyn = resolve_integrand(self.mod_dyn.dyn_eval)
        f_obs, e_obs = resolve_integrand(self.mod_obs.meas_eval)
        h_dyn, h_obs = self.tf_dyn._handle_for(e_dyn), self.tf_obs._handle_for(e_obs)
        flags = 0
        if x0_mean is None and x0_cov is None:
            m0, P0 = np.ascontiguousarray(self.x0_mean, dtype=np.float64), np.ascontiguousarray(self._initial_cov(), dtype=np.float64)
        else:
            m0 = np.ascontiguousarray(np.broadcast_to(self.x0_mean, (B, D)) if x0_mean is None else x0_mean, dtype=np.float64)
            P0 = np.ascontiguousarray(np.broadcast_to(self._initial_cov(), (B, D, D)) if x0_cov is None else x0_cov, dtype=np.float64)
            flags |= 2           # SSMQ_PIPED_X0_PER_TRAJECTORY
        fm, fP = _lib.pinned_empty((D, T, B)), _lib.pinned_empty((D, D, T, B))
        if fm is None or fP is None:
            fm, fP = np.empty((D, T, B)), np.empty((D, D, T, B))
        else:
            flags |= 1           # SSMQ_PIPED_OUT_PINNED
        st = np.empty(B, dtype=np.int32)
        gqg, pg = _lib.as_c(self.G.dot(self.q_cov).dot(self.G.T))
        rr, pr = _lib.as_c(self.r_cov)
        yc = np.ascontiguousarray(data)
        dp = lambda a: a.ctypes.data_as(_lib.c_double_p)       # noqa: E731
        rc = lib.ssmq_filter_forward_piped(ctypes.c_void_p(h_dyn), ctypes.byref(f_dyn), ctypes.c_void_p(h_obs), ctypes.byref(f_obs), B, T,
                                           dp(yc), dp(m0), dp(P0), pg, pr, ctypes.c_void_p(fm.ctypes.data),
                                           ctypes.c_void_p(fP.ctypes.data), ctypes.c_void_p(st.ctypes.data), flags, 0)
        if rc == -3:             # SSMQ_E_UNSUPPORTED
            return None
        _lib.check(rc, 'ssmq_filter_forward_piped')
        self.status = st
        if raise_on_failure and st.any():
            b = int(np.flatnonzero(st)[0])
            raise np.linalg.LinAlgError('Matrix is not positive definite (trajectory {}, step {})'.format(b, int(st[b]) - 1))
        self.fi_mean, self.fi_cov = fm, fP
        return fm, fP

    def forward_pass_batch(self, data, x0_mean=None, x0_cov=None, raise_on_failure=True, smooth=False):
        """data (dim_y, T, B).  Optional per-trajectory initial moments x0_mean (B, D), x0_cov (B, D, D)."""
        lib = _lib.load()
        data = np.asarray(data, dtype=np.float64)
        self._data = data
        Y, T, B = data.shape
        D = self.mod_dyn.dim_state
        ld = (B + 63) // 64 * 64
        if not smooth and self._additive and B > 0 and T > 0:
            done = self._forward_pass_piped(lib, data, x0_mean, x0_cov, raise_on_failure)
            if done is not None:
                return done
        # (dim_y, T, B) -> planes [T][Y][ld]: a row permutation, done between the caller's array and the library's pinned
        # staging block (`ssmq_upload_planes`)
        d_y = _lib.scratch(8 * T * Y * ld)
        _lib.upload_study(data, Y, ld, d_y)
        m0 = np.broadcast_to(self.x0_mean, (B, D)) if x0_mean is None else np.asarray(x0_mean, dtype=np.float64)
        P0 = np.broadcast_to(self._initial_cov(), (B, D, D)) if x0_cov is None else np.asarray(x0_cov, dtype=np.float64)
        mbuf = np.zeros((D, ld))
        mbuf[:, :B] = m0.T
        Pbuf = np.zeros((D * D, ld))
        Pbuf[:, :B] = P0.reshape(B, D * D).T
        if ld > B:       # padding lanes are never read (b >= B), keep them PD anyway
            Pbuf[:, B:] = np.eye(D).reshape(-1, 1)
        d_m0, d_P0 = _lib.scratch(mbuf.nbytes), _lib.scratch(Pbuf.nbytes)
        d_m0.upload(mbuf)
        d_P0.upload(Pbuf)
        d_fm, d_fP = _lib.scratch(8 * T * D * ld), _lib.scratch(8 * T * D * D * ld)
        d_st = _lib.scratch(4 * ld)
        f_dyn, e_dyn = resolve_integrand(self.mod_dyn.dyn_eval)
        f_obs, e_obs = resolve_integrand(self.mod_obs.meas_eval)
        h_dyn, h_obs = self.tf_dyn._handle_for(e_dyn), self.tf_obs._handle_for(e_obs)
        if smooth:
            d_sm, d_sP = _lib.scratch(8 * T * D * ld), _lib.scratch(8 * T * D * D * ld)
            if not self._additive:
                self._launch_aug(lib, h_dyn, f_dyn, h_obs, f_obs, B, ld, T, d_y, d_m0, d_P0, d_fm, d_fP, d_st, d_sm, d_sP)
            else:
                gqg, pg = _lib.as_c(self.G.dot(self.q_cov).dot(self.G.T))
                rr, pr = _lib.as_c(self.r_cov)
                _lib.check(lib.ssmq_filter_smooth_dev(ctypes.c_void_p(h_dyn), ctypes.byref(f_dyn), ctypes.c_void_p(h_obs),
                                                      ctypes.byref(f_obs), B, ld, T, ctypes.c_void_p(d_y.ptr),
                                                      ctypes.c_void_p(d_m0.ptr), ctypes.c_void_p(d_P0.ptr), pg, pr,
                                                      ctypes.c_void_p(d_fm.ptr), ctypes.c_void_p(d_fP.ptr),
                                                      ctypes.c_void_p(d_sm.ptr), ctypes.c_void_p(d_sP.ptr),
                                                      ctypes.c_void_p(d_st.ptr)), 'ssmq_filter_smooth_dev')
            self.sm_mean = _lib.download_study(d_sm, (D,), T, B, ld)
            self.sm_cov = _lib.download_study(d_sP, (D, D), T, B, ld)
            d_sm.free()
            d_sP.free()
        else:
            self._launch(lib, h_dyn, f_dyn, h_obs, f_obs, B, ld, T, d_y, d_m0, d_P0, d_fm, d_fP, d_st)
        fm = _lib.download_study(d_fm, (D,), T, B, ld)       # planes [T][D][ld] -> (D, T, B), via pinned staging
        fP = _lib.download_study(d_fP, (D, D), T, B, ld)
        self.status = d_st.download((ld,), dtype=np.int32)[:B]
        for buf in (d_y, d_m0, d_P0, d_fm, d_fP, d_st):
            buf.free()
        if raise_on_failure and self.status.any():
            b = int(np.flatnonzero(self.status)[0])
            raise np.linalg.LinAlgError('Matrix is not positive definite (trajectory {}, step {})'.format(
                b, int(self.status[b]) - 1))
        self.fi_mean, self.fi_cov = fm, fP
        return self.fi_mean, self.fi_cov


def run_filters(algs, data, x0_mean=None, x0_cov=None, raise_on_failure=True):
    """Several filters over the SAME measurements as one device launch - the loop `for alg in algs: alg.forward_pass(y)` of the
    reference's studies (research/bsq/bsq_ungm.py:132-137, research/tpq/tpq_base.py:175-192), for data of shape (dim_y, T, B).
    Returns [(fi_mean (D, T, B), fi_cov (D, D, T, B)), ...] in the order of `algs` and leaves `fi_mean / fi_cov / status` on
    every filter, exactly what `alg.forward_pass_batch(data)` would have produced (the same kernels run, concurrently:
    `ssmq_filter_forward_multi_dev`).  The filters must be additive-noise Gaussian / Studentian filters; they may differ in
    models and state dimension, the measurements are uploaded once."""
    lib = _lib.load()
    data = np.asarray(data, dtype=np.float64)
    Y, T, B = data.shape
    ld = (B + 63) // 64 * 64
    for a in algs:
        if not isinstance(a, GaussianInference) or not a._additive:
            raise NotImplementedError('run_filters: additive-noise Gaussian / Studentian filters only')
        if a.mod_obs.dim_out != Y:
            raise ValueError('run_filters: every filter must take the same measurements')
    d_y = _lib.scratch(8 * T * Y * ld)
    _lib.upload_study(data, Y, ld, d_y)
    jobs = (_lib.FilterJob * len(algs))()
    keep, bufs = [], []
    for i, a in enumerate(algs):
        a._data = data
        D = a.mod_dyn.dim_state
        m0 = np.broadcast_to(a.x0_mean, (B, D)) if x0_mean is None else np.asarray(x0_mean, dtype=np.float64)
        P0 = np.broadcast_to(a._initial_cov(), (B, D, D)) if x0_cov is None else np.asarray(x0_cov, dtype=np.float64)
        mbuf = np.zeros((D, ld))
        mbuf[:, :B] = m0.T
        Pbuf = np.zeros((D * D, ld))
        Pbuf[:, :B] = P0.reshape(B, D * D).T
        if ld > B:
            Pbuf[:, B:] = np.eye(D).reshape(-1, 1)
        d_m0, d_P0 = _lib.scratch(mbuf.nbytes), _lib.scratch(Pbuf.nbytes)
        d_m0.upload(mbuf)
        d_P0.upload(Pbuf)
        d_fm, d_fP, d_st = _lib.scratch(8 * T * D * ld), _lib.scratch(8 * T * D * D * ld), _lib.scratch(4 * ld)
        bufs.append((d_m0, d_P0, d_fm, d_fP, d_st))
        f_dyn, e_dyn = resolve_integrand(a.mod_dyn.dyn_eval)
        f_obs, e_obs = resolve_integrand(a.mod_obs.meas_eval)
        student = isinstance(a, StudentianInference)
        gqg, pg = _lib.as_c(a.G.dot(a.q_smat if student else a.q_cov).dot(a.G.T))
        rr, pr = _lib.as_c(a.r_smat if student else a.r_cov)
        keep += [f_dyn, f_obs, gqg, rr]
        j = jobs[i]
        j.h_dyn, j.f_dyn = a.tf_dyn._handle_for(e_dyn), ctypes.pointer(f_dyn)
        j.h_obs, j.f_obs = a.tf_obs._handle_for(e_obs), ctypes.pointer(f_obs)
        j.B, j.ld, j.T = B, ld, T
        j.d_y, j.d_m0, j.d_P0 = d_y.ptr, d_m0.ptr, d_P0.ptr
        j.GQG, j.R = pg, pr
        j.d_fm, j.d_fP, j.d_status = d_fm.ptr, d_fP.ptr, d_st.ptr
        if student:
            sc, ps = _lib.as_c(a.scale_sequence(T))
            keep.append(sc)
            j.scale, j.dof = ps, float(a.dof)
    _lib.check(lib.ssmq_filter_forward_multi_dev(len(algs), jobs), 'ssmq_filter_forward_multi_dev')
    out, failed = [], None
    for a, (d_m0, d_P0, d_fm, d_fP, d_st) in zip(algs, bufs):
        D = a.mod_dyn.dim_state
        a.fi_mean = _lib.download_study(d_fm, (D,), T, B, ld)
        a.fi_cov = _lib.download_study(d_fP, (D, D), T, B, ld)
        a.status = d_st.download((ld,), dtype=np.int32)[:B]
        for buf in (d_m0, d_P0, d_fm, d_fP, d_st):
            buf.free()
        if a.status.any() and failed is None:
            b = int(np.flatnonzero(a.status)[0])
            failed = '{}: Matrix is not positive definite (trajectory {}, step {})'.format(type(a).__name__, b, int(a.status[b]) - 1)
        out.append((a.fi_mean, a.fi_cov))
    d_y.free()
    if raise_on_failure and failed:
        raise np.linalg.LinAlgError(failed)
    return out


class CubatureKalman(GaussianInference):
    """ssinf.py:360-371."""

    def __init__(self, dyn, obs):
        super().__init__(dyn, obs, SphericalRadialTransform(dyn.dim_in), SphericalRadialTransform(obs.dim_in))


class ExtendedKalman(GaussianInference):
    """Extended Kalman filter and smoother (ssinf.py:347-357): both transforms are linearisations around the mean.  Runs for
    the models whose Jacobians the reference implements (its own test skips the others: tests/test_ssinf.py:96-101)."""

    def __init__(self, dyn, obs):
        super().__init__(dyn, obs, LinearizationTransform(dyn.dim_in), LinearizationTransform(obs.dim_in))


class UnscentedKalman(GaussianInference):
    """ssinf.py:374-402."""

    def __init__(self, dyn, obs, kappa=None, alpha=1.0, beta=2.0):
        super().__init__(dyn, obs, UnscentedTransform(dyn.dim_in, kappa=kappa, alpha=alpha, beta=beta),
                         UnscentedTransform(obs.dim_in, kappa=kappa, alpha=alpha, beta=beta))


class GaussHermiteKalman(GaussianInference):
    """ssinf.py:405-420."""

    def __init__(self, dyn, obs, deg=3):
        super().__init__(dyn, obs, GaussHermiteTransform(dyn.dim_in, degree=deg),
                         GaussHermiteTransform(obs.dim_in, degree=deg))


class GaussianProcessKalman(GaussianInference):
    """ssinf.py:423-463."""

    def __init__(self, dyn, obs, kern_par_dyn, kern_par_obs, kernel='rbf', points='ut', point_hyp=None):
        t_dyn = GaussianProcessTransform(dyn.dim_in, dyn.dim_state, kern_par_dyn, kernel, points, point_hyp)
        t_obs = GaussianProcessTransform(obs.dim_in, obs.dim_out, kern_par_obs, kernel, points, point_hyp)
        super().__init__(dyn, obs, t_dyn, t_obs)


class BayesSardKalman(GaussianInference):
    """ssinf.py:466-502."""

    def __init__(self, dyn, obs, kern_par_dyn, kern_par_obs, mulind_dyn=2, mulind_obs=2, points='ut', point_hyp=None):
        t_dyn = BayesSardTransform(dyn.dim_in, dyn.dim_state, kern_par_dyn, mulind_dyn, points, point_hyp)
        t_obs = BayesSardTransform(obs.dim_in, obs.dim_out, kern_par_obs, mulind_obs, points, point_hyp)
        super().__init__(dyn, obs, t_dyn, t_obs)


class StudentProcessKalman(GaussianInference):
    """ssinf.py:505-552.  As in the reference the transforms are built with dim_out = 1 (I_out = eye(1)), so the whole
    scaled (E, E) model-variance matrix is added for E > 1."""

    def __init__(self, dyn, obs, kern_par_dyn, kern_par_obs, kernel='rbf', points='ut', point_hyp=None, nu=3.0):
        t_dyn = StudentTProcessTransform(dyn.dim_in, 1, kern_par_dyn, kernel, points, point_hyp, nu=nu)
        t_obs = StudentTProcessTransform(obs.dim_in, 1, kern_par_obs, kernel, points, point_hyp, nu=nu)
        super().__init__(dyn, obs, t_dyn, t_obs)


class StudentianInference(GaussianInference):
    """Additive-noise Studentian filter (ssinf.py:555-736): the state and measurement are jointly Student-t; the moment
    transforms are fed scale matrices and the measurement update rescales the filtered scale matrix by
    (dof + delta'delta) / (dof + dim_y).  `forward_pass*` return the filtered mean and the reference's `x_cov_fi`."""

    def __init__(self, mod_dyn, mod_obs, tf_dyn, tf_obs, dof=4.0, fixed_dof=True):
        assert isinstance(mod_dyn, TransitionModel) and isinstance(mod_obs, MeasurementModel)
        assert isinstance(tf_dyn, MomentTransform) and isinstance(tf_obs, MomentTransform)
        if not (mod_dyn.noise_additive and mod_obs.noise_additive):
            raise NotImplementedError('the device filter loop covers additive-noise models')
        if dof <= 2.0:
            dof = 4.0
        self.mod_dyn, self.mod_obs, self.tf_dyn, self.tf_obs = mod_dyn, mod_obs, tf_dyn, tf_obs
        # NB: as in the reference, get_stats() hands back SCALE matrices that are then treated as covariances
        self.x0_mean, self.x0_cov, self.x0_dof = mod_dyn.init_rv.get_stats()
        self.q_mean, self.q_cov, self.q_dof = mod_dyn.noise_rv.get_stats()
        self.r_mean, self.r_cov, self.r_dof = mod_obs.noise_rv.get_stats()
        self.G = mod_dyn.noise_gain
        scale = (dof - 2) / dof
        self.x_smat_0 = scale * self.x0_cov
        self.q_smat = scale * self.q_cov
        self.r_smat = scale * self.r_cov
        self.dof, self.fixed_dof = dof, fixed_dof
        self.fi_mean = self.fi_cov = None
        self.status = None

    def _initial_cov(self):
        return self.x_smat_0

    def backward_pass_batch(self):
        raise NotImplementedError('the reference has no Student smoother either (ssinf.py:738-740)')

    def scale_sequence(self, steps):
        """(dof_pr - 2) / dof_pr of every time update (ssinf.py:652-660; dof_fi grows by dim_out per update, :735)."""
        out = np.zeros(steps)
        dof_fi = self.x0_dof
        for k in range(steps):
            if self.fixed_dof:
                dof_pr = min(dof_fi, self.q_dof, self.r_dof)
                out[k] = (dof_pr - 2) / dof_pr
            else:
                out[k] = (self.dof - 2) / self.dof
            dof_fi += self.mod_obs.dim_out
        return out

    def _launch(self, lib, h_dyn, f_dyn, h_obs, f_obs, B, ld, T, d_y, d_m0, d_P0, d_fm, d_fP, d_st):
        gqg, pg = _lib.as_c(self.G.dot(self.q_smat).dot(self.G.T))
        rr, pr = _lib.as_c(self.r_smat)
        sc, ps = _lib.as_c(self.scale_sequence(T))
        _lib.check(lib.ssmq_student_filter_forward_dev(
            ctypes.c_void_p(h_dyn), ctypes.byref(f_dyn), ctypes.c_void_p(h_obs), ctypes.byref(f_obs), B, ld, T,
            ctypes.c_void_p(d_y.ptr), ctypes.c_void_p(d_m0.ptr), ctypes.c_void_p(d_P0.ptr), pg, pr, ps,
            ctypes.c_double(self.dof), ctypes.c_void_p(d_fm.ptr), ctypes.c_void_p(d_fP.ptr),
            ctypes.c_void_p(d_st.ptr)), 'ssmq_student_filter_forward_dev')


class FullySymmetricStudent(StudentianInference):
    """Student filter with the fully-symmetric rule (ssinf.py:743-790)."""

    def __init__(self, dyn, obs, degree=3, kappa=None, dof=4.0, fixed_dof=True):
        dyn_dof = min(dyn.init_rv.dof, dyn.noise_rv.dof)
        obs_dof = min(dyn_dof, obs.noise_rv.dof)
        t_dyn = FullySymmetricStudentTransform(dyn.dim_in, degree, kappa, dyn_dof)
        t_obs = FullySymmetricStudentTransform(obs.dim_in, degree, kappa, obs_dof)
        super().__init__(dyn, obs, t_dyn, t_obs, dof, fixed_dof)


class StudentProcessStudent(StudentianInference):
    """Student-t process quadrature Student filter (ssinf.py:793-857).  The reference builds its weights with the
    Monte-Carlo 'rbf-student' kernel (RNG-dependent, bq/bqkern.py:457-536), which is out of scope here: construct it with
    the plain RBF kernel and assign the Monte-Carlo weights (tf.wm / tf.Wc / tf.Wcc / tf.model.model_var / tf.model.iK)
    as data, exactly as research/tpq/tpq_ungm.py:109-124 does."""

    def __init__(self, dyn, obs, kern_par_dyn, kern_par_obs, point_par=None, dof=4.0, fixed_dof=True, dof_tp=4.0):
        assert kern_par_dyn.shape[1] == dyn.dim_in + 1 and kern_par_obs.shape[1] == obs.dim_in + 1
        point_par = {} if point_par is None else point_par
        pp_dyn, pp_obs = dict(point_par), dict(point_par)
        pp_dyn.update({'dof': dyn.noise_rv.dof})
        pp_obs.update({'dof': obs.noise_rv.dof})
        t_dyn = StudentTProcessTransform(dyn.dim_in, 1, kern_par_dyn, 'rbf', 'fs', pp_dyn, nu=dof_tp)
        t_obs = StudentTProcessTransform(obs.dim_in, 1, kern_par_obs, 'rbf', 'fs', pp_obs, nu=dof_tp)
        super().__init__(dyn, obs, t_dyn, t_obs, dof, fixed_dof)


class MarginalInference(GaussianInference):
    """Gaussian filter with the moment-transform (kernel) parameters marginalised (ssinf.py:1034-1273; the reference
    calls it experimental).  Per measurement: Laplace approximation of the log-parameter posterior (BFGS on
    -log N(y | m_y(theta), P_y(theta)) - log N(theta | prior)), then a spherical-radial rule over that posterior mixing
    the theta-conditioned state posteriors.  Every evaluation "weights(theta) -> two transforms -> update" runs on the
    device through `ssmq_gp_theta_step`, batched over theta: one call per finite-difference gradient
    (param_dim + 1 items) and one per marginalisation (2 param_dim items).  The optimiser itself is SciPy on the host, as
    in the reference.  GP-quadrature transforms; dynamics with additive noise or with the noise as an argument (the
    augmented moments of ssinf.py:1174-1176); additive measurement models - the reference builds the measurement transform
    on `dim_state` inputs (ssinf.py:1288) and cannot run a non-additive one either.

    Reference quirks kept: the measurement update evaluates both transforms at time index k, not k - 1
    (ssinf.py:112 passes k); the mixture covariance is the weighted sum of the conditional covariances, without the
    spread-of-the-means term (ssinf.py:1114-1115); the predictive moments the smoother uses come from the generic time
    update at index k - 1 (ssinf.py:104-107) with whatever weights the dynamics transform was LAST given - the dummy unit
    parameters at the first step, afterwards those of the last parameter point of the previous step's marginalisation
    (BQTransform.apply keeps the weights of its last `kern_par`, bq/bqmtran.py:93-95)."""

    def __init__(self, dyn, obs, tf_dyn, tf_obs, par_mean=None, par_cov=None):
        super().__init__(dyn, obs, tf_dyn, tf_obs)
        if not self.mod_obs.noise_additive:
            raise NotImplementedError('marginalised filter: the measurement transform is built on dim_state inputs '
                                      '(ssinf.py:1288); a measurement model that takes its noise as an argument does not '
                                      'run in the reference either')
        self.param_dyn_dim = self.mod_dyn.dim_in + 1
        self.param_obs_dim = self.mod_obs.dim_state + 1
        self.param_dim = self.param_dyn_dim + self.param_obs_dim
        self.param_prior_mean = np.zeros(self.param_dim) if par_mean is None else np.asarray(par_mean, dtype=float)
        self.param_prior_cov = np.eye(self.param_dim) if par_cov is None else np.asarray(par_cov, dtype=float)
        self.param_mean, self.param_cov = self.param_prior_mean, self.param_prior_cov
        self.param_jitter = 1e-8 * np.eye(self.param_dim)
        self.param_upts = SphericalRadialTransform.unit_sigma_points(self.param_dim)
        self.param_wts = SphericalRadialTransform.weights(self.param_dim)
        self.param_pts_num = self.param_upts.shape[1]
        self.x_mean_fi, self.x_cov_fi = self.x0_mean, self.x0_cov
        self.fd_step = 1.4901161193847656e-08      # SciPy's default forward-difference step for BFGS
        self._last_theta = None                    # parameters the reference's transforms would hold by now
        self.pr_mean = self.pr_cov = self.pr_xx_cov = None

    def reset(self):
        super().reset()
        self.param_mean, self.param_cov = self.param_prior_mean, self.param_prior_cov
        self.x_mean_fi, self.x_cov_fi = self.x0_mean, self.x0_cov
        self.pr_mean = self.pr_cov = self.pr_xx_cov = None
        # _last_theta stays: the reference's reset() does not touch the weights its transforms were last given either

    def _augment(self, mean, cov):
        """ssinf.py:271-272 / 1174-1176: [mean; q_mean], blockdiag(cov, Q) for dynamics that take their noise as an argument;
        mean (D,) / cov (D, D) or per-item (P, D) / (P, D, D)."""
        if self.mod_dyn.noise_additive:
            return mean, cov
        mean, cov = np.asarray(mean, dtype=np.float64), np.asarray(cov, dtype=np.float64)
        qm, qc = np.atleast_1d(self.q_mean), np.atleast_2d(self.q_cov)
        D, dq = self.mod_dyn.dim_state, qm.shape[0]
        m = np.concatenate((mean, np.broadcast_to(qm, mean.shape[:-1] + (dq,))), axis=-1)
        c = np.zeros(cov.shape[:-2] + (D + dq, D + dq))
        c[..., :D, :D] = cov
        c[..., D:, D:] = qc
        return m, c

    def _theta_static(self):
        """What rarely changes between calls of `theta_step` (the marginalised filter makes hundreds per time step and the
        ctypes conversions were a third of a call): integrand descriptors, transform handles, the noise terms, and a
        prototype of the entry point that takes the array addresses as plain integers.  Rebuilt when a model constant, a
        state index or a noise covariance has been changed since (compared by value: research code assigns them after
        construction)."""
        state_index = getattr(self.mod_obs, 'state_index', None)
        key = (self.mod_dyn._par(), self.mod_obs._par(), None if state_index is None else tuple(state_index))
        st = getattr(self, '_theta_cache', None)
        if st is not None and st['key'] == key and np.array_equal(st['q_cov'], self.q_cov) and \
                np.array_equal(st['rr'], self.r_cov) and np.array_equal(st['G'], self.G):
            # The handles follow a transform object that was re-assigned and points / I_out that were replaced (research code
            # assigns alg.tf_dyn / tf_obs and their attributes after construction - a cached pointer would go stale or dangle).
            # What the theta step reads from a handle is its shape, its unit points and the model-variance mode - the weights are
            # made on the device per parameter item - so that is what is compared here (two tiny tobytes()); the full comparison
            # of _handle_for (six arrays per handle, 5 us per call of a 65 us step) runs only when one of them differs.
            sig = self._theta_signature()
            if sig != st.get('sig'):
                st['h_dyn'], st['h_obs'] = self.tf_dyn._handle_for(st['e_dyn']), self.tf_obs._handle_for(st['e_obs'])
                st['sig'] = sig
            return st
        lib = _lib.load()
        f_dyn, e_dyn = resolve_integrand(self.mod_dyn.dyn_eval)
        f_obs, e_obs = resolve_integrand(self.mod_obs.meas_eval)
        vp = ctypes.c_void_p
        proto = ctypes.CFUNCTYPE(ctypes.c_int, vp, vp, vp, vp, ctypes.c_int64, vp, vp, ctypes.c_double, vp, vp, ctypes.c_int,
                                 vp, ctypes.c_int, ctypes.c_double, vp, vp, vp, vp, vp, vp)
        st = dict(key=key, fn=proto(('ssmq_gp_theta_step', lib)), f_dyn=f_dyn, f_obs=f_obs, e_dyn=e_dyn, e_obs=e_obs,
                  pf_dyn=ctypes.addressof(f_dyn), pf_obs=ctypes.addressof(f_obs),
                  h_dyn=self.tf_dyn._handle_for(e_dyn), h_obs=self.tf_obs._handle_for(e_obs),
                  q_cov=np.array(self.q_cov, dtype=np.float64), G=np.array(self.G, dtype=np.float64),
                  gqg=(np.ascontiguousarray(self.G.dot(self.q_cov).dot(self.G.T), dtype=np.float64)
                       if self.mod_dyn.noise_additive else None),
                  rr=np.array(self.r_cov, dtype=np.float64, order='C'))
        st['sig'] = self._theta_signature()
        self._theta_cache = st
        return st

    def _theta_signature(self):
        """Identity of the two transform objects plus the bytes of what `ssmq_gp_theta_step` takes from their handles."""
        td, to = self.tf_dyn, self.tf_obs
        return (id(td), id(to), td.model.points.tobytes(), to.model.points.tobytes(), td.I_out.shape, to.I_out.shape,
                td.model.points.shape, to.model.points.shape)

    def theta_step(self, theta, mean, cov, y, time):
        """theta (P, param_dim) log-parameters; mean (D,) / cov (D, D) shared by all items or (P, D) / (P, D, D);
        y (Y,) or (P, Y).  Returns conditional posterior means (P, D), covariances (P, D, D), log-likelihoods (P,) and
        the status flags (P,) of `ssmq_gp_theta_step`."""
        c = self._theta_static()
        theta = np.asarray(theta, dtype=np.float64)
        if theta.ndim < 2:
            theta = theta.reshape(1, -1)
        P = theta.shape[0]
        D = self.mod_dyn.dim_state
        par = np.exp(theta)
        pd = np.ascontiguousarray(par[:, :self.param_dyn_dim])
        po = np.ascontiguousarray(par[:, self.param_dyn_dim:])
        mean, cov = self._augment(mean, cov)
        mean = np.ascontiguousarray(mean, dtype=np.float64)
        cov = np.ascontiguousarray(cov, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        om, oc, ll = np.empty((P, D)), np.empty((P, D, D)), np.empty(P)
        st = np.zeros(P, dtype=np.int32)
        gqg = c['gqg']
        _lib.check(c['fn'](c['h_dyn'], c['pf_dyn'], c['h_obs'], c['pf_obs'], P, pd.ctypes.data, po.ctypes.data,
                           float(self.tf_dyn.model.kernel.jitter),
                           mean.ctypes.data, cov.ctypes.data, 1 if mean.ndim == 1 else 0, y.ctypes.data,
                           1 if y.ndim == 1 else 0, float(time), None if gqg is None else gqg.ctypes.data,
                           c['rr'].ctypes.data, om.ctypes.data, oc.ctypes.data, ll.ctypes.data, st.ctypes.data),
                   'ssmq_gp_theta_step')
        return om, oc, ll, st

    def _param_log_prior(self, theta):
        """log N(theta | param_mean, param_cov) (ssinf.py:1200-1218); host arithmetic on param_dim numbers."""
        d = np.atleast_2d(theta) - self.param_mean
        L = np.linalg.cholesky(self.param_cov)
        v = np.linalg.solve(L, d.T)
        return -0.5 * ((v ** 2).sum(axis=0) + 2 * np.log(np.diag(L)).sum() + self.param_dim * np.log(2 * np.pi))

    def _param_log_likelihood(self, theta, y, k):
        """ssinf.py:1153-1198 for one or many theta rows."""
        ll = self.theta_step(theta, self.x_mean_fi, self.x_cov_fi, y, k)[2]
        return ll if np.ndim(theta) == 2 else float(ll[0])

    def _param_neg_log_posterior(self, theta, y, k):
        """ssinf.py:1220-1241."""
        val = -self._param_log_likelihood(theta, y, k) - self._param_log_prior(theta)
        return val if np.ndim(theta) == 2 else float(val[0])

    def _param_objective_and_gradient(self, theta, y, k):
        """Objective and its forward-difference gradient from ONE theta-batched device call (param_dim + 1 items)."""
        pts = np.vstack((theta, theta + self.fd_step * np.eye(self.param_dim)))
        val = self._param_neg_log_posterior(pts, y, k)
        val = np.where(np.isfinite(val), val, np.inf)
        return float(val[0]), (val[1:] - val[0]) / ((theta + self.fd_step) - theta)

    def _param_posterior_moments(self, y, k):
        """Laplace approximation of the parameter posterior (ssinf.py:1243-1273)."""
        from scipy.optimize import minimize
        res = minimize(self._param_objective_and_gradient, self.param_mean, (y, k), method='BFGS', jac=True)
        self.param_mean, self.param_cov = res.x, res.hess_inv + self.param_jitter

    def _state_posterior_moments(self, theta, y, k):
        """ssinf.py:1117-1151."""
        m, c, _, st = self.theta_step(theta, self.x_mean_fi, self.x_cov_fi, y, k)
        if st.any():
            raise np.linalg.LinAlgError('Matrix is not positive definite (theta item {}, flags {})'.format(
                int(np.flatnonzero(st)[0]), int(st[np.flatnonzero(st)[0]])))
        return (m, c) if np.ndim(theta) == 2 else (m[0], c[0])

    def _measurement_update(self, y, time=None, laplace=None):
        """ssinf.py:1083-1115: all 2 param_dim theta points in one device call.  laplace = (mean, cov) of the parameter
        posterior replaces the optimiser run (a caller that has them already, e.g. to continue from another
        implementation's iterates)."""
        if laplace is None:
            self._param_posterior_moments(y, time)
        else:
            self.param_mean, self.param_cov = np.asarray(laplace[0], dtype=float), np.asarray(laplace[1], dtype=float)
        chol = np.linalg.cholesky(self.param_cov)
        param_pts = self.param_mean[:, None] + chol.dot(self.param_upts)
        mean, cov = self._state_posterior_moments(param_pts.T, y, time)
        self.x_mean_fi = np.einsum('ji,j->i', mean, self.param_wts)
        self.x_cov_fi = np.einsum('kij,k->ij', cov, self.param_wts)
        self._last_theta = param_pts[:, -1].copy()       # the parameters the reference's transforms are left with

    def _predictive_moments(self, time):
        """The generic time update of forward_pass (ssinf.py:104-107 -> :254-295), dynamics half: what the smoother later
        reads as pr_mean / pr_cov / pr_xx_cov.  One device transform with the weights the reference's dynamics transform
        holds at this point (see the class docstring)."""
        par = None if self._last_theta is None else np.exp(self._last_theta[:self.param_dyn_dim])[None, :]
        m_in, P_in = self._augment(np.asarray(self.x_mean_fi, dtype=float), np.asarray(self.x_cov_fi, dtype=float))
        m_pr, P_pr, C = self.tf_dyn.apply(self.mod_dyn.dyn_eval, m_in, P_in, np.atleast_1d(float(time)), par)
        if self.mod_dyn.noise_additive:
            P_pr = P_pr + self.G.dot(self.q_cov).dot(self.G.T)
        return m_pr, P_pr, C[:, :self.mod_dyn.dim_state]

    def forward_pass(self, data, keep_predictive=True):
        """data (dim_y, T) -> (D, T), (D, D, T).  ssinf.py:66-118.  The generic time update of step k only feeds the
        smoother (its moments are overwritten by the theta-conditioned ones inside the measurement update):
        keep_predictive=False skips it."""
        data = np.asarray(data, dtype=np.float64)
        T = data.shape[1]
        D = self.mod_dyn.dim_state
        fm, fP = np.zeros((D, T)), np.zeros((D, D, T))
        pm, pP, pC = np.zeros((D, T)), np.zeros((D, D, T)), np.zeros((D, D, T))
        for k in range(1, T + 1):
            if keep_predictive:
                pm[:, k - 1], pP[..., k - 1], pC[..., k - 1] = self._predictive_moments(k - 1)
            self._measurement_update(data[:, k - 1], k)
            fm[:, k - 1], fP[..., k - 1] = self.x_mean_fi, self.x_cov_fi
        self.fi_mean, self.fi_cov = fm, fP
        self.pr_mean, self.pr_cov, self.pr_xx_cov = (pm, pP, pC) if keep_predictive else (None, None, None)
        return fm, fP

    def backward_pass(self):
        """RTS smoothing of the last forward_pass (ssinf.py:120-147, inherited by the reference's MarginalInference): the
        recursion on the device (`ssmq_rts_backward_dev`) over the moments the forward pass kept."""
        assert self.fi_mean is not None and self.pr_mean is not None, 'run forward_pass (keep_predictive=True) first'
        lib = _lib.load()
        D, T = self.fi_mean.shape
        ld = 64

        def planes(a, n):             # (n..., T) -> [T][n][ld], trajectory 0
            buf = np.zeros((T, n, ld))
            buf[:, :, 0] = np.asarray(a, dtype=np.float64).reshape(n, T).T
            if n == D * D:
                buf[:, :, 1:] = np.eye(D).reshape(1, -1, 1)          # padding lanes are never read; keep them PD anyway
            d = _lib.DeviceBuffer(buf.nbytes)
            d.upload(buf)
            return d
        d_fm, d_fP = planes(self.fi_mean, D), planes(self.fi_cov, D * D)
        d_pm, d_pP, d_pC = planes(self.pr_mean, D), planes(self.pr_cov, D * D), planes(self.pr_xx_cov, D * D)
        d_sm, d_sP, d_st = _lib.DeviceBuffer(8 * T * D * ld), _lib.DeviceBuffer(8 * T * D * D * ld), _lib.DeviceBuffer(4 * ld)
        d_st.zero()
        _lib.check(lib.ssmq_rts_backward_dev(D, 1, ld, T, *(ctypes.c_void_p(b.ptr) for b in (d_fm, d_fP, d_pm, d_pP, d_pC, d_sm,
                                                                                             d_sP, d_st))), 'ssmq_rts_backward_dev')
        sm = d_sm.download((T, D, ld))[:, :, 0].T
        sP = d_sP.download((T, D, D, ld))[..., 0].transpose(1, 2, 0)
        st = int(d_st.download((ld,), dtype=np.int32)[0])
        for b in (d_fm, d_fP, d_pm, d_pP, d_pC, d_sm, d_sP, d_st):
            b.free()
        if st:
            raise np.linalg.LinAlgError('Matrix is not positive definite (a predictive covariance of the smoother)')
        self.sm_mean, self.sm_cov = np.ascontiguousarray(sm), np.ascontiguousarray(sP)
        return self.sm_mean, self.sm_cov

    def laplace_batch(self, mean, cov, y, time, prior_mean, prior_cov):
        """The Laplace step (ssinf.py:1243-1273) of B trajectories at once: B BFGS runs in lock step, every round one
        theta-batched device call (`ssmq_gp_marginal_laplace_batch`, csrc/ssmq_marginal.hip).  mean (B, D) / cov (B, D, D)
        filtered moments, y (B, Y), prior_mean (B, P) / prior_cov (B, P, P).  Returns the posterior modes (B, P), the BFGS
        inverse Hessians (B, P, P), status (B,), iterations (B,) and the number of device calls."""
        c = self._theta_static()
        lib = _lib.load()
        mean, cov = self._augment(np.asarray(mean, dtype=np.float64), np.asarray(cov, dtype=np.float64))
        mean, cov = np.ascontiguousarray(mean), np.ascontiguousarray(cov)
        y = np.ascontiguousarray(y, dtype=np.float64)
        B, P = mean.shape[0], self.param_dim
        pm, pc = np.ascontiguousarray(prior_mean, dtype=np.float64), np.ascontiguousarray(prior_cov, dtype=np.float64)
        theta = pm.copy()
        hinv = np.empty((B, P, P))
        st, it = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
        rounds = ctypes.c_int64(0)
        dp = lambda a: None if a is None else a.ctypes.data_as(_lib.c_double_p)       # noqa: E731
        ip = lambda a: a.ctypes.data_as(_lib.c_int32_p)       # noqa: E731
        _lib.check(lib.ssmq_gp_marginal_laplace_batch(c['h_dyn'], ctypes.byref(c['f_dyn']), c['h_obs'], ctypes.byref(c['f_obs']), B,
                                                      float(self.tf_dyn.model.kernel.jitter), dp(mean), dp(cov), dp(y), float(time),
                                                      dp(c['gqg']), dp(c['rr']), dp(pm), dp(pc), float(self.fd_step), dp(theta),
                                                      dp(hinv), ip(st), ip(it), ctypes.byref(rounds)),
                   'ssmq_gp_marginal_laplace_batch')
        return theta, hinv, st, it, int(rounds.value)

    def forward_pass_batch(self, data, **kwargs):
        """data (dim_y, T, B) -> (D, T, B), (D, D, T, B): the Monte-Carlo loop around forward_pass (research/tpq/tpq_base.py:175-192;
        every trajectory from the prior, as `reset()` leaves the filter), ONE call of `ssmq_gp_marginal_filter_batch`
        (csrc/ssmq_marginal.hip): every trajectory walks its own Laplace steps (BFGS), mixtures over the parameter points and time
        steps, and each device round serves whatever the unfinished trajectories are waiting for - the number of rounds is the
        longest trajectory's, not the sum over the steps of the slowest one's (forward_pass_batch_stepwise).
        `batch_failed[b]` = the step at which trajectory b failed (where forward_pass raises LinAlgError), 0 otherwise; its
        moments are NaN from that step on.  `batch_stats`: device rounds, BFGS iterations, theta items."""
        c = self._theta_static()
        lib = _lib.load()
        data = np.asarray(data, dtype=np.float64)
        Y, T, B = data.shape
        D, P = self.mod_dyn.dim_state, self.param_dim
        self.reset()
        y = np.ascontiguousarray(data.transpose(2, 1, 0))                      # (B, T, Y)
        dp = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(_lib.c_double_p)   # noqa: E731
        additive = self.mod_dyn.noise_additive
        qm = None if additive else np.ascontiguousarray(np.atleast_1d(self.q_mean), dtype=np.float64)
        qc = None if additive else np.ascontiguousarray(np.atleast_2d(self.q_cov), dtype=np.float64)
        x0m, x0c = np.ascontiguousarray(self.x0_mean, dtype=np.float64), np.ascontiguousarray(self.x0_cov, dtype=np.float64)
        pm0, pc0 = np.ascontiguousarray(self.param_prior_mean, dtype=np.float64), np.ascontiguousarray(self.param_prior_cov, dtype=np.float64)
        upts, uwts = np.ascontiguousarray(self.param_upts, dtype=np.float64), np.ascontiguousarray(self.param_wts, dtype=np.float64)
        fm, fP = np.empty((B, T, D)), np.empty((B, T, D, D))
        failed = np.zeros(B, dtype=np.int32)
        th, pcl = np.empty((B, P)), np.empty((B, P, P))
        stats = (ctypes.c_int64 * 3)()
        keep = (y, qm, qc, x0m, x0c, pm0, pc0, upts, uwts)                     # alive for the duration of the call
        _lib.check(lib.ssmq_gp_marginal_filter_batch(
            c['h_dyn'], ctypes.byref(c['f_dyn']), c['h_obs'], ctypes.byref(c['f_obs']), B, T, float(self.tf_dyn.model.kernel.jitter),
            dp(y), dp(x0m), dp(x0c), dp(qm), dp(qc), dp(c['gqg']), dp(c['rr']), dp(pm0), dp(pc0), dp(upts), dp(uwts),
            int(self.param_pts_num), float(self.fd_step), float(self.param_jitter[0, 0]), dp(fm), dp(fP),
            failed.ctypes.data_as(_lib.c_int32_p), dp(th), dp(pcl), stats), 'ssmq_gp_marginal_filter_batch')
        del keep
        self.batch_failed = (failed & 0xffff).astype(np.int64) if T < 65536 else failed.astype(np.int64)
        # why (include/ssmq.h): 1 prior not PD, 2 / 3 Laplace covariance not finite / not PD, 4 a mixture point's step failed, 5 mixture not finite
        self.batch_failed_reason = (failed >> 16).astype(np.int64) if T < 65536 else np.zeros(B, dtype=np.int64)
        self.batch_stats = dict(rounds=int(stats[0]), iterations=int(stats[1]), items=int(stats[2]), fallbacks=0)
        self.param_mean, self.param_cov = th[-1], pcl[-1]
        self.batch_param_mean, self.batch_param_cov = th, pcl      # every trajectory's last parameter posterior (B, P), (B, P, P)
        self.fi_mean = np.ascontiguousarray(fm.transpose(2, 1, 0))
        self.fi_cov = np.ascontiguousarray(fP.transpose(2, 3, 1, 0))
        if B and T and not failed[-1]:
            self.x_mean_fi, self.x_cov_fi = fm[-1, -1].copy(), fP[-1, -1].copy()
        return self.fi_mean, self.fi_cov

    def forward_pass_batch_stepwise(self, data, **kwargs):
        """The same with the trajectories in lock step PER TIME STEP (round 4's first version, kept as a second route to the same
        numbers): the Laplace step of every trajectory (`laplace_batch`: B x (param_dim + 1) theta items per device call), then the
        marginalisation over the 2 param_dim parameter sigma points of every trajectory in ONE call (B x 2 param_dim items).
        `batch_failed[b]` = the step at which trajectory b failed (where forward_pass raises LinAlgError), 0 otherwise; its
        moments are NaN from that step on.  `batch_stats`: device rounds, BFGS iterations."""
        data = np.asarray(data, dtype=np.float64)
        Y, T, B = data.shape
        D, P = self.mod_dyn.dim_state, self.param_dim
        self.reset()
        xm = np.tile(np.asarray(self.x0_mean, dtype=float), (B, 1))
        xP = np.tile(np.asarray(self.x0_cov, dtype=float), (B, 1, 1))
        pm = np.tile(self.param_prior_mean, (B, 1))
        pc = np.tile(self.param_prior_cov, (B, 1, 1))
        fm, fP = np.zeros((D, T, B)), np.zeros((D, D, T, B))
        npts = self.param_pts_num
        self.batch_stats = dict(rounds=0, fallbacks=0, iterations=0)
        # failed[b] = step at which trajectory b left the batch (0: still in it).  Where the serial loop raises LinAlgError for a
        # trajectory (a kernel matrix / covariance that is not positive definite at one of its parameter points), the batch keeps
        # going: NaN moments from that step on, the trajectory parked on the prior so that it costs nothing further.
        failed = np.zeros(B, dtype=np.int64)
        pts = None
        for k in range(1, T + 1):
            y = np.ascontiguousarray(data[:, k - 1, :].T)
            theta, hinv, st, it, rounds = self.laplace_batch(xm, xP, y, k, pm, pc)
            self.batch_stats['rounds'] += rounds
            self.batch_stats['iterations'] += int(it.sum())
            pm_new, pc_new = theta, hinv + self.param_jitter
            bad = (st == _lib.BFGS_PRIOR_NOT_PD) | ~np.all(np.isfinite(pm_new), axis=1) | ~np.all(np.isfinite(pc_new), axis=(1, 2))
            # a Laplace covariance that is not positive definite: numpy.linalg.cholesky would raise in _measurement_update
            ok_c = np.all(np.linalg.eigvalsh(np.where(bad[:, None, None], np.eye(P), 0.5 * (pc_new + pc_new.transpose(0, 2, 1)))) > 0, axis=1)
            bad |= ~ok_c
            pm_new[bad], pc_new[bad] = self.param_prior_mean, self.param_prior_cov
            pm, pc = pm_new, pc_new
            chol = np.linalg.cholesky(pc)
            pts = pm[:, :, None] + chol @ self.param_upts                      # (B, P, 2 P)
            items = np.ascontiguousarray(pts.transpose(0, 2, 1)).reshape(B * npts, P)
            m, c, _, sts = self.theta_step(items, np.repeat(xm, npts, axis=0), np.repeat(xP, npts, axis=0),
                                           np.repeat(y, npts, axis=0), k)
            bad |= sts.reshape(B, npts).any(axis=1)
            xm = np.einsum('bjd,j->bd', m.reshape(B, npts, D), self.param_wts)
            xP = np.einsum('bjde,j->bde', c.reshape(B, npts, D, D), self.param_wts)
            bad |= ~np.all(np.isfinite(xm), axis=1) | ~np.all(np.isfinite(xP), axis=(1, 2))
            newly = bad & (failed == 0)
            failed[newly] = k
            out = failed > 0
            xm[out], xP[out] = self.x0_mean, self.x0_cov
            pm[out], pc[out] = self.param_prior_mean, self.param_prior_cov
            fm[:, k - 1, :], fP[:, :, k - 1, :] = xm.T, xP.transpose(1, 2, 0)
            fm[:, k - 1, out], fP[:, :, k - 1, out] = np.nan, np.nan
        self.batch_failed = failed
        self.x_mean_fi, self.x_cov_fi, self.param_mean, self.param_cov = xm[-1], xP[-1], pm[-1], pc[-1]
        self._last_theta = pts[-1, :, -1].copy() if pts is not None else None
        self.fi_mean, self.fi_cov = fm, fP
        return fm, fP

    def forward_pass_serial(self, data, **kwargs):
        """One trajectory after another, one SciPy BFGS run per trajectory and step: the reference's own loop
        (research/tpq/tpq_base.py:175-192 around ssinf.py:66-118); kept as the yardstick of forward_pass_batch."""
        data = np.asarray(data, dtype=np.float64)
        out = []
        for b in range(data.shape[2]):
            self.reset()
            out.append(self.forward_pass(data[..., b], keep_predictive=False))
        self.fi_mean = np.stack([o[0] for o in out], axis=-1)
        self.fi_cov = np.stack([o[1] for o in out], axis=-1)
        return self.fi_mean, self.fi_cov

    def backward_pass_batch(self):
        raise NotImplementedError('the marginalised filter smooths one trajectory at a time: forward_pass + backward_pass')

    def kernel_name(self):
        return 'per theta batch: k_weights x2 | k_pack_wide_consts x2 | k_apply_wave x2 | k_kalman_update | k_gauss_logpdf'


class MarginalizedGaussianProcessKalman(MarginalInference):
    """ssinf.py:1276-1292.  The transforms are built with dim_out = 1 as in the reference, so model_var * I_out is
    broadcast over the whole covariance."""

    def __init__(self, dyn, obs, kernel='rbf', points='ut', point_hyp=None, par_mean=None, par_cov=None):
        kpar_dyn = np.ones((1, dyn.dim_in + 1))
        kpar_obs = np.ones((1, obs.dim_state + 1))
        t_dyn = GaussianProcessTransform(dyn.dim_in, 1, kpar_dyn, kernel, points, point_hyp)
        t_obs = GaussianProcessTransform(obs.dim_state, 1, kpar_obs, kernel, points, point_hyp)
        super().__init__(dyn, obs, t_dyn, t_obs, par_mean, par_cov)
