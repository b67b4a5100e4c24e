"""
Integrand models of Bayesian quadrature: kernel + point set -> quadrature weights (reference: ssmtoybox/bq/bqmod.py).

The weights are computed on the device (`ssmq_weights_gp` / `ssmq_weights_bs`, ssmtoybox_amd/csrc/ssmq_weights.hip);
this module keeps the reference's attribute surface: `points`, `kernel`, `dim_in`, `num_pts`, `q`, `Q`, `R`, `iK`,
`model_var`, `integral_var`, `nu`, `mulind` (bq/bqmod.py:85-106) and the `bq_weights(par, *args)` /
`exp_model_variance` / `integral_variance` methods the reference's tests and research scripts call.
Hyper-parameter optimisation, prediction, plotting and the multi-output models are out of scope.
"""
import numpy as np

from .. import _lib
from ..mtran import (SphericalRadialTransform, UnscentedTransform, GaussHermiteTransform,
                     FullySymmetricStudentTransform)
from .bqkern import RBFGauss, device_gp_weights


def n_sum_k(n, k):
    """Multi-indices of total degree k as columns of an (n, .) integer array, in the column order the reference's
    `utils.n_sum_k` produces (utils.py:459-475) - the order fixes the column order of the Vandermonde matrix and with
    it the Bayes-Sard weights.  Built degree by degree: level 1 is e_0 .. e_{n-1}; level d + 1 raises the first n - 1
    columns of level d by every e_j with j >= the column's position, then raises every remaining column by e_{n-1}.
    For k <= 2 that enumerates all n-tuples summing to k in lexicographic order of the index pair; for k >= 3 it is the
    reference's (incomplete) set, kept as is (pinned by tests/golden/g1_points.npz: nsumk_*)."""
    if k < 0:
        raise ValueError('k must be non-negative')
    if k == 0:
        return np.zeros((n, 1), dtype=int)
    unit = [tuple(int(r == j) for r in range(n)) for j in range(n)]
    level = list(unit)
    for _ in range(k - 1):
        head = [tuple(a + b for a, b in zip(level[i], unit[j])) for i in range(n - 1) for j in range(i, n)]
        tail = [tuple(a + b for a, b in zip(col, unit[n - 1])) for col in level[n - 1:]]
        level = head + tail
    return np.array(level, dtype=int).T.reshape(n, -1)


class Model:
    """Kernel + point set (bq/bqmod.py:15-106)."""

    _supported_points_ = ['sr', 'ut', 'gh', 'fs']
    _supported_kernels_ = ['rbf']

    def __init__(self, dim, kern_par, kern_str, point_str, point_par, estimate_par):
        self.kernel = Model.get_kernel(dim, kern_str, kern_par)
        self.points = Model.get_points(dim, point_str, point_par)
        self.estimate_par = estimate_par
        self.str_pts = point_str
        self.str_pts_par = str(point_par)
        self.dim_in, self.num_pts = self.points.shape
        self.eye_d, self.eye_n = np.eye(self.dim_in), np.eye(self.num_pts)
        self.q, self.Q, self.R, self.iK = None, None, None, None
        self.model_var = None
        self.integral_var = None

    def __str__(self):
        return '{} {}\n{} {}'.format(type(self.kernel).__name__, self.kernel.par, self.str_pts, self.str_pts_par)

    @staticmethod
    def get_points(dim, points, point_par):
        """bq/bqmod.py:340-382; unknown strings print a message and return None, as the reference does."""
        points = points.lower()
        if points not in Model._supported_points_:
            print('Points {} not supported. Supported points are {}.'.format(points, Model._supported_points_))
            return None
        point_par = {} if point_par is None else point_par
        if points == 'sr':
            return SphericalRadialTransform.unit_sigma_points(dim)
        if points == 'ut':
            return UnscentedTransform.unit_sigma_points(dim, **point_par)
        if points == 'gh':
            return GaussHermiteTransform.unit_sigma_points(dim, **point_par)
        return FullySymmetricStudentTransform.unit_sigma_points(dim, **point_par)

    @staticmethod
    def get_kernel(dim, kernel, par):
        """bq/bqmod.py:384-423 ('rq' and 'rbf-student' are not on this path)."""
        kernel = kernel.lower()
        if kernel not in Model._supported_kernels_:
            print('Kernel {} not supported. Supported kernels are {}.'.format(kernel, Model._supported_kernels_))
            return None
        return RBFGauss(dim, par)


class GaussianProcessModel(Model):
    """GP quadrature weights (bq/bqmod.py:426-535)."""

    def __init__(self, dim, kern_par, kern_str, point_str, point_par=None, estimate_par=False):
        super().__init__(dim, kern_par, kern_str, point_str, point_par, estimate_par)

    def bq_weights(self, par, *args):
        """wm = q iK, Wc = iK Q iK (symmetrised), Wcc = R iK and the model / integral variances (bq/bqmod.py:495-523)."""
        par = self.kernel.get_parameters(par)
        w = device_gp_weights(self.points, par[:1], self.kernel.jitter)
        self.q, self.Q, self.R, self.iK = w['q'][0], w['Q'][0], w['R'][0], w['iK'][0]
        self.model_var = float(w['model_var'][0])
        self.integral_var = float(w['integral_var'][0])
        return w['wm'][0], w['Wc'][0], w['Wcc'][0], self.model_var, self.integral_var

    def bq_weights_batch(self, pars):
        """theta-batched weights: pars (P, 1 + D) -> dict with a leading P axis (one workgroup per row)."""
        return device_gp_weights(self.points, np.atleast_2d(pars), self.kernel.jitter)

    def exp_model_variance(self, par, *args):
        """alpha^2 (1 - tr(Q iK)) with iK the inverse of the SCALED kernel matrix (bq/bqmod.py:525-528: eval_inv_dot is
        called with its default scaling=True there) and Q the unscaled expectation - both from the device, the trace of
        their product on the host."""
        par = self.kernel.get_parameters(par)
        iK = self.kernel.eval_inv_dot(par, self.points)
        Q = self.kernel.exp_x_kxkx(par, par, self.points)
        return float(self.kernel.exp_x_kxx(par) * (1 - np.trace(Q.dot(iK))))

    def integral_variance(self, par, *args):
        """bq/bqmod.py:530-535."""
        par = self.kernel.get_parameters(par)
        return float(device_gp_weights(self.points, par[:1], self.kernel.jitter)['integral_var'][0])


class StudentTProcessModel(GaussianProcessModel):
    """Student-t process: GP weights, data-dependent model variance (bq/bqmod.py:1060-1190)."""

    def __init__(self, dim, kern_par, kern_str, point_str, point_par=None, estimate_par=False, nu=4.0):
        super().__init__(dim, kern_par, kern_str, point_str, point_par, estimate_par)
        self.nu = 3.0 if nu < 2 else nu

    def exp_model_variance(self, par, *args):
        """(nu - 2 + fx iK fx') / (nu - 2 + N) * model_var with the cached scaling=False inverse
        (bq/bqmod.py:1132-1160, estimate_par=False branch).  Host arithmetic on (E, N) data for callers that ask for the
        number; `apply()` computes the same quantity inside the device kernel."""
        fcn_obs = np.squeeze(args[0])
        scale = (self.nu - 2 + fcn_obs.dot(self.iK).dot(fcn_obs.T)) / (self.nu - 2 + self.num_pts)
        return scale * self.model_var

    def integral_variance(self, par, *args):
        fcn_obs = np.squeeze(args[0])
        scale = (self.nu - 2 + fcn_obs.dot(self.iK).dot(fcn_obs.T)) / (self.nu - 2 + self.num_pts)
        return scale * self.integral_var


class BayesSardModel(Model):
    """GP with a multivariate polynomial prior mean (bq/bqmod.py:599-1057)."""

    def __init__(self, dim, kern_par, multi_ind=2, point_str='ut', point_par=None, estimate_par=False):
        super().__init__(dim, kern_par, 'rbf', point_str, point_par, estimate_par)
        if type(multi_ind) is int:
            self.mulind = np.hstack([n_sum_k(dim, td) for td in range(multi_ind + 1)])
        elif type(multi_ind) is np.ndarray:
            self.mulind = multi_ind
        else:
            raise ValueError('Multi-index error: multi-index has to be either int or ndarray')

    def bq_weights(self, par, multi_ind=None):
        """bq/bqmod.py:893-992.  NOTE the reference crashes when handed an int multi-index here (SURVEY.md appendix
        B-2); this build falls back to the expanded `self.mulind` for anything that is not an ndarray."""
        if not isinstance(multi_ind, np.ndarray):
            multi_ind = self.mulind
        par = self.kernel.get_parameters(par)
        if multi_ind.shape[0] != self.dim_in:
            raise ValueError('Dimension mismatch {:d} != {:d}. Dimension of monomials must be equal to the dimension'
                             ' of the sigma-points.'.format(multi_ind.shape[0], self.dim_in))
        nb = multi_ind.shape[1]
        if nb > self.num_pts:
            raise ValueError('Number of basis functions needs to be lower than or equal to the number of points.'
                             'You supplied {:d} basis functions and {:d} points.'.format(nb, self.num_pts))
        lib = _lib.load()
        x, px = _lib.as_c(self.points)
        p, pp = _lib.as_c(par[:1])
        mi = np.ascontiguousarray(multi_ind, dtype=np.int32)
        D, N = x.shape
        out = {k: _lib.out_c(s) for k, s in (('wm', (N,)), ('Wc', (N, N)), ('Wcc', (D, N)), ('iK', (N, N)),
                                             ('q', (N,)), ('Q', (N, N)), ('R', (D, N)), ('mv', (1,)), ('iv', (1,)))}
        st = np.zeros(1, dtype=np.int32)
        rc = _lib.check(lib.ssmq_weights_bs(D, N, px, pp, 1, float(self.kernel.jitter),
                                            mi.ctypes.data_as(_lib.c_int32_p), nb, out['wm'][1], out['Wc'][1],
                                            out['Wcc'][1], out['iK'][1], out['q'][1], out['Q'][1], out['R'][1],
                                            out['mv'][1], out['iv'][1], st.ctypes.data_as(_lib.c_int32_p)),
                        'ssmq_weights_bs')
        if rc > 0:
            raise np.linalg.LinAlgError('Bayes-Sard weights: matrix not positive definite / singular (code {})'.format(
                int(st[0])))
        self.q, self.iK = out['q'][0], out['iK'][0]
        if nb < N:
            self.Q, self.R = out['Q'][0], out['R'][0]
        self.model_var = float(out['mv'][0][0])
        self.integral_var = float(out['iv'][0][0])
        return out['wm'][0], out['Wc'][0], out['Wcc'][0], self.model_var, self.integral_var

    def _moments(self, multi_ind, x=None, par=None, want=('px',)):
        """The polynomial expectations behind the weights (`ssmq_bs_moments`), one array per name in `want`."""
        mi = np.ascontiguousarray(multi_ind, dtype=np.int32)
        if mi.ndim != 2:
            raise ValueError('multi-index matrix must be (dim, num_basis)')
        D, NB = mi.shape
        N = 0 if x is None else np.atleast_2d(x).shape[1]
        shapes = {'px': (NB,), 'xpx': (D, NB), 'pxpx': (NB, NB), 'kxpx': (N, NB)}
        out = {k: _lib.out_c(shapes[k]) for k in want}
        xs = _lib.as_c(np.atleast_2d(np.asarray(x, dtype=np.float64)))[1] if x is not None else None
        pp = _lib.as_c(self.kernel.get_parameters(par)[:1])[1] if par is not None else None
        ptr = lambda k: out[k][1] if k in out else None          # noqa: E731
        _lib.check(_lib.load().ssmq_bs_moments(D, N, xs, pp, mi.ctypes.data_as(_lib.c_int32_p), NB, ptr('px'), ptr('xpx'),
                                               ptr('pxpx'), ptr('kxpx'), None), 'ssmq_bs_moments')
        return [out[k][0] for k in want]

    def _exp_x_px(self, multi_ind):
        """bq/bqmod.py:635-662: E[p_q(x)], (Q,)."""
        return self._moments(multi_ind, want=('px',))[0]

    def _exp_x_xpx(self, multi_ind):
        """bq/bqmod.py:664-698: E[x p(x)'], (D, Q)."""
        return self._moments(multi_ind, want=('xpx',))[0]

    def _exp_x_pxpx(self, multi_ind):
        """bq/bqmod.py:700-731: E[p(x) p(x)'], (Q, Q)."""
        return self._moments(multi_ind, want=('pxpx',))[0]

    def _exp_x_kxpx(self, par, multi_ind, x):
        """bq/bqmod.py:733-797: E[k(x, x_n) p_q(x)], (N, Q), on the device."""
        return self._moments(multi_ind, x=x, par=par, want=('kxpx',))[0]

    def _variances(self, pars, multi_ind=None):
        """theta-batched (model_var, integral_var) with the semantics of the reference's stand-alone methods
        (`ssmq_variances_bs`): pars (P, 1 + D)."""
        if not isinstance(multi_ind, np.ndarray):
            multi_ind = self.mulind
        lib = _lib.load()
        x, px = _lib.as_c(self.points)
        p, pp = _lib.as_c(np.atleast_2d(np.asarray(pars, dtype=np.float64)))
        mi = np.ascontiguousarray(multi_ind, dtype=np.int32)
        D, N = x.shape
        P = p.shape[0]
        if p.shape[1] != D + 1:
            raise ValueError('kernel parameters must have 1 + dim entries per row')
        mv, pmv = _lib.out_c((P,))
        iv, piv = _lib.out_c((P,))
        st = np.zeros(P, dtype=np.int32)
        rc = _lib.check(lib.ssmq_variances_bs(D, N, px, pp, P, float(self.kernel.jitter),
                                              mi.ctypes.data_as(_lib.c_int32_p), mi.shape[1], pmv, piv,
                                              st.ctypes.data_as(_lib.c_int32_p)), 'ssmq_variances_bs')
        if rc > 0:
            raise np.linalg.LinAlgError('Bayes-Sard variances: matrix not positive definite (parameter row {}, code {})'
                                        .format(rc - 1, int(st[rc - 1])))
        return mv, iv

    def exp_model_variance(self, par, mulind=None):
        """bq/bqmod.py:995-1026 (not the value bq_weights() returns: no jitter on V' iK V, general formula always)."""
        return float(self._variances(self.kernel.get_parameters(par)[:1], mulind)[0][0])

    def integral_variance(self, par, mulind=None):
        """bq/bqmod.py:1028-1050."""
        return float(self._variances(self.kernel.get_parameters(par)[:1], mulind)[1][0])

    def exp_model_variance_batch(self, pars, mulind=None):
        """Length-scale sweeps (research/bsq/bsq_ungm.py:244-282) in one launch: pars (P, 1 + D) -> (P,)."""
        return self._variances(pars, mulind)[0]

    def integral_variance_batch(self, pars, mulind=None):
        return self._variances(pars, mulind)[1]

