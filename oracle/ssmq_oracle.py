"""
CPU oracle (NumPy) for the sigma-point / Bayesian-quadrature moment-transform path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it; it is the *checker*, never the thing measured or shipped.
The product path lives in `ssmtoybox_amd/` and runs hand-written HIP kernels through the C-ABI in
`include/ssmq.h`; it never falls back to this module.

It is a plain restatement, written from the reference's published algorithm, of (all paths under
/root/reference/ssmtoybox, cited per function):
    point sets            mtran.py:171-204, 234-293, 315-360, 405-578
    RBF kernel + moments  bq/bqkern.py:329-424, utils.py:385-409
    K^-1 by Cholesky      bq/bqkern.py:38-64, 96-120
    GP / TP / BS weights  bq/bqmod.py:495-523, 893-992, 1132-1160
    moment transforms     bq/bqmtran.py:60-109, 158-223, 394-415 ; mtran.py:105-149
    integrands            ssmod.py (closed-form dynamics / measurement functions)
    Gaussian / Student filter recursions   ssinf.py:66-118, 254-323, 634-736 (non-additive noise :271-295, RTS :120-147)
    marginalised-filter theta step         ssinf.py:1117-1198
    performance metrics                    utils.py:18-148 (aggregation research/tpq/tpq_base.py:154-172)
    BS model / integral variance           bq/bqmod.py:995-1050
    simulators                             ssmod.py:168-199, 1011-1039, with THIS BUILD's counter-based generator
                                           (Philox4x32-10, pinned by the Random123 known-answer vectors)

Parity pinning: every function here is checked in tests/test_oracle_golden.py against golden vectors that were
produced by importing the reference itself in the build container (tests/golden/make_golden.py; fixtures
tests/golden/*.npz) and against the reference's own known-answer tests.
All arithmetic is IEEE fp64.
"""
import math

import numpy as np
from numpy.polynomial.hermite_e import hermegauss, hermeval
from scipy.linalg import cho_factor, cho_solve

# --------------------------------------------------------------------------------------------------------------
# point sets (a14)
# --------------------------------------------------------------------------------------------------------------


def _default_kappa(dim, kappa):
    # mtran.py:254,284: kappa defaults to max(3 - dim, 0)
    return max(3.0 - dim, 0.0) if kappa is None else kappa


def points_ut(dim, kappa=None, alpha=1.0):
    """Unscented unit points [0 | c I | -c I], c = sqrt(dim + lambda).  mtran.py:234-257."""
    kappa = _default_kappa(dim, kappa)
    lam = alpha ** 2 * (dim + kappa) - dim
    c = np.sqrt(dim + lam)
    pts = np.zeros((dim, 2 * dim + 1))
    pts[:, 1:dim + 1] = c * np.eye(dim)
    pts[:, dim + 1:] = -c * np.eye(dim)
    return pts


def weights_ut(dim, kappa=None, alpha=1.0, beta=2.0):
    """Unscented mean / covariance weights.  mtran.py:259-293."""
    kappa = _default_kappa(dim, kappa)
    lam = alpha ** 2 * (dim + kappa) - dim
    wm = np.full(2 * dim + 1, 1.0 / (2.0 * (dim + lam)))
    wc = wm.copy()
    wm[0] = lam / (dim + lam)
    wc[0] = wm[0] + (1 - alpha ** 2 + beta)
    return wm, wc


def points_sr(dim):
    """Spherical-radial unit points sqrt(dim) [I | -I].  mtran.py:187-204."""
    c = np.sqrt(dim)
    return np.concatenate((c * np.eye(dim), -c * np.eye(dim)), axis=1)


def weights_sr(dim):
    """mtran.py:171-185."""
    return np.full(2 * dim, 1.0 / (2.0 * dim))


def _cartesian_rows(vec, dim):
    # all dim-tuples drawn from vec, last coordinate varying fastest (sklearn.utils.extmath.cartesian order,
    # which mtran.py:336,359 relies on)
    grids = np.meshgrid(*([vec] * dim), indexing='ij')
    return np.stack([g.reshape(-1) for g in grids], axis=1)


def points_gh(dim, degree=3):
    """Gauss-Hermite product grid.  mtran.py:338-360."""
    x, _ = hermegauss(degree)
    return _cartesian_rows(x, dim).T


def weights_gh(dim, degree=3):
    """GH weights deg!/(deg^2 He_{deg-1}(x)^2), product over dimensions.  mtran.py:315-336."""
    x, _ = hermegauss(degree)
    coef = [0] * (degree - 1) + [1]
    w = math.factorial(degree) / (degree ** 2 * hermeval(x, coef) ** 2)
    return np.prod(_cartesian_rows(w, dim), axis=1)


def _fs_generator_set(dim, gen):
    """Fully symmetric set for a generator with equal entries, in the reference's column order.
    mtran.py:522-578 (recursive symmetric_set; for each leading index i, for each sub-point: +u then -u)."""
    if len(gen) == 0:
        return np.zeros((dim, 1))
    cols = []
    for i in range(dim):
        if len(gen) == 1:
            u = np.zeros(dim)
            u[i] = gen[0]
            cols += [u, -u]
        else:
            sub = _fs_generator_set(dim - i - 1, gen[1:])
            if dim - i - 1 == 0:
                continue
            for j in range(sub.shape[1]):
                u = np.zeros(dim)
                u[i] = gen[0]
                u[i + 1:] = sub[:, j]
                cols += [u, -u]
    if not cols:
        return np.zeros((dim, 0))
    return np.stack(cols, axis=1)


def points_fs(dim, degree=3, kappa=None, dof=4.0):
    """Fully-symmetric Student-t unit points, degree 3 or 5.  mtran.py:473-520."""
    if degree not in (3, 5):
        degree = 3
    kappa = _default_kappa(dim, kappa)
    dof = max(dof, degree)
    i2 = dof / (dof - 2)
    if degree == 3:
        u = np.sqrt(i2 * (dim + kappa))
        pts = np.zeros((dim, 2 * dim + 1))
        pts[:, 1:dim + 1] = np.eye(dim)
        pts[:, dim + 1:] = -np.eye(dim)
        return u * pts
    i4 = 3 * dof ** 2 / ((dof - 2) * (dof - 4))
    u = np.sqrt(i4 / i2)
    return np.concatenate((_fs_generator_set(dim, []), _fs_generator_set(dim, [u]),
                           _fs_generator_set(dim, [u, u])), axis=1)


def weights_fs(dim, degree=3, kappa=None, dof=4.0):
    """mtran.py:405-471."""
    if degree not in (3, 5):
        degree = 3
    kappa = _default_kappa(dim, kappa)
    dof = max(dof, degree)
    if degree == 3:
        w = np.full(2 * dim + 1, 1 / (2 * (dim + kappa)))
        w[0] = kappa / (dim + kappa)
        return w
    i2 = dof / (dof - 2)
    i22 = dof ** 2 / ((dof - 2) * (dof - 4))
    i4 = 3 * i22
    a0 = 1 - dim * (i2 / i4) ** 2 * (i4 - 0.5 * (dim - 1) * i22)
    a1 = 0.5 * (i2 / i4) ** 2 * (i4 - (dim - 1) * i22)
    a11 = 0.25 * (i2 / i4) ** 2 * i22
    return np.concatenate(([a0], np.full(2 * dim, a1), np.full(2 * dim * (dim - 1), a11)))


def unit_points(dim, kind, point_par=None):
    """Dispatch by the reference's strings 'ut' | 'sr' | 'gh' | 'fs'.  bq/bqmod.py:340-382."""
    point_par = {} if point_par is None else dict(point_par)
    kind = kind.lower()
    if kind == 'ut':
        return points_ut(dim, **point_par)
    if kind == 'sr':
        return points_sr(dim)
    if kind == 'gh':
        return points_gh(dim, **point_par)
    if kind == 'fs':
        return points_fs(dim, **point_par)
    raise ValueError(kind)


# --------------------------------------------------------------------------------------------------------------
# RBF kernel, its Gaussian expectations and K^-1 (a9, a10, a11)
# --------------------------------------------------------------------------------------------------------------


def _pairwise_maha(a, b, v=None):
    """Weighted squared distances of the rows of a and b by the expansion |a|^2 + |b|^2 - 2 a.b  (utils.py:385-409).
    v is the diagonal of the weight matrix (the reference only ever passes diagonal V on this path)."""
    av = a if v is None else a * v
    bv = b if v is None else b * v
    a2 = np.sum(av * a, axis=1)
    b2 = np.sum(bv * b, axis=1)
    return (a2[:, None] + b2[None, :]) - 2 * av.dot(b.T)


def _split_par(par):
    """par = [alpha, ell_1..ell_D] -> alpha, 1/ell (vector).  bq/bqkern.py:438-454."""
    par = np.asarray(par, dtype=float).reshape(-1)
    return par[0], par[1:] ** -1


def rbf_eval(par, x1, x2=None, scaling=True):
    """K_ij = exp(2 log(alpha) - maha/2) on length-scale-normalised points.  bq/bqkern.py:329-343."""
    alpha, sil = _split_par(par)
    if not scaling:
        alpha = 1.0
    x2 = x1 if x2 is None else x2
    a = (sil[:, None] * x1).T
    b = (sil[:, None] * x2).T
    return np.exp(2 * np.log(alpha) - 0.5 * _pairwise_maha(a, b))


def rbf_eval_diag(par, x1, x2, scaling=True):
    """k(x1_i, x2_i) through the difference form of the `diag=True` branch.  bq/bqkern.py:338-341."""
    alpha, sil = _split_par(par)
    if not scaling:
        alpha = 1.0
    dx = sil[:, None] * x1 - sil[:, None] * x2
    return np.exp(2 * np.log(alpha) - 0.5 * np.sum(dx * dx, axis=0))


def rbf_chol(par, x, jitter=1e-8, scaling=True):
    """Lower Cholesky factor of K + jitter I.  bq/bqkern.py:122-142."""
    return np.linalg.cholesky(rbf_eval(par, x, scaling=scaling) + jitter * np.eye(x.shape[1]))


def rbf_inv_dot(par, x, b, jitter=1e-8, scaling=True):
    """sym((K + jitter I)^-1 b) for a square b: _cho_inv symmetrises whatever it solved for.  bq/bqkern.py:38-64, 96-120."""
    ia = cho_solve(cho_factor(rbf_eval(par, x, scaling=scaling) + jitter * np.eye(x.shape[1])), b)
    return 0.5 * (ia + ia.T)


def gp_exp_model_variance(par, x, jitter=1e-8):
    """GaussianProcessModel.exp_model_variance as a call: the inverse of the SCALED kernel matrix (eval_inv_dot's default)
    with the unscaled Q.  bq/bqmod.py:525-528."""
    iK = rbf_inv(par, x, jitter, scaling=True)
    return rbf_kxx(par) * (1 - np.trace(rbf_Q(par, par, x).dot(iK)))


def rbf_q(par, x, scaling=False):
    """Kernel mean q_n = alpha^2 det(Lam^-1 + I)^-1/2 exp(-x_n'(Lam+I)^-1 x_n / 2).  bq/bqkern.py:345-356."""
    alpha, sil = _split_par(par)
    if not scaling:
        alpha = 1.0
    inv_lam = sil ** 2
    lam = inv_lam ** -1
    c = alpha ** 2 * np.prod(inv_lam + 1.0) ** -0.5
    xl = (1.0 / (lam + 1.0))[:, None] * x
    return c * np.exp(-0.5 * np.sum(x * xl, axis=0))


def rbf_R(par, x):
    """R = q * (Lam + I)^-1 x  (D, N).  bq/bqkern.py:358-364."""
    _, sil = _split_par(par)
    lam = sil ** -2
    mu = (1.0 / (lam + 1.0))[:, None] * x
    return rbf_q(par, x)[None, :] * mu


def rbf_Q(par0, par1, x, scaling=False):
    """Q_ij = det(r)^-1/2 exp(xi_i + xi'_j + maha(Lam0^-1 x_i, -Lam1^-1 x_j; r^-1)/2), r = Lam0^-1 + Lam1^-1 + I.
    bq/bqkern.py:366-415."""
    a0, sil0 = _split_par(par0)
    a1, sil1 = _split_par(par1)
    if not scaling:
        a0, a1 = 1.0, 1.0
    il0, il1 = sil0 ** 2, sil1 ** 2
    z0 = sil0[:, None] * x
    z1 = sil1[:, None] * x
    xi0 = 2 * np.log(a0) - 0.5 * np.sum(z0 * z0, axis=0)
    xi1 = 2 * np.log(a1) - 0.5 * np.sum(z1 * z1, axis=0)
    y0 = il0[:, None] * x
    y1 = il1[:, None] * x
    r = il0 + il1 + 1.0
    n = (xi0[:, None] + xi1[None, :]) + 0.5 * _pairwise_maha(y0.T, -y1.T, 1.0 / r)
    return np.prod(r) ** -0.5 * np.exp(n)


def rbf_kxx(par):
    """E_x[k(x,x)] = alpha^2.  bq/bqkern.py:417-419."""
    return _split_par(par)[0] ** 2


def rbf_kbar(par):
    """E_xx'[k(x,x')] = alpha^2 det(2 Lam^-1 + I)^-1/2.  bq/bqkern.py:421-424."""
    alpha, sil = _split_par(par)
    return alpha ** 2 * np.prod(2 * sil ** 2 + 1.0) ** -0.5


def chol_inverse(a):
    """Symmetrised inverse of an SPD matrix through its Cholesky factor.  bq/bqkern.py:38-64."""
    ia = cho_solve(cho_factor(a), np.eye(a.shape[0]))
    return 0.5 * (ia + ia.T)


def rbf_inv(par, x, jitter=1e-8, scaling=True):
    """(K + jitter I)^-1.  bq/bqkern.py:96-120 (jitter default: bq/bqkern.py:322)."""
    return chol_inverse(rbf_eval(par, x, scaling=scaling) + jitter * np.eye(x.shape[1]))


# --------------------------------------------------------------------------------------------------------------
# quadrature weights (a12, a13)
# --------------------------------------------------------------------------------------------------------------


def _sym_if_needed(w):
    # bq/bqmod.py:520-521, 990-991
    return w if np.array_equal(w, w.T) else 0.5 * (w + w.T)


def gp_weights(par, x, jitter=1e-8):
    """GP quadrature weights.  bq/bqmod.py:495-523.
    Returns dict(wm, Wc, Wcc, iK, q, Q, R, model_var, integral_var)."""
    par = np.asarray(par, dtype=float).reshape(-1)
    iK = rbf_inv(par, x, jitter, scaling=False)
    q = rbf_q(par, x)
    Q = rbf_Q(par, par, x)
    R = rbf_R(par, x)
    wm = q.dot(iK)
    Wc = iK.dot(Q).dot(iK)
    Wcc = R.dot(iK)
    model_var = rbf_kxx(par) * (1 - np.trace(Q.dot(iK)))
    integral_var = rbf_kbar(par) - q.dot(iK).dot(q)
    return dict(wm=wm, Wc=_sym_if_needed(Wc), Wcc=Wcc, iK=iK, q=q, Q=Q, R=R, model_var=model_var,
                integral_var=integral_var)


def n_sum_k(n, k):
    """All n-tuples of non-negative integers summing to k, in the reference's column order.  utils.py:459-475."""
    if k == 0:
        return np.zeros((n, 1), dtype=int)
    if k == 1:
        return np.eye(n, dtype=int)
    a = n_sum_k(n, k - 1)
    eye = np.eye(n, dtype=int)
    cols = []
    for i in range(n - 1):
        for j in range(i, n):
            cols.append(a[:, i] + eye[:, j])
    # the reference pre-allocates n(n+1)/2 - 1 columns for this part and fills them in this order
    width = (n * (1 + n) // 2) - 1
    temp = np.zeros((n, width), dtype=int)
    for c, col in enumerate(cols):
        temp[:, c] = col
    return np.concatenate((temp, a[:, n - 1:] + eye[:, -1, None]), axis=1)


def total_degree_multi_index(dim, degree):
    """Monomials of total degree <= degree.  bq/bqmod.py:621-628."""
    return np.concatenate([n_sum_k(dim, td) for td in range(degree + 1)], axis=1)


def vandermonde(mul_ind, x):
    """V[n, b] = prod_d x[d, n] ** mul_ind[d, b].  utils.py:478-502."""
    return np.prod(x[:, :, None] ** mul_ind[:, None, :], axis=0)


def _dfact(n):
    """Double factorial with (-1)!! = 0!! = 1 (bq/bqmod.py:656-661)."""
    n = int(n)
    r = 1
    while n > 1:
        r *= n
        n -= 2
    return r


def poly_px(mi):
    """E[p_q(x)] under N(0, I).  bq/bqmod.py:635-662."""
    d, nq = mi.shape
    out = np.zeros(nq)
    for q in range(nq):
        if np.all(mi[:, q] % 2 == 0):
            out[q] = np.prod([_dfact(mi[k, q] - 1) for k in range(d)])
    return out


def poly_xpx(mi):
    """E[x_e p_q(x)].  bq/bqmod.py:664-698 (note: the factor is alpha_e, the reference's formula)."""
    d, nq = mi.shape
    out = np.zeros((d, nq))
    for e in range(d):
        others = np.arange(d) != e
        for q in range(nq):
            rest = mi[others, q]
            if (mi[e, q] + 1) % 2 == 0 and np.all(rest % 2 == 0):
                out[e, q] = mi[e, q] * np.prod([_dfact(a - 1) for a in rest])
    return out


def poly_pxpx(mi):
    """E[p_r(x) p_q(x)].  bq/bqmod.py:700-731."""
    d, nq = mi.shape
    out = np.zeros((nq, nq))
    for r in range(nq):
        for q in range(nq):
            s = mi[:, r] + mi[:, q]
            if np.all(s % 2 == 0):
                out[r, q] = np.prod([_dfact(a - 1) for a in s])
    return out


def poly_kxpx(par, mi, x):
    """E[k(x, x_n) p_q(x)] in closed form (per-dimension product).  bq/bqmod.py:733-797."""
    d, nq = mi.shape
    npts = x.shape[1]
    _, sil = _split_par(par)
    ell = sil ** -2   # NB: the reference's `ell` here is 1/(1/ell)^2 = ell^2 ... kept as written there
    out = np.zeros((npts, nq))
    for n in range(npts):
        for q in range(nq):
            prod = 1.0
            for k in range(d):
                a = int(mi[k, q])
                ea = ell[k] * (1 + ell[k] ** 2) ** (-(1 + a) / 2) * np.exp(-x[k, n] ** 2 / (2 * (1 + ell[k] ** 2)))
                eb = 0.0
                for m in range(a // 2 + 1):
                    c = math.factorial(a) / ((2 ** m) * math.factorial(m) * math.factorial(a - 2 * m))
                    eb += c * (ell[k] ** (2 * m)) * ((x[k, n] / np.sqrt(1 + ell[k] ** 2)) ** (a - 2 * m))
                prod *= ea * eb
            out[n, q] = prod
    return out


def bs_weights(par, x, mi, jitter=1e-8):
    """Bayes-Sard quadrature weights (unisolvent and general branch).  bq/bqmod.py:893-992."""
    par = np.asarray(par, dtype=float).reshape(-1)
    mi = np.asarray(mi)
    npts = x.shape[1]
    nq = mi.shape[1]
    iK = rbf_inv(par, x, jitter, scaling=False)
    V = vandermonde(mi, x)
    iViKV = cho_solve(cho_factor(V.T.dot(iK).dot(V) + 1e-8 * np.eye(nq)), np.eye(nq))
    px, xpx, pxpx = poly_px(mi), poly_xpx(mi), poly_pxpx(mi)
    kxpx = poly_kxpx(par, mi, x)
    q = rbf_q(par, x)
    kbar = rbf_kbar(par)
    ks2 = par[0] ** 2
    Q = R = None
    if nq == npts:
        iV = np.linalg.solve(V, np.eye(nq))
        wm = iV.T.dot(px)
        Wc = iV.T.dot(pxpx).dot(iV)
        Wcc = xpx.dot(iV)
        model_var = ks2 * (1 - np.trace(kxpx.T.dot(iV.T) + kxpx.dot(iV) - pxpx.dot(iViKV)))
        integral_var = kbar - q.dot(iV.T).dot(px) - px.dot(iV).dot(q) + px.dot(iViKV).dot(px)
    elif nq < npts:
        Q = rbf_Q(par, par, x)
        R = rbf_R(par, x)
        Z = V.T.dot(iK)
        A = V.dot(iViKV)
        b = Z.dot(q) - px
        Bm = Z.dot(Q).dot(Z.T) + pxpx - Z.dot(kxpx) - kxpx.T.dot(Z.T)
        Dm = R.dot(Z.T) - xpx
        wm = iK.dot(q - A.dot(b))
        Wc = iK.dot(Q - A.dot(Bm).dot(A.T)).dot(iK)
        Wcc = (R - Dm.dot(A.T)).dot(iK)
        model_var = ks2 * (1 - np.trace(Q.dot(iK)) + np.trace(Bm.dot(iViKV)))
        integral_var = kbar - q.dot(iK).dot(q) + b.dot(iViKV).dot(b)
    else:
        raise ValueError('more basis functions than points')
    return dict(wm=wm, Wc=_sym_if_needed(Wc), Wcc=Wcc, iK=iK, q=q, Q=Q, R=R, model_var=model_var,
                integral_var=integral_var, px=px, xpx=xpx, pxpx=pxpx, kxpx=kxpx, V=V)


def bs_variances(par, x, mi, jitter=1e-8):
    """BayesSardModel.exp_model_variance / integral_variance (bq/bqmod.py:995-1050): the general-branch formulas for any
    point set, and V' iK V inverted WITHOUT jitter - not the numbers bq_weights() returns beside the weights."""
    par = np.atleast_2d(np.asarray(par, dtype=float))
    alpha = par[0, 0]
    iK = rbf_inv(par, x, jitter, scaling=False)
    q, Q = rbf_q(par, x), rbf_Q(par, par, x)
    V = vandermonde(mi, x)
    px, pxpx, kxpx = poly_px(mi), poly_pxpx(mi), poly_kxpx(par, mi, x)
    Z = V.T.dot(iK)
    iG = chol_inverse(Z.dot(V))
    B = Z.dot(Q).dot(Z.T) + pxpx - Z.dot(kxpx) - kxpx.T.dot(Z.T)
    b = Z.dot(q) - px
    return (alpha ** 2 * (1 - np.trace(Q.dot(iK)) + np.trace(B.dot(iG))),
            rbf_kbar(par) - q.dot(iK).dot(q) + b.dot(iG).dot(b))


# --------------------------------------------------------------------------------------------------------------
# integrands (a4): closed-form functions of ssmod.py, additive-noise evaluation (zero noise) unless noted
# --------------------------------------------------------------------------------------------------------------

F_UNGM_DYN, F_UNGM_MEAS, F_UNGMNA_DYN, F_UNGMNA_MEAS = 1, 2, 3, 4
F_PENDULUM_DYN, F_PENDULUM_MEAS, F_REENTRY1D_DYN, F_RANGE_MEAS = 5, 6, 7, 8
F_REENTRY2D_DYN, F_RADAR2D_MEAS, F_CT_DYN, F_BEARING_MEAS = 9, 10, 11, 12
F_CTRS_DYN, F_CV_DYN, F_REENTRY2D_BIAS_DYN, F_SMOOTH10D_DYN = 13, 14, 15, 16


def integrand(fid, x, t, p=()):
    """Evaluate integrand `fid` at one input column x; t is the time index, p the model constants.
    Formulas follow ssmod.py: UNGM :268-269,:1060-1061 (NA :299-300,:1085-1086); pendulum :357-358,:1114-1115;
    reentry-1D :424-427; range :1147-1149; reentry-2D :530-564; radar :1227-1252; CT :675-690; bearing :1189-1195;
    CTRS :755-774; CV :839-846.  F_REENTRY2D_BIAS_DYN is this build's synthetic 6-D case:
    reentry-2D on states 0..4 plus a pass-through sixth state."""
    if fid == F_UNGM_DYN:
        return np.array([0.5 * x[0] + 25 * (x[0] / (1 + x[0] ** 2)) + 8 * np.cos(1.2 * t)])
    if fid == F_UNGM_MEAS:
        return np.array([0.05 * x[0] ** 2])
    if fid == F_UNGMNA_DYN:   # input [x, q]
        return np.array([0.5 * x[0] + 25 * (x[0] / (1 + x[0] ** 2)) + 8 * x[1] * np.cos(1.2 * t)])
    if fid == F_UNGMNA_MEAS:  # input [x, r]
        return np.array([0.05 * x[1] * x[0] ** 2])
    if fid == F_PENDULUM_DYN:
        dt, g = p[0], 9.81
        return np.array([x[0] + x[1] * dt, x[1] - g * dt * np.sin(x[0])])
    if fid == F_PENDULUM_MEAS:
        return np.array([np.sin(x[0])])
    if fid == F_REENTRY1D_DYN:
        dt, gam = p[0], 1 / 6.096
        return np.array([x[0] - dt * x[1], x[1] - dt * np.exp(-gam * x[0]) * x[1] ** 2 * x[2], x[2]])
    if fid == F_RANGE_MEAS:
        sx, sy = 30.0, 30.0
        return np.array([np.sqrt(sx ** 2 + (x[0] - sy) ** 2)])
    if fid in (F_REENTRY2D_DYN, F_REENTRY2D_BIAS_DYN):
        dt = p[0]
        r0, h0, gm0, b0 = 6374.0, 13.406, 3.9860e5, -0.59783
        b = b0 * np.exp(x[4])
        rr = np.sqrt(x[0] ** 2 + x[1] ** 2)
        vv = np.sqrt(x[2] ** 2 + x[3] ** 2)
        dr = b * np.exp((r0 - rr) / h0) * vv
        gr = -gm0 / rr ** 3
        out = [x[0] + dt * x[2], x[1] + dt * x[3], x[2] + dt * (dr * x[2] + gr * x[0]),
               x[3] + dt * (dr * x[3] + gr * x[1]), x[4]]
        if fid == F_REENTRY2D_BIAS_DYN:
            out.append(x[5])
        return np.array(out)
    if fid == F_RADAR2D_MEAS:
        lx, ly = (p[0], p[1]) if len(p) >= 2 else (0.0, 0.0)
        return np.array([np.sqrt((x[0] - lx) ** 2 + (x[1] - ly) ** 2), np.arctan2(x[1] - ly, x[0] - lx)])
    if fid == F_CT_DYN:
        dt = p[0]
        om = x[4]
        a, b = np.sin(om * dt), np.cos(om * dt)
        c, d = np.sin(om * dt) / om, (1 - np.cos(om * dt)) / om
        return np.array([x[0] + c * x[1] - d * x[3], b * x[1] - a * x[3], d * x[1] + x[2] + c * x[3],
                         a * x[1] + b * x[3], x[4]])
    if fid == F_BEARING_MEAS:
        sp = np.asarray(p, dtype=float).reshape(-1, 2)
        return np.arctan2(x[1] - sp[:, 1], x[0] - sp[:, 0])
    if fid == F_CTRS_DYN:     # input [x(5), q(2)], non-additive
        dt = p[0]
        s, q = x[:5], x[5:7]
        if s[4] == 0:
            f = np.array([dt * s[2] * np.cos(s[3]), dt * s[2] * np.sin(s[3]), dt * q[0],
                          dt * s[3] + 0.5 * dt ** 2 * q[1], dt * q[1]])
        else:
            c = s[2] / s[4]
            f = np.array([c * (np.sin(s[3] + s[4] * dt) - np.sin(s[3])) + 0.5 * dt ** 2 * np.cos(s[3]) * q[0],
                          c * (-np.cos(s[3] + s[4] * dt) + np.cos(s[3])) + 0.5 * dt ** 2 * np.sin(s[3]) * q[0],
                          dt * q[0], dt * s[3] + 0.5 * dt ** 2 * q[1], dt * q[1]])
        return s + f
    if fid == F_CV_DYN:
        dt = p[0]
        return np.array([x[0] + dt * x[1], x[1], x[2] + dt * x[3], x[3]])
    if fid == F_SMOOTH10D_DYN:    # this build's synthetic 10-D case (Bayes-Sard quadrature at D = 10)
        return np.concatenate((np.sin(x[:5]) + x[5:10] ** 2, x[5:10] * np.cos(x[:5])))
    raise ValueError(fid)


def eval_columns(fid, x, t, p=(), state_index=None):
    """fx[:, n] = f(x[:, n]) for all sigma points (bq/bqmtran.py:132-156); optional measurement sub-state selection
    (ssmod.py:990-991)."""
    if state_index is not None:
        x = x[np.asarray(state_index)]
    return np.stack([integrand(fid, x[:, n], t, p) for n in range(x.shape[1])], axis=1)


# --------------------------------------------------------------------------------------------------------------
# moment transforms (a1-a8)
# --------------------------------------------------------------------------------------------------------------


def moments_bq(fx, chol, wm, Wc, Wcc, emv_diag):
    """mean = fx wm; cov = fx Wc fx' - mean mean' + diag(emv); ccov = fx Wcc' L'.
    bq/bqmtran.py:158-223 (uncentred covariance, :199)."""
    mean_f = fx.dot(wm)
    cov_f = fx.dot(Wc).dot(fx.T) - np.outer(mean_f, mean_f) + np.diag(np.atleast_1d(emv_diag) * np.ones(fx.shape[0]))
    cov_fx = fx.dot(Wcc.T).dot(chol.T)
    return mean_f, cov_f, cov_fx


def tp_emv_diag(fx, iK, model_var, nu):
    """Diagonal of the data-dependent TP expected model variance (nu - 2 + fx iK fx')/(nu - 2 + N) * model_var.
    bq/bqmtran.py:394-415 with bq/bqmod.py:1132-1160 (`* I_out` keeps the diagonal only)."""
    n = fx.shape[1]
    s = np.einsum('en,nm,em->e', fx, iK, fx)
    return (nu - 2 + s) / (nu - 2 + n) * model_var


def moments_sigma(fx, x, mean, wm, wc_diag):
    """Classical centred form.  mtran.py:141-149."""
    mean_f = fx.dot(wm)
    dfx = fx - mean_f[:, None]
    cov_f = (dfx * wc_diag).dot(dfx.T)
    cov_fx = (dfx * wc_diag).dot((x - mean[:, None]).T)
    return mean_f, cov_f, cov_fx


def apply_bq(fid, mean, cov, t, pts, w, p=(), state_index=None, tp_nu=None):
    """One BQ moment transform.  bq/bqmtran.py:60-109.  `w` is a dict as returned by gp_weights / bs_weights."""
    chol = np.linalg.cholesky(cov)
    x = mean[:, None] + chol.dot(pts)
    fx = eval_columns(fid, x, t, p, state_index)
    mv = w['model_var']
    emv = np.ones(fx.shape[0]) * (np.diag(mv) if np.ndim(mv) == 2 else mv)
    if tp_nu is not None:
        emv = tp_emv_diag(fx, w['iK'], emv, tp_nu)
    return moments_bq(fx, chol, w['wm'], w['Wc'], w['Wcc'], emv)


def apply_sigma(fid, mean, cov, t, pts, wm, wc_diag, p=(), state_index=None):
    """One classical sigma-point transform.  mtran.py:105-149."""
    chol = np.linalg.cholesky(cov)
    x = mean[:, None] + chol.dot(pts)
    fx = eval_columns(fid, x, t, p, state_index)
    return moments_sigma(fx, x, mean, wm, wc_diag)


def jacobian(fid, x, t, p=()):
    """Jacobian of integrand `fid` w.r.t. its own inputs at column x, for the models whose dyn_fcn_dx / meas_fcn_dx the
    reference implements (every other one returns None there): ssmod.py:271-272 (UNGM), :305-306 (UNGM, noise as input),
    :363-365 (pendulum), :848-852 (constant velocity - the TRANSPOSE of the transition matrix, as written there),
    :1063-1064, :1088-1089, :1117-1118 (their measurement functions)."""
    if fid == F_UNGM_DYN:
        return np.array([[0.5 + 25 * (1 - x[0] ** 2) / (1 + x[0] ** 2) ** 2]])
    if fid == F_UNGMNA_DYN:
        return np.array([[0.5 + 25 * (1 - x[0] ** 2) / (1 + x[0] ** 2) ** 2, 8 * np.cos(1.2 * t)]])
    if fid == F_PENDULUM_DYN:
        return np.array([[1.0, p[0]], [-9.81 * p[0] * np.cos(x[0]), 1.0]])
    if fid == F_CV_DYN:
        dt = p[0]
        return np.array([[1, dt, 0, 0], [0, 1, 0, 0], [0, 0, 1, dt], [0, 0, 0, 1]], dtype=float).T
    if fid == F_UNGM_MEAS:
        return np.array([[0.1 * x[0]]])
    if fid == F_UNGMNA_MEAS:
        return np.array([[0.1 * x[1] * x[0], 0.05 * x[0] ** 2]])
    if fid == F_PENDULUM_MEAS:
        return np.array([[np.cos(x[0])]])
    return None


def apply_linear(fid, mean, cov, t, p=(), state_index=None):
    """Linearisation transform, mtran.py:49-59: mean_f = f(mean), cov_fx = J cov, cov_f = cov_fx J'.  The Jacobian is placed
    into the columns of the full input as MeasurementModel.meas_eval does (ssmod.py:985-1009): through the state index, or -
    without one - by `out[:, None] = jac`, which numpy BROADCASTS when jac has one column and the input more (the pendulum's
    measurement on its 2-D state gets cos(x0) in both columns); kept."""
    D = mean.shape[0]
    xs = mean if state_index is None else mean[np.asarray(state_index)]
    mean_f = integrand(fid, xs, t, p)
    js = jacobian(fid, xs, t, p)
    if js is None:
        raise ValueError('integrand {} has no Jacobian'.format(fid))
    J = np.zeros((mean_f.shape[0], D))
    if state_index is not None:
        J[:, np.asarray(state_index)[:js.shape[1]]] = js
    elif js.shape[1] == D:
        J[:] = js
    elif js.shape[1] == 1:
        J[:] = js                      # broadcast over the columns
    else:
        raise ValueError('a Jacobian of 1 < columns < D without a state index cannot be placed')
    cov_fx = J.dot(cov)
    return mean_f, cov_fx.dot(J.T), cov_fx


# --------------------------------------------------------------------------------------------------------------
# filter recursions around the path (callers)
# --------------------------------------------------------------------------------------------------------------


def kalman_update(m_pr, P_pr, y_mean, P_y, P_yx, y):
    """Gaussian measurement update.  ssinf.py:297-323 (covariance left unsymmetrised, :323)."""
    gain = cho_solve(cho_factor(P_y), P_yx).T
    return m_pr + gain.dot(y - y_mean), P_pr - gain.dot(P_y).dot(gain.T)


def gaussian_filter(y, m0, P0, Qn, Rn, G, tf_dyn, tf_obs):
    """Forward pass of an additive-noise Gaussian sigma-point / BQ filter.  ssinf.py:66-118, 254-323.
    y: (dim_y, T).  tf_dyn / tf_obs: callables (mean, cov, t) -> (mean_f, cov_f, cov_fx).
    Both transforms of step k use time index k - 1 (ssinf.py:104, 276-288).
    Returns filtered means (D, T), covariances (D, D, T) and the predictive moments."""
    dim, steps = m0.shape[0], y.shape[1]
    fm, fP = np.zeros((dim, steps)), np.zeros((dim, dim, steps))
    pm, pP, pC = np.zeros((dim, steps)), np.zeros((dim, dim, steps)), np.zeros((dim, dim, steps))
    m, P = m0.copy(), P0.copy()
    GQG = G.dot(Qn).dot(G.T)
    for k in range(1, steps + 1):
        m_pr, P_pr, C_xx = tf_dyn(m, P, k - 1)
        P_pr = P_pr + GQG
        y_mean, P_y, P_yx = tf_obs(m_pr, P_pr, k - 1)
        P_y = P_y + Rn
        m, P = kalman_update(m_pr, P_pr, y_mean, P_y, P_yx, y[:, k - 1])
        fm[:, k - 1], fP[..., k - 1] = m, P
        pm[:, k - 1], pP[..., k - 1], pC[..., k - 1] = m_pr, P_pr, C_xx
    return fm, fP, pm, pP, pC


def gaussian_filter_aug(y, m0, P0, q_mean, Qn, r_mean, Rn, G, tf_dyn, tf_obs, dyn_additive, obs_additive):
    """Forward pass of a Gaussian filter whose models may take the noise as an argument.  ssinf.py:254-323:
    non-additive model -> the transform sees [mean; noise_mean], blockdiag(cov, noise_cov) (:271-272, :282-283) and
    the cross-covariance is cut to the first dim_state columns (:294-295); additive model -> noise covariance added
    after the transform (:278-279, :290-291).  Raises LinAlgError where the reference does."""
    dim, steps = m0.shape[0], y.shape[1]
    fm, fP = np.zeros((dim, steps)), np.zeros((dim, dim, steps))
    m, P = m0.copy(), P0.copy()

    def aug(mean, cov, nm, nc):
        da = mean.shape[0] + nm.shape[0]
        ca = np.zeros((da, da))
        ca[:mean.shape[0], :mean.shape[0]] = cov
        ca[mean.shape[0]:, mean.shape[0]:] = nc
        return np.concatenate((mean, nm)), ca

    for k in range(steps):
        if dyn_additive:
            m_pr, P_pr, _ = tf_dyn(m, P, k)
            P_pr = P_pr + G.dot(Qn).dot(G.T)
        else:
            m_pr, P_pr, _ = tf_dyn(*aug(m, P, q_mean, Qn), k)
        if obs_additive:
            y_mean, P_y, P_yx = tf_obs(m_pr, P_pr, k)
            P_y = P_y + Rn
        else:
            y_mean, P_y, P_yx = tf_obs(*aug(m_pr, P_pr, r_mean, Rn), k)
        P_yx = P_yx[:, :dim]
        m, P = kalman_update(m_pr, P_pr, y_mean, P_y, P_yx, y[:, k])
        fm[:, k], fP[..., k] = m, P
    return fm, fP


def student_scale_sequence(steps, dim_y, x0_dof, q_dof, r_dof, dof=4.0, fixed_dof=True):
    """(dof_pr - 2) / dof_pr used by each time update of a Studentian filter; the filtered dof grows by dim_y per
    measurement update (ssinf.py:652-660, 735-736), so the sequence depends on the degrees of freedom only."""
    out = np.zeros(steps)
    dof_fi = x0_dof
    for k in range(steps):
        if fixed_dof:
            dof_pr = min(dof_fi, q_dof, r_dof)
            out[k] = (dof_pr - 2) / dof_pr
        else:
            out[k] = (dof - 2) / dof
        dof_fi += dim_y
    return out


def student_filter(y, m0, S0_scale, q_scale, r_scale, G, tf_dyn, tf_obs, scale_seq, dof=4.0):
    """Forward pass of an additive-noise Studentian filter.  ssinf.py:634-736.
    S0_scale / q_scale / r_scale are the SCALE matrices the reference's StudentRV.get_stats() returns (it stores them in
    variables called *_cov); the filter multiplies them by (dof - 2) / dof (ssinf.py:617-621).
    Returns the filtered means (D, T) and the reference's x_cov_fi (D, D, T)."""
    dim, steps = m0.shape[0], y.shape[1]
    c = (dof - 2) / dof
    smat, q_smat, r_smat = c * S0_scale, c * q_scale, c * r_scale
    GqG = G.dot(q_smat).dot(G.T)
    m = m0.copy()
    fm, fP = np.zeros((dim, steps)), np.zeros((dim, dim, steps))
    for k in range(steps):
        sc = scale_seq[k]
        m_pr, P_pr, _ = tf_dyn(m, smat, k)
        S_pr = sc * P_pr + GqG
        y_mean, P_y, P_yx = tf_obs(m_pr, S_pr, k)
        S_y = sc * P_y + r_smat
        S_yx = sc * P_yx
        gain = cho_solve(cho_factor(S_y), S_yx).T
        dy = y[:, k] - y_mean
        m = m_pr + gain.dot(dy)
        P = S_pr - gain.dot(S_y).dot(gain.T)
        delta = np.linalg.solve(np.linalg.cholesky(S_y), dy)
        smat = (dof + delta.dot(delta)) / (dof + y.shape[0]) * P
        fm[:, k], fP[..., k] = m, P
    return fm, fP


def rts_smoother(fm, fP, pm, pP, pC):
    """Backward pass with the reference's indexing (ssinf.py:120-147, 325-344): arrays hold steps
    1..T; the loop `for k in range(N - 2, 0, -1)` starts from the last filtered estimate, pairs it with the predictive
    moments of index k + 1 and leaves the last two smoothed steps equal to the filtered ones."""
    T = fm.shape[1]
    sm, sP = fm.copy(), fP.copy()
    ms, Ps = fm[:, T - 1], fP[..., T - 1]
    for k in range(T - 2, 0, -1):
        gain = cho_solve(cho_factor(pP[..., k]), pC[..., k]).T
        ms = fm[:, k - 1] + gain.dot(ms - pm[:, k])
        Ps = fP[..., k - 1] + gain.dot(Ps - pP[..., k]).dot(gain.T)
        sm[:, k - 1], sP[..., k - 1] = ms, Ps
    return sm, sP


# --------------------------------------------------------------------------------------------------------------
# performance metrics (utils.py:18-148) and their Monte-Carlo aggregation (research/tpq/tpq_base.py:154-172)
# --------------------------------------------------------------------------------------------------------------

def squared_error(x, m):
    """utils.py:18-38."""
    return (x - m) ** 2


def mse_matrix(x, m):
    """Sample mean-square-error matrix of one time step.  utils.py:41-64.  x, m: (D, M)."""
    dx = x - m
    return dx.dot(dx.T) / m.shape[1]


def neg_log_likelihood(x, m, P):
    """utils.py:123-148 (sign * logdet of slogdet, explicit inverse)."""
    dx = x - m
    sign, logdet = np.linalg.slogdet(P)
    return 0.5 * (sign * logdet + dx.dot(np.linalg.inv(P)).dot(dx) + x.shape[0] * np.log(2 * np.pi))


def mat_sqrt(a):
    """Cholesky factor, or u sqrt(s) from an SVD when `a` is not positive definite.  utils.py:412-433."""
    try:
        return np.linalg.cholesky(a)
    except np.linalg.LinAlgError:
        u, sv, _ = np.linalg.svd(a)
        return u.dot(np.diag(np.sqrt(sv)))


def log_cred_ratio(x, m, P, mse):
    """utils.py:66-120."""
    dx = x - m
    a = np.linalg.solve(mat_sqrt(P), dx)
    b = np.linalg.solve(mat_sqrt(mse), dx)
    return 10 * (np.log10(a.dot(a)) - np.log10(b.dot(b)))


def error_sums(x, fm, fP, ok=None):
    """Per-time-step SUMS over the trajectories with ok[b] (the quantities research/tpq/tpq_base.py:154-160 averages):
    x, fm (D, T, B); fP (D, D, T, B).  Returns dict se (T, D), rmse (T,), nll (T,), mse (T, D, D), n_ok (T,), n_pd (T,)
    - the layout of ssmq_error_sums_dev.  The nll term is the reference's formula for ANY nonsingular P (inv + slogdet,
    utils.py:143-148); n_pd counts the entries that entered it (all but those with a singular P)."""
    D, T, B = fm.shape
    ok = np.ones(B, dtype=bool) if ok is None else np.asarray(ok, dtype=bool)
    se, rmse, nll, mse = np.zeros((T, D)), np.zeros(T), np.zeros(T), np.zeros((T, D, D))
    n_ok, n_pd = np.zeros(T), np.zeros(T)
    for k in range(T):
        for b in np.flatnonzero(ok):
            dx = x[:, k, b] - fm[:, k, b]
            se[k] += squared_error(x[:, k, b], fm[:, k, b])
            rmse[k] += np.sqrt(dx.dot(dx))
            mse[k] += np.outer(dx, dx)
            n_ok[k] += 1
            try:
                term = neg_log_likelihood(x[:, k, b], fm[:, k, b], fP[..., k, b])
            except np.linalg.LinAlgError:
                continue
            nll[k] += term
            n_pd[k] += 1
    return dict(se=se, rmse=rmse, nll=nll, mse=mse, n_ok=n_ok, n_pd=n_pd)


def lcr_sums(x, fm, fP, mse_global, ok=None):
    """Per-time-step sums of the log credibility ratio against the given (T, D, D) MSE matrices, over trajectories with
    ok[b] (layout of ssmq_lcr_sums_dev): dict lcr (T,), n (T,).  A P that is not positive definite goes through mat_sqrt's
    SVD branch, as in the reference (utils.py:426-432)."""
    D, T, B = fm.shape
    ok = np.ones(B, dtype=bool) if ok is None else np.asarray(ok, dtype=bool)
    lcr, n = np.zeros(T), np.zeros(T)
    for k in range(T):
        for b in np.flatnonzero(ok):
            lcr[k] += log_cred_ratio(x[:, k, b], fm[:, k, b], fP[..., k, b], mse_global[k])
            n[k] += 1
    return dict(lcr=lcr, n=n)


# --------------------------------------------------------------------------------------------------------------
# marginalised GP-quadrature filter: one theta-conditioned step (ssinf.py:1117-1198)
# --------------------------------------------------------------------------------------------------------------

def gauss_logpdf(y, mean, cov):
    """log N(y | mean, cov) (scipy.stats.multivariate_normal.logpdf at ssinf.py:1198), Cholesky route."""
    L = np.linalg.cholesky(cov)
    v = np.linalg.solve(L, y - mean)
    return -0.5 * (v.dot(v) + 2 * np.log(np.diag(L)).sum() + y.shape[0] * np.log(2 * np.pi))


def marginal_theta_step(par_dyn, par_obs, m, P, y, t, fid_dyn, fid_obs, pts_dyn, pts_obs, GQG, R, p_dyn=(), p_obs=(),
                        emv_broadcast=False):
    """State posterior N(x_k | y_1:k, theta) and log N(y_k | y_1:k-1, theta) for ONE set of kernel parameters
    (`_state_posterior_moments` ssinf.py:1117-1151, `_param_log_likelihood` :1153-1198): both transforms are
    re-weighted with the given parameters (bq/bqmtran.py:93-95), time index `t` for both.
    emv_broadcast: the marginalised filter builds its transforms with dim_out = 1 (ssinf.py:1290-1291), so
    model_var * I_out is a 1 x 1 array broadcast over the whole covariance."""
    def tf(fid, mean, cov, pts, par, p):
        w = gp_weights(np.atleast_2d(par), pts)
        mf, cf, cfx = apply_bq(fid, mean, cov, t, pts, dict(w, model_var=0.0), p)
        E = mf.shape[0]
        return mf, cf + (w['model_var'] * (np.ones((E, E)) if emv_broadcast else np.eye(E))), cfx
    m_pr, P_pr, _ = tf(fid_dyn, m, P, pts_dyn, par_dyn, p_dyn)
    P_pr = P_pr + GQG
    y_mean, P_y, P_yx = tf(fid_obs, m_pr, P_pr, pts_obs, par_obs, p_obs)
    P_y = P_y + R
    mean, cov = kalman_update(m_pr, P_pr, y_mean, P_y, P_yx, y)
    return mean, cov, gauss_logpdf(y, y_mean, P_y)


# --------------------------------------------------------------------------------------------------------------
# synthetic trajectories (ssmod.py:168-199, 1011-1039) with the build's counter-based generator
# --------------------------------------------------------------------------------------------------------------

def philox4x32_10(ctr, key):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11).  ctr: four and
    key: two arrays (or ints) of 32-bit words; returns four uint64 arrays holding 32-bit words.  Pinned by the Random123
    known-answer vectors in tests/test_oracle_golden.py."""
    c = [np.asarray(v, dtype=np.uint64) & np.uint64(0xffffffff) for v in ctr]
    k = [np.asarray(v, dtype=np.uint64) & np.uint64(0xffffffff) for v in key]
    m32 = np.uint64(0xffffffff)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k[0], p1 & m32, (p0 >> np.uint64(32)) ^ c[3] ^ k[1], p0 & m32]
        k = [(k[0] + np.uint64(0x9E3779B9)) & m32, (k[1] + np.uint64(0xBB67AE85)) & m32]
    return c


def normal_pair(seed, traj, step, tag):
    """Two standard normals per (global trajectory index, time step, tag): Box-Muller on two 53-bit uniforms built from
    one Philox block, exactly as ssmq_simulate.hip does."""
    traj = np.asarray(traj, dtype=np.uint64)
    c = philox4x32_10([traj, traj >> np.uint64(32), np.full(traj.shape, step, dtype=np.uint64),
                       np.full(traj.shape, tag, dtype=np.uint64)], [seed & 0xffffffff, (seed >> 32) & 0xffffffff])
    u1 = ((((c[0] >> np.uint64(5)) << np.uint64(26)) | (c[1] >> np.uint64(6))).astype(np.float64) + 0.5) / 2.0 ** 53
    u2 = ((((c[2] >> np.uint64(5)) << np.uint64(26)) | (c[3] >> np.uint64(6))).astype(np.float64) + 0.5) / 2.0 ** 53
    r = np.sqrt(-2.0 * np.log(u1))
    return r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)


def gauss_vectors(seed, traj, step, purpose, mean, chol):
    """mean + chol z for every trajectory in `traj`: (n, B)."""
    n = mean.shape[0]
    z = np.zeros((n, np.size(traj)))
    for j in range(0, n, 2):
        z0, z1 = normal_pair(seed, traj, step, (purpose << 16) | (j >> 1))
        z[j] = z0
        if j + 1 < n:
            z[j + 1] = z1
    return mean[:, None] + chol.dot(z)


def simulate(fid_dyn, fid_obs, steps, B, x0_mean, x0_cov, q_mean, q_cov, r_mean, r_cov, G=None, p_dyn=(), p_obs=(),
             dyn_additive=True, obs_additive=True, state_index=None, seed=0, traj_offset=0):
    """x (D, steps, B), y (Y, steps, B):  x[0] ~ N(x0);  x[k] = dyn_fcn(x[k-1], q[k-1], k-1) (ssmod.py:196-198);
    y[k] = meas_fcn(x[k], r[k], k+1) (ssmod.py:1036-1038).  Additive models: f(x) + G q / h(x) + r."""
    D, dq, dr = x0_mean.shape[0], q_mean.shape[0], r_mean.shape[0]
    G = np.eye(D, dq) if G is None else G
    traj = np.arange(B, dtype=np.uint64) + np.uint64(traj_offset)
    L0, Lq, Lr = np.linalg.cholesky(x0_cov), np.linalg.cholesky(q_cov), np.linalg.cholesky(r_cov)
    x = gauss_vectors(seed, traj, 0, 0, x0_mean, L0)
    xs, ys = [], []
    for k in range(steps):
        xs.append(x)
        r = gauss_vectors(seed, traj, k, 2, r_mean, Lr)
        xa = x if obs_additive else np.vstack((x, r))
        sel = xa if state_index is None else xa[np.asarray(state_index)]
        h = np.stack([integrand(fid_obs, sel[:, b], k + 1, p_obs) for b in range(B)], axis=1)
        ys.append(h + r if obs_additive else h)
        if k + 1 == steps:
            break
        q = gauss_vectors(seed, traj, k, 1, q_mean, Lq)
        xa = x if dyn_additive else np.vstack((x, q))
        f = np.stack([integrand(fid_dyn, xa[:, b], k, p_dyn) for b in range(B)], axis=1)
        x = f + G.dot(q) if dyn_additive else f
    return np.stack(xs, axis=1), np.stack(ys, axis=1)


# --------------------------------------------------------------------------------------------------------------
# the simulators' other random variables and the continuous-time dynamics (restatement of ssmq_simulate.hip's generator;
# the DISTRIBUTIONS are the reference's: utils.py:254-299 gauss_mixture, :349-382 multivariate_t; ssmod.py:201-244)
# --------------------------------------------------------------------------------------------------------------
RV_GAUSS, RV_STUDENT, RV_MIXTURE = 0, 1, 2


def uniform_one(seed, traj, step, tag):
    """One 53-bit uniform in (0, 1) per (global trajectory index, time step, tag): words 0 and 1 of the Philox block."""
    traj = np.asarray(traj, dtype=np.uint64)
    c = philox4x32_10([traj, traj >> np.uint64(32), np.full(traj.shape, step, dtype=np.uint64),
                       np.full(traj.shape, tag, dtype=np.uint64)], [seed & 0xffffffff, (seed >> 32) & 0xffffffff])
    return ((((c[0] >> np.uint64(5)) << np.uint64(26)) | (c[1] >> np.uint64(6))).astype(np.float64) + 0.5) / 2.0 ** 53


def gamma_mt(seed, traj, step, base, shape):
    """Gamma(shape, 1), shape >= 1, by Marsaglia-Tsang: attempt t takes its normal from tag base | (0x100 + t), its uniform
    from base | (0x180 + t); at most 16 attempts."""
    traj = np.asarray(traj, dtype=np.uint64)
    d = shape - 1.0 / 3.0
    c = 1.0 / np.sqrt(9.0 * d)
    v = np.ones(traj.shape)
    done = np.zeros(traj.shape, dtype=bool)
    for t in range(16):
        x, _ = normal_pair(seed, traj, step, base | (0x100 + t))
        u = uniform_one(seed, traj, step, base | (0x180 + t))
        w = 1.0 + c * x
        cand = w * w * w
        with np.errstate(invalid='ignore', divide='ignore'):
            acc = (cand > 0.0) & (np.log(u) < 0.5 * x * x + d - d * cand + d * np.log(np.where(cand > 0, cand, 1.0)))
        fall = np.where(np.abs(cand) > 0.0, np.abs(cand), 1.0)
        v = np.where(done, v, np.where(acc, cand, fall))
        done = done | acc
        if done.all():
            break
    return d * v


def sample_rv(rv, seed, traj, step, purpose):
    """(dim, B) draws of a random variable given as dict(kind, mean (K, n), chol (K, n, n), alpha (K,), dof)."""
    traj = np.asarray(traj, dtype=np.uint64)
    mean, chol = np.atleast_2d(rv['mean']), np.asarray(rv['chol']).reshape(-1, rv['mean'].shape[-1], rv['mean'].shape[-1])
    n, B = mean.shape[1], traj.size
    comp = np.zeros(B, dtype=int)
    if rv['kind'] == RV_MIXTURE:
        u = uniform_one(seed, traj, step, (purpose << 16) | 0x200)
        cum = np.cumsum(rv['alpha'])
        comp = np.minimum((u[:, None] >= cum[None, :]).sum(axis=1), len(cum) - 1)
    z = np.zeros((n, B))
    for j in range(0, n, 2):
        z0, z1 = normal_pair(seed, traj, step, (purpose << 16) | (j >> 1))
        z[j] = z0
        if j + 1 < n:
            z[j + 1] = z1
    scale = np.ones(B)
    if rv['kind'] == RV_STUDENT:
        g = gamma_mt(seed, traj, step, purpose << 16, 0.5 * rv['dof']) * (2.0 / rv['dof'])
        scale = 1.0 / np.sqrt(g)
    return mean[comp].T + np.einsum('bij,jb->ib', chol[comp], z) * scale


def gauss_rv(mean, cov):
    return dict(kind=RV_GAUSS, mean=np.atleast_2d(np.asarray(mean, dtype=float)), chol=np.linalg.cholesky(np.atleast_2d(cov))[None])


def student_rv(mean, scale, dof):
    return dict(kind=RV_STUDENT, mean=np.atleast_2d(np.asarray(mean, dtype=float)), chol=np.linalg.cholesky(np.atleast_2d(scale))[None],
                dof=float(dof))


def mixture_rv(means, covs, alphas):
    a = np.asarray(alphas, dtype=float)
    return dict(kind=RV_MIXTURE, mean=np.stack([np.atleast_1d(m) for m in means]).astype(float),
                chol=np.stack([np.linalg.cholesky(np.atleast_2d(c)) for c in covs]), alpha=a / a.sum())


def integrand_cont(fid, x, q):
    """dx/dt of the models that define dyn_fcn_cont: reentry-1D ssmod.py:429-432, reentry-2D :569-585, CTRS :779-780."""
    if fid == F_REENTRY1D_DYN:
        gam = 1 / 6.096
        return np.array([-x[1] + q[0], -np.exp(-gam * x[0]) * x[1] ** 2 * x[2] + q[1], q[2]])
    if fid == F_REENTRY2D_DYN:
        r0, h0, gm0, b0 = 6374.0, 13.406, 3.9860e5, -0.59783
        b = b0 * np.exp(x[4])
        rr, vv = np.sqrt(x[0] ** 2 + x[1] ** 2), np.sqrt(x[2] ** 2 + x[3] ** 2)
        dr = b * np.exp((r0 - rr) / h0) * vv
        gr = -gm0 / rr ** 3
        return np.array([x[2], x[3], dr * x[2] + gr * x[0] + q[0], dr * x[3] + gr * x[1] + q[1], q[2]])
    if fid == F_CTRS_DYN:
        return np.array([x[2] * np.cos(x[3]), x[2] * np.sin(x[3]), 0 * x[0], x[4], 0 * x[0]])
    raise ValueError(fid)


def simulate_rv(fid_dyn, fid_obs, steps, B, x0, q, r, G=None, p_dyn=(), p_obs=(), dyn_additive=True, obs_additive=True,
                state_index=None, seed=0, traj_offset=0, continuous_dt=None):
    """`simulate` with arbitrary random variables (dicts as sample_rv takes them) and, with continuous_dt, the
    Euler-Maruyama recursion of simulate_continuous (ssmod.py:201-244: the initial state is not among the columns)."""
    D, dq = x0['mean'].shape[1], q['mean'].shape[1]
    G = np.eye(D, dq) if G is None else G
    traj = np.arange(B, dtype=np.uint64) + np.uint64(traj_offset)
    x = sample_rv(x0, seed, traj, 0, 0)
    xs, ys = [], []
    for k in range(steps):
        if continuous_dt is not None:
            dt = continuous_dt
            qk = (np.sqrt(dt) / dt) * sample_rv(q, seed, traj, k, 1)
            x = x + dt * integrand_cont(fid_dyn, x, qk)
        xs.append(x)
        if fid_obs is not None:
            rk = sample_rv(r, seed, traj, k, 2)
            xa = x if obs_additive else np.vstack((x, rk))
            sel = xa if state_index is None else xa[np.asarray(state_index)]
            h = np.stack([integrand(fid_obs, sel[:, b], k + 1, p_obs) for b in range(B)], axis=1)
            ys.append(h + rk if obs_additive else h)
        if k + 1 == steps or continuous_dt is not None:
            continue
        qk = sample_rv(q, seed, traj, k, 1)
        xa = x if dyn_additive else np.vstack((x, qk))
        f = np.stack([integrand(fid_dyn, xa[:, b], k, p_dyn) for b in range(B)], axis=1)
        x = f + G.dot(qk) if dyn_additive else f
    return np.stack(xs, axis=1), (np.stack(ys, axis=1) if ys else None)
