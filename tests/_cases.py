"""Shared case tables: golden-fixture model names -> integrand ids / constants of this build (oracle numbering ==
include/ssmq.h numbering)."""
import numpy as np

from oracle import ssmq_oracle as orc

SENSORS = np.vstack((1000 * np.eye(2), -1000 * np.eye(2))).astype(float)

# name -> (fid, params, state_index, dim_in, dim_out)
MODELS = {
    'ungm_dyn': (orc.F_UNGM_DYN, (), None, 1, 1),
    'ungm_meas': (orc.F_UNGM_MEAS, (), None, 1, 1),
    'ungmna_dyn': (orc.F_UNGMNA_DYN, (), None, 2, 1),
    'ungmna_meas': (orc.F_UNGMNA_MEAS, (), None, 2, 1),
    'pend_dyn': (orc.F_PENDULUM_DYN, (0.01,), None, 2, 2),
    'pend_meas': (orc.F_PENDULUM_MEAS, (), None, 2, 1),
    'reentry_dyn': (orc.F_REENTRY2D_DYN, (0.1,), None, 5, 5),
    'radar_meas': (orc.F_RADAR2D_MEAS, (0.0, 0.0), None, 5, 2),
    'ct_dyn': (orc.F_CT_DYN, (0.1,), None, 5, 5),
    'bearing_meas': (orc.F_BEARING_MEAS, tuple(SENSORS.reshape(-1)), (0, 2), 5, 4),
    'cv_dyn': (orc.F_CV_DYN, (0.1,), None, 4, 4),
    'reentry1d_dyn': (orc.F_REENTRY1D_DYN, (0.1,), None, 3, 3),
    'range_meas': (orc.F_RANGE_MEAS, (), None, 3, 1),
    'ctrs_dyn': (orc.F_CTRS_DYN, (0.05,), None, 7, 5),
}

SIGMA_TF = ('ut', 'sr', 'gh', 'fs')
BQ_TF = ('gpq', 'gpqsr', 'tpq', 'tpq1', 'bsq')


def rel_err(a, b):
    """max |a - b| / max |b| (norm-wise relative error; 0/0 -> 0)."""
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    den = np.max(np.abs(b)) if b.size else 0.0
    num = np.max(np.abs(a - b)) if b.size else 0.0
    return num / den if den > 0 else num


def cov_err(P, Pref, mask=None):
    """Entry-scaled covariance error: max |P_ij - Pref_ij| / sqrt(|Pref_ii Pref_jj|), matrix axes first ((D, D, ...)).
    A norm-wise error would be blind to the small blocks of an ill-scaled covariance (reentry: 1e-6-sized position /
    velocity variances next to an O(1) one).  mask (shape of the trailing axes) selects the items compared."""
    P, Pref = np.asarray(P, dtype=float), np.asarray(Pref, dtype=float)
    D = Pref.shape[0]
    d = np.sqrt(np.abs(Pref[np.arange(D), np.arange(D)]))          # (D, ...)
    e = np.abs(P - Pref) / (d[:, None] * d[None, :])
    if mask is not None:
        e = e[:, :, mask]
    return float(np.max(e)) if e.size else 0.0


def mean_err(m, mref, mask=None):
    """Row-scaled mean error: max_i max |m_i - mref_i| / max |mref_i| (state axis first): every state component against
    its own magnitude, not against the largest one."""
    m, mref = np.asarray(m, dtype=float), np.asarray(mref, dtype=float)
    if mask is not None:
        m, mref = m[:, mask], mref[:, mask]
    if mref.size == 0:
        return 0.0
    ax = tuple(range(1, mref.ndim))
    den = np.max(np.abs(mref), axis=ax)
    num = np.max(np.abs(m - mref), axis=ax)
    return float(np.max(np.where(den > 0, num / np.where(den > 0, den, 1.0), num)))


def mean_err_sigma(m, mref, Pref, mask=None):
    """Mean error in standard deviations: max |m_i - mref_i| / sqrt(Pref_ii), layouts (D, ...) / (D, D, ...) - the scale
    on which a difference between two filters matters, and independent of the magnitude of the state."""
    m, mref, Pref = np.asarray(m, dtype=float), np.asarray(mref, dtype=float), np.asarray(Pref, dtype=float)
    D = mref.shape[0]
    e = np.abs(m - mref) / np.sqrt(np.abs(Pref[np.arange(D), np.arange(D)]))
    if mask is not None:
        e = e[:, mask]
    return float(np.max(e)) if e.size else 0.0


# measured value / bar of every recorded comparison of a test session (tests/conftest.py writes them to
# gpurun_out/parity_stats.json so that the tolerances in the tests can be traced to what actually holds)
STATS = []


def within(value, tol, what):
    STATS.append((str(what), float(value), float(tol)))
    return bool(value < tol)


def _recorded():
    """(what -> largest value measured) from the parity statistics committed under profiles/ by earlier rounds."""
    import glob
    import json
    import os
    out = {}
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles')
    for path in sorted(glob.glob(os.path.join(root, 'r0*_parity_stats.json'))):
        try:
            for r in json.load(open(path)):
                out[r['what']] = max(out.get(r['what'], 0.0), float(r['measured']))
        except (OSError, ValueError, KeyError):
            pass
    return out


RECORDED = _recorded()


def capped(bar, what, factor=1e3, floor=1e-13):
    """A condition-number bar (64 cond eps, 8 cond^2 eps) is a worst case and can sit many orders above what the
    kernels deliver; where an earlier round RECORDED the measured value of this very comparison, the bar is at most
    1e3 x that (never below `floor`), so a regression of a few digits fails instead of hiding under the worst case."""
    rec = RECORDED.get(str(what))
    return bar if rec is None else min(bar, max(factor * rec, floor))


# ---------------------------------------------------------------------------------------------------------------
# tolerance of one moment transform against the NumPy reference (north star: 1e-10 relative, fp64)
# ---------------------------------------------------------------------------------------------------------------
RTOL = 1e-10


def moment_scales(mf_ref, cf_ref, cfx_ref, cov_in):
    """Scales the 1e-10 relative bar is taken against.  mean: max |mean_f|.  cov_f: the uncentred BQ covariance
    subtracts mean mean' from fx Wc fx' (bq/bqmtran.py:199), so its rounding error is relative to max|mean_f|^2, not to
    the (possibly much smaller) result - the reference's own result carries that noise.
    cov_fx: max |mean_f| * sqrt(max |cov|)."""
    ms = max(float(np.max(np.abs(mf_ref))), 1e-300)
    cs = max(float(np.max(np.abs(cf_ref))), ms ** 2)
    xs = max(float(np.max(np.abs(cfx_ref))), ms * float(np.sqrt(np.max(np.abs(cov_in)))))
    return ms, cs, xs


def assert_moments_close(got, ref, cov_in, rtol=RTOL, what=''):
    mf, cf, cfx = got
    mf_r, cf_r, cfx_r = ref
    ms, cs, xs = moment_scales(mf_r, cf_r, cfx_r, cov_in)
    e1 = float(np.max(np.abs(mf - mf_r))) / ms
    e2 = float(np.max(np.abs(cf - cf_r))) / cs
    e3 = float(np.max(np.abs(cfx - cfx_r))) / xs
    assert np.all(np.isfinite(mf)) and np.all(np.isfinite(cf)) and np.all(np.isfinite(cfx)), what
    assert e1 <= rtol and e2 <= rtol and e3 <= rtol, (what, e1, e2, e3)
    return max(e1, e2, e3)
