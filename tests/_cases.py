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
