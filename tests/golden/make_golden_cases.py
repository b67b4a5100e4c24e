"""Case tables shared by make_golden.py (which needs the reference) and the tests (which must not)."""
import numpy as np

GP_CASES = [
    # tag, dim, ell, point_str, point_par
    ('d1_l3_ut00', 1, 3.0, 'ut', {'kappa': 0, 'alpha': 1}),
    ('d1_l3_ut', 1, 3.0, 'ut', None),
    ('d1_l1_ut', 1, 1.0, 'ut', None),
    ('d1_l03_ut', 1, 0.3, 'ut', None),
    ('d1_l3_gh5', 1, 3.0, 'gh', {'degree': 5}),
    ('d2_l3_ut', 2, 3.0, 'ut', None),
    ('d2_l3_sr', 2, 3.0, 'sr', None),
    ('d2_l1_gh3', 2, 1.0, 'gh', {'degree': 3}),
    ('d5_l3_ut', 5, 3.0, 'ut', None),
    ('d5_l25_ut', 5, 25.0, 'ut', None),
    ('d6_l3_ut', 6, 3.0, 'ut', None),
    ('d6_l25_ut', 6, 25.0, 'ut', None),
    ('d5_l3_fs5', 5, 3.0, 'fs', {'degree': 5}),
    ('d10_l3_fs5', 10, 3.0, 'fs', {'degree': 5}),
    ('d10_l3_ut', 10, 3.0, 'ut', None),
]


def _ut_mi(d):
    return np.hstack((np.zeros((d, 1)), np.eye(d), 2 * np.eye(d))).astype(int)


BS_CASES = [
    # tag, dim, point_str, point_par, multi-index (None = total degree <= 2), ell
    ('d1_ut', 1, 'ut', None, np.array([[0, 1, 2]]), 1.0),
    ('d1_ut_l3', 1, 'ut', None, np.array([[0, 1, 2]]), 3.0),
    ('d1_gh5', 1, 'gh', {'degree': 5}, np.array([[0, 1, 2, 3, 4]]), 1.0),
    ('d1_gh5_q3', 1, 'gh', {'degree': 5}, np.array([[0, 1, 2]]), 3.0),
    ('d2_ut', 2, 'ut', None, np.array([[0, 1, 0, 2, 0], [0, 0, 1, 0, 2]]), 1.0),
    ('d2_gh3', 2, 'gh', {'degree': 3}, np.array([[0, 1, 0, 1, 2, 0, 1, 2, 2], [0, 0, 1, 1, 0, 2, 2, 1, 2]]), 1.0),
    ('d2_gh3_td2', 2, 'gh', {'degree': 3}, None, 3.0),
    ('d5_ut', 5, 'ut', None, _ut_mi(5), 3.0),
    ('d5_fs5_td2', 5, 'fs', {'degree': 5}, None, 3.0),
    ('d10_ut', 10, 'ut', None, _ut_mi(10), 3.0),
    ('d10_fs5_td2', 10, 'fs', {'degree': 5}, None, 3.0),
]


def gp_par(dim, ell, alpha=1.0, aniso=False):
    ells = ell * (1.0 + 0.1 * np.arange(dim)) if aniso else ell * np.ones(dim)
    return np.concatenate(([alpha], ells))[None, :]
