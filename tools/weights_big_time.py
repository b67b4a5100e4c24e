#!/usr/bin/env python3
"""Quadrature weights on point sets beyond the CU-resident route (N > 201): the many-workgroup route (k_wb_*) against the
single-workgroup one (SSMQ_WEIGHTS_ONE_WG=1) - wall clock per call and the difference of every output."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402

amd.set_device(0)
g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'g12_large_weights.npz'))


def build(kind, dim, par, mi, pstr, ppar):
    t0 = time.perf_counter()
    tf = (amd.BayesSardTransform(dim, dim, par, mi, pstr, ppar) if kind == 'bs'
          else amd.GaussianProcessTransform(dim, dim, par, 'rbf', pstr, ppar))
    return tf, time.perf_counter() - t0


for tag, pstr, ppar in (('d6_gh3_l15', 'gh', {'degree': 3}), ('d10_fs7_l3', 'fs', {'degree': 7})):
    pts, par, mi = g[tag + '_pts'], g[tag + '_par'], g[tag + '_mi']
    dim, N = pts.shape
    for kind in ('gp', 'bs'):
        os.environ.pop('SSMQ_WEIGHTS_ONE_WG', None)
        build(kind, dim, par, mi, pstr, ppar)
        new, t_new = build(kind, dim, par, mi, pstr, ppar)
        os.environ['SSMQ_WEIGHTS_ONE_WG'] = '1'
        old, t_old = build(kind, dim, par, mi, pstr, ppar)
        os.environ.pop('SSMQ_WEIGHTS_ONE_WG', None)
        rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())  # noqa: E731
        print('%s %s N=%d: many workgroups %.1f ms, one workgroup %.1f ms; max rel diff wm %.1e Wc %.1e Wcc %.1e iK %.1e mv %.1e iv %.1e'
              % (tag, kind, N, 1e3 * t_new, 1e3 * t_old, rel(new.wm, old.wm), rel(new.Wc, old.Wc), rel(new.Wcc, old.Wcc),
                 rel(new.model.iK, old.model.iK), abs(new.model.model_var - old.model.model_var),
                 abs(new.model.integral_var - old.model.integral_var)), flush=True)
