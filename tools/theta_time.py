"""Latency of the theta-batched step of the marginalised GPQ filter (`ssmq_gp_theta_step`): wall clock per
call at a few batch sizes, and a whole forward pass of MarginalizedGaussianProcessKalman on UNGM (T steps)."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssmtoybox_amd import ssinf, ssmod, _lib     # noqa: E402


def model(name):
    if name == 'ungm':
        dyn = ssmod.UNGMTransition(ssmod.GaussRV(1), ssmod.GaussRV(1, cov=np.array([[10.0]])))
        obs = ssmod.UNGMMeasurement(ssmod.GaussRV(1), 1)
    else:
        dyn = ssmod.Pendulum2DTransition(ssmod.GaussRV(2, mean=np.array([1.5, 0.0]), cov=0.01 * np.eye(2)),
                                         ssmod.GaussRV(2, cov=0.01 * np.eye(2)), 0.01)
        obs = ssmod.Pendulum2DMeasurement(ssmod.GaussRV(1, cov=np.array([[0.1]])), 2)
    return dyn, obs


for name in ('ungm', 'pendulum'):
    dyn, obs = model(name)
    alg = ssinf.MarginalizedGaussianProcessKalman(dyn, obs, 'rbf', 'sr')
    D = dyn.dim_state
    rng = np.random.default_rng(0)
    y = rng.standard_normal(obs.dim_out)
    for P in (1, alg.param_dim + 1, 2 * alg.param_dim, 256, 4096, 65536):
        theta = 0.1 * rng.standard_normal((P, alg.param_dim))
        for _ in range(3):
            alg.theta_step(theta, np.zeros(D), np.eye(D), y, 1)
        n = 200 if P <= 4096 else 20
        t0 = time.perf_counter()
        for _ in range(n):
            alg.theta_step(theta, np.zeros(D), np.eye(D), y, 1)
        dt = (time.perf_counter() - t0) / n
        print('%-9s P=%6d  %9.1f us per call  %10.3e theta-steps/s' % (name, P, dt * 1e6, P / dt), flush=True)
    T = int(os.environ.get('T', 30))
    x = dyn.simulate_discrete(T, 1)
    yy = obs.simulate_measurements(x)
    alg.forward_pass(yy[..., 0])          # first pass: library warm-up (arena, code objects)
    alg.reset()
    t0 = time.perf_counter()
    alg.forward_pass(yy[..., 0])
    print('%-9s forward_pass T=%d: %.3f s (%.1f ms per time step)' % (name, T, time.perf_counter() - t0,
                                                                      (time.perf_counter() - t0) / T * 1e3), flush=True)
