"""Where a theta-batched step spends its time: 300 calls at P = param_dim + 1 on the pendulum model (run under
`rocprofv3 --kernel-trace --stats`), plus host-side wall clock split into the library call and the Python around it."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssmtoybox_amd import ssinf, ssmod     # noqa: E402

dyn = ssmod.Pendulum2DTransition(ssmod.GaussRV(2, mean=np.array([1.5, 0.0]), cov=0.01 * np.eye(2)),
                                 ssmod.GaussRV(2, cov=0.01 * np.eye(2)), 0.01)
obs = ssmod.Pendulum2DMeasurement(ssmod.GaussRV(1, cov=np.array([[0.1]])), 2)
alg = ssinf.MarginalizedGaussianProcessKalman(dyn, obs, 'rbf', 'sr')
rng = np.random.default_rng(0)
P = int(os.environ.get('P', alg.param_dim + 1))
theta = 0.1 * rng.standard_normal((P, alg.param_dim))
y = rng.standard_normal(1)
for _ in range(5):
    alg.theta_step(theta, np.zeros(2), np.eye(2), y, 1)
t0 = time.perf_counter()
for _ in range(300):
    alg.theta_step(theta, np.zeros(2), np.eye(2), y, 1)
print('P=%d: %.1f us per call' % (P, (time.perf_counter() - t0) / 300 * 1e6))
