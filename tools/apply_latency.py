"""Latency of the drop-in call: one MomentTransform.apply() (B = 1) and small batches through the host-buffer entry point
(`ssmq_apply_batch`), wall clock including the ctypes wrapper.  The reference needs 60-120 us per apply() at these shapes."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd                    # noqa: E402
from ssmtoybox_amd import ssmod                # noqa: E402

cases = []
dyn = ssmod.UNGMTransition(ssmod.GaussRV(1), ssmod.GaussRV(1, cov=np.array([[10.0]])))
cases.append(('ungm gpq D=1 N=3', amd.GaussianProcessTransform(1, 1, np.array([[1.0, 3.0]])), dyn.dyn_eval, 1))
m0 = np.array([6500.4, 349.14, -1.8093, -6.7967, 0.6932])
re = ssmod.ReentryVehicle2DTransition(ssmod.GaussRV(5, m0, np.diag([1e-6, 1e-6, 1e-6, 1e-6, 1.0])), ssmod.GaussRV(3))
cases.append(('reentry gpq D=5 N=11', amd.GaussianProcessTransform(5, 5, np.array([[1.0] + [25.0] * 5])), re.dyn_eval, 5))
cases.append(('reentry ut  D=5 N=11', amd.UnscentedTransform(5), re.dyn_eval, 5))
rng = np.random.default_rng(0)
for name, tf, f, D in cases:
    mean = m0[:D] if D == 5 else np.zeros(1)
    cov = np.diag([1e-6, 1e-6, 1e-6, 1e-6, 1.0])[:D, :D] if D == 5 else np.eye(1)
    for _ in range(10):
        tf.apply(f, mean, cov, np.atleast_1d(1.0))
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n):
        tf.apply(f, mean, cov, np.atleast_1d(1.0))
    dt = (time.perf_counter() - t0) / n
    print('%-22s apply()            %8.1f us per call' % (name, dt * 1e6), flush=True)
    for B in (64, 1024, 4096, 16384, 100000):
        means = np.repeat(mean[None], B, axis=0) + 0.0
        covs = np.repeat(cov[None], B, axis=0) + 0.0
        for _ in range(3):
            tf.apply_batch(f, means, covs, 1.0)
        n = 200 if B <= 4096 else 20
        t0 = time.perf_counter()
        for _ in range(n):
            tf.apply_batch(f, means, covs, 1.0)
        dt = (time.perf_counter() - t0) / n
        print('%-22s apply_batch B=%-6d %8.1f us per call  %.3e transforms/s' % (name, B, dt * 1e6, B / dt), flush=True)


def polar2cartesian(x, pars):
    return x[0] * np.array([np.cos(x[1]), np.sin(x[1])])


mt = amd.BayesSardTransform(2, 2, np.array([[1.0, 1, 1]]), multi_ind=np.array([[0, 1, 0, 2, 0], [0, 0, 1, 0, 2]]), point_str='ut')
mean, cov = np.array([1, np.pi / 2]), np.diag([0.05 ** 2, (np.pi / 10) ** 2])
for _ in range(10):
    mt.apply(polar2cartesian, mean, cov, np.atleast_1d(0))
t0 = time.perf_counter()
for _ in range(1000):
    mt.apply(polar2cartesian, mean, cov, np.atleast_1d(0))
dt = (time.perf_counter() - t0) / 1000
t0 = time.perf_counter()
for _ in range(1000):
    np.apply_along_axis(polar2cartesian, 0, np.zeros((2, 5)), None)
df = (time.perf_counter() - t0) / 1000
print('python callable (polar -> cartesian, BSQ D=2 N=5) apply() %8.1f us per call, of which %.1f us evaluate f in Python'
      % (dt * 1e6, df * 1e6), flush=True)
