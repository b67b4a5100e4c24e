"""Host / device split of one theta-batched step (pendulum, P = param_dim + 1): the Python wrapper `theta_step` against the
bare C call with prepared arguments."""
import ctypes
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssmtoybox_amd import ssinf, ssmod, _lib     # noqa: E402
from ssmtoybox_amd.mtran import resolve_integrand  # noqa: E402

dyn = ssmod.Pendulum2DTransition(ssmod.GaussRV(2, mean=np.array([1.5, 0.0]), cov=0.01 * np.eye(2)),
                                 ssmod.GaussRV(2, cov=0.01 * np.eye(2)), 0.01)
obs = ssmod.Pendulum2DMeasurement(ssmod.GaussRV(1, cov=np.array([[0.1]])), 2)
alg = ssinf.MarginalizedGaussianProcessKalman(dyn, obs, 'rbf', 'sr')
rng = np.random.default_rng(0)
P = int(os.environ.get('P', alg.param_dim + 1))
theta = 0.1 * rng.standard_normal((P, alg.param_dim))
y = rng.standard_normal(1)
m0, P0 = np.zeros(2), np.eye(2)
for _ in range(5):
    alg.theta_step(theta, m0, P0, y, 1)
n = 2000
t0 = time.perf_counter()
for _ in range(n):
    alg.theta_step(theta, m0, P0, y, 1)
t_py = (time.perf_counter() - t0) / n
lib = _lib.load()
pd, ppd = _lib.as_c(np.exp(theta[:, :alg.param_dyn_dim]))
po, ppo = _lib.as_c(np.exp(theta[:, alg.param_dyn_dim:]))
mean, pm = _lib.as_c(m0)
cov, pc = _lib.as_c(P0)
yy, py = _lib.as_c(y)
f_dyn, e_dyn = resolve_integrand(alg.mod_dyn.dyn_eval)
f_obs, e_obs = resolve_integrand(alg.mod_obs.meas_eval)
h_dyn, h_obs = alg.tf_dyn._handle_for(e_dyn), alg.tf_obs._handle_for(e_obs)
gqg, pg = _lib.as_c(alg.G.dot(alg.q_cov).dot(alg.G.T))
rr, pr = _lib.as_c(alg.r_cov)
om, pom = _lib.out_c((P, 2))
oc, poc = _lib.out_c((P, 2, 2))
ll, pll = _lib.out_c((P,))
st = np.zeros(P, dtype=np.int32)
args = (ctypes.c_void_p(h_dyn), ctypes.byref(f_dyn), ctypes.c_void_p(h_obs), ctypes.byref(f_obs), P, ppd, ppo,
        float(alg.tf_dyn.model.kernel.jitter), pm, pc, 1, py, 1, 1.0, pg, pr, pom, poc, pll, st.ctypes.data_as(_lib.c_int32_p))
t0 = time.perf_counter()
for _ in range(n):
    lib.ssmq_gp_theta_step(*args)
t_c = (time.perf_counter() - t0) / n
print('P=%d: wrapper %.1f us, bare C call %.1f us, Python share %.1f us' % (P, t_py * 1e6, t_c * 1e6, (t_py - t_c) * 1e6))
