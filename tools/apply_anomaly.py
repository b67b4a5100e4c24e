import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd
from ssmtoybox_amd import ssmod
m0 = np.array([6500.4, 349.14, -1.8093, -6.7967, 0.6932])
re = ssmod.ReentryVehicle2DTransition(ssmod.GaussRV(5, m0, np.diag([1e-6, 1e-6, 1e-6, 1e-6, 1.0])), ssmod.GaussRV(3))
tf = amd.GaussianProcessTransform(5, 5, np.array([[1.0] + [25.0] * 5]))
print(tf.kernel_name(re.dyn_eval))
cov = np.diag([1e-6, 1e-6, 1e-6, 1e-6, 1.0])
for B in (8192, 12000, 16320, 16384, 16448, 20000, 32768, 65536):
    means = np.repeat(m0[None], B, axis=0) + 0.0
    covs = np.repeat(cov[None], B, axis=0) + 0.0
    ts = []
    for _ in range(8):
        t0 = time.perf_counter()
        tf.apply_batch(re.dyn_eval, means, covs, 1.0)
        ts.append((time.perf_counter() - t0) * 1e6)
    print(B, ' '.join('%.0f' % t for t in ts), flush=True)
