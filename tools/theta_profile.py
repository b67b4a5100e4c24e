"""cProfile of a marginalised-filter forward pass (where the wall clock of ssinf.MarginalInference goes)."""
import cProfile
import os
import pstats
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssmtoybox_amd import ssinf, ssmod     # noqa: E402

dyn = ssmod.UNGMTransition(ssmod.GaussRV(1), ssmod.GaussRV(1, cov=np.array([[10.0]])))
obs = ssmod.UNGMMeasurement(ssmod.GaussRV(1), 1)
alg = ssinf.MarginalizedGaussianProcessKalman(dyn, obs, 'rbf', 'sr')
x = dyn.simulate_discrete(30, 1)
y = obs.simulate_measurements(x)
alg.forward_pass(y[..., 0])
alg.reset()
pr = cProfile.Profile()
pr.enable()
alg.forward_pass(y[..., 0])
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
