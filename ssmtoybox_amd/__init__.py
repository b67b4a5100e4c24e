"""
ssmtoybox_amd - MI355X-native sigma-point / Bayesian-quadrature moment transforms.

A drop-in for ONE path of jacobnzw/SSMToybox: `MomentTransform.apply()` and the quadrature-weight construction behind
it (ssmtoybox/mtran.py + ssmtoybox/bq/*), as hand-written HIP kernels (gfx950) reached through a C ABI (include/ssmq.h)
and ctypes.  No PyTorch, no NumPy fallback: without libssmq.so and a GPU the compute calls raise.
"""
from ._lib import SsmqError, device_count, set_device, device_name  # noqa: F401
from .mtran import (MomentTransform, SigmaPointTransform, UnscentedTransform, SphericalRadialTransform,  # noqa: F401
                    GaussHermiteTransform, FullySymmetricStudentTransform, MonteCarloTransform, LinearizationTransform)
from .bq.bqmtran import (BQTransform, GaussianProcessTransform, BayesSardTransform,  # noqa: F401
                         StudentTProcessTransform)

__version__ = '0.1.0'
