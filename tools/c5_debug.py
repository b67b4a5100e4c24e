import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd
from ssmtoybox_amd import ssmod as sm
from ssmtoybox_amd.bq.bqmod import n_sum_k
rng = np.random.default_rng(19)
B = 96
means = rng.standard_normal((B, 10))
a = rng.standard_normal((B, 10, 10)) / np.sqrt(10)
covs = np.einsum('bij,bkj->bik', a, a) + 0.1 * np.eye(10)
model = sm.Smooth10DTransition()
mi = np.hstack([n_sum_k(10, k) for k in range(3)])
tf = amd.BayesSardTransform(10, 10, np.array([[1.0] + [3.0] * 10]), mi, 'fs', {'degree': 5})
os.environ.pop('SSMQ_NO_FUSED_COV', None)
mf, cf, cfx = tf.apply_batch(model.dyn_eval, means, covs, 0.0)
os.environ['SSMQ_NO_FUSED_COV'] = '1'
mf2, cf2, cfx2 = tf.apply_batch(model.dyn_eval, means, covs, 0.0)
print('mean diff', np.abs(mf - mf2).max(), 'ccov diff', np.abs(cfx - cfx2).max())
d = np.abs(cf - cf2)
print('cov diff max', d.max(), 'scale', np.abs(cf2).max())
for b in (0, 1, 2, 3, 7, 8):
    print(b, 'bad entries:\n', (d[b] > 1e-9 * np.abs(cf2).max()).astype(int))
