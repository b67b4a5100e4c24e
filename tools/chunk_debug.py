#!/usr/bin/env python3
"""Why a strip run differs from the whole pass (if it does): which trajectories, which blocks, from which step."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from benchlib.workloads import FilterBench  # noqa: E402

amd.set_device(0)
wl_name, filt, B, T = os.environ.get('WL', 'ct'), os.environ.get('FILT', 'tpqkf'), int(os.environ.get('B', '70000')), int(os.environ.get('T', '6'))
os.environ['SSMQ_FUSED_CHUNKED'] = '0'
wl = FilterBench(amd, B, T, 5, wl_name, filt)
print('whole pass:', wl.alg.kernel_name(B))
wl.step()
ref = wl.results()
wl.step()
ref2 = wl.results()
print('whole pass twice equal:', all(np.array_equal(a, b, equal_nan=True) for a, b in zip(ref, ref2)))
for mode in os.environ.get('MODES', '1,1,700,1,1000,1').split(','):
    os.environ['SSMQ_FUSED_CHUNKED'] = mode
    wl.d_fm.upload(np.zeros((T, wl.D, wl.ld)))
    wl.step()
    got = wl.results()
    same = [np.array_equal(g, r, equal_nan=True) for g, r in zip(got, ref)]
    msg = 'mode %-5s %s  fm/fP/status equal: %s' % (mode, wl.alg.kernel_name(B).split('<')[0], same)
    if not all(same):
        neq = ~((got[0] == ref[0]) | (np.isnan(got[0]) & np.isnan(ref[0])))          # (D, T, B)
        bad_b = np.flatnonzero(neq.any(axis=(0, 1)))
        first_step = [int(np.flatnonzero(neq[:, :, b].any(axis=0))[0]) for b in bad_b[:2000]]
        blocks = np.unique(bad_b // 64)
        rel = np.nanmax(np.abs(got[0][:, :, bad_b] - ref[0][:, :, bad_b]) / (np.abs(ref[0][:, :, bad_b]) + 1e-300))
        msg += '\n   %d trajectories differ in %d blocks (first blocks %s); first differing step: %s; lanes in first block: %s; max rel diff %.2e; status differs: %d' % (
            bad_b.size, blocks.size, blocks[:12].tolist(), np.bincount(first_step).tolist(), (bad_b[bad_b // 64 == blocks[0]] % 64).tolist()[:16], rel,
            int((got[2] != ref[2]).sum()))
        n_blocks = (B + 63) // 64
        strips = 1024 if mode == '1' else int(mode)
        tot = n_blocks * T
        cut = sorted(set(int(tot * s // strips) // T for s in range(1, strips) if (tot * s // strips) % T))
        msg += '\n   blocks that straddle two strips: %d; differing blocks among them: %d' % (len(cut), len(set(cut) & set(blocks.tolist())))
    print(msg)
