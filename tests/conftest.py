import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run by the driver with -m gpu)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np
    gdir = os.path.join(ROOT, 'tests', 'golden')
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(gdir, name + '.npz')))
        return cache[name]
    return load


def pytest_sessionfinish(session, exitstatus):
    """Dump the recorded (what, measured, bar) triples of the parity tests (tests/_cases.py::within)."""
    import json
    try:
        from tests._cases import STATS
    except Exception:
        return
    out = os.path.join(ROOT, 'gpurun_out')
    if STATS and os.path.isdir(out):
        with open(os.path.join(out, 'parity_stats.json'), 'w') as f:
            json.dump([dict(what=w, measured=v, bar=t) for w, v, t in STATS], f, indent=1)
