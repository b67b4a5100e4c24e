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
        path = os.path.join(out, 'parity_stats.json')
        # a partial run (-k, -x after a failure, one file) never replaces the record of a fuller one: its triples go to
        # parity_stats_partial.json instead (round 5 committed the 40 triples of a partial rerun over the full run's ~1 050)
        try:
            have = len(json.load(open(path)))
        except (OSError, ValueError):
            have = -1
        opt = session.config.option
        partial = bool(getattr(opt, 'keyword', '')) or any('::' in a for a in session.config.args) or exitstatus != 0
        if len(STATS) < have or partial:
            path = os.path.join(out, 'parity_stats_partial.json')
        with open(path, 'w') as f:
            json.dump([dict(what=w, measured=v, bar=t) for w, v, t in STATS], f, indent=1)
