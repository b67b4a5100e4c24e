"""ctypes access to oracle/libssmq_oracle.so (the C restatement of the hot loop).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product package."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
DP = ctypes.POINTER(ctypes.c_double)


class Integrand(ctypes.Structure):
    _fields_ = [('fid', ctypes.c_int), ('npar', ctypes.c_int), ('nidx', ctypes.c_int), ('par', ctypes.c_double * 16),
                ('idx', ctypes.c_int * 16)]

    @classmethod
    def make(cls, fid, par=(), idx=None):
        s = cls()
        idx = [] if idx is None else list(idx)
        s.fid, s.npar, s.nidx = int(fid), len(par), len(idx)
        for i, p in enumerate(par):
            s.par[i] = float(p)
        for i, k in enumerate(idx):
            s.idx[i] = int(k)
        return s


class Transform(ctypes.Structure):
    _fields_ = [('form', ctypes.c_int), ('D', ctypes.c_int), ('E', ctypes.c_int), ('N', ctypes.c_int),
                ('emv_broadcast', ctypes.c_int), ('tp_nu', ctypes.c_double), ('pts', DP), ('wm', DP), ('Wc', DP),
                ('Wcc', DP), ('emv', DP), ('iK', DP), ('f', Integrand)]


FLAGS = 'gcc -O2 -fopenmp (portable build, oracle/Makefile: all)'


def load():
    """The portable build (oracle/Makefile `all`), or the file SSMQ_ORACLE_LIB names (the sanitizer build of
    tools/oracle_asan.sh)."""
    global _lib
    if _lib is None:
        path = os.environ.get('SSMQ_ORACLE_LIB') or os.path.join(_HERE, 'libssmq_oracle.so')
        if not os.path.exists(path):
            subprocess.check_call(['make', '-C', _HERE])
        _lib = ctypes.CDLL(path)
        _lib.orc_max_threads.restype = ctypes.c_int
    return _lib


def use_native():
    """For bench.py's cpu_baseline leg: (re)build the port with -O3 -march=native ON THIS HOST (`make -B native`; a file
    built on another machine may use instructions this one lacks) and switch to it.  Returns the compiler line, or the
    portable build's when the native build is not possible (no compiler on the box)."""
    global _lib, FLAGS
    if os.environ.get('SSMQ_ORACLE_PORTABLE'):      # A/B of the two builds on one host (tools): keep the portable -O2 file
        load()
        return FLAGS
    try:
        subprocess.check_call(['make', '-B', '-C', _HERE, 'native'], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lib = ctypes.CDLL(os.path.join(_HERE, 'libssmq_oracle_native.so'))
        lib.orc_max_threads.restype = ctypes.c_int
        _lib = lib
        FLAGS = open(os.path.join(_HERE, 'libssmq_oracle_native.flags')).read().strip() + ' (built on this host)'
    except (OSError, subprocess.CalledProcessError):
        load()
    return FLAGS


def cpu_model():
    """First `model name` of /proc/cpuinfo (the host the baseline is timed on)."""
    try:
        for line in open('/proc/cpuinfo'):
            if line.lower().startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _p(a):
    return None if a is None else a.ctypes.data_as(DP)


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def make_transform(form, D, E, pts, wm, Wc, Wcc=None, emv=None, emv_broadcast=0, tp_nu=0.0, iK=None, integrand=None):
    """Returns (Transform struct, keep-alive list)."""
    arrs = [_c(pts), _c(wm), _c(Wc), _c(Wcc), _c(emv), _c(iK)]
    t = Transform()
    t.form, t.D, t.E, t.N, t.emv_broadcast, t.tp_nu = form, D, E, arrs[0].shape[1], emv_broadcast, tp_nu
    t.pts, t.wm, t.Wc, t.Wcc, t.emv, t.iK = [_p(a) for a in arrs]
    if integrand is not None:
        t.f = integrand
    t._keep = arrs          # the struct holds raw pointers into these arrays: they live as long as it does
    return t, arrs


def apply_batch(t, mean, cov, time, threads=1):
    lib = load()
    mean, cov = _c(mean), _c(cov)
    B = mean.shape[0]
    time = _c(np.broadcast_to(np.asarray(time, dtype=float).reshape(-1), (B,)) if np.size(time) in (1, B) else time)
    mf, cf, cfx = np.empty((B, t.E)), np.empty((B, t.E, t.E)), np.empty((B, t.E, t.D))
    st = np.zeros(B, dtype=np.int32)
    lib.orc_apply_batch(t.form, t.D, t.E, t.N, ctypes.byref(t.f), ctypes.c_int64(B), _p(mean), _p(cov), _p(time), 1,
                        t.pts, t.wm, t.Wc, t.Wcc, t.emv, t.emv_broadcast, ctypes.c_double(t.tp_nu), t.iK, _p(mf),
                        _p(cf), _p(cfx), st.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), int(threads))
    return mf, cf, cfx, st


def filter_forward(t_dyn, t_obs, y, m0, P0, GQG, R, threads=1):
    """y (B, T, Y) -> fm (B, T, D), fP (B, T, D, D), status (B,)."""
    lib = load()
    y, m0, P0, GQG, R = _c(y), _c(m0), _c(P0), _c(GQG), _c(R)
    B, T, Y = y.shape
    D = t_dyn.D
    fm, fP = np.full((B, T, D), np.nan), np.full((B, T, D, D), np.nan)
    st = np.zeros(B, dtype=np.int32)
    lib.orc_filter_forward(ctypes.byref(t_dyn), ctypes.byref(t_obs), ctypes.c_int64(B), int(T), _p(y), _p(m0), _p(P0),
                           _p(GQG), _p(R), _p(fm), _p(fP), st.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                           int(threads))
    return fm, fP, st


def max_threads():
    return load().orc_max_threads()
