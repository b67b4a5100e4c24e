"""
ctypes binding of libssmq.so (C ABI: include/ssmq.h) - the only way this package computes anything.

There is no CPU fallback: if the shared library is missing, or no gfx950 device is usable, loading / the first compute
call raises `SsmqError`.  PyTorch is not involved; device memory is owned by the library.
"""
import ctypes
import os

import numpy as np

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int32_p = ctypes.POINTER(ctypes.c_int32)

ABI_VERSION = 102          # include/ssmq.h SSMQ_VERSION this binding's struct layouts and prototypes were written for
SSMQ_MAX_FPAR = 16
SSMQ_MAX_FIDX = 16
FORM_BQ, FORM_SIGMA = 0, 1
EMV_DIAG, EMV_BROADCAST = 0, 1

# integrand ids (include/ssmq.h enum ssmq_integrand_id)
F_UNGM_DYN, F_UNGM_MEAS, F_UNGMNA_DYN, F_UNGMNA_MEAS = 1, 2, 3, 4
F_PENDULUM_DYN, F_PENDULUM_MEAS, F_REENTRY1D_DYN, F_RANGE_MEAS = 5, 6, 7, 8
F_REENTRY2D_DYN, F_RADAR2D_MEAS, F_CT_DYN, F_BEARING_MEAS = 9, 10, 11, 12
F_CTRS_DYN, F_CV_DYN, F_REENTRY2D_BIAS_DYN, F_SMOOTH10D_DYN = 13, 14, 15, 16


class SsmqError(RuntimeError):
    pass


class Integrand(ctypes.Structure):
    """struct ssmq_integrand."""
    _fields_ = [('id', ctypes.c_int32), ('n_par', ctypes.c_int32), ('n_idx', ctypes.c_int32),
                ('reserved', ctypes.c_int32), ('par', ctypes.c_double * SSMQ_MAX_FPAR),
                ('idx', ctypes.c_int32 * SSMQ_MAX_FIDX)]

    @classmethod
    def make(cls, fid, par=(), idx=None):
        par = [float(p) for p in par]
        idx = [] if idx is None else [int(i) for i in idx]
        if len(par) > SSMQ_MAX_FPAR or len(idx) > SSMQ_MAX_FIDX:
            raise ValueError('too many integrand constants / state indices')
        s = cls()
        s.id, s.n_par, s.n_idx = int(fid), len(par), len(idx)
        for i, p in enumerate(par):
            s.par[i] = p
        for i, k in enumerate(idx):
            s.idx[i] = k
        return s


class Rv(ctypes.Structure):
    """struct ssmq_rv: a Gaussian / Student-t / Gaussian-mixture random variable for the device simulator."""
    _fields_ = [('kind', ctypes.c_int32), ('dim', ctypes.c_int32), ('n_comp', ctypes.c_int32), ('reserved', ctypes.c_int32),
                ('dof', ctypes.c_double), ('mean', c_double_p), ('chol', c_double_p), ('alpha', c_double_p)]


RV_GAUSS, RV_STUDENT, RV_MIXTURE = 0, 1, 2


class FilterJob(ctypes.Structure):
    """struct ssmq_filter_job: one filter of ssmq_filter_forward_multi_dev (the arguments of ssmq_filter_forward_dev /
    ssmq_student_filter_forward_dev)."""
    _fields_ = [('h_dyn', ctypes.c_void_p), ('f_dyn', ctypes.POINTER(Integrand)), ('h_obs', ctypes.c_void_p),
                ('f_obs', ctypes.POINTER(Integrand)), ('B', ctypes.c_int64), ('ld', ctypes.c_int64), ('T', ctypes.c_int32),
                ('reserved', ctypes.c_int32), ('d_y', ctypes.c_void_p), ('d_m0', ctypes.c_void_p), ('d_P0', ctypes.c_void_p),
                ('GQG', c_double_p), ('R', c_double_p), ('d_fm', ctypes.c_void_p), ('d_fP', ctypes.c_void_p),
                ('d_status', ctypes.c_void_p), ('scale', c_double_p), ('dof', ctypes.c_double)]

_PROTOTYPES = {
    # name: (restype, argtypes)
    'ssmq_version': (ctypes.c_int, []),
    'ssmq_last_error': (ctypes.c_char_p, []),
    'ssmq_device_count': (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    'ssmq_set_device': (ctypes.c_int, [ctypes.c_int]),
    'ssmq_device_name': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    'ssmq_device_pci_bus_id': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    'ssmq_malloc': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]),
    'ssmq_free': (ctypes.c_int, [ctypes.c_void_p]),
    'ssmq_memcpy_h2d': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    'ssmq_memcpy_d2h': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    'ssmq_memcpy_d2d': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    'ssmq_memset': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]),
    'ssmq_sync': (ctypes.c_int, []),
    'ssmq_aos_to_soa': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                       ctypes.c_int64]),
    'ssmq_soa_to_aos': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                       ctypes.c_int64]),
    'ssmq_event_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p)]),
    'ssmq_event_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'ssmq_event_record': (ctypes.c_int, [ctypes.c_void_p]),
    'ssmq_event_elapsed_ms': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]),
    'ssmq_status_first': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]),
    'ssmq_weights_gp': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, ctypes.c_int,
                                       ctypes.c_double] + [c_double_p] * 9 + [c_int32_p]),
    'ssmq_weights_tp': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, ctypes.c_int,
                                       ctypes.c_double] + [c_double_p] * 9 + [c_int32_p]),
    'ssmq_weights_bs': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, ctypes.c_int,
                                       ctypes.c_double, c_int32_p, ctypes.c_int] + [c_double_p] * 9 + [c_int32_p]),
    'ssmq_variances_bs': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, ctypes.c_int, ctypes.c_double,
                                         c_int32_p, ctypes.c_int, c_double_p, c_double_p, c_int32_p]),
    'ssmq_transform_create': (ctypes.c_void_p, [ctypes.c_int] * 4 + [c_double_p] * 5 + [ctypes.c_int, ctypes.c_double,
                                                                                       c_double_p]),
    'ssmq_transform_create_linear': (ctypes.c_void_p, [ctypes.c_int, ctypes.c_int]),
    'ssmq_transform_update': (ctypes.c_int, [ctypes.c_void_p] + [c_double_p] * 5 + [ctypes.c_int, ctypes.c_double,
                                                                                    c_double_p]),
    'ssmq_transform_destroy': (None, [ctypes.c_void_p]),
    'ssmq_transform_dims': (ctypes.c_int, [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int)] * 3),
    'ssmq_apply_batch': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Integrand), ctypes.c_int64, c_double_p,
                                        c_double_p, c_double_p, ctypes.c_int, c_double_p, c_double_p, c_double_p,
                                        c_int32_p]),
    'ssmq_apply_batch_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Integrand), ctypes.c_int64,
                                            ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p]),
    'ssmq_apply_kernel_name': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Integrand), ctypes.c_char_p,
                                              ctypes.c_int]),
    'ssmq_sigma_points_batch': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, c_double_p, c_double_p, c_double_p,
                                               c_double_p, c_int32_p]),
    'ssmq_apply_fx_batch': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64] + [c_double_p] * 7),
    'ssmq_fxwc_batch_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                           ctypes.c_int64, ctypes.POINTER(ctypes.c_int)]),
    'ssmq_kalman_update_dev': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64] +
                               [ctypes.c_void_p] * 9),
    'ssmq_filter_forward_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Integrand), ctypes.c_void_p,
                                               ctypes.POINTER(Integrand), ctypes.c_int64, ctypes.c_int64,
                                               ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               c_double_p, c_double_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p]),
    'ssmq_filter_forward_multi_dev': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(FilterJob)]),
    'ssmq_filter_forward_piped': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Integrand), ctypes.c_void_p, ctypes.POINTER(Integrand),
                                                 ctypes.c_int64, ctypes.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    'ssmq_pinned_alloc': (ctypes.c_int, [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]),
    'ssmq_pinned_free': (ctypes.c_int, [ctypes.c_void_p]),
    'ssmq_pinned_is_block': (ctypes.c_int, [ctypes.c_void_p]),
    'ssmq_filter_forward_aug_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Integrand), ctypes.c_void_p,
                                                   ctypes.POINTER(Integrand), ctypes.c_int, ctypes.c_int64,
                                                   ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                   ctypes.c_void_p, c_double_p, c_double_p, ctypes.c_int, c_double_p,
                                                   c_double_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                   ctypes.c_void_p]),
    'ssmq_filter_smooth_aug_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Integrand), ctypes.c_void_p,
                                                  ctypes.POINTER(Integrand), ctypes.c_int, ctypes.c_int64,
                                                  ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                  ctypes.c_void_p, c_double_p, c_double_p, ctypes.c_int, c_double_p,
                                                  c_double_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'ssmq_filter_smooth_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Integrand), ctypes.c_void_p,
                                              ctypes.POINTER(Integrand), ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_double_p, c_double_p,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_void_p]),
    'ssmq_rts_backward_dev': (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int] + [ctypes.c_void_p] * 8),
    'ssmq_student_filter_forward_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Integrand), ctypes.c_void_p,
                                                       ctypes.POINTER(Integrand), ctypes.c_int64, ctypes.c_int64,
                                                       ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                       c_double_p, c_double_p, c_double_p, ctypes.c_double,
                                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'ssmq_gp_theta_step': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Integrand), ctypes.c_void_p,
                                          ctypes.POINTER(Integrand), ctypes.c_int64, c_double_p, c_double_p,
                                          ctypes.c_double, c_double_p, c_double_p, ctypes.c_int, c_double_p, ctypes.c_int,
                                          ctypes.c_double, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                          c_int32_p]),
    'ssmq_gp_marginal_laplace_batch': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Integrand), ctypes.c_void_p,
                                                      ctypes.POINTER(Integrand), ctypes.c_int64, ctypes.c_double, c_double_p,
                                                      c_double_p, c_double_p, ctypes.c_double, c_double_p, c_double_p,
                                                      c_double_p, c_double_p, ctypes.c_double, c_double_p, c_double_p,
                                                      c_int32_p, c_int32_p, ctypes.POINTER(ctypes.c_int64)]),
    'ssmq_gp_theta_step_times': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Integrand), ctypes.c_void_p,
                                                ctypes.POINTER(Integrand), ctypes.c_int64, c_double_p, c_double_p,
                                                ctypes.c_double, c_double_p, c_double_p, ctypes.c_int, c_double_p, ctypes.c_int,
                                                c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                                c_int32_p]),
    'ssmq_gp_marginal_filter_batch': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Integrand), ctypes.c_void_p,
                                                     ctypes.POINTER(Integrand), ctypes.c_int64, ctypes.c_int, ctypes.c_double,
                                                     c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                                     c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, ctypes.c_int,
                                                     ctypes.c_double, ctypes.c_double, c_double_p, c_double_p, c_int32_p,
                                                     c_double_p, c_double_p, ctypes.POINTER(ctypes.c_int64)]),
    'ssmq_bfgs_lockstep_host': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double,
                                               c_double_p, c_double_p, c_int32_p, c_int32_p, ctypes.POINTER(ctypes.c_int64)]),
    'ssmq_points_count': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_int]),
    'ssmq_points': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, c_double_p,
                                   c_double_p]),
    'ssmq_simulate_dev': (ctypes.c_int, [ctypes.POINTER(Integrand), ctypes.POINTER(Integrand), ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                         ctypes.c_int64, ctypes.c_int, c_double_p, c_double_p, c_double_p, c_double_p,
                                         c_double_p, c_double_p, c_double_p, ctypes.c_uint64, ctypes.c_uint64,
                                         ctypes.c_void_p, ctypes.c_void_p]),
    'ssmq_simulate_rv_dev': (ctypes.c_int, [ctypes.POINTER(Integrand), ctypes.POINTER(Integrand), ctypes.c_int, ctypes.c_int,
                                            ctypes.POINTER(Rv), ctypes.POINTER(Rv), ctypes.POINTER(Rv), c_double_p,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_double, ctypes.c_uint64, ctypes.c_uint64,
                                            ctypes.c_void_p, ctypes.c_void_p]),
    'ssmq_error_sums_width': (ctypes.c_int, [ctypes.c_int]),
    'ssmq_error_sums_dev': (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_double_p]),
    'ssmq_lcr_sums_dev': (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_double_p, c_double_p]),
    'ssmq_filter_kernel_name': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Integrand), ctypes.c_void_p,
                                               ctypes.POINTER(Integrand), ctypes.c_char_p, ctypes.c_int]),
    'ssmq_filter_kernel_name_batch': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Integrand), ctypes.c_void_p,
                                                     ctypes.POINTER(Integrand), ctypes.c_int64, ctypes.c_char_p, ctypes.c_int]),
    'ssmq_rbf_eval': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, c_double_p,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p]),
    'ssmq_rbf_factor': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_double, c_double_p, c_double_p, c_double_p, c_int32_p]),
    'ssmq_rbf_exp_kxkx': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, c_double_p, ctypes.c_int,
                                         c_double_p]),
    'ssmq_bs_moments': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, c_int32_p, ctypes.c_int,
                                       c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    'ssmq_current_device': (ctypes.c_int, []),
    'ssmq_upload_planes': (ctypes.c_int, [c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                          ctypes.c_void_p]),
    'ssmq_download_planes': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                            c_double_p]),
    'ssmq_comm_unique_id': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    'ssmq_comm_init': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]),
    'ssmq_comm_abandon_init': (ctypes.c_int, []),
    'ssmq_comm_rank': (ctypes.c_int, []),
    'ssmq_comm_world': (ctypes.c_int, []),
    'ssmq_allreduce_sum': (ctypes.c_int, [c_double_p, ctypes.c_int64]),
    'ssmq_allreduce_max': (ctypes.c_int, [c_double_p, ctypes.c_int64]),
    'ssmq_comm_barrier': (ctypes.c_int, []),
    'ssmq_comm_destroy': (ctypes.c_int, []),
}

BFGS_MAXITER, BFGS_PRECISION_LOSS, BFGS_NAN, BFGS_FALLBACK, BFGS_PRIOR_NOT_PD = 1, 2, 3, 100, 101     # include/ssmq.h

EXPORTED_SYMBOLS = tuple(sorted(_PROTOTYPES))

_lib = None


def library_path():
    return os.environ.get('SSMQ_LIBRARY', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libssmq.so'))


def load():
    """Load libssmq.so and declare the prototypes.  Raises SsmqError if the library is absent - never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise SsmqError('libssmq.so not found at {} - build it with __graft_entry__.build() / '
                        'make -C ssmtoybox_amd/csrc; this package has no CPU fallback'.format(path))
    lib = ctypes.CDLL(path)
    lib.ssmq_version.restype = ctypes.c_int
    have = lib.ssmq_version()
    if have != ABI_VERSION:
        # struct ssmq_integrand changed size between 100 and 101 (idx[8] -> idx[16]): a stale library would read past it
        raise SsmqError('{} reports ABI version {}, this binding needs {} - rebuild it (make -C ssmtoybox_amd/csrc)'.format(
            path, have, ABI_VERSION))
    for name, (res, args) in _PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().ssmq_last_error().decode('utf-8', 'replace')


def check(rc, what=''):
    """Negative return codes are errors; non-negative ones are passed through (0 ok, >0 first non-PD item + 1)."""
    if rc < 0:
        raise SsmqError('{} failed (code {}): {}'.format(what or 'libssmq call', rc, last_error()))
    return rc


def device_count():
    n = ctypes.c_int(0)
    rc = load().ssmq_device_count(ctypes.byref(n))
    return n.value if rc == 0 else 0


def set_device(dev):
    if '_scratch_pool' in globals():
        _scratch_pool.flush()         # blocks of the device the library is leaving
    check(load().ssmq_set_device(int(dev)), 'ssmq_set_device')


def device_name():
    buf = ctypes.create_string_buffer(256)
    check(load().ssmq_device_name(buf, 256), 'ssmq_device_name')
    return buf.value.decode()


def device_pci_bus_id():
    buf = ctypes.create_string_buffer(64)
    check(load().ssmq_device_pci_bus_id(buf, 64), 'ssmq_device_pci_bus_id')
    return buf.value.decode()


def sync():
    check(load().ssmq_sync(), 'ssmq_sync')


def as_c(a):
    """C-contiguous fp64 view/copy and its ctypes pointer (keep the array alive while the pointer is used)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(c_double_p)


def out_c(shape):
    a = np.empty(shape, dtype=np.float64)
    return a, a.ctypes.data_as(c_double_p)


class DeviceBuffer:
    """A block of HBM owned through ssmq_malloc / ssmq_free."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = ctypes.c_void_p()
        check(load().ssmq_malloc(ctypes.byref(p), ctypes.c_size_t(max(self.nbytes, 8))), 'ssmq_malloc')
        self.ptr = p.value

    def at(self, byte_offset):
        return ctypes.c_void_p(self.ptr + int(byte_offset))

    def upload(self, host, byte_offset=0):
        host = np.ascontiguousarray(host)
        check(load().ssmq_memcpy_h2d(self.at(byte_offset), host.ctypes.data_as(ctypes.c_void_p),
                                     ctypes.c_size_t(host.nbytes)), 'ssmq_memcpy_h2d')

    def download(self, shape, dtype=np.float64, byte_offset=0):
        out = np.empty(shape, dtype=dtype)
        check(load().ssmq_memcpy_d2h(out.ctypes.data_as(ctypes.c_void_p), self.at(byte_offset),
                                     ctypes.c_size_t(out.nbytes)), 'ssmq_memcpy_d2h')
        return out

    def zero(self):
        check(load().ssmq_memset(ctypes.c_void_p(self.ptr), 0, ctypes.c_size_t(self.nbytes)), 'ssmq_memset')

    def free(self):
        if self.ptr:
            load().ssmq_free(ctypes.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _ScratchPool:
    """Device blocks handed back by the batch entry points, kept for the next call: hipFree synchronises the device and unmaps
    (1.1 ms for the six buffers of a forward_pass_batch at B = 1e4, T = 100 - half of that call, tools/api_breakdown.py).  At
    most `limit` bytes are kept; a block serves requests between half its size and its size.  Flushed when the library moves to
    another device."""
    limit = 4 << 30

    def __init__(self):
        import threading
        self.blocks = []          # (capacity, ptr), oldest first
        self.cached = 0
        self.lock = threading.Lock()     # threads have their own streams (include/ssmq.h): a block comes back only after the
                                         # thread that used it has synchronised (every caller downloads its results first)

    def take(self, nbytes):
        with self.lock:
            best = None
            for i, (cap, ptr) in enumerate(self.blocks):
                if nbytes <= cap <= 2 * nbytes + 4096 and (best is None or cap < self.blocks[best][0]):
                    best = i
            if best is None:
                return None
            cap, ptr = self.blocks.pop(best)
            self.cached -= cap
            return cap, ptr

    def give(self, cap, ptr):
        drop = []
        with self.lock:
            self.blocks.append((cap, ptr))
            self.cached += cap
            while self.cached > self.limit and self.blocks:
                c, p = self.blocks.pop(0)
                self.cached -= c
                drop.append(p)
        for p in drop:
            load().ssmq_free(ctypes.c_void_p(p))

    def flush(self):
        with self.lock:
            blocks, self.blocks, self.cached = self.blocks, [], 0
        for c, p in blocks:
            load().ssmq_free(ctypes.c_void_p(p))


_scratch_pool = _ScratchPool()


class ScratchBuffer(DeviceBuffer):
    """A DeviceBuffer whose free() returns the block to the pool instead of the driver (contents are undefined on reuse, as
    they are after hipMalloc)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        got = _scratch_pool.take(max(self.nbytes, 8))
        if got is None:
            DeviceBuffer.__init__(self, nbytes)
            self.capacity = max(self.nbytes, 8)
        else:
            self.capacity, self.ptr = got

    def free(self):
        if self.ptr:
            _scratch_pool.give(self.capacity, self.ptr)
            self.ptr = None


def scratch(nbytes):
    return ScratchBuffer(nbytes)


class _PinnedOwner:
    """Owns one block of ssmq_pinned_alloc and presents it through the array interface; the block goes back to the library's pool
    when the last ndarray that views it is collected (np.asarray(owner).base is the owner)."""
    __slots__ = ('ptr', 'nbytes', '__array_interface__')

    def __init__(self, shape):
        n = 1
        for v in shape:
            n *= int(v)
        p = ctypes.c_void_p()
        check(load().ssmq_pinned_alloc(ctypes.c_size_t(max(8 * n, 8)), ctypes.byref(p)), 'ssmq_pinned_alloc')
        self.ptr, self.nbytes = p.value, 8 * n
        self.__array_interface__ = {'data': (p.value, False), 'shape': tuple(int(v) for v in shape), 'typestr': '<f8', 'version': 3}
        _pinned_out[0] += self.nbytes

    def __del__(self):
        try:
            if self.ptr:
                _pinned_out[0] -= self.nbytes
                if _lib is not None:
                    _lib.ssmq_pinned_free(ctypes.c_void_p(self.ptr))
        except Exception:       # noqa: BLE001  (interpreter shutdown)
            pass
        self.ptr = None


PINNED_RESULT_LIMIT = 256 << 20      # results larger than this (or once four times this much is out) are ordinary pageable arrays
_pinned_out = [0]


def pinned_empty(shape):
    """An ndarray in page-locked memory (what `forward_pass_batch` returns when it pipelines its transfers: the copy engine
    writes the result in place).  An ordinary array to its user - writable, owned by nobody else; the block returns to the
    library's pool when the array is collected.  None when the request is over the limit for pinned results."""
    n = 1
    for v in shape:
        n *= int(v)
    if 8 * n > PINNED_RESULT_LIMIT or _pinned_out[0] + 8 * n > 4 * PINNED_RESULT_LIMIT:
        return None
    return np.asarray(_PinnedOwner(shape))


def upload_study(arr, n_elem, ld, dst):
    """arr (n_elem..., T, B) host array (reference layout) -> planes [T][n_elem][ld] at DeviceBuffer dst."""
    arr = np.ascontiguousarray(arr, dtype=np.float64)
    T, B = arr.shape[-2], arr.shape[-1]
    check(load().ssmq_upload_planes(arr.ctypes.data_as(c_double_p), T, int(n_elem), B, int(ld), ctypes.c_void_p(dst.ptr)),
          'ssmq_upload_planes')


def download_study(src, shape_elem, T, B, ld):
    """planes [T][prod(shape_elem)][ld] at DeviceBuffer src -> host array shape_elem + (T, B) (reference layout)."""
    out = np.empty(tuple(shape_elem) + (int(T), int(B)))
    n = int(np.prod(shape_elem)) if len(shape_elem) else 1
    check(load().ssmq_download_planes(ctypes.c_void_p(src.ptr), int(T), n, int(B), int(ld), out.ctypes.data_as(c_double_p)),
          'ssmq_download_planes')
    return out


class SoA:
    """fp64 planes [n_elem][ld] in HBM - the library's batch layout (element e of trajectory b at e * ld + b)."""

    def __init__(self, n_elem, batch, ld=None):
        self.n, self.B = int(n_elem), int(batch)
        self.ld = int(ld) if ld is not None else (self.B + 63) // 64 * 64
        self.buf = DeviceBuffer(8 * self.n * self.ld)

    @property
    def ptr(self):
        return ctypes.c_void_p(self.buf.ptr)

    def plane_ptr(self, first_elem):
        return self.buf.at(8 * int(first_elem) * self.ld)

    @classmethod
    def from_host(cls, arr, ld=None):
        """arr: (B, ...) in the reference layout (trajectory-major)."""
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        b = arr.shape[0]
        n = int(np.prod(arr.shape[1:])) if arr.ndim > 1 else 1
        out = cls(n, b, ld)
        if b:
            tmp = DeviceBuffer(arr.nbytes)
            tmp.upload(arr)
            check(load().ssmq_aos_to_soa(ctypes.c_void_p(tmp.ptr), out.ptr, n, b, out.ld), 'ssmq_aos_to_soa')
            sync()
            tmp.free()
        return out

    def to_host(self, shape_tail=None):
        """Back to the reference layout: (B,) + shape_tail."""
        tail = (self.n,) if shape_tail is None else tuple(shape_tail)
        if self.B == 0:
            return np.empty((0,) + tail)
        tmp = DeviceBuffer(8 * self.n * self.B)
        check(load().ssmq_soa_to_aos(self.ptr, ctypes.c_void_p(tmp.ptr), self.n, self.B, self.ld), 'ssmq_soa_to_aos')
        out = tmp.download((self.B,) + tail)
        tmp.free()
        return out


class Event:
    def __init__(self):
        p = ctypes.c_void_p()
        check(load().ssmq_event_create(ctypes.byref(p)), 'ssmq_event_create')
        self.ptr = p.value

    def record(self):
        check(load().ssmq_event_record(ctypes.c_void_p(self.ptr)), 'ssmq_event_record')

    def elapsed_ms(self, stop):
        ms = ctypes.c_float(0)
        check(load().ssmq_event_elapsed_ms(ctypes.c_void_p(self.ptr), ctypes.c_void_p(stop.ptr), ctypes.byref(ms)),
              'ssmq_event_elapsed_ms')
        return ms.value

    def __del__(self):
        try:
            if self.ptr:
                load().ssmq_event_destroy(ctypes.c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass
