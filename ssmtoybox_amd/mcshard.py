"""
Sharding of independent Monte-Carlo trajectories across ranks (one process per GPU) and the final aggregation of the
error sums - the only collective of the path.

Trajectories never interact inside the moment-transform / filter path (the reference loops `for imc in range(mc)`,
research/tpq/tpq_base.py:187-189), so each rank owns a contiguous slice of the MC index and no data-path collective
exists.  Phase 1 all-reduces per-time-step sums (squared error, RMSE, NLL, MSE matrix, counts) that a reduction kernel
produced from the filter's own output buffers; phase 2 - only if the log-credibility ratio is wanted, because it needs
the GLOBAL MSE matrix per step (utils.py:113-120 via research/tpq/tpq_base.py:167-169) - all-reduces the LCR sums.
Messages are a few KB: latency-bound, one fused all-reduce per phase.

The collective itself is `ssmq_allreduce_sum` of the C ABI: RCCL over xGMI, opened by libssmq at run time, no PyTorch
(`RcclComm`).  Rendezvous needs only the launcher's environment (RANK, WORLD_SIZE, MASTER_PORT): rank 0 creates the
128-byte RCCL id and leaves it in a file that the other ranks of the node pick up.  `TorchComm` wraps a
`torch.distributed` process group instead - used by the CPU tests (gloo, no GPU, no RCCL) and for rehearsals with several
ranks on one GPU.
"""
import ctypes
import os
import time

import numpy as np

from . import _lib


def shard_bounds(total, rank, world):
    """Contiguous slice [lo, hi) of `total` trajectories owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# ---------------------------------------------------------------------------------------------------------------
# communicators
# ---------------------------------------------------------------------------------------------------------------
class SingleComm:
    """One process: every reduction is the identity."""
    rank, world = 0, 1

    def allreduce_sum(self, flat):
        return flat

    def allreduce_max(self, flat):
        return flat

    def barrier(self):
        if _lib.device_count() > 0:
            _lib.sync()

    def close(self):
        pass


def _id_file(rank_env=None):
    """Where rank 0 leaves the RCCL id for the other ranks of this launch.  All ranks of one node are children of one
    launcher process, so (parent pid, MASTER_PORT) names the launch; SSMQ_RCCL_ID_FILE overrides."""
    explicit = os.environ.get('SSMQ_RCCL_ID_FILE')
    if explicit:
        return explicit
    tmp = os.environ.get('TMPDIR', '/tmp')
    # an elastic restart keeps (ppid, port, run id): the restart count tells the attempts apart
    return os.path.join(tmp, 'ssmq_rccl_{}_{}_{}_{}.id'.format(
        os.getppid(), os.environ.get('MASTER_PORT', '0'), os.environ.get('TORCHELASTIC_RUN_ID', 'none'),
        os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')))


class RcclComm:
    """RCCL communicator behind the C ABI (ssmq_comm_*).  One per process; the device must be selected first."""

    def __init__(self, rank, world, id_file=None, timeout_s=300.0, force=False, init_timeout_s=240.0):
        """force: create an RCCL communicator even for world = 1 (rehearsal of the RCCL path on one GPU)."""
        lib = _lib.load()
        self.rank, self.world = int(rank), int(world)
        self._file = id_file or _id_file()
        self._rccl = self.world > 1 or force
        buf = ctypes.create_string_buffer(128)
        if self._rccl:
            if self.rank == 0:
                # nothing of an earlier attempt under the same name may be taken for this one's
                import glob
                for stale in [self._file] + glob.glob(self._file + '.st*'):
                    try:
                        os.unlink(stale)
                    except OSError:
                        pass
                _lib.check(lib.ssmq_comm_unique_id(buf, 128), 'ssmq_comm_unique_id')
                tmp = self._file + '.tmp{}'.format(os.getpid())
                with open(tmp, 'wb') as f:
                    f.write(buf.raw)
                os.replace(tmp, self._file)          # atomic: readers see all 128 bytes or no file
            else:
                t0 = time.time()
                while True:
                    try:
                        # a file left behind by an earlier launch that happened to share (ppid, port) is older than this
                        # process: ignore it
                        if os.path.getmtime(self._file) >= _process_start_time() - 1.0:
                            with open(self._file, 'rb') as f:
                                raw = f.read()
                            if len(raw) == 128:
                                break
                    except OSError:
                        pass
                    if _fresh_content(self._file + '.st0') == b'0':
                        raise _lib.SsmqError('RCCL rendezvous: rank 0 could not create the id')
                    if time.time() - t0 > timeout_s:
                        raise _lib.SsmqError('RCCL rendezvous: no id file {} after {} s'.format(self._file, timeout_s))
                    time.sleep(0.01)
                buf = ctypes.create_string_buffer(raw, 128)
        # ncclCommInitRank is collective: if a peer never arrives it blocks for good.  It runs on a helper thread (ctypes
        # releases the GIL) so that this rank can give up after init_timeout_s and report the failure (open_comm then
        # moves every rank to the gloo fallback); a thread left behind in RCCL is abandoned (hung_init).
        self.hung_init = False
        if self._rccl and (self.world > 1 or os.environ.get('SSMQ_RCCL_INIT_THREAD') == '1'):   # (env: one-rank rehearsal)
            import threading
            box = {}
            dev = lib.ssmq_current_device()              # HIP's current device is per thread: select it over there too

            def _init():
                try:
                    if dev >= 0:
                        _lib.check(lib.ssmq_set_device(dev), 'ssmq_set_device')
                    box['rc'] = lib.ssmq_comm_init(self.rank, self.world, buf, 128)
                    box['err'] = _lib.last_error() if box['rc'] else ''
                except Exception as e:                   # noqa: BLE001
                    box['rc'], box['err'] = -1, str(e)
            th = threading.Thread(target=_init, daemon=True)
            th.start()
            th.join(init_timeout_s)
            if th.is_alive():
                self.hung_init = True
                lib.ssmq_comm_abandon_init()             # the thread still holds stdout on stderr: take it back
                raise _lib.SsmqError('RCCL communicator: ncclCommInitRank did not return within {} s'.format(init_timeout_s))
            if box.get('rc'):
                raise _lib.SsmqError('ssmq_comm_init failed (code {}): {}'.format(box['rc'], box.get('err', '')))
        else:
            _lib.check(lib.ssmq_comm_init(self.rank, self.world, buf if self._rccl else None, 128), 'ssmq_comm_init')

    def _reduce(self, fn, flat):
        flat = np.ascontiguousarray(flat, dtype=np.float64).copy()
        if self._rccl and flat.size:
            _lib.check(fn(flat.ctypes.data_as(_lib.c_double_p), flat.size), 'ssmq_allreduce')
        return flat

    def allreduce_sum(self, flat):
        return self._reduce(_lib.load().ssmq_allreduce_sum, flat)

    def allreduce_max(self, flat):
        return self._reduce(_lib.load().ssmq_allreduce_max, flat)

    def barrier(self):
        _lib.check(_lib.load().ssmq_comm_barrier(), 'ssmq_comm_barrier')

    def close(self):
        self.barrier()                                # nobody is still looking for the id file
        _lib.check(_lib.load().ssmq_comm_destroy(), 'ssmq_comm_destroy')
        for path in ([self._file] if self.rank == 0 and self._rccl else []) + [getattr(self, 'status_file', None)]:
            try:
                if path:
                    os.unlink(path)
            except OSError:
                pass


def _fresh_content(path):
    """Content of a rendezvous file written by this launch (not older than the launcher), else None."""
    try:
        if os.path.getmtime(path) >= _process_start_time() - 1.0:
            with open(path, 'rb') as f:
                return f.read()
    except OSError:
        pass
    return None


def _process_start_time():
    try:
        return os.stat('/proc/{}'.format(os.getppid())).st_ctime
    except OSError:
        return 0.0


class TorchComm:
    """A torch.distributed process group as the communicator (CPU tests over gloo; rehearsals)."""

    def __init__(self, dist, device=None):
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device or ('cuda' if dist.get_backend() == 'nccl' else 'cpu')

    def _reduce(self, flat, op):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(flat, dtype=np.float64).copy()).to(self.device)
        if self.world > 1:
            self.dist.all_reduce(t, op=op)
        return t.cpu().numpy()

    def allreduce_sum(self, flat):
        return self._reduce(flat, self.dist.ReduceOp.SUM)

    def allreduce_max(self, flat):
        return self._reduce(flat, self.dist.ReduceOp.MAX)

    def barrier(self):
        if _lib.device_count() > 0:
            _lib.sync()
        self.dist.barrier()

    def close(self):
        self.dist.barrier()


def open_comm(rank, world, force_rccl=False, consensus_timeout_s=120.0, log=None):
    """Communicator of a launched job: RCCL behind the C ABI; if creating it fails on ANY rank, every rank falls back to
    a torch.distributed gloo group (the collective is a few KB at the end of the run, never the data path) and says so.
    The ranks agree through one status file each next to the id file, so no rank is left waiting inside a collective
    that others never entered."""
    if world <= 1 and not force_rccl:
        return SingleComm()
    comm, err = None, ''
    try:
        comm = RcclComm(rank, world, force=force_rccl)
    except Exception as e:                                   # noqa: BLE001 - any failure means "no RCCL on this rank"
        err = '{}: {}'.format(type(e).__name__, e)
    hung = 'did not return' in err
    if world <= 1:
        if comm is None:
            raise _lib.SsmqError('RCCL communicator: ' + err)
        return comm
    base = _id_file()
    with open('{}.st{}.tmp'.format(base, rank), 'wb') as f:
        f.write(b'1' if comm is not None else b'0')
    os.replace('{}.st{}.tmp'.format(base, rank), '{}.st{}'.format(base, rank))
    t0, states = time.time(), {}
    while len(states) < world and time.time() - t0 < consensus_timeout_s:
        for r in range(world):
            if r not in states:
                v = _fresh_content('{}.st{}'.format(base, r))
                if v in (b'0', b'1'):
                    states[r] = v == b'1'
        if len(states) < world:
            time.sleep(0.01)
    if len(states) == world and all(states.values()):
        comm.status_file = '{}.st{}'.format(base, rank)
        return comm
    why = err or 'rank(s) {} reported no RCCL communicator'.format(
        sorted(set(range(world)) - {r for r, ok in states.items() if ok}))
    (log or (lambda m: None))('rank {}: RCCL unavailable ({}); all-reduce falls back to gloo'.format(rank, why))
    import torch.distributed as dist
    dist.init_process_group('gloo')                          # every rank is past the status files once this returns
    try:
        os.unlink('{}.st{}'.format(base, rank))
    except OSError:
        pass
    fb = TorchComm(dist)
    fb.fallback_reason = why
    fb.abandoned_rccl_thread = hung      # the caller should leave with os._exit once its output is written
    return fb


def _as_comm(comm):
    """None -> single process; an object with is_initialized() (torch.distributed) -> TorchComm; else as given."""
    if comm is None:
        return SingleComm()
    if hasattr(comm, 'is_initialized'):
        return TorchComm(comm) if comm.is_initialized() else SingleComm()
    return comm


# ---------------------------------------------------------------------------------------------------------------
# device-side sums of this rank's trajectories
# ---------------------------------------------------------------------------------------------------------------
def device_error_sums(D, B, ld, T, d_x, d_fm, d_fP, d_status=None):
    """Phase-1 sums of this rank's trajectories, reduced on the device (`ssmq_error_sums_dev`).
    d_x, d_fm: DeviceBuffer planes [T][D][ld]; d_fP [T][D*D][ld] (the buffers the filter wrote); d_status [ld] or None.
    Returns a dict of arrays: se (T, D) squared error (utils.py:18-38), rmse (T,) sum of ||x - m|| (the quantity
    research/tpq/tpq_base.py:158-159 averages), nll (T,) negative log-likelihood (utils.py:123-148), mse (T, D, D) outer
    products (utils.py:41-64), n_ok (T,) trajectories counted, n_pd (T,) of them in the nll sum (all whose P is
    nonsingular: the reference's formula does not need a positive-definite P), n_all (T,) trajectories this rank ran
    (so that what was left out is known after the reduction)."""
    lib = _lib.load()
    W = lib.ssmq_error_sums_width(D)
    if W < 0:
        raise _lib.SsmqError('ssmq_error_sums_width: dimension {} out of range'.format(D))
    sums, ps = _lib.out_c((T, W))
    _lib.check(lib.ssmq_error_sums_dev(D, B, ld, T, ctypes.c_void_p(d_x.ptr), ctypes.c_void_p(d_fm.ptr),
                                       ctypes.c_void_p(d_fP.ptr), ctypes.c_void_p(d_status.ptr if d_status else None),
                                       ps), 'ssmq_error_sums_dev')
    return dict(se=sums[:, :D].copy(), rmse=sums[:, D].copy(), nll=sums[:, D + 1].copy(),
                mse=sums[:, D + 2:D + 2 + D * D].reshape(T, D, D).copy(), n_ok=sums[:, D + 2 + D * D].copy(),
                n_pd=sums[:, D + 3 + D * D].copy(), n_all=np.full(T, float(B)))


def device_lcr_sums(D, B, ld, T, d_x, d_fm, d_fP, mse_global, d_status=None, reg=1e-6):
    """Phase 2 (`ssmq_lcr_sums_dev`): sums over this rank's trajectories of the log credibility ratio (utils.py:66-120)
    per time step, given the GLOBAL MSE matrices (T, D, D) (+ reg I as research/tpq/tpq_base.py:161-167 does).
    Returns dict lcr (T,), n (T,)."""
    lib = _lib.load()
    M, pM = _lib.as_c(np.asarray(mse_global, dtype=np.float64) + reg * np.eye(D))
    sums, ps = _lib.out_c((T, 2))
    _lib.check(lib.ssmq_lcr_sums_dev(D, B, ld, T, ctypes.c_void_p(d_x.ptr), ctypes.c_void_p(d_fm.ptr),
                                     ctypes.c_void_p(d_fP.ptr), ctypes.c_void_p(d_status.ptr if d_status else None),
                                     pM, ps), 'ssmq_lcr_sums_dev')
    return dict(lcr=sums[:, 0].copy(), n=sums[:, 1].copy())


def _pack(sums, keys):
    return np.concatenate([np.asarray(sums[k], dtype=np.float64).reshape(-1) for k in keys])


def _unpack(flat, sums, keys):
    out, pos = {}, 0
    for k in keys:
        n = int(np.asarray(sums[k]).size)
        out[k] = flat[pos:pos + n].reshape(np.asarray(sums[k]).shape)
        pos += n
    return out


def allreduce_sums(sums, comm=None):
    """Sum a dict of arrays over all ranks with ONE all-reduce of the packed buffer.  comm: RcclComm / TorchComm /
    an initialised torch.distributed module / None (single process)."""
    comm = _as_comm(comm)
    keys = sorted(sums)
    return _unpack(comm.allreduce_sum(_pack(sums, keys)), sums, keys)


def finalize(total, strict=False):
    """Global averages from all-reduced phase-1 sums: rmse_avg (T,), nll_avg (T,), mse (T, D, D), rmse_total (), count ()
    trajectories aggregated, and what was LEFT OUT of them: excluded_failed (T,) trajectories whose filter had failed
    (lost positive definiteness: the reference raises LinAlgError there and has no result for the run at all),
    excluded_not_pd (T,) finished trajectories whose filtered covariance at that step is singular (no NLL term:
    numpy.linalg.inv raises there).  The reference averages over all runs (utils.py:123-148 via research/tpq/tpq_base.py:154-172), so an average
    taken over fewer runs is not the same statistic: with strict=True the averages of a step with exclusions are NaN."""
    n = np.maximum(total['n_ok'], 1.0)
    T = total['rmse'].shape[0]
    n_all = total.get('n_all', total['n_ok'])
    out = dict(rmse_avg=total['rmse'] / n, nll_avg=total['nll'] / np.maximum(total['n_pd'], 1.0),
               mse=total['mse'] / n[:, None, None],
               rmse_total=float(np.sqrt(total['se'].sum() / max(total['n_ok'].sum(), 1.0))),
               count=float(total['n_ok'].max()) if T else 0.0,
               excluded_failed=n_all - total['n_ok'], excluded_not_pd=total['n_ok'] - total['n_pd'])
    if strict:
        bad = out['excluded_failed'] > 0
        out['rmse_avg'] = np.where(bad, np.nan, out['rmse_avg'])
        out['mse'] = np.where(bad[:, None, None], np.nan, out['mse'])
        out['nll_avg'] = np.where(bad | (out['excluded_not_pd'] > 0), np.nan, out['nll_avg'])
        if bad.any():
            out['rmse_total'] = float('nan')
    return out


def finalize_lcr(total):
    """Average log credibility ratio per time step (the inclination indicator's summand) from all-reduced phase-2 sums."""
    return total['lcr'] / np.maximum(total['n'], 1.0)
