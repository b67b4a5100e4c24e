"""Rank set-up of bench.py: the communicator from the launcher's environment, the final aggregation (the path's only
collectives), and the self-spawned launch of `python bench.py --gpus N`."""
import json
import os
import sys
import time

import numpy as np

from .common import ROOT

def make_comm():
    """Communicator from the launcher's environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*): RCCL behind the C ABI
    (default; no PyTorch), or SSMQ_BENCH_BACKEND=gloo - a torch.distributed gloo group, for rehearsals with several ranks
    on one GPU or none.  Returns (comm, rank, world, local_rank)."""
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    backend = os.environ.get('SSMQ_BENCH_BACKEND', 'rccl')
    from ssmtoybox_amd import mcshard, _lib
    have = _lib.device_count()
    ndev = max(have, 1)
    local_rank %= ndev
    if have > 0:                                 # (none: the stand-in ranks of tests/test_rccl_stub.py on a machine without a GPU)
        _lib.set_device(local_rank)
    launched = world > 1 or ('RANK' in os.environ and 'MASTER_PORT' in os.environ)
    if not launched:
        return mcshard.SingleComm(), 0, 1, local_rank
    why = ''
    if backend != 'gloo' and world > ndev and os.environ.get('SSMQ_BENCH_FORCE_RCCL') != '1':
        # RCCL refuses two ranks on one device; every rank sees the same device count, so all of them take this branch
        backend, why = 'gloo', '{} ranks on {} device(s)'.format(world, ndev)
        if rank == 0:
            sys.stderr.write('bench.py: {} - RCCL needs one device per rank, all-reduce over gloo\n'.format(why))
    if backend == 'gloo':
        import torch.distributed as dist
        dist.init_process_group('gloo')
        comm = mcshard.TorchComm(dist)
        comm.fallback_reason = why
        return comm, rank, world, local_rank
    comm = mcshard.open_comm(rank, world, force_rccl=os.environ.get('SSMQ_BENCH_FORCE_RCCL') == '1',
                             log=lambda m: sys.stderr.write(m + '\n'))
    return comm, rank, world, local_rank


def final_aggregation(comm, rank, world, loc, lcr_sums_of, pass_ms_dev, B):
    """What every rank does after its timed passes - the path's only collectives:
    phase 1: this rank's per-time-step error sums `loc` (mcshard.device_error_sums: reduced on the device from the filter's
    output buffers), ONE all-reduce of the packed buffer; phase 2: log credibility ratio against the GLOBAL per-step MSE matrix
    (`lcr_sums_of(mse)` -> this rank's sums), a second all-reduce; then the per-rank launch times and trajectory counts (one
    slot per rank, summed) and what the final collective costs: the packed phase-1 buffer all-reduced 20 times after a common
    start (every rank takes part: collective calls).  Shared by main() and the stand-in ranks of tests/test_rccl_stub.py."""
    from ssmtoybox_amd import mcshard
    agg = mcshard.finalize(mcshard.allreduce_sums(loc, comm))
    lcr = mcshard.finalize_lcr(mcshard.allreduce_sums(lcr_sums_of(agg['mse']), comm))
    slot = np.zeros(2 * world)
    slot[rank], slot[world + rank] = pass_ms_dev, B
    slot = comm.allreduce_sum(slot)
    n_packed = sum(int(np.asarray(v).size) for v in loc.values())
    lat = []
    comm.barrier()
    for _ in range(20):
        t1 = time.perf_counter()
        comm.allreduce_sum(np.zeros(n_packed))
        lat.append(time.perf_counter() - t1)
    return dict(agg=agg, lcr=lcr, slot=slot, allreduce_us=float(np.median(lat)) * 1e6, n_packed=n_packed)


def free_port():
    import socket
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def child_env(rank, world, port, id_file, base=None):
    """Environment of rank `rank` of a self-spawned launch: what torch.distributed.run would export (RANK, LOCAL_RANK,
    WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR, MASTER_PORT) plus the explicit rendezvous file of the RCCL id, so the
    ranks do not depend on sharing a parent pid."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), SSMQ_RCCL_ID_FILE=id_file, SSMQ_BENCH_CHILD='1')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC only on this pool (RCCL across processes)
    return env


def needs_launcher(gpus, env=None):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: this process starts the N ranks itself."""
    env = os.environ if env is None else env
    return gpus > 1 and 'WORLD_SIZE' not in env and 'RANK' not in env


def launch_ranks(gpus, argv, timeout_s=1500.0, script=None):
    """Start `gpus` fresh processes of this file (one rank per GPU), relay rank 0's JSON line, return the exit code.
    This process never touches the GPU (children are started with subprocess, not exec)."""
    import subprocess
    import tempfile
    tmp = tempfile.mkdtemp(prefix='ssmq_bench_')
    id_file = os.path.join(tmp, 'rccl.id')
    port = free_port()
    me = os.path.abspath(script or os.path.join(ROOT, 'bench.py'))
    procs = []
    for r in range(gpus):
        procs.append(subprocess.Popen([sys.executable, me] + list(argv), env=child_env(r, gpus, port, id_file),
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    rc, line = 0, None
    t_end = time.time() + timeout_s
    try:
        text, _ = procs[0].communicate(timeout=max(1.0, t_end - time.time()))
        for ln in text.decode('utf-8', 'replace').splitlines():
            if ln.startswith('{'):
                line = ln
            elif ln.strip():
                sys.stderr.write(ln + '\n')
        for pr in procs:
            pr.wait(timeout=max(1.0, t_end - time.time()))
    except subprocess.TimeoutExpired:
        sys.stderr.write('bench.py: ranks did not finish within {:.0f} s\n'.format(timeout_s))
        rc = 124
    for r, pr in enumerate(procs):
        if pr.poll() is None:
            pr.kill()
            pr.wait()
        if pr.returncode and not rc:
            sys.stderr.write('bench.py: rank {} exited with code {}\n'.format(r, pr.returncode))
            rc = pr.returncode if pr.returncode > 0 else 1
    for name in os.listdir(tmp):
        try:
            os.unlink(os.path.join(tmp, name))
        except OSError:
            pass
    try:
        os.rmdir(tmp)
    except OSError:
        pass
    if line is None:
        sys.stderr.write('bench.py: rank 0 printed no result line\n')
        return rc or 1
    out = json.loads(line)
    out.setdefault('config', {})['launcher'] = 'bench.py --gpus {}: {} child processes, one rank per GPU'.format(gpus, gpus)
    print(json.dumps(out))
    if out.get('n_gpus') != gpus:
        sys.stderr.write('bench.py: result line reports n_gpus = {} for --gpus {}\n'.format(out.get('n_gpus'), gpus))
        return rc or 1
    return rc



def rank_devices(comm, rank, world):
    """Every rank's device (name, PCI bus id) gathered on all ranks with ONE small all-reduce (each rank fills its own slot of a
    zero array with the bytes of its strings, the sum is the gather), plus the communicator's own idea of its size
    (ssmq_comm_world).  On --gpus N > 1 the record shows at a glance whether N DISTINCT devices took part."""
    from ssmtoybox_amd import _lib
    W = 96
    try:
        name, bus = _lib.device_name(), _lib.device_pci_bus_id()
    except Exception as e:                         # the stand-in ranks of tests/test_rccl_stub.py have no device
        name, bus = 'no device ({})'.format(type(e).__name__), 'none:{}'.format(rank)
    text = (bus + '|' + name).encode()[:W]
    slot = np.zeros(world * W)
    slot[rank * W:rank * W + len(text)] = np.frombuffer(text, dtype=np.uint8)
    slot = comm.allreduce_sum(slot) if world > 1 else slot
    per_rank = []
    for r in range(world):
        b = bytes(int(round(v)) for v in slot[r * W:(r + 1) * W]).rstrip(b'\x00').decode('utf-8', 'replace')
        per_rank.append(b)
    buses = [p.split('|')[0] for p in per_rank]
    names = sorted(set(p.split('|', 1)[1] if '|' in p else p for p in per_rank))
    try:
        cw = int(_lib.load().ssmq_comm_world())
    except Exception:
        cw = None
    return {'per_rank': per_rank, 'distinct': len(set(buses)), 'comm_world': cw,
            'compact': '{} x {} [{}]'.format(world, ' / '.join(names)[:60], ','.join(buses)[:120])}
