"""N > 1 path on CPU: MC trajectories sharded over 2 ranks (gloo), error sums all-reduced, against the single-process
result.  The per-rank filter and the per-rank error sums come from the oracle here (no GPU in this test; on the GPU box
they are `ssmq_filter_forward_dev` and `ssmq_error_sums_dev` / `ssmq_lcr_sums_dev`); what is under test is the sharding
and the two-phase aggregation that bench.py uses unchanged with backend nccl (RCCL)."""
import os
import socket

import numpy as np
import pytest

from ssmtoybox_amd import mcshard


def test_shard_bounds_cover_everything():
    for total in (0, 1, 7, 10000, 100001):
        for world in (1, 2, 3, 8):
            spans = [mcshard.shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _study(B=48, T=30, seed=3):
    from bench import simulate_ungm
    from oracle import ssmq_oracle as orc, c_oracle as co
    x, y = simulate_ungm(B, T, seed)
    par = np.array([1.0, 3.0])
    pts = orc.points_ut(1)
    w = orc.gp_weights(par, pts)
    one = np.ones((1, 1))
    td, k1 = co.make_transform(0, 1, 1, pts, w['wm'], w['Wc'], w['Wcc'], w['model_var'] * one,
                               integrand=co.Integrand.make(orc.F_UNGM_DYN))
    to, k2 = co.make_transform(0, 1, 1, pts, w['wm'], w['Wc'], w['Wcc'], w['model_var'] * one,
                               integrand=co.Integrand.make(orc.F_UNGM_MEAS))
    fm, fP, st = co.filter_forward(td, to, np.ascontiguousarray(y.T[:, :, None]), np.zeros(1), one, 10.0 * one, one)
    return x[None], fm.transpose(2, 1, 0), fP.transpose(2, 3, 1, 0), st == 0     # (D,T,B), (D,T,B), (D,D,T,B)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import ssmq_oracle as orc
    x, fm, fP, ok = _study()
    lo, hi = mcshard.shard_bounds(fm.shape[2], rank, world)
    loc = orc.error_sums(x[..., lo:hi], fm[..., lo:hi], fP[..., lo:hi], ok[lo:hi])
    tot = mcshard.finalize(mcshard.allreduce_sums(loc, dist))
    lcr = mcshard.allreduce_sums(orc.lcr_sums(x[..., lo:hi], fm[..., lo:hi], fP[..., lo:hi],
                                              tot['mse'] + 1e-6 * np.eye(fm.shape[0]), ok[lo:hi]), dist)
    if rank == 0:
        q.put((tot, lcr))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_aggregation_matches_single_process():
    import torch.multiprocessing as mp
    from oracle import ssmq_oracle as orc
    x, fm, fP, ok = _study()
    ref = mcshard.finalize(orc.error_sums(x, fm, fP, ok))
    ref_lcr = orc.lcr_sums(x, fm, fP, ref['mse'] + 1e-6 * np.eye(fm.shape[0]), ok)
    # direct formulas of the reference's metrics on the same data (utils.py:18-148)
    assert np.isclose(ref['rmse_avg'][5], np.mean(np.sqrt(((x - fm) ** 2).sum(axis=0))[5]))
    d = (x - fm)[0, 7, 0]
    p = fP[0, 0, 7, 0]
    assert np.isclose(orc.error_sums(x[..., :1], fm[..., :1], fP[..., :1])['nll'][7],
                      0.5 * (np.log(p) + d * d / p + np.log(2 * np.pi)))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    tot, lcr = q.get(timeout=120)
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    for k in ('rmse_avg', 'nll_avg', 'mse'):
        assert np.allclose(tot[k], ref[k], rtol=1e-12, atol=1e-12), k
    assert np.isclose(tot['rmse_total'], ref['rmse_total'], rtol=1e-12) and tot['count'] == ref['count']
    assert np.allclose(lcr['lcr'], ref_lcr['lcr'], rtol=1e-10, atol=1e-10) and np.array_equal(lcr['n'], ref_lcr['n'])
    assert mcshard.finalize_lcr(lcr).shape == (fm.shape[1],)


def _fallback_worker(rank, world, port, idfile, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      SSMQ_RCCL_ID_FILE=idfile)
    notes = []
    comm = mcshard.open_comm(rank, world, log=notes.append, consensus_timeout_s=60.0)
    out = comm.allreduce_sum(np.array([1.0 + rank, 10.0]))
    mx = comm.allreduce_max(np.array([float(rank)]))
    if rank == 0:
        q.put((type(comm).__name__, getattr(comm, 'fallback_reason', ''), out, mx, notes))
    comm.close()


def test_launched_ranks_agree_on_gloo_when_rccl_is_unavailable(tmp_path):
    """No GPU here, so no rank can create an RCCL communicator: both must notice (status files) and meet in a gloo group
    instead of one of them raising while the other waits - the N > 1 launch of bench.py goes through the same function."""
    import torch.multiprocessing as mp
    from ssmtoybox_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip('RCCL is available on this machine; the fallback is not taken')
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_fallback_worker, args=(r, 2, port, str(tmp_path / 'id'), q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    name, why, out, mx, notes = q.get(timeout=180)
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    assert name == 'TorchComm' and why and notes
    assert np.array_equal(out, [3.0, 20.0]) and np.array_equal(mx, [1.0])


# ---- bench.py --gpus N: the self-spawned launch (no GPU needed: the ranks here are a stand-in script) ----------------
_RANK_STUB = '''
import json, os, sys
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
assert os.environ['LOCAL_RANK'] == os.environ['RANK'] and os.environ['MASTER_ADDR'] == '127.0.0.1'
assert os.environ['SSMQ_RCCL_ID_FILE'].endswith('rccl.id') and int(os.environ['MASTER_PORT']) > 0
# every rank leaves a mark next to the id file: the ranks share ONE rendezvous directory
open(os.environ['SSMQ_RCCL_ID_FILE'] + '.seen%d' % rank, 'w').close()
mode = sys.argv[1]
if mode == 'fail' and rank == world - 1:
    sys.exit(3)
if rank == 0:
    import time
    d = os.path.dirname(os.environ['SSMQ_RCCL_ID_FILE'])
    t0 = time.time()
    while len([n for n in os.listdir(d) if '.seen' in n]) < world and time.time() - t0 < 30:
        time.sleep(0.01)
    print('a banner line some library printed')
    print(json.dumps({'n_gpus': 1 if mode == 'one' else world, 'argv': sys.argv[1:],
                      'seen': len([n for n in os.listdir(d) if '.seen' in n]), 'config': {}}))
else:
    print('rank', rank, 'says hello')          # must not reach the launcher's stdout
'''


def test_bench_gpus_flag_starts_that_many_ranks(tmp_path, capfd):
    import json
    import bench
    assert bench.needs_launcher(8, {}) and not bench.needs_launcher(1, {})
    assert not bench.needs_launcher(8, {'WORLD_SIZE': '8', 'RANK': '3'})          # under torch.distributed.run: a rank
    env = bench.child_env(2, 4, 29511, '/x/rccl.id', {'PATH': '/bin'})
    assert (env['RANK'], env['LOCAL_RANK'], env['WORLD_SIZE'], env['MASTER_PORT']) == ('2', '2', '4', '29511')
    assert env['SSMQ_RCCL_ID_FILE'] == '/x/rccl.id' and env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and env['PATH'] == '/bin'
    stub = tmp_path / 'rank.py'
    stub.write_text(_RANK_STUB)
    capfd.readouterr()
    assert bench.launch_ranks(3, ['ok', '--steps', '5'], timeout_s=60, script=str(stub)) == 0
    out, err = capfd.readouterr()
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out                                     # ONE JSON line on stdout, banners and peers on stderr
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 3 and rec['seen'] == 3 and rec['argv'] == ['ok', '--steps', '5']
    assert '3 child processes' in rec['config']['launcher']
    assert 'banner line' in err and 'says hello' in err
    assert bench.launch_ranks(2, ['fail'], timeout_s=60, script=str(stub)) == 3        # a rank that dies fails the launch
    capfd.readouterr()
    assert bench.launch_ranks(2, ['one'], timeout_s=60, script=str(stub)) == 1         # one GPU measured N times: refused
    assert 'n_gpus = 1 for --gpus 2' in capfd.readouterr()[1]
