"""
The RCCL branch of the multi-GPU path end to end on a machine without GPUs: `bench.launch_ranks` (what `python bench.py
--gpus N` runs) starts N = 8 stand-in ranks (tests/_standin_rank.py); each goes through bench.make_comm -> mcshard.open_comm
-> RcclComm (id file, ncclCommInitRank, status-file consensus) and bench.final_aggregation (two-phase all-reduce, per-rank
slots, latency loop) with a stand-in librccl (tests/stub_rccl/rccl_stub.c: the five symbols csrc/ssmq_comm.hip binds,
reducing through files) in place of the real one.  No 8-GPU node was available in any round: this is the N = 8 execution of
that code there is; it says nothing about xGMI bandwidth or scaling.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope='module')
def stub_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp('rccl_stub')
    subprocess.check_call(['gcc', '-O1', '-shared', '-fPIC', '-o', str(d / 'librccl.so.1'),
                           os.path.join(ROOT, 'tests', 'stub_rccl', 'rccl_stub.c')])
    return str(d)


def run_launch(stub_dir, world, tmp_path, extra_env=None):
    env = dict(os.environ)
    env.update(LD_LIBRARY_PATH=stub_dir + os.pathsep + env.get('LD_LIBRARY_PATH', ''), SSMQ_COMM_HOST_STAGING='1',
               SSMQ_BENCH_FORCE_RCCL='1', TMPDIR=str(tmp_path), PYTHONPATH=ROOT)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'SSMQ_BENCH_BACKEND'):
        env.pop(k, None)
    env.update(extra_env or {})
    code = ('import sys; sys.path.insert(0, {root!r}); import bench; '
            'raise SystemExit(bench.launch_ranks({world}, [], timeout_s=240.0, script={script!r}))').format(
                root=ROOT, world=world, script=os.path.join(ROOT, 'tests', '_standin_rank.py'))
    p = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def expected(world):
    from tests._standin_rank import local_sums
    from ssmtoybox_amd import mcshard
    tot = None
    for r in range(world):
        loc = local_sums(r)
        tot = loc if tot is None else {k: tot[k] + loc[k] for k in loc}
    agg = mcshard.finalize(tot)
    lcr = sum(np.full(7, 0.5 + r) * agg['mse'][:, 0, 0] for r in range(world)) / sum(100.0 - r for r in range(world))
    return agg, lcr


@pytest.mark.parametrize('world', [2, 8])
def test_rccl_branch_with_stub_library(stub_dir, tmp_path, world):
    rc, out, err = run_launch(stub_dir, world, tmp_path)
    assert rc == 0 and out is not None, err[-2000:]
    assert out['n_gpus'] == world and out['collective'] == 'RcclComm' and out['fallback'] == '', (out, err[-2000:])
    agg, lcr = expected(world)
    # the stub reduces in rank order: the same sums as a plain loop over the ranks, to rounding of the order of summation
    assert np.allclose(out['rmse_avg'], agg['rmse_avg'], rtol=1e-13, atol=0) and np.allclose(out['nll_avg'], agg['nll_avg'], rtol=1e-12)
    assert out['count'] == agg['count'] and np.allclose(out['lcr'], lcr, rtol=1e-12)
    assert out['per_rank_ms'] == [1.0 + r for r in range(world)] and out['per_rank_B'] == [100.0 - r for r in range(world)]
    slowest = 0.01 * (1 + max(r % 3 for r in range(world)))          # the stand-in's "timed passes": max over ranks
    assert out['elapsed_max'] >= out['own_elapsed'] and out['elapsed_max'] >= slowest - 1e-3 and out['allreduce_us'] > 0
    assert out['n_packed'] == 7 * (2 + 1 + 1 + 4 + 1 + 1 + 1)
    assert 'launcher' in out['config']                          # (bench.launch_ranks stamps the line it relays)
    left = [f for f in os.listdir(tmp_path) if 'rccl' in f]
    assert not left, left                                        # id file, status files and the stub's directory are gone


def test_one_rank_without_rccl_moves_every_rank_to_gloo(stub_dir, tmp_path):
    """ncclCommInitRank fails on rank 3 of 4: the status files carry that to every rank, all of them leave RCCL and the
    aggregation runs over a torch.distributed gloo group - same numbers, and the result line says which collective ran."""
    world = 4
    rc, out, err = run_launch(stub_dir, world, tmp_path, {'RCCL_STUB_FAIL_RANK': '3', 'MASTER_ADDR': '127.0.0.1'})
    assert rc == 0 and out is not None, err[-2000:]
    assert out['n_gpus'] == world and out['collective'] == 'TorchComm' and 'RCCL' in err, (out, err[-1500:])
    agg, lcr = expected(world)
    assert np.allclose(out['rmse_avg'], agg['rmse_avg'], rtol=1e-12) and np.allclose(out['lcr'], lcr, rtol=1e-12)
