#!/usr/bin/env python3
"""One rank of `bench.py --gpus N` WITHOUT a GPU: everything bench.py does around its timed passes - communicator from the
launcher's environment (bench.make_comm -> mcshard.open_comm: RCCL id file, ncclCommInitRank on every rank, status-file
consensus, gloo fallback), common-start barrier, max-over-ranks time, the two-phase aggregation and the latency loop
(bench.final_aggregation) - with host-made sums in place of the device reduction.  Started by bench.launch_ranks (the
launcher of `python bench.py --gpus N`) from tests/test_rccl_stub.py, with a stand-in librccl on LD_LIBRARY_PATH."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def local_sums(rank, T=7, D=2, B=100):
    """What mcshard.device_error_sums would return for this rank's trajectories (seeded per rank: the test recomputes them)."""
    rng = np.random.default_rng(1000 + rank)
    return dict(se=rng.random((T, D)), rmse=rng.random(T), nll=rng.standard_normal(T), mse=rng.random((T, D, D)),
                n_ok=np.full(T, float(B - rank)), n_pd=np.full(T, float(B - rank - 1)), n_all=np.full(T, float(B)))


def main():
    result_out = os.fdopen(os.dup(1), 'w')
    comm, rank, world, local_rank = bench.make_comm()
    comm.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (1 + rank % 3))                       # the "timed passes"
    elapsed = time.perf_counter() - t0
    comm.barrier()
    elapsed_max = float(comm.allreduce_max(np.array([elapsed]))[0])
    loc = local_sums(rank)
    fa = bench.final_aggregation(comm, rank, world, loc,
                                 lambda mse: dict(lcr=np.full(7, 0.5 + rank) * mse[:, 0, 0], n=np.full(7, 100.0 - rank)),
                                 1.0 + rank, 100 - rank)
    if rank == 0:
        agg = fa['agg']
        result_out.write(json.dumps({
            'n_gpus': world, 'collective': type(comm).__name__, 'fallback': getattr(comm, 'fallback_reason', ''),
            'elapsed_max': elapsed_max, 'own_elapsed': elapsed,
            'rmse_avg': [float(v) for v in agg['rmse_avg']], 'nll_avg': [float(v) for v in agg['nll_avg']],
            'count': agg['count'], 'lcr': [float(v) for v in fa['lcr']],
            'per_rank_ms': [float(v) for v in fa['slot'][:world]], 'per_rank_B': [float(v) for v in fa['slot'][world:]],
            'allreduce_us': fa['allreduce_us'], 'n_packed': fa['n_packed']}) + '\n')
        result_out.flush()
    comm.close()
    if getattr(comm, 'abandoned_rccl_thread', False):
        os._exit(0)


if __name__ == '__main__':
    main()
