#!/bin/bash
# two bench ranks on ONE GPU over gloo: rehearsal of the N > 1 path (sharding, barriers, two-phase reduce)
set -e
export SSMQ_BENCH_BACKEND=gloo
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 3 --no-mt6 --no-cpu-baseline > gpurun_out/rehearse2.json 2> gpurun_out/rehearse2.err || { tail -20 gpurun_out/rehearse2.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open('gpurun_out/rehearse2.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('n_gpus', 'value', 'ms_per_step', 'rmse', 'nll', 'inclination_indicator', 'trajectories_aggregated')})
PY
