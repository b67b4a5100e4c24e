#!/bin/bash
# one rank through the launcher with the RCCL backend: the exact command shape the driver uses for N > 1
set -e
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps 50 --warmup 5 --no-mt6 --no-cpu-baseline > gpurun_out/rehearse1.json 2> gpurun_out/rehearse1.err || { tail -20 gpurun_out/rehearse1.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open('gpurun_out/rehearse1.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('n_gpus', 'value', 'ms_per_step', 'rmse', 'nll', 'inclination_indicator', 'trajectories_aggregated')})
PY
