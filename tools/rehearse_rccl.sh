#!/bin/bash
# The driver's launch shape for bench.py --gpus N, on ONE GPU: (1) one rank with a real RCCL communicator through the
# C ABI (no PyTorch in the worker), (2) two ranks sharing the GPU over a torch.distributed gloo group.
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
SSMQ_BENCH_FORCE_RCCL=1 SSMQ_RCCL_INIT_THREAD=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 \
    bench.py --gpus 1 --steps 5 --warmup 2 --no-mt6 --no-cpu-baseline > gpurun_out/rehearse_rccl1.json 2> gpurun_out/rehearse_rccl1.err
SSMQ_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29532 \
    bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/rehearse_gloo2.json 2> gpurun_out/rehearse_gloo2.err
cut -c1-600 gpurun_out/rehearse_rccl1.json; echo; cut -c1-600 gpurun_out/rehearse_gloo2.json
