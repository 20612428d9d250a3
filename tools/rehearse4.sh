#!/bin/bash
set -e
mkdir -p gpurun_out
export MSSEG_DIST_BACKEND=gloo MSSEG_BENCH_ONE_DEVICE=1
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 4 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r4_unet.json 2> gpurun_out/r4_unet.err
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 4 --workload sliding_window --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r4_sw.json 2> gpurun_out/r4_sw.err
cut -c1-330 gpurun_out/r4_unet.json gpurun_out/r4_sw.json
