#!/bin/bash
# one-GPU rehearsal of the data-parallel step structure (tools/, not part of the product)
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_engine.py -x -q > gpurun_out/ov_tests.log 2>&1
python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/ov_single.json 2> gpurun_out/ov_single.err
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --split-graph > gpurun_out/ov_split.json 2> gpurun_out/ov_split.err
MSSEG_NO_GRAD_OVERLAP=1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --split-graph > gpurun_out/ov_split_nooverlap.json 2> gpurun_out/ov_split_nooverlap.err
MSSEG_DIST_BACKEND=gloo MSSEG_BENCH_ONE_DEVICE=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ov_gloo2.json 2> gpurun_out/ov_gloo2.err
tail -3 gpurun_out/ov_tests.log
cat gpurun_out/ov_single.json gpurun_out/ov_split.json gpurun_out/ov_split_nooverlap.json gpurun_out/ov_gloo2.json | cut -c1-400
