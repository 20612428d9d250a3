#!/bin/bash
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -x -q > gpurun_out/pc_tests.log 2>&1 || { tail -40 gpurun_out/pc_tests.log; exit 1; }
tail -2 gpurun_out/pc_tests.log
python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/pc_new.json 2> gpurun_out/pc_new.err
MSSEG_NO_NORM_POOL=1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/pc_old.json 2> gpurun_out/pc_old.err
python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/pc_new2.json 2> gpurun_out/pc_new2.err
python bench.py --workload sliding_window --no-cpu-baseline > gpurun_out/pc_sw.json 2> gpurun_out/pc_sw.err
cut -c1-160 gpurun_out/pc_new.json gpurun_out/pc_old.json gpurun_out/pc_new2.json gpurun_out/pc_sw.json
