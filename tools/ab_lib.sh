#!/bin/bash
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q > gpurun_out/ab_tests.log 2>&1 || { tail -30 gpurun_out/ab_tests.log; exit 1; }
tail -1 gpurun_out/ab_tests.log
for i in 1 2; do
MSSEG_LIB=$PWD/tools/libnew.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/ab_new$i.json 2> gpurun_out/ab_new$i.err
MSSEG_LIB=$PWD/tools/libold.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/ab_old$i.json 2> gpurun_out/ab_old$i.err
done
cut -c1-160 gpurun_out/ab_new1.json gpurun_out/ab_old1.json gpurun_out/ab_new2.json gpurun_out/ab_old2.json
