#!/bin/bash
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -x -q > gpurun_out/pe_tests.log 2>&1 || { tail -30 gpurun_out/pe_tests.log; exit 1; }
tail -1 gpurun_out/pe_tests.log
for i in 1 2; do
python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/pe_new$i.json 2> gpurun_out/pe_new$i.err
MSSEG_NO_PLANAR_DGRAD=1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/pe_old$i.json 2> gpurun_out/pe_old$i.err
done
cut -c1-160 gpurun_out/pe_new1.json gpurun_out/pe_old1.json gpurun_out/pe_new2.json gpurun_out/pe_old2.json
