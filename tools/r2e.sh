#!/bin/bash
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2e_tests.log 2>&1; echo "suite rc=$?"; tail -4 gpurun_out/r2e_tests.log
python bench.py --no-cpu-baseline > gpurun_out/r2e_unet.json 2>gpurun_out/r2e_unet.err; cut -c1-200 gpurun_out/r2e_unet.json
MSSEG_NO_DECONV_FAST=1 python bench.py --no-cpu-baseline | cut -c60-200
MSSEG_NO_HEAD_FUSE=1 python bench.py --no-cpu-baseline | cut -c60-200
MSSEG_NO_HEAD_FUSE=1 MSSEG_NO_DECONV_FAST=1 python bench.py --no-cpu-baseline | cut -c60-200
python bench.py --no-cpu-baseline --workload sliding_window > gpurun_out/r2e_sw.json 2>gpurun_out/r2e_sw.err; cut -c1-200 gpurun_out/r2e_sw.json
bash tools/prof.sh r2e_prof_unet --steps 20 --warmup 5 --no-graph > /dev/null 2>&1; cut -c1-150 gpurun_out/r2e_prof_unet/summary.txt | head -12
