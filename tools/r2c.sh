#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "deconv or head" > gpurun_out/r2c_k.log 2>&1; echo rc=$?; tail -5 gpurun_out/r2c_k.log
timeout -k 10 800 python -m pytest tests/test_gpu_baseline.py -x -q -s > gpurun_out/r2c_baseline.log 2>&1; echo rc=$?; tail -3 gpurun_out/r2c_baseline.log
python bench.py --no-cpu-baseline > gpurun_out/r2c_unet.json 2>gpurun_out/r2c_unet.err; cut -c1-200 gpurun_out/r2c_unet.json
python bench.py --no-cpu-baseline --workload sliding_window > gpurun_out/r2c_sw.json 2>gpurun_out/r2c_sw.err; cut -c1-200 gpurun_out/r2c_sw.json
