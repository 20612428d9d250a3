#!/bin/bash
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2g_tests.log 2>&1; echo "suite rc=$?"; tail -4 gpurun_out/r2g_tests.log
python bench.py --no-cpu-baseline > gpurun_out/r2g_unet.json 2>gpurun_out/r2g_unet.err; cut -c1-200 gpurun_out/r2g_unet.json
