#!/bin/bash
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/fc_tests.log 2>&1 || { tail -40 gpurun_out/fc_tests.log; exit 1; }
tail -2 gpurun_out/fc_tests.log
python bench.py --no-cpu-baseline > gpurun_out/fc_unet.json 2> gpurun_out/fc_unet.err
python bench.py --workload swin_unetr --no-cpu-baseline > gpurun_out/fc_swin.json 2> gpurun_out/fc_swin.err
cut -c1-220 gpurun_out/fc_unet.json gpurun_out/fc_swin.json
