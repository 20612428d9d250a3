#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_engine.py -x -q -k "overlap or two_ranks" > gpurun_out/r2w_t.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/r2w_t.log
timeout -k 10 300 python bench.py --workload swin_unetr --split-graph --no-cpu-baseline > gpurun_out/r2w_swin_split.json 2> gpurun_out/r2w_swin_split.err; cut -c1-200 gpurun_out/r2w_swin_split.json; tail -2 gpurun_out/r2w_swin_split.err
timeout -k 10 300 python bench.py --workload swin_unetr --no-cpu-baseline > gpurun_out/r2w_swin.json 2> gpurun_out/r2w_swin.err; cut -c1-200 gpurun_out/r2w_swin.json
