#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "dwconv3 or batchnorm" > gpurun_out/r2s_k.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r2s_k.log
timeout -k 10 600 python -m pytest tests/test_gpu_swin.py -x -q -s -k "swindepth" > gpurun_out/r2s_s.log 2>&1; echo "rc=$?"; grep "SwinDepth" gpurun_out/r2s_s.log; tail -4 gpurun_out/r2s_s.log
