#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "interp or kv_attention or dropout3d" > gpurun_out/r2t_k.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/r2t_k.log
timeout -k 10 600 python -m pytest tests/test_gpu_swin.py -x -q -s -k "segformer" > gpurun_out/r2t_s.log 2>&1; echo "rc=$?"; grep "SegFormer3D" gpurun_out/r2t_s.log; tail -6 gpurun_out/r2t_s.log
