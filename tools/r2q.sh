#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_engine.py -q -s -k "unetrc" > gpurun_out/r2q_tests.log 2>&1; echo "rc=$?"; grep "UNETRC\|rel-L2" gpurun_out/r2q_tests.log; tail -5 gpurun_out/r2q_tests.log
