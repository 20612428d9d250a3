#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_engine.py -x -q -k "eval_model_and_test_model or drivers" > gpurun_out/r2x_t.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r2x_t.log
