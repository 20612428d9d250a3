#!/bin/bash
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r2n_tests.log 2>&1; echo "suite rc=$?"; tail -6 gpurun_out/r2n_tests.log
