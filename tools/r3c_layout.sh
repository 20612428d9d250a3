#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_swin.py -x -q -k "layout or official" > gpurun_out/r3c_layout_pytest.log 2>&1; rc=$?; tail -4 gpurun_out/r3c_layout_pytest.log
[ $rc -ne 0 ] && { tail -40 gpurun_out/r3c_layout_pytest.log; exit $rc; }
for r in 1 2; do python bench.py --workload swin_unetr_official --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | cut -c1-200; done
