#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -s -k "shortcut_gradient" > gpurun_out/r3c_accum_pytest.log 2>&1; rc=$?; grep -E "rel. L2|fp32|passed|failed|Error" gpurun_out/r3c_accum_pytest.log | tail -8
[ $rc -ne 0 ] && { tail -30 gpurun_out/r3c_accum_pytest.log; exit $rc; }
bash tools/r3_ab_swin.sh r3c_ab_accum swin_unetr "MSSEG_NO_DGRAD_ACCUM=1" "-" 2>&1 | head -3
