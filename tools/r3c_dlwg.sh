#!/bin/bash
# transposed-conv weight gradient on the one-pass kernel (gather mode of csrc/linear_wgrad.hip) + 48-wide stem: tests, same-box A/B
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "deconv or linear or stem" > gpurun_out/r3c_dlwg_pytest.log 2>&1; rc=$?; tail -5 gpurun_out/r3c_dlwg_pytest.log
[ $rc -ne 0 ] && exit $rc
bash tools/r3_ab_swin.sh r3c_ab_dlwg_swin swin_unetr "MSSEG_NO_DECONV_LWG=1 MSSEG_NO_STEM=1" "MSSEG_NO_STEM=1" "-" > gpurun_out/r3c_ab_dlwg_swin.txt 2>&1; head -42 gpurun_out/r3c_ab_dlwg_swin.txt
