#!/bin/bash
# sliced-output transposed-conv kernels (csrc/deconv_k2s2_gen.hip): tests, then same-box A/B against the generic path
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "deconv" > gpurun_out/r3c_deconv_pytest.log 2>&1; rc=$?; tail -5 gpurun_out/r3c_deconv_pytest.log
[ $rc -ne 0 ] && exit $rc
bash tools/r3_ab.sh r3c_ab_deconv_unet "MSSEG_NO_DECONV_GEN=1" "-"
bash tools/r3_ab_swin.sh r3c_ab_deconv_swin swin_unetr "MSSEG_NO_DECONV_GEN=1" "-"
