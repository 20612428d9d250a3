#!/bin/bash
# 1x1x1 one-channel conv on the stem kernels + inference K-split at the 24^3 level: tests, Swin-UNETR step, sliding-window A/B
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gather or stem" > gpurun_out/r3c_s3_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r3c_s3_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_baseline.py tests/test_gpu_engine.py -x -q -k "sliding" > gpurun_out/r3c_s3_pytest2.log 2>&1; rc=$?; tail -3 gpurun_out/r3c_s3_pytest2.log
[ $rc -ne 0 ] && exit $rc
python bench.py --workload swin_unetr --no-cpu-baseline --steps 10 --warmup 3 2> /dev/null | cut -c1-200
for r in 1 2; do
  MSSEG_NO_KSPLIT_INFER=1 python bench.py --workload sliding_window --no-cpu-baseline 2> /dev/null | cut -c1-130
  python bench.py --workload sliding_window --no-cpu-baseline 2> /dev/null | cut -c1-130
done
