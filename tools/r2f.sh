#!/bin/bash
# direct small-grid conv: correctness tests then A/B timing per shape (MSSEG_K3DIRECT_MAX=0 -> tile kernels)
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv3d_k3 or fused or dgrad" > gpurun_out/r2f_k.log 2>&1; echo rc=$?; tail -3 gpurun_out/r2f_k.log
for shp in "128 256 6" "256 256 6" "64 128 12" "128 128 12" "256 128 12" "32 64 24" "64 64 24" "128 64 24"; do
  set -- $shp
  for mode in fwd fwdstats; do
    a=$(MSSEG_K3DIRECT_MAX=100000 python tools/bench_conv.py $mode $1 $2 $3 50 | head -1)
    b=$(MSSEG_K3DIRECT_MAX=0 python tools/bench_conv.py $mode $1 $2 $3 50 | head -1)
    echo "direct: $a"; echo "tiles : $b"
  done
done
