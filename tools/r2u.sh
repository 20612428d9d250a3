#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_swin.py tests/test_gpu_kernels.py -x -q -k "segformer or interp or kv_attention" > gpurun_out/r2u_t.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r2u_t.log
for w in segformer3d swin_depth; do
  timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline > gpurun_out/r2u_$w.json 2> gpurun_out/r2u_$w.err; echo "$w rc=$?"; cut -c1-330 gpurun_out/r2u_$w.json; tail -2 gpurun_out/r2u_$w.err
done
bash tools/prof.sh r2u_prof_segformer --workload segformer3d --steps 5 --warmup 2 --no-graph > /dev/null 2>&1; head -8 gpurun_out/r2u_prof_segformer/summary.txt | cut -c1-150; tail -2 gpurun_out/r2u_prof_segformer/summary.txt | cut -c1-100
