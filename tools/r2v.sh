#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_swin.py tests/test_gpu_kernels.py -x -q -k "segformer or swindepth or dwconv" > gpurun_out/r2v_t.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r2v_t.log
for w in segformer3d swin_depth; do
  timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline > gpurun_out/r2v_$w.json 2> gpurun_out/r2v_$w.err; echo "$w rc=$?"; cut -c1-200 gpurun_out/r2v_$w.json; tail -2 gpurun_out/r2v_$w.err
done
bash tools/prof.sh r2v_prof_swindepth --workload swin_depth --steps 5 --warmup 2 --no-graph > /dev/null 2>&1; head -8 gpurun_out/r2v_prof_swindepth/summary.txt | cut -c1-150; tail -2 gpurun_out/r2v_prof_swindepth/summary.txt | cut -c1-100
