#!/bin/bash
for w in segformer3d swin_depth; do
  timeout -k 10 400 python bench.py --workload $w > gpurun_out/r2u_$w.json 2> gpurun_out/r2u_$w.err; echo "$w rc=$?"; cut -c1-260 gpurun_out/r2u_$w.json; tail -2 gpurun_out/r2u_$w.err
done
bash tools/prof.sh r2u_prof_segformer --workload segformer3d --steps 5 --warmup 2 > /dev/null 2>&1; head -14 gpurun_out/r2u_prof_segformer/summary.txt | cut -c1-150
bash tools/prof.sh r2u_prof_swindepth --workload swin_depth --steps 5 --warmup 2 > /dev/null 2>&1; head -10 gpurun_out/r2u_prof_swindepth/summary.txt | cut -c1-150
