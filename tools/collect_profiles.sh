#!/bin/bash
# copy the evidence set of tools/r3_final.sh <tag> from gpurun_out/ (scratch) into profiles/ (tracked): tools/collect_profiles.sh <tag>
tag=${1:-r3c}
for w in unet swin_unetr swin_unetr_official segformer3d sliding_window; do
  d=gpurun_out/${tag}_prof_$w
  [ -d $d ] || continue
  cp $(ls -t $d/runc/*_kernel_stats.csv | head -1) profiles/${tag}_${w}_rocprofv3_kernel_stats.csv
  cp $d/summary.txt profiles/${tag}_${w}_rocprofv3_stats_summary.txt
  [ -s gpurun_out/${tag}_bench_$w.json ] && cp gpurun_out/${tag}_bench_$w.json profiles/${tag}_${w}_bench_line.json
done
[ -s gpurun_out/${tag}_step_traffic.json ] && cp gpurun_out/${tag}_step_traffic.json profiles/${tag}_unet_step_traffic.json
ls profiles | grep "^${tag}_" | wc -l
