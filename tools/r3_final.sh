#!/bin/bash
# round-3 evidence set from ONE box and one call: kernel-trace stats (eager launches) + the plain bench line of the workloads,
# whole-step HBM traffic of the UNet step (FETCH_SIZE / WRITE_SIZE in separate counter passes).  usage: tools/r3_final.sh <tag>
tag=${1:-r3}
mkdir -p gpurun_out
for w in unet swin_unetr swin_unetr_official segformer3d; do
  extra=""; [ $w = unet ] && extra="--no-sliding-window"
  bash tools/prof.sh ${tag}_prof_$w --workload $w --steps 10 --warmup 3 --no-graph $extra > /dev/null 2>&1; tail -1 gpurun_out/${tag}_prof_$w/summary.txt | cut -c1-100
  timeout -k 10 500 python bench.py --workload $w > gpurun_out/${tag}_bench_$w.json 2> gpurun_out/${tag}_bench_$w.err; cut -c1-170 gpurun_out/${tag}_bench_$w.json
done
MSSEG_NO_SW_GRAPH= bash tools/prof.sh ${tag}_prof_sliding_window --workload sliding_window --steps 1 --warmup 1 > /dev/null 2>&1; tail -1 gpurun_out/${tag}_prof_sliding_window/summary.txt | cut -c1-100
timeout -k 10 500 python bench.py --workload sliding_window > gpurun_out/${tag}_bench_sliding_window.json 2> gpurun_out/${tag}_bench_sliding_window.err; cut -c1-170 gpurun_out/${tag}_bench_sliding_window.json
bash tools/pmc_step.sh ${tag}_step --workload unet --steps 5 --warmup 2 --no-sliding-window 2>&1 | tail -16
