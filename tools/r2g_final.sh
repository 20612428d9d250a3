#!/bin/bash
# end-of-round evidence (r2g): kernel-trace stats (eager launches) + the plain bench line of every workload
mkdir -p gpurun_out
for w in unet swin_unetr swin_unetr_official segformer3d swin_depth swinception; do
  bash tools/prof.sh r2g_prof_$w --workload $w --steps 10 --warmup 3 --no-graph > /dev/null 2>&1; tail -1 gpurun_out/r2g_prof_$w/summary.txt | cut -c1-120
  timeout -k 10 400 python bench.py --workload $w > gpurun_out/r2g_bench_$w.json 2> gpurun_out/r2g_bench_$w.err; cut -c1-160 gpurun_out/r2g_bench_$w.json
done
MSSEG_NO_SW_GRAPH= bash tools/prof.sh r2g_prof_sliding_window --workload sliding_window --steps 1 --warmup 1 > /dev/null 2>&1; tail -1 gpurun_out/r2g_prof_sliding_window/summary.txt | cut -c1-120
timeout -k 10 500 python bench.py --workload sliding_window > gpurun_out/r2g_bench_sliding_window.json 2> gpurun_out/r2g_bench_sliding_window.err; cut -c1-160 gpurun_out/r2g_bench_sliding_window.json
