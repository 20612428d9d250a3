#!/bin/bash
# rocprofv3 kernel-trace stats of the three workloads (eager launches so that every row is a whole number of calls)
tag=${1:-r2a}
bash tools/prof.sh ${tag}_prof_unet --steps 20 --warmup 5 --no-graph > /dev/null 2>&1; tail -2 gpurun_out/${tag}_prof_unet/summary.txt | cut -c1-200
bash tools/prof.sh ${tag}_prof_swin --workload swin_unetr --steps 10 --warmup 3 --no-graph > /dev/null 2>&1; tail -2 gpurun_out/${tag}_prof_swin/summary.txt | cut -c1-200
MSSEG_NO_SW_GRAPH= bash tools/prof.sh ${tag}_prof_sw --workload sliding_window --steps 1 --warmup 1 > /dev/null 2>&1; tail -2 gpurun_out/${tag}_prof_sw/summary.txt | cut -c1-200
