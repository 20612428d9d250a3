#!/bin/bash
# session baseline: gpu tests + eager kernel-trace profiles of the three headline workloads + their bench lines
tag=${1:-r3c}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/${tag}_pytest.log
for w in unet swin_unetr; do
  extra=""; [ $w = unet ] && extra="--no-sliding-window"
  bash tools/prof.sh ${tag}_prof_$w --workload $w --steps 10 --warmup 3 --no-graph $extra > /dev/null 2>&1; tail -1 gpurun_out/${tag}_prof_$w/summary.txt | cut -c1-100
  timeout -k 10 500 python bench.py --workload $w --no-cpu-baseline --no-sliding-window > gpurun_out/${tag}_bench_$w.json 2> gpurun_out/${tag}_bench_$w.err; cut -c1-170 gpurun_out/${tag}_bench_$w.json
done
