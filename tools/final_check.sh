#!/bin/bash
# the driver's round-end sequence on one box: GPU tests, smoke(), the default bench line, the other workloads, a profile
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/fin_tests.log 2>&1 || { tail -40 gpurun_out/fin_tests.log; exit 1; }
tail -1 gpurun_out/fin_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/fin_smoke.log 2>&1 || { tail -20 gpurun_out/fin_smoke.log; exit 1; }
tail -1 gpurun_out/fin_smoke.log
python bench.py > gpurun_out/fin_bench.json 2> gpurun_out/fin_bench.err
cat gpurun_out/fin_bench.json
python bench.py --workload swin_unetr --no-cpu-baseline > gpurun_out/fin_swin.json 2> gpurun_out/fin_swin.err
python bench.py --workload sliding_window --no-cpu-baseline > gpurun_out/fin_sw.json 2> gpurun_out/fin_sw.err
cut -c1-200 gpurun_out/fin_swin.json gpurun_out/fin_sw.json
bash tools/prof.sh profr5 --steps 20 --warmup 5 --no-graph > /dev/null 2>&1
tail -2 gpurun_out/profr5/summary.txt | cut -c1-200
