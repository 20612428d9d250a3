#!/bin/bash
# round-2 GPU check: new baseline-shape tests first, then the whole GPU suite, smoke, the three bench workloads
mkdir -p gpurun_out
tag=${1:-r2a}
timeout -k 10 900 python -m pytest tests/test_gpu_baseline.py -x -q -s > gpurun_out/${tag}_baseline.log 2>&1; echo "baseline rc=$?"; tail -5 gpurun_out/${tag}_baseline.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_baseline.py > gpurun_out/${tag}_tests.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/${tag}_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1; tail -1 gpurun_out/${tag}_smoke.log
timeout -k 10 300 python bench.py > gpurun_out/${tag}_unet.json 2> gpurun_out/${tag}_unet.err; cut -c1-400 gpurun_out/${tag}_unet.json
timeout -k 10 300 python bench.py --workload swin_unetr > gpurun_out/${tag}_swin.json 2> gpurun_out/${tag}_swin.err; cut -c1-300 gpurun_out/${tag}_swin.json
timeout -k 10 400 python bench.py --workload sliding_window > gpurun_out/${tag}_sw.json 2> gpurun_out/${tag}_sw.err; cut -c1-300 gpurun_out/${tag}_sw.json
tail -3 gpurun_out/${tag}_*.err
