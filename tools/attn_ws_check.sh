#!/bin/bash
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_swin.py -x -q > gpurun_out/aw_tests.log 2>&1 || { tail -30 gpurun_out/aw_tests.log; exit 1; }
tail -2 gpurun_out/aw_tests.log
python tools/bench_attn.py > gpurun_out/aw_new.log 2>&1
MSSEG_ATTN_BWD_NO_WS=1 python tools/bench_attn.py > gpurun_out/aw_old.log 2>&1
NO_DTAB=1 python tools/bench_attn.py > gpurun_out/aw_nodtab.log 2>&1
grep bwd gpurun_out/aw_new.log gpurun_out/aw_old.log gpurun_out/aw_nodtab.log
python bench.py --workload swin_unetr --no-cpu-baseline > gpurun_out/aw_swin_new.json 2> gpurun_out/aw_swin_new.err
MSSEG_ATTN_BWD_NO_WS=1 python bench.py --workload swin_unetr --no-cpu-baseline > gpurun_out/aw_swin_old.json 2> gpurun_out/aw_swin_old.err
cut -c1-200 gpurun_out/aw_swin_new.json gpurun_out/aw_swin_old.json
