#!/bin/bash
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -k "pool" > gpurun_out/pd_tests.log 2>&1 || { tail -30 gpurun_out/pd_tests.log; exit 1; }
tail -1 gpurun_out/pd_tests.log
for i in 1 2; do
python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/pd_new$i.json 2> gpurun_out/pd_new$i.err
MSSEG_NO_POOL_BWD_FUSE=1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/pd_fwdonly$i.json 2> gpurun_out/pd_fwdonly$i.err
done
cut -c1-160 gpurun_out/pd_new1.json gpurun_out/pd_fwdonly1.json gpurun_out/pd_new2.json gpurun_out/pd_fwdonly2.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pd_prof -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pd_prof.log 2>&1
