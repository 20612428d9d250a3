#!/bin/bash
# usage: tools_prof.sh <tag> [bench args]   (runs on the GPU box via gpurun)
tag=$1; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag/run.log 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/stats.py gpurun_out/$tag
