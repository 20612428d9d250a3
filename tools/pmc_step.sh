#!/bin/bash
# whole-step HBM traffic: FETCH_SIZE and WRITE_SIZE in separate counter passes over eager steps of a bench workload
# usage: tools/pmc_step.sh <tag> [bench args]
tag=$1; shift
for ctr in FETCH_SIZE WRITE_SIZE; do
  mkdir -p $GRAFT_REPO_ROOT/gpurun_out/${tag}_$ctr
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_$ctr -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-graph "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_$ctr/run.log 2>&1)
done
cd $GRAFT_REPO_ROOT && python3 tools/pmc_step.py $tag
